#!/usr/bin/env python3
# per-frame comparison of the default k_select ordering against ARIA_SELECT_SORT=bitonic on the bench sequence
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

def child(tag):
    import torch
    import aria_slam_amd as A
    W, H, NF, PAIRS = 640, 480, 2000, 4096
    B = 2 * PAIRS
    host = torch.empty((B, H, W), dtype=torch.uint8).pin_memory()
    A.synth_sequence(1, PAIRS, W, H, out=host.numpy())
    img = host.cuda()
    e = A.OrbHipExtractor(max_features=NF, max_width=W, max_height=H, max_batch=B)
    cap = e.kp_capacity()
    kps = torch.zeros((B, cap, 24), dtype=torch.uint8, device="cuda")
    desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(B, dtype=torch.int32, device="cuda")
    e.extract_batch_device(img.data_ptr(), B, W, H, kps.data_ptr(), desc.data_ptr(), cnt.data_ptr(), cap)
    torch.cuda.synchronize(); e.check()
    np.save("/tmp/cs_%s_kps.npy" % tag, kps.cpu().numpy()); np.save("/tmp/cs_%s_cnt.npy" % tag, cnt.cpu().numpy())
    np.save("/tmp/cs_%s_desc.npy" % tag, desc.cpu().numpy())

if len(sys.argv) > 1:
    child(sys.argv[1]); sys.exit(0)
env = dict(os.environ)
subprocess.check_call([sys.executable, __file__, "new"], env=env)
env["ARIA_SELECT_SORT"] = "bitonic"
subprocess.check_call([sys.executable, __file__, "old"], env=env)
ka, kb = np.load("/tmp/cs_new_kps.npy"), np.load("/tmp/cs_old_kps.npy")
ca, cb = np.load("/tmp/cs_new_cnt.npy"), np.load("/tmp/cs_old_cnt.npy")
da, db = np.load("/tmp/cs_new_desc.npy"), np.load("/tmp/cs_old_desc.npy")
bad = [f for f in range(len(ca)) if ca[f] != cb[f] or (ka[f, :ca[f]] != kb[f, :cb[f]]).any() or (da[f, :ca[f]] != db[f, :cb[f]]).any()]
print("differing frames:", len(bad), bad[:20])
for f in bad[:4]:
    print("frame", f, "counts", ca[f], cb[f])
    n = min(ca[f], cb[f])
    rows = np.nonzero((ka[f, :n] != kb[f, :n]).any(axis=1))[0]
    print("  rows differing:", len(rows), rows[:20])
    for r in rows[:6]:
        a = ka[f, r].view(np.float32)[:5]; b = kb[f, r].view(np.float32)[:5]
        print("   row", r, "new", a, ka[f, r, 20:24].view(np.int32), "old", b, kb[f, r, 20:24].view(np.int32))
