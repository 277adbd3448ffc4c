#!/usr/bin/env python3
"""Debug aid: batch extraction on one stream WHILE the matcher runs on another (bench.py's two-stream schedule), compared frame
by frame with a batch extraction that ran alone. Prints what differs (keypoint rows, descriptor bytes)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import aria_slam_amd as A

W, H, NF, B = 640, 480, 2000, int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda", 0)
host = torch.empty((B, H, W), dtype=torch.uint8)
A.synth_sequence(1, B // 2, W, H, out=host.numpy())
img = host.to(dev)
se, sm = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
e = A.OrbHipExtractor(max_features=NF, max_width=W, max_height=H, max_batch=B, stream=se.cuda_stream)
m = A.HipMatcher(stream=sm.cuda_stream, max_query=4096, max_train=4096)
cap = e.kp_capacity()


def bufs():
    return (torch.zeros((B, cap, 24), dtype=torch.uint8, device=dev), torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev),
            torch.zeros(B, dtype=torch.int32, device=dev))


k0, d0, c0 = bufs()
with torch.cuda.stream(se):
    e.extract_batch_device(img, B, W, H, k0, d0, c0, cap)
torch.cuda.synchronize()
mt = torch.empty((B, cap, 12), dtype=torch.uint8, device=dev)
nm = torch.zeros(B, dtype=torch.int32, device=dev)
k1, d1, c1 = bufs()
for rep in range(3):
    with torch.cuda.stream(sm):
        for _ in range(2):
            m.match_batch_device(d0.data_ptr() + cap * 32, c0.data_ptr() + 4, d0, c0, B - 1, cap * 32, 0.75,
                                 mt.data_ptr() + cap * 12, nm.data_ptr() + 4, cap)
    with torch.cuda.stream(se):
        e.extract_batch_device(img, B, W, H, k1, d1, c1, cap)
    torch.cuda.synchronize()
    a, b = c0.cpu().numpy(), c1.cpu().numpy()
    ka, kb = k0.cpu().numpy(), k1.cpu().numpy()
    da, db = d0.cpu().numpy(), d1.cpu().numpy()
    bad = []
    for f in range(B):
        n = int(a[f])
        if a[f] != b[f] or ka[f, :n].tobytes() != kb[f, :n].tobytes() or da[f, :n].tobytes() != db[f, :n].tobytes():
            bad.append(f)
    print("rep", rep, "frames differing from the run alone:", len(bad), bad[:10])
    for f in bad[:4]:
        n, n2 = int(a[f]), int(b[f])
        x = np.frombuffer(ka[f, :n].tobytes(), dtype=np.float32).reshape(-1, 6)
        y = np.frombuffer(kb[f, :n2].tobytes(), dtype=np.float32).reshape(-1, 6)
        mrow = min(n, n2)
        rows = np.nonzero((x[:mrow] != y[:mrow]).any(axis=1))[0]
        drows = np.nonzero((da[f, :mrow] != db[f, :mrow]).any(axis=1))[0]
        print("  frame", f, "counts", n, n2, "keypoint rows differing", len(rows), "descriptor rows differing", len(drows))
        for i in rows[:3]:
            print("    row", int(i), "alone", x[i][:5], int(x[i][5:6].view(np.int32)[0]), "beside", y[i][:5], int(y[i][5:6].view(np.int32)[0]))
        for i in drows[:3]:
            if i not in rows:
                print("    desc row", int(i), "oct", int(x[i][5:6].view(np.int32)[0]), "xy", x[i][:2], "bytes differing", int((da[f, i] != db[f, i]).sum()))
