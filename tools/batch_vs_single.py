#!/usr/bin/env python3
"""Debug aid: the batch entry point against the single-frame entry point, frame by frame (bench.py's synthetic sequence).
Usage: tools/batch_vs_single.py W H NFEATURES FRAMES  -> prints the frames whose keypoints / descriptors differ."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import aria_slam_amd as A

W, H, NF, B = (int(v) for v in sys.argv[1:5])
host = torch.empty((B, H, W), dtype=torch.uint8)
A.synth_sequence(1, B // 2, W, H, out=host.numpy())
img = host.cuda()
e = A.OrbHipExtractor(max_features=NF, max_width=W, max_height=H, max_batch=B)
cap = e.kp_capacity()
kps = torch.zeros((B, cap, 24), dtype=torch.uint8, device="cuda")
desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
cnt = torch.zeros(B, dtype=torch.int32, device="cuda")
e.extract_batch_device(img, B, W, H, kps, desc, cnt, cap)
torch.cuda.synchronize()
print("batch FAST/blur kernel:", e.fast_blur_kernel())
kh, dh, ch = kps.cpu().numpy(), desc.cpu().numpy(), cnt.cpu().numpy()
s = A.OrbHipExtractor(max_features=NF, max_width=W, max_height=H)
bad = 0
for f in range(B):
    r = s.extract(host[f].numpy())
    n = int(ch[f])
    k1 = kh[f, :n].tobytes()
    k2 = np.ascontiguousarray(r["keypoints"]).tobytes()
    d1 = dh[f, :n].tobytes()
    d2 = np.ascontiguousarray(r["descriptors"]).tobytes()
    if k1 != k2 or d1 != d2:
        bad += 1
        kk1 = np.frombuffer(k1, dtype=np.float32).reshape(-1, 6) if len(k1) else np.zeros((0, 6), np.float32)
        kk2 = np.frombuffer(k2, dtype=np.float32).reshape(-1, 6) if len(k2) else np.zeros((0, 6), np.float32)
        msg = f"frame {f}: batch {n} keypoints, single {len(kk2)}"
        m = min(len(kk1), len(kk2))
        diff = np.nonzero((kk1[:m] != kk2[:m]).any(axis=1))[0]
        if len(diff):
            i = int(diff[0])
            msg += f"; first differing row {i}: batch {kk1[i][:4]} oct {kk1[i][5:6].view(np.int32)} single {kk2[i][:4]} oct {kk2[i][5:6].view(np.int32)}; {len(diff)} rows differ"
        elif d1 != d2:
            msg += "; keypoints equal, descriptors differ"
        print(msg)
print("frames differing:", bad, "of", B)
