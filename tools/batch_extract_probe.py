#!/usr/bin/env python3
# band kernel phase stamps (ARIA_STAMPS=<level>) for a batch of frames at a given size
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import aria_slam_amd as A
W, H, NF, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
host = torch.empty((B, H, W), dtype=torch.uint8).pin_memory()
A.synth_sequence(1, B // 2, W, H, out=host.numpy())
img = host.cuda()
e = A.OrbHipExtractor(max_features=NF, max_width=W, max_height=H, max_batch=B)
cap = e.kp_capacity()
kps = torch.zeros((B, cap, 24), dtype=torch.uint8, device="cuda")
desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
cnt = torch.zeros(B, dtype=torch.int32, device="cuda")
for _ in range(2):
    e.extract_batch_device(img.data_ptr(), B, W, H, kps.data_ptr(), desc.data_ptr(), cnt.data_ptr(), cap)
    torch.cuda.synchronize()
e.close()
