#!/usr/bin/env python3
# k_select phase stamps (ARIA_SEL_STAMPS=1) for one frame at a time and for a 256-frame pass
import os, sys
os.environ["ARIA_SEL_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import aria_slam_amd as A
seq = A.synth_sequence(1, 256, 640, 480)
e = A.OrbHipExtractor(max_features=2000, max_width=640, max_height=480, max_batch=256)
for i in range(3):
    e.extract(seq[i])
e.close()
print("-- batch of 2048", file=sys.stderr)
import torch
B = 2048
host = torch.empty((B, 480, 640), dtype=torch.uint8).pin_memory()
A.synth_sequence(1, B // 2, 640, 480, out=host.numpy())
img = host.cuda()
e = A.OrbHipExtractor(max_features=2000, max_width=640, max_height=480, max_batch=B)
cap = e.kp_capacity()
kps = torch.zeros((B, cap, 24), dtype=torch.uint8, device="cuda")
desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
cnt = torch.zeros(B, dtype=torch.int32, device="cuda")
for _ in range(2):
    e.extract_batch_device(img.data_ptr(), B, 640, 480, kps.data_ptr(), desc.data_ptr(), cnt.data_ptr(), cap)
    torch.cuda.synchronize()
e.close()
