#!/usr/bin/env python3
"""Extract + match two golden cases on the GPU and compare with tests/golden/golden.json (no oracle needed).
Used by tests/test_gpu_variants.py to check alternative kernel paths selected through environment variables."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import aria_slam_amd as A  # noqa: E402


def sha(x):
    return hashlib.sha256(np.ascontiguousarray(x).tobytes()).hexdigest()


def main():
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))
    bad = 0
    for key in ("s1_640x480_n2000", "s3_333x251_n300", "s7_752x480_n1000"):
        g = gold[key]
        s, wh, n = key.split("_")
        w, h = (int(v) for v in wh.split("x"))
        a, b = A.synth_frame_pair(int(s[1:]), w, h)
        e = A.OrbHipExtractor(max_features=int(n[1:]), max_width=w, max_height=h)
        m = A.HipMatcher()
        fa, fb = e.extract(a), e.extract(b)
        ok = (sha(fa["keypoints"]) == g["kp_a"] and sha(fa["descriptors"]) == g["desc_a"] and
              sha(fb["keypoints"]) == g["kp_b"] and sha(fb["descriptors"]) == g["desc_b"] and
              sha(m.match(fb, fa, None, 0.75)) == g["matches"])
        print(key, "OK" if ok else "MISMATCH")
        bad += 0 if ok else 1
        e.close()
        m.close()
        # the same pair through the device-resident batch entry point (its own FAST/blur kernel: the streaming one by default)
        import torch
        dev = torch.device("cuda", 0)
        st = torch.cuda.Stream(device=dev)
        e = A.OrbHipExtractor(max_features=int(n[1:]), max_width=w, max_height=h, max_batch=2, stream=st.cuda_stream)
        cap = e.kp_capacity()
        imgs = torch.from_numpy(np.stack([a, b])).to(dev)
        kps = torch.zeros((2, cap, 24), dtype=torch.uint8, device=dev)
        desc = torch.zeros((2, cap, 32), dtype=torch.uint8, device=dev)
        cnt = torch.zeros((2,), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        e.extract_batch_device(imgs, 2, w, h, kps, desc, cnt, cap)
        e.check()
        c, k, d = cnt.cpu().numpy(), kps.cpu().numpy(), desc.cpu().numpy()
        ok = (sha(k[0, :c[0]]) == g["kp_a"] and sha(d[0, :c[0]]) == g["desc_a"] and sha(k[1, :c[1]]) == g["kp_b"] and
              sha(d[1, :c[1]]) == g["desc_b"])
        print(key, "batch (%s)" % e.fast_blur_kernel(), "OK" if ok else "MISMATCH")
        bad += 0 if ok else 1
        e.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
