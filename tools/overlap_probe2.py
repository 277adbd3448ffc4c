#!/usr/bin/env python3
"""Probe: one extractor handle over 8192 frames vs two handles (own scratch, own stream) over 4096 frames each, running
concurrently; matcher excluded. Prints frames/s of both."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import aria_slam_amd as A  # noqa: E402

def main():
    W, H, NF, B, steps = 640, 480, 2000, 8192, 4
    dev = torch.device("cuda", 0)
    host = torch.empty((B, H, W), dtype=torch.uint8)
    A.synth_sequence(1, B // 2, W, H, out=host.numpy())
    images = host.to(dev); del host
    for nsplit in (1, 2, 4):
        streams = [torch.cuda.Stream(device=dev) for _ in range(nsplit)]
        n = B // nsplit
        exts = [A.OrbHipExtractor(max_features=NF, stream=s.cuda_stream, max_width=W, max_height=H, max_batch=n) for s in streams]
        cap = exts[0].kp_capacity()
        kps = torch.empty((B, cap, 24), dtype=torch.uint8, device=dev)
        desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
        counts = torch.zeros((B,), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        def step():
            for i, e in enumerate(exts):
                e.extract_batch_device(images.data_ptr() + i * n * W * H, n, W, H, kps.data_ptr() + i * n * cap * 24,
                                       desc.data_ptr() + i * n * cap * 32, counts.data_ptr() + i * n * 4, cap)
        step(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps): step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        for e in exts: e.check()
        print("%d handle(s): %.0f frames/s (extract only), checksum %d" % (nsplit, B * steps / dt, int(desc.sum(dtype=torch.int64).item())))
        for e in exts: e.close()

if __name__ == "__main__":
    main()
