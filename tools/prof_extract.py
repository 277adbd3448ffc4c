#!/usr/bin/env python3
"""Small driver for rocprofv3 runs: extract (and optionally match) a resident batch a few times."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=256)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--features", type=int, default=2000)
    ap.add_argument("--chunk", type=int, default=256)
    ap.add_argument("--match", action="store_true")
    ap.add_argument("--calibrate", action="store_true", help="also run a 1 GiB device copy (PMC FETCH_SIZE calibration)")
    a = ap.parse_args()
    import torch
    import aria_slam_amd as A
    dev = torch.device("cuda", 0)
    B = 2 * a.pairs
    host = torch.empty((B, a.height, a.width), dtype=torch.uint8)
    A.synth_sequence(1, a.pairs, a.width, a.height, out=host.numpy())
    images = host.to(dev)
    work = torch.cuda.Stream(device=dev)          # one real stream for both handles (handle 0 would give each its own)
    torch.cuda.set_stream(work)
    stream = work.cuda_stream
    ext = A.OrbHipExtractor(max_features=a.features, stream=stream, max_width=a.width, max_height=a.height, max_batch=a.chunk)
    mat = A.HipMatcher(stream=stream)
    cap = ext.kp_capacity()
    kps = torch.empty((B, cap, 24), dtype=torch.uint8, device=dev)
    desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    counts = torch.zeros((B,), dtype=torch.int32, device=dev)
    matches = torch.empty((B, cap, 12), dtype=torch.uint8, device=dev)
    nm = torch.zeros((B,), dtype=torch.int32, device=dev)
    for _ in range(a.iters):
        ext.extract_batch_device(images, B, a.width, a.height, kps, desc, counts, cap)
        if a.match:
            mat.match_batch_device(desc.data_ptr() + cap * 32, counts.data_ptr() + 4, desc, counts, B - 1, cap * 32, 0.75,
                                   matches.data_ptr() + cap * 12, nm.data_ptr() + 4, cap)
    if a.calibrate:
        src = torch.empty(2 ** 30, dtype=torch.uint8, device=dev).fill_(1)
        dst = torch.empty_like(src)
        torch.cuda.synchronize()
        dst.copy_(src)
        # the same gigabyte once more with ONE DWORD per lane (k_fast_blur_stream's row loads): tools/microbench/copy_dword.hip,
        # built by tools/pmc_traffic.sh next to this script's output
        so = os.environ.get("ARIA_CALIB_COPY_SO")
        if so and os.path.exists(so):
            import ctypes
            lib = ctypes.CDLL(so)
            lib.calib_copy_dword.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
            torch.cuda.synchronize()
            assert lib.calib_copy_dword(src.data_ptr(), dst.data_ptr(), src.numel(), stream) == 0
    torch.cuda.synchronize()
    ext.check()
    mat.sync()
    print("ok", int(counts.sum().item()))


if __name__ == "__main__":
    main()
