#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry points (IFeatureExtractor::extract + IMatcher::match semantics):
one frame in flight, pageable host buffers in and out, stream sync per call. Times the C-ABI calls themselves
(preallocated outputs), i.e. what the C++ adapters see."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aria_slam_amd as A  # noqa: E402


def main():
    W, H, NF = 640, 480, 2000
    seq = A.synth_sequence(1, 32, W, H)
    e = A.OrbHipExtractor(max_features=NF, max_width=W, max_height=H)
    m = A.HipMatcher()
    L = e._L
    cap = e.kp_capacity()
    kp = [np.empty(cap, A.KP_DTYPE) for _ in range(2)]
    ds = [np.empty((cap, 32), np.uint8) for _ in range(2)]
    mt = np.empty(cap, A.MATCH_DTYPE)
    n = [C.c_int(), C.c_int()]
    nm = C.c_int()

    def ext(i, slot):
        rc = L.aria_orb_extract(e._h, seq[i].ctypes.data, W, H, W, kp[slot].ctypes.data, ds[slot].ctypes.data, cap, C.byref(n[slot]))
        assert rc == 0

    for i in range(8):
        ext(i, 0)
    reps = 20
    t0 = time.perf_counter()
    for rep in range(reps):
        for i in range(len(seq)):
            ext(i, 0)
    t1 = time.perf_counter()
    ext(0, 1)
    tm = 0.0
    for rep in range(reps // 2):
        for i in range(len(seq)):
            s = i & 1
            ext(i, s)
            ta = time.perf_counter()
            rc = L.aria_matcher_match(m._h, ds[s].ctypes.data, n[s].value, ds[1 - s].ctypes.data, n[1 - s].value,
                                      C.c_float(0.75), mt.ctypes.data, cap, C.byref(nm))
            tm += time.perf_counter() - ta
            assert rc == 0
    t2 = time.perf_counter()
    N1, N2 = reps * len(seq), (reps // 2) * len(seq)
    print("aria_orb_extract (host buffers)           : %.1f us/frame = %.0f frames/s" % (1e6 * (t1 - t0) / N1, N1 / (t1 - t0)))
    print("aria_orb_extract + aria_matcher_match     : %.1f us/frame = %.0f frames/s (match calls alone %.1f us, %d..%d keypoints)"
          % (1e6 * (t2 - t1) / N2, N2 / (t2 - t1), 1e6 * tm / N2, min(n[0].value, n[1].value), max(n[0].value, n[1].value)))
    # device hand-off (getGpuDescriptors / matchGpu): (a) two waits per frame, own streams; (b) the match queued behind
    # extractAsync on ONE stream, one wait per frame -- with the host time of each of the four calls
    import torch
    dev = torch.device("cuda", 0)
    for shared in (False, True):
        st = torch.cuda.Stream(device=dev).cuda_stream if shared else None
        e2 = A.OrbHipExtractor(max_features=NF, max_width=W, max_height=H, stream=st)
        m2 = A.HipMatcher(stream=st)
        rc = L.aria_orb_extract(e2._h, seq[0].ctypes.data, W, H, W, kp[1].ctypes.data, ds[1].ctypes.data, cap, C.byref(n[1]))
        assert rc == 0
        _, d_desc, d_cnt, n0, rows = e2.device_result()
        m2.retain_device(d_desc, n0)
        seg = [0.0] * 4
        for rep in range(reps // 2 + 1):
            if rep == 1:
                t3 = time.perf_counter()
                seg = [0.0] * 4
            for i in range(len(seq)):
                s = i & 1
                if shared:
                    a = time.perf_counter()
                    rc = L.aria_orb_extract_async(e2._h, seq[i].ctypes.data, W, H, W)
                    b = time.perf_counter()
                    rc |= L.aria_matcher_match_device_async(m2._h, d_desc, d_cnt, rows, 1, C.c_float(0.75))
                    c = time.perf_counter()
                    rc |= L.aria_orb_sync(e2._h, kp[s].ctypes.data, ds[s].ctypes.data, cap, C.byref(n[s]))
                    d = time.perf_counter()
                    rc |= L.aria_matcher_finish(m2._h, n[s].value, mt.ctypes.data, cap, C.byref(nm))
                    f = time.perf_counter()
                    seg = [seg[0] + b - a, seg[1] + c - b, seg[2] + d - c, seg[3] + f - d]
                else:
                    rc = L.aria_orb_extract(e2._h, seq[i].ctypes.data, W, H, W, kp[s].ctypes.data, ds[s].ctypes.data, cap, C.byref(n[s]))
                    a = time.perf_counter()
                    rc |= L.aria_matcher_match_device(m2._h, d_desc, n[s].value, None, n[1 - s].value, C.c_float(0.75), mt.ctypes.data, cap, C.byref(nm))
                    seg[0] += time.perf_counter() - a
                assert rc == 0, rc
        t4 = time.perf_counter()
        if shared:
            print("extract_async + match_device_async + sync + finish (one stream): %.1f us/frame; calls: extract_async %.1f, "
                  "match_async %.1f, orb_sync %.1f, finish %.1f us" % (1e6 * (t4 - t3) / N2, *[1e6 * x / N2 for x in seg]))
        else:
            print("aria_orb_extract + aria_matcher_match_device (own streams)      : %.1f us/frame (match calls alone %.1f us)"
                  % (1e6 * (t4 - t3) / N2, 1e6 * seg[0] / N2))
        e2.close()
        m2.close()
    # GPU-side stage times of one frame in flight (drained HIP-event brackets: adds host time, so not a rate)
    e.set_profiling(True)
    m.set_profiling(True)
    for i in range(len(seq)):
        s = i & 1
        ext(i, s)
        L.aria_matcher_match(m._h, ds[s].ctypes.data, n[s].value, ds[1 - s].ctypes.data, n[1 - s].value,
                             C.c_float(0.75), mt.ctypes.data, cap, C.byref(nm))
    pe, fr = e.get_profile()
    pm, pr = m.get_profile()
    print("single-frame GPU stage times (us/frame): " +
          ", ".join("%s %.1f" % (k, 1e3 * v[0] / max(fr, 1)) for k, v in pe.items()) + " | " +
          ", ".join("%s %.1f" % (k, 1e3 * v[0] / max(pr, 1)) for k, v in pm.items()))


if __name__ == "__main__":
    main()
