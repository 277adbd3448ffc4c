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
