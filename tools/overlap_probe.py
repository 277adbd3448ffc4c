#!/usr/bin/env python3
"""How much would pipelining the matcher of step s beside the extractor of step s+1 buy? ("next".)
Same workload as bench.py (one 8192-frame pass per step), no profiling brackets. Serial: both handles on one stream.
Pipelined: extractor on stream E, matcher on stream M, two sets of output buffers; M waits for E's "set b described"
event, E waits for M's "set b matched" event before overwriting set b two steps later. Results are checked to be the
same in both modes."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import aria_slam_amd as A  # noqa: E402


def main():
    W, H, NF, pairs, steps = 640, 480, 2000, 4096, 4
    B = 2 * pairs
    dev = torch.device("cuda", 0)
    host = torch.empty((B, H, W), dtype=torch.uint8)
    A.synth_sequence(1, pairs, W, H, out=host.numpy())
    images = host.to(dev)
    del host
    se, sm = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    out = {}
    for mode in ("serial", "pipelined"):
        ext = A.OrbHipExtractor(max_features=NF, stream=se.cuda_stream, max_width=W, max_height=H, max_batch=B)
        mat = A.HipMatcher(stream=(se if mode == "serial" else sm).cuda_stream)
        cap = ext.kp_capacity()
        sets = []
        for _ in range(2):
            sets.append(dict(kps=torch.empty((B, cap, 24), dtype=torch.uint8, device=dev),
                             desc=torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev),
                             counts=torch.zeros((B,), dtype=torch.int32, device=dev),
                             matches=torch.empty((B, cap, 12), dtype=torch.uint8, device=dev),
                             nm=torch.zeros((B,), dtype=torch.int32, device=dev),
                             described=torch.cuda.Event(), matched=torch.cuda.Event()))
        torch.cuda.synchronize()

        def step(s):
            d = sets[s & 1]
            with torch.cuda.stream(se):
                if mode == "pipelined":
                    se.wait_event(d["matched"])          # set b is free again (recorded two steps ago; no-op at first)
                ext.extract_batch_device(images, B, W, H, d["kps"], d["desc"], d["counts"], cap)
                d["described"].record(se)
            ms = se if mode == "serial" else sm
            with torch.cuda.stream(ms):
                if mode == "pipelined":
                    sm.wait_event(d["described"])
                mat.match_batch_device(d["desc"].data_ptr() + cap * 32, d["counts"].data_ptr() + 4, d["desc"], d["counts"],
                                       B - 1, cap * 32, 0.75, d["matches"].data_ptr() + cap * 12, d["nm"].data_ptr() + 4, cap)
                d["matched"].record(ms)

        step(0)
        step(1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in range(steps):
            step(s)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ext.check()
        mat.sync()
        out[mode] = (B * steps / dt, sets[(steps - 1) & 1]["nm"].cpu().numpy().copy(), sets[(steps - 1) & 1]["counts"].cpu().numpy().copy())
        print("%-9s: %.0f frames/s" % (mode, out[mode][0]))
        ext.close()
        mat.close()
    same = np.array_equal(out["serial"][1], out["pipelined"][1]) and np.array_equal(out["serial"][2], out["pipelined"][2])
    print("results identical:", same, " gain: %.1f %%" % (100.0 * (out["pipelined"][0] / out["serial"][0] - 1.0)))
    sys.exit(0 if same else 1)


if __name__ == "__main__":
    main()
