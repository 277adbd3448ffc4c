#!/usr/bin/env python3
"""Per-level table of the batch FAST/blur launches from a rocprofv3 kernel trace (rocpd .db or *_kernel_trace.csv).

usage: level_times.py <trace> <W> <H> <frames per launch> [nfeatures]
A pass of the extractor launches the FAST/blur kernel once per pyramid level, in level order. Every run of 8 consecutive
batch launches (k_fast_blur_stream, or k_fast_blur_band with frames-per-launch rows in its grid) is one pass; the table
lists, per level, the MEDIAN over the passes: us per launch, us per Mpx, and for the streaming kernel the waves and the share
of lanes that carry pixels. The last lines give the sum over the levels (= us per frame of the stage), its mean per launch
(= bench.py's roofline.avg_launch_ms) and the algorithmic-byte rate (2P + (P - p0)) / time against the 8 TB/s peak
(= roofline.achieved, roofline.frac).
"""
import csv
import sqlite3
import statistics
import sys


def launches(path):
    if path.endswith(".db"):
        c = sqlite3.connect(path)
        tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
        kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
        ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol_")][0]
        q = f"select s.kernel_name, d.grid_size_x, d.grid_size_y, d.workgroup_size_x, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"
        for r in c.execute(q):
            yield r[0], int(r[1]), int(r[2]), int(r[3]), int(r[4]), int(r[5])
    else:
        rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
        for r in rows:
            yield (r["Kernel_Name"], int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0), int(r.get("Grid_Size_Y", 1) or 1),
                   int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 0)) or 0), int(r["Start_Timestamp"]), int(r["End_Timestamp"]))


def level_dims(W, H):
    import numpy as np
    f32 = np.float32
    out = []
    for l in range(8):
        inv = f32(1.0) / f32(np.float64(f32(1.2)) ** l)
        out.append((int(np.rint(f32(W) * inv)), int(np.rint(f32(H) * inv))))
    return out


def main():
    path, W, H, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    dims = level_dims(W, H)
    ev = []
    for name, gx, gy, wg, t0, t1 in launches(path):
        if "k_fast_blur_stream" in name:
            ev.append((name, gx, gy, wg, t0, t1))
        elif "k_fast_blur_band" in name and gy == B:
            ev.append((name, gx, gy, wg, t0, t1))
    # passes = runs of 8 consecutive launches (level order). The streaming kernel is launched by batches only; the band
    # kernel's batch launches were selected by their grid (frames in y)
    passes = [ev[i:i + 8] for i in range(0, len(ev) - len(ev) % 8, 8)]
    if passes and "stream" in passes[0][0][0]:
        U0 = (dims[0][0] + 3) // 4 + 2
        passes = [p for p in passes if (p[0][1] // 64) * 62 >= B * U0 * 0.9 and (p[0][1] // 64) * 62 < B * U0 * 40]
    if not passes:
        print("no batch FAST/blur passes of %d frames found" % B)
        return 1
    kern = "k_fast_blur_stream" if "stream" in passes[0][0][0] else "k_fast_blur_band"
    print("# %s, %dx%d, %d frames per launch, %d passes (median per level)" % (kern, W, H, B, len(passes)))
    print("# level  size        us/launch   us/Mpx   waves    lanes with pixels")
    tot = 0.0
    P = sum(w * h for w, h in dims)
    for l in range(8):
        us = statistics.median((p[l][5] - p[l][4]) / 1e3 for p in passes)
        lw, lh = dims[l]
        waves = passes[0][l][1] * passes[0][l][2] // 64
        util = ""
        if kern == "k_fast_blur_stream":
            D = (lw + 3) // 4
            util = "%.1f %%" % (100.0 * D / (D + 2) * 62 / 64)
        print("L%d     %4dx%-4d  %10.1f  %7.3f  %7d   %s" % (l, lw, lh, us, us / (lw * lh * B / 1e6), waves, util))
        tot += us
    alg = (2 * P + (P - dims[0][0] * dims[0][1])) * B
    print("# sum over levels %.1f us = %.3f us per frame; mean per launch %.4f ms" % (tot, tot / B, tot / 8 / 1e3))
    print("# algorithmic bytes per pass (2P + (P - p0)) x frames = %d -> %.1f GB/s = %.4f of 8000 GB/s" % (alg, alg / (tot * 1e-6) / 1e9, alg / (tot * 1e-6) / 1e9 / 8000.0))
    return 0


if __name__ == "__main__":
    sys.exit(main())
