#!/usr/bin/env python3
"""Where do the two readings of cv::ORB's pyramid level size differ?

    mode 0 (this repository's default):  cvRound(dim * (1.0f / scale))     -- reciprocal, then a float multiply
    mode 1 (SURVEY.md A.1's wording):    cvRound(dim / scale)              -- one float division

scale = (float)pow((double)1.2f, level), dim an image width or height, all arithmetic in float32 as in orb.cpp, cvRound =
round half to even. The sweep covers every dim in [64, 2047] (the library's size range, kMaxDim) and levels 1..7; level 0
never differs. Output: the list of (dim, level) where the two differ, and whether any size named in BASELINE.json is hit.
CPU only; no oracle, no GPU. Run:  python tools/level_size_sweep.py > profiles/level_size_sweep.txt
"""
import numpy as np

BASELINE_DIMS = {640: "640x480 (configs[1], [2])", 480: "640x480 / 752x480", 752: "752x480 (configs[0], [4], EuRoC)",
                 1408: "1408x1408 (configs[3], Aria RGB)"}


def cv_round(x):
    return int(np.rint(np.float32(x)))      # rint = round half to even, like cvRound (lrint / cvtss2si)


def main():
    f32 = np.float32
    scales = [f32(np.float64(f32(1.2)) ** l) for l in range(8)]
    diffs = []
    for level in range(1, 8):
        s = scales[level]
        inv = f32(1.0) / s
        for dim in range(64, 2048):
            a = cv_round(f32(dim) * inv)
            b = cv_round(f32(dim) / s)
            if a != b:
                diffs.append((dim, level, a, b))
    print("# tools/level_size_sweep.py: cvRound(dim * (1.0f / scale)) vs cvRound(dim / scale), dim in [64, 2047], levels 1..7")
    print("# %d of %d (dim, level) combinations differ" % (len(diffs), 7 * (2048 - 64)))
    print("# dim level  mode0  mode1")
    for d in diffs:
        print("%5d %5d %6d %6d" % d)
    hit = [d for d in diffs if d[0] in BASELINE_DIMS]
    print("# BASELINE.json sizes affected: %s" % (", ".join("%s at level %d" % (BASELINE_DIMS[d[0]], d[1]) for d in hit) or "none"))


if __name__ == "__main__":
    main()
