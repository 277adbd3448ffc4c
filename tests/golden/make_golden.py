#!/usr/bin/env python3
"""Generates tests/golden/golden.json.

There is nothing to import from the reference (its hot path is OpenCV calls and OpenCV is absent), so these
goldens are produced by THIS repo's CPU oracle on seeded synthetic frames. They pin (a) the synthetic generator,
(b) the oracle against accidental change, and (c) give the GPU tests a fixture that does not need the oracle
at run time. They do NOT pin the oracle against real OpenCV (parity unpinned, see oracle/orb_oracle.h).

Usage: python tests/golden/make_golden.py            (rewrites golden.json and golden_seed1_640x480.npz)
       python tests/golden/make_golden.py --bench    (rewrites bench_prefix.json: oracle digests of the first 256
                                                      frames of bench.py's default sequence, seeds 1..128, and of
                                                      their (frame, previous frame) matches -- what bench.py's
                                                      `verified` compares the timed batch path with)
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import aria_slam_amd as A            # noqa: E402  (generator only; no GPU needed)
from oracle import oracle_py as O    # noqa: E402


def sha(x):
    return hashlib.sha256(np.ascontiguousarray(x).tobytes()).hexdigest()


def main():
    g = {}
    a, b = A.synth_frame_pair(1, 640, 480)
    g["synth_seed1_a"], g["synth_seed1_b"] = sha(a), sha(b)
    cases = [(1, 640, 480, 2000), (2, 640, 480, 2000), (7, 752, 480, 1000), (3, 333, 251, 300), (11, 1408, 1408, 4000)]
    for seed, w, h, nf in cases:
        fa, fb = A.synth_frame_pair(seed, w, h)
        p = O.default_params(nf)
        ka, da = O.orb_extract(fa, p)
        kb, db = O.orb_extract(fb, p)
        m = O.match_ratio(db, da, 0.75)
        key = "s%d_%dx%d_n%d" % (seed, w, h, nf)
        g[key] = {"image_a": sha(fa), "image_b": sha(fb), "n_a": len(ka), "n_b": len(kb),
                  "kp_a": sha(ka), "desc_a": sha(da), "kp_b": sha(kb), "desc_b": sha(db),
                  "n_matches": len(m), "matches": sha(m),
                  "per_level_a": np.bincount(ka["octave"], minlength=8).tolist(),
                  "first_kp_a": [float(ka[0][f]) for f in ("x", "y", "size", "angle", "response")] + [int(ka[0]["octave"])],
                  "first_desc_a": da[0].tolist()}
        if (seed, w, h) == (1, 640, 480):
            np.savez_compressed(os.path.join(HERE, "golden_seed1_640x480.npz"), kp_a=ka, desc_a=da, kp_b=kb,
                                desc_b=db, matches=m)
    json.dump(g, open(os.path.join(HERE, "golden.json"), "w"), indent=1, sort_keys=True)
    print("wrote", os.path.join(HERE, "golden.json"))


def _bench_pair(args):
    seed, w, h, nf = args
    a, b = A.synth_frame_pair(seed, w, h)
    p = O.default_params(nf)
    ka, da = O.orb_extract(a, p)
    kb, db = O.orb_extract(b, p)
    return seed, (ka, da), (kb, db)


def bench_prefix(n_frames=256, w=640, h=480, nf=2000, ratio=0.75):
    """Digests exactly as bench.py's frame_digest / match_digest compute them (first 16 hex chars kept)."""
    import multiprocessing as mp
    sys.path.insert(0, ROOT)
    import bench as B
    with mp.get_context("fork").Pool(min(8, os.cpu_count() or 1)) as pool:
        res = sorted(pool.map(_bench_pair, [(1 + s, w, h, nf) for s in range(n_frames // 2)]))
    frames = []
    for _, fa, fb in res:
        frames += [fa, fb]
    fd = [B.frame_digest(len(k), k, d)[:16] for k, d in frames]
    md = [""]
    for f in range(1, n_frames):
        m = O.match_ratio(frames[f][1], frames[f - 1][1], ratio)
        md.append(B.match_digest(len(m), m)[:16])
    path = os.path.join(HERE, "bench_prefix.json")
    g = json.load(open(path)) if os.path.exists(path) else {}
    key = "%dx%d_n%d" % (w, h, nf)
    full = g.get(key, {}).get("full", {})
    g[key] = {"ratio": ratio, "seed0": 1, "frames": n_frames, "frame": fd, "match": md, "full": full,
              "made_by": "tests/golden/make_golden.py --bench (oracle/orb_oracle.cpp); 'full' = checksum of checksums "
                         "of the whole default sequence as printed by bench.py on the GPU (HIP path whose prefix "
                         "equals these oracle digests)"}
    json.dump(g, open(path, "w"), indent=0, sort_keys=True)
    print("wrote", path)


if __name__ == "__main__":
    if "--bench" in sys.argv:
        bench_prefix()
    else:
        main()
