#!/usr/bin/env python3
"""Reduce the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_traffic.sh to HBM bytes per frame per kernel."""
import glob
import json
import os
import sys

import pandas as pd


def load(d):
    f = sorted(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1]
    df = pd.read_csv(f)
    df["k"] = df["Kernel_Name"].str.extract(r"(k_\w+)")[0]
    return df


def main():
    root, chunk = sys.argv[1], int(sys.argv[2])
    fe, wr = load(os.path.join(root, "pmc_FETCH_SIZE")), load(os.path.join(root, "pmc_WRITE_SIZE"))
    frames = int(sys.argv[3]) if len(sys.argv) > 3 else 1024 * 3      # prof_extract.py --pairs 512 --iters 3
    # calibration: the 1 GiB torch copy (elementwise copy kernel, 16 B per lane) must fetch 2^30 bytes
    cal = fe[fe.Kernel_Name.str.contains("elementwise|copy|Copy", regex=True) & ~fe.Kernel_Name.str.contains("calib_copy_dword") & (fe.Counter_Value > 100000)]
    if not len(cal):
        sys.exit("pmc_traffic.py: the 1 GiB calibration copy (prof_extract.py --calibrate) is not in the FETCH_SIZE pass: "
                 "refusing to guess the gfx950 FETCH_SIZE correction factor")
    cal_kb = float(cal.Counter_Value.max())
    factor = (2 ** 30 / 1024.0) / cal_kb
    out = {"chunk": chunk, "frames": frames, "fetch_calibration": {"copy_bytes": 2 ** 30, "FETCH_SIZE_kb_reported": cal_kb, "factor": round(factor, 3)},
           "kernels": {}}
    # one-dword-per-lane loads (k_fast_blur_stream's rows): their own factor, from the dword calibration copy when it ran
    cal4 = fe[fe.Kernel_Name.str.contains("calib_copy_dword")]
    factor4 = None
    if len(cal4):
        factor4 = (2 ** 30 / 1024.0) / float(cal4.Counter_Value.max())
        w4 = wr[wr.Kernel_Name.str.contains("calib_copy_dword")]
        out["fetch_calibration_dword_loads"] = {"copy_bytes": 2 ** 30, "FETCH_SIZE_kb_reported": float(cal4.Counter_Value.max()),
                                                "factor": round(factor4, 3),
                                                "WRITE_SIZE_kb_reported": float(w4.Counter_Value.max()) if len(w4) else None}
    for k in sorted(set(fe.k.dropna())):
        fkb = fe[fe.k == k].Counter_Value.sum()
        wkb = wr[wr.k == k].Counter_Value.sum()
        n = frames if k != "k_ratio_compact" else frames
        if k == "k_calib_copy_dword":
            continue
        kf = factor4 if (k == "k_fast_blur_stream" and factor4) else factor
        out["kernels"][k] = {"fetch_bytes_per_frame_raw": round(fkb * 1024 / n), "fetch_bytes_per_frame": round(fkb * 1024 * kf / n),
                             "write_bytes_per_frame": round(wkb * 1024 / n), "fetch_factor": round(kf, 3)}
    # the build these counters belong to: sha256 of the FAST/blur kernels' source files (bench.py prints traffic_stale when
    # the tree's differ) and, where a checkout is present, HEAD
    import hashlib
    import subprocess
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out["kernel_source_sha256"] = {}
    for k, f in (("k_fast_blur_stream", "fast_blur_stream.hip"), ("k_fast_blur_band", "fast_blur_band.hip")):
        path = os.path.join(repo, "aria_slam_amd", "csrc", f)
        if os.path.isfile(path):
            out["kernel_source_sha256"][k] = hashlib.sha256(open(path, "rb").read()).hexdigest()
    try:
        out["git_head"] = subprocess.check_output(["git", "-C", repo, "rev-parse", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        out["git_head"] = None          # the GPU box gets a snapshot without .git
    dst = os.path.join(root, "pmc_traffic_%d.json" % chunk)
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
