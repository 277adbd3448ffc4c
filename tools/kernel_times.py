#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per-frame MEDIAN kernel time (warm-up outliers dominate means)."""
import glob
import os
import sys

import pandas as pd


def main():
    d, frames_per_pass = sys.argv[1], int(sys.argv[2])
    f = sorted(glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]
    df = pd.read_csv(f)
    df["dur"] = (df.End_Timestamp - df.Start_Timestamp) / 1e3
    df["k"] = df["Kernel_Name"].str.extract(r"(k_\w+)")[0].fillna("other")
    df = df[df.k != "other"]
    g = df.groupby(["k", "Grid_Size_X", "Grid_Size_Y", "Workgroup_Size_X"]).dur.agg(["count", "median", "min", "mean"])
    per = g.groupby("k")[["median", "min", "mean"]].sum() / frames_per_pass
    print(per.round(3).to_string())
    print("sum of medians: %.3f us/frame" % per["median"].sum())


if __name__ == "__main__":
    main()
