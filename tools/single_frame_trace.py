#!/usr/bin/env python3
"""Timeline of ONE single-frame aria_orb_extract (hipGraph replay): run under
   rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 tools/single_frame_trace.py run
then   python3 tools/single_frame_trace.py show <dir>   prints start offset / duration of every kernel of the last call."""
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import aria_slam_amd as A
    seq = A.synth_sequence(1, 4, 640, 480)
    e = A.OrbHipExtractor(max_features=2000, max_width=640, max_height=480)
    for rep in range(6):
        for img in seq:
            e.extract(img)
    e.close()


def show(d):
    import pandas as pd
    f = sorted(glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]
    df = pd.read_csv(f).sort_values("Start_Timestamp")
    df["k"] = df["Kernel_Name"].str.extract(r"(k_\w+)")[0].fillna(df["Kernel_Name"].str.slice(0, 30))
    # the last call = the kernels after the last k_describe-before-last ... take the trailing kernels up to 2 k_describe back
    idx = df.index[df.k == "k_describe"].tolist()
    last_start = df.index.get_loc(idx[-3]) + 1 if len(idx) >= 3 else 0
    sub = df.iloc[last_start:]
    t0 = sub.Start_Timestamp.min()
    for _, r in sub.iterrows():
        print("%-22s grid %6d x %-5d start %8.1f us  dur %7.1f us" % (r.k, r.Grid_Size_X // max(r.Workgroup_Size_X, 1), r.Grid_Size_Y,
                                                                    (r.Start_Timestamp - t0) / 1e3, (r.End_Timestamp - r.Start_Timestamp) / 1e3))
    print("span %.1f us" % ((sub.End_Timestamp.max() - t0) / 1e3))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        show(sys.argv[2])
