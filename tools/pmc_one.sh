#!/bin/bash
# One counter pass over the extractor + matcher kernels: tools/pmc_one.sh "CTR1 CTR2 ..."  -> per-kernel sums on stdout
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_one
timeout -k 5 200 rocprofv3 --pmc $1 --kernel-trace --output-format csv -d $R/gpurun_out/pmc_one -- python3 $R/tools/prof_extract.py --pairs 512 --iters 2 --chunk 1024 --match > $R/gpurun_out/pmc_one.log 2>&1 || { echo "pass failed"; tail -3 $R/gpurun_out/pmc_one.log; exit 1; }
python3 - "$R/gpurun_out/pmc_one" <<'PY'
import glob, os, sys
import pandas as pd
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1]
df = pd.read_csv(f)
df["k"] = df["Kernel_Name"].str.extract(r"(k_\w+)")[0]
t = df.dropna(subset=["k"]).groupby(["k", "Counter_Name"]).Counter_Value.agg(["sum", "count"])
print(t.to_string())
PY
