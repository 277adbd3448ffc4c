#!/bin/bash
# Round 2: SQ counters of the FAST/blur kernels (and the rest of the extractor) for one build/environment.
# Usage (GPU box): tools/pmc_sq_r2.sh <tag> [ENV=VAL ...]   -> gpurun_out/pmc_sq_<tag>.txt (per-kernel sums over 2048 frames)
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
tag=$1; shift
for kv in "$@"; do export "$kv"; done
out=$R/gpurun_out/pmc_sq_$tag.txt
echo "# rocprofv3 --pmc, separate passes; tools/prof_extract.py --pairs 1024 --iters 1 --chunk 1024 (2048 frames, 640x480/2000); env: $*" > $out
n=0
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES" "SQ_INSTS_LDS SQ_WAIT_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES"; do
  n=$((n+1)); rm -rf $R/gpurun_out/pmc_r2_$tag_$n
  timeout -k 5 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_r2_${tag}_$n -- python3 $R/tools/prof_extract.py --pairs 1024 --iters 1 --chunk 1024 > $R/gpurun_out/pmc_r2_${tag}_$n.log 2>&1 || { echo "pass $n ($set) failed" >> $out; tail -3 $R/gpurun_out/pmc_r2_${tag}_$n.log; continue; }
  python3 - "$R/gpurun_out/pmc_r2_${tag}_$n" >> $out <<'PY'
import glob, os, sys
import pandas as pd
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1]
df = pd.read_csv(f)
df["k"] = df["Kernel_Name"].str.extract(r"(k_\w+)")[0]
df = df.dropna(subset=["k"])
df["dur_us"] = (df["End_Timestamp"] - df["Start_Timestamp"]) / 1e3 if "End_Timestamp" in df else 0.0
t = df.groupby(["k", "Counter_Name"]).agg(sum=("Counter_Value", "sum"), dispatches=("Counter_Value", "count"))
print(t.to_string())
PY
done
cat $out
