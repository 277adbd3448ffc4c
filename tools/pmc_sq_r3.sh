#!/bin/bash
# Round 3: SQ counters of the extractor kernels for one build/environment at a batch large enough to fill the chip.
# Usage (GPU box): tools/pmc_sq_r3.sh <tag> [ENV=VAL ...]   -> gpurun_out/pmc_sq_<tag>.txt (per-kernel sums over 4096 frames, 640x480/2000)
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
tag=$1; shift
for kv in "$@"; do export "$kv"; done
out=$R/gpurun_out/pmc_sq_$tag.txt
echo "# rocprofv3 --pmc, separate passes; tools/prof_extract.py --pairs 2048 --iters 1 --chunk 4096 (4096 frames, 640x480/2000); env: $*" > $out
n=0
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM"; do
  n=$((n+1)); rm -rf $R/gpurun_out/pmc_r3_${tag}_$n
  timeout -k 5 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_r3_${tag}_$n -- python3 $R/tools/prof_extract.py --pairs 2048 --iters 1 --chunk 4096 > $R/gpurun_out/pmc_r3_${tag}_$n.log 2>&1 || { echo "pass $n ($set) failed" >> $out; tail -3 $R/gpurun_out/pmc_r3_${tag}_$n.log; continue; }
  python3 - "$R/gpurun_out/pmc_r3_${tag}_$n" >> $out <<'PY'
import glob, os, sys
import pandas as pd
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1]
df = pd.read_csv(f)
df["k"] = df["Kernel_Name"].str.extract(r"(k_\w+)")[0]
df = df.dropna(subset=["k"])
t = df.groupby(["k", "Counter_Name"]).agg(sum=("Counter_Value", "sum"), dispatches=("Counter_Value", "count"))
print(t.to_string())
PY
  rm -rf $R/gpurun_out/pmc_r3_${tag}_$n
done
cat $out
