#!/bin/bash
# SQ counters of the matcher/extractor kernels (issue/wait/matrix-pipe breakdown), a few counters per pass.
# Usage (on the GPU box): tools/pmc_sq.sh   -> gpurun_out/pmc_sq_<n>/ + gpurun_out/pmc_sq.txt
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
n=0
: > $R/gpurun_out/pmc_sq.txt
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM" "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_INSTS_FLAT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE"; do
  n=$((n+1)); rm -rf $R/gpurun_out/pmc_sq_$n
  timeout -k 5 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq_$n -- python3 $R/tools/prof_extract.py --pairs 512 --iters 2 --chunk 1024 --match > $R/gpurun_out/pmc_sq_$n.log 2>&1 || { echo "pass $n ($set) failed"; tail -3 $R/gpurun_out/pmc_sq_$n.log; continue; }
  python3 - "$R/gpurun_out/pmc_sq_$n" >> $R/gpurun_out/pmc_sq.txt <<'PY'
import glob, os, sys
import pandas as pd
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1]
df = pd.read_csv(f)
df["k"] = df["Kernel_Name"].str.extract(r"(k_\w+)")[0]
t = df.dropna(subset=["k"]).groupby(["k", "Counter_Name"]).Counter_Value.agg(["sum", "count"])
print(t.to_string())
PY
done
cat $R/gpurun_out/pmc_sq.txt
