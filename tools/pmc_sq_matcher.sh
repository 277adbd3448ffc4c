#!/bin/bash
# SQ counters of the matcher kernel alone (k_knn2_mfma over 2048 pairs of 2000 x 2000 descriptors), separate --pmc passes.
# Usage (GPU box): tools/pmc_sq_matcher.sh <tag>   -> gpurun_out/pmc_sq_matcher_<tag>.txt
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
tag=$1; out=$R/gpurun_out/pmc_sq_matcher_$tag.txt
echo "# rocprofv3 --pmc, separate passes; tools/prof_extract.py --pairs 1024 --iters 1 --chunk 1024 --match (2047 pairs, 2000 x 2000 x 256 bit)" > $out
n=0
for set in "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES" "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_INSTS_SALU SQ_WAVES"; do
  n=$((n+1)); rm -rf $R/gpurun_out/pmc_m_${tag}_$n
  timeout -k 5 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_m_${tag}_$n -- python3 $R/tools/prof_extract.py --pairs 1024 --iters 1 --chunk 1024 --match > $R/gpurun_out/pmc_m_${tag}_$n.log 2>&1 || { echo "pass $n ($set) failed" >> $out; tail -3 $R/gpurun_out/pmc_m_${tag}_$n.log >> $out; continue; }
  python3 - "$R/gpurun_out/pmc_m_${tag}_$n" >> $out <<'PY'
import glob, os, sys
import pandas as pd
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1]
df = pd.read_csv(f)
df["k"] = df["Kernel_Name"].str.extract(r"(k_\w+)")[0]
df = df[df.k.isin(["k_knn2_mfma", "k_knn2_fp4", "k_ratio_compact"])]
t = df.groupby(["k", "Counter_Name"]).agg(sum=("Counter_Value", "sum"), dispatches=("Counter_Value", "count"))
print(t.to_string())
PY
done
cat $out
