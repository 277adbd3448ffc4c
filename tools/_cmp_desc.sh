#!/bin/bash
# kernel-trace medians of the extractor kernels for two trees (current, _cmp/old)
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for T in "$R" "$R/_cmp/old"; do
  rm -rf $R/gpurun_out/cmp_kt
  timeout -k 5 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/cmp_kt -- python3 $T/tools/prof_extract.py --pairs 1024 --iters 3 --chunk 2048 > /dev/null 2>&1
  echo "== $T"; python3 $R/tools/kernel_times.py $R/gpurun_out/cmp_kt 2048
done
