#!/bin/bash
# kernel-trace of the extractor + matcher on a resident batch; prints per-frame median kernel times
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
C=${1:-1024}
rm -rf $R/gpurun_out/ktc_$C
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ktc_$C -- python3 $R/tools/prof_extract.py --pairs 512 --iters 6 --chunk $C --match > $R/gpurun_out/ktc_$C.log 2>&1 || { echo "rocprof run failed"; tail -5 $R/gpurun_out/ktc_$C.log; exit 1; }
python3 $R/tools/kernel_times.py $R/gpurun_out/ktc_$C $C
