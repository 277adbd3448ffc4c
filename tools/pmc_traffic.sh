#!/bin/bash
# HBM traffic per kernel from the PMC counters (MI355X_MICROARCH.md "HBM"): FETCH_SIZE and WRITE_SIZE in SEPARATE
# passes (they do not fit one pass), each with --kernel-trace only. A 1 GiB torch copy in the same process calibrates
# the gfx950 FETCH_SIZE under-count (wide coalesced reads are tallied at half their bytes).
# Usage (on the GPU box): tools/pmc_traffic.sh <chunk>   -> gpurun_out/pmc_traffic_<chunk>.json
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
C=${1:-1024}
# dword-per-lane calibration copy (the streaming FAST/blur kernel's access width)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -shared -fPIC $R/tools/microbench/copy_dword.hip -o $R/gpurun_out/libcopy_dword.so 2> /dev/null && export ARIA_CALIB_COPY_SO=$R/gpurun_out/libcopy_dword.so
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$ctr
  timeout -k 5 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$ctr -- python3 $R/tools/prof_extract.py --pairs ${PAIRS:-512} --iters ${ITERS:-3} --chunk $C --match --calibrate > $R/gpurun_out/pmc_$ctr.log 2>&1 || { echo "pmc run $ctr failed"; tail -5 $R/gpurun_out/pmc_$ctr.log; exit 1; }
done
python3 $R/tools/pmc_traffic.py $R/gpurun_out $C $(( 2 * ${PAIRS:-512} * ${ITERS:-3} ))
