#!/bin/bash
# Round-end evidence, run on the GPU box: bench.py (with cpu_baseline), rocprofv3 --kernel-trace --stats of the same
# command, PMC traffic passes (FETCH_SIZE / WRITE_SIZE, separate passes), SQ counter passes of the final kernels.
# Results land in gpurun_out/ (copy the summaries into profiles/).
# Usage: tools/refresh_profiles.sh <tag>
T=${1:-r2_x}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R && timeout -k 10 400 python3 bench.py > $O/${T}_bench.log 2> $O/${T}_bench.err || { echo bench failed; tail -5 $O/${T}_bench.err; exit 1; }
tail -1 $O/${T}_bench.log > $O/${T}_bench.json
cd /tmp; export TMPDIR=/tmp
rm -rf $O/${T}_stats
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_stats -- python3 $R/bench.py --no-cpu-baseline > $O/${T}_stats.log 2>&1 || { echo rocprof failed; tail -5 $O/${T}_stats.log; exit 1; }
f=$(ls -t $O/${T}_stats/*/*_kernel_stats.csv | head -1); cp $f $O/${T}_kernel_stats.csv
grep '^{' $O/${T}_stats.log | tail -1 > $O/${T}_bench_under_rocprof.json
cd $R && bash tools/pmc_traffic.sh 1024 > $O/${T}_pmc.log 2>&1 || { echo pmc failed; tail -5 $O/${T}_pmc.log; exit 1; }
cd $R && bash tools/pmc_sq_r2.sh ${T} > /dev/null 2>&1; cp $O/pmc_sq_${T}.txt $O/${T}_sq_counters.txt
echo refreshed $T
