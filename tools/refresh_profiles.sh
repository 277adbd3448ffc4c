#!/bin/bash
# Round-end evidence, run on the GPU box. Everything bench.py's roofline block states can be recomputed from the files this
# writes (copy them from gpurun_out/ into profiles/):
#   <T>_bench.json                    python bench.py (with cpu_baseline)
#   <T>_bench_under_rocprof.json      the same command under rocprofv3 --kernel-trace (its own clock: slower)
#   <T>_kernel_stats_by_shape.csv     kernel statistics of THAT run split by launch shape (tools/kernel_stats_by_shape.py):
#                                     the 8192-frame batch launches and the single-frame launches of the host-path leg apart
#   <T>_levels_640x480.txt            per-level table of the batch FAST/blur launches of THAT run (tools/level_times.py):
#                                     us per launch, us per Mpx, waves, lanes with pixels; sum = stage us/frame, mean per
#                                     launch = roofline.avg_launch_ms, algorithmic GB/s = roofline.achieved
#   <T>_levels_1408x1408.txt          the same for 1408x1408 / 4000 kp, 1024 frames per launch (prof_extract.py)
#   <T>_pmc_traffic.json              HBM bytes per frame and kernel from PMC FETCH_SIZE / WRITE_SIZE (separate passes)
#   <T>_sq_counters.txt               SQ counter passes of the extractor kernels (tools/pmc_sq_r3.sh, 4096 frames)
# Usage: tools/refresh_profiles.sh <tag>
T=${1:-r4_x}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
# PMC traffic FIRST, installed as profiles/pmc_traffic_r4.json (with the sha256 of the kernel source it was measured on), so that
# the bench line below carries this build's traffic and traffic_stale: false
cd $R && PAIRS=2048 ITERS=1 bash tools/pmc_traffic.sh 4096 > $O/${T}_pmc.log 2>&1 || { echo pmc failed; tail -5 $O/${T}_pmc.log; exit 1; }
cp $O/pmc_traffic_4096.json $O/${T}_pmc_traffic.json; cp $O/pmc_traffic_4096.json $R/profiles/pmc_traffic_r4.json
cd $R && timeout -k 10 500 python3 bench.py > $O/${T}_bench.log 2> $O/${T}_bench.err || { echo bench failed; tail -5 $O/${T}_bench.err; exit 1; }
tail -1 $O/${T}_bench.log > $O/${T}_bench.json
cd /tmp; export TMPDIR=/tmp
rm -rf $O/${T}_stats
timeout -k 10 500 rocprofv3 --kernel-trace -d $O/${T}_stats -o run -- python3 $R/bench.py --no-cpu-baseline > $O/${T}_stats.log 2>&1 || { echo rocprof failed; tail -5 $O/${T}_stats.log; exit 1; }
grep '^{' $O/${T}_stats.log | tail -1 > $O/${T}_bench_under_rocprof.json
db=$(find $O/${T}_stats -name "*.db" | head -1)
python3 $R/tools/kernel_stats_by_shape.py $db 500 > $O/${T}_kernel_stats_by_shape.csv
python3 $R/tools/level_times.py $db 640 480 8192 > $O/${T}_levels_640x480.txt
rm -rf $O/${T}_stats1408
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/${T}_stats1408 -o run -- python3 $R/tools/prof_extract.py --pairs 512 --iters 3 --chunk 1024 --width 1408 --height 1408 --features 4000 > $O/${T}_stats1408.log 2>&1 || { echo rocprof 1408 failed; tail -5 $O/${T}_stats1408.log; exit 1; }
python3 $R/tools/level_times.py $(find $O/${T}_stats1408 -name "*.db" | head -1) 1408 1408 1024 > $O/${T}_levels_1408x1408.txt
# ... and at BASELINE configs[3]'s per-GPU batch (4096 frames per launch: every level has >= 6 rounds of waves)
rm -rf $O/${T}_stats1408b
timeout -k 10 400 rocprofv3 --kernel-trace -d $O/${T}_stats1408b -o run -- python3 $R/tools/prof_extract.py --pairs 2048 --iters 2 --chunk 4096 --width 1408 --height 1408 --features 4000 > $O/${T}_stats1408b.log 2>&1 && python3 $R/tools/level_times.py $(find $O/${T}_stats1408b -name "*.db" | head -1) 1408 1408 4096 > $O/${T}_levels_1408x1408_4096frames.txt
rm -rf $O/${T}_stats1408b
cd $R && bash tools/pmc_sq_r3.sh ${T} > /dev/null 2>&1; cp $O/pmc_sq_${T}.txt $O/${T}_sq_counters.txt
rm -rf $O/${T}_stats $O/${T}_stats1408
# the other configurations (BASELINE configs[3] at its per-GPU size, the EuRoC size, the north star's second size at 2000 kp)
cd $R && timeout -k 10 400 python3 bench.py --width 1408 --height 1408 --features 4000 --pairs 2048 --no-cpu-baseline 2> /dev/null | tail -1 > $O/${T}_bench_1408x1408_4000kp_4096frames.json
cd $R && timeout -k 10 300 python3 bench.py --width 752 --height 480 --features 1000 --no-cpu-baseline 2> /dev/null | tail -1 > $O/${T}_bench_752x480_1000kp.json
cd $R && timeout -k 10 300 python3 bench.py --width 1408 --height 1408 --features 2000 --pairs 1024 --no-cpu-baseline 2> /dev/null | tail -1 > $O/${T}_bench_1408x1408_2000kp.json
cd $R && bash tools/pmc_sq_matcher.sh ${T} > /dev/null 2>&1; cp $O/pmc_sq_matcher_${T}.txt $O/${T}_sq_counters_matcher.txt

# the streaming kernel by waves per SIMD (variants library: LDS padding), and the C++ batch driver's rate
cd $R && bash tools/occupancy_probe_stream.sh 8192 > $O/${T}_occupancy_stream.txt 2>&1
cd $R && timeout -k 10 300 python3 tools/euroc_batch_rate.py > $O/${T}_euroc_batch_rate.txt 2>&1
echo refreshed $T with occupancy scan and batch driver rate
