#!/bin/bash
# What would a fourth wave per SIMD buy k_fast_blur_stream? TIMING PROBE with wrong results: build/ab/libprobe4w.so
# (tools/build_ab.sh probe4w -DARIA_PROBE_4WAVES: no pyramid step -> 127 VGPRs, short lists -> 10 192 B of LDS per wave) run at
# 3 waves per SIMD (LDS padded to 13 KB per wave) and at 4 (unpadded). Only LEVEL 0 is comparable (its input is the caller's
# image; the coarser raw levels are never written in this build). Usage (GPU box): tools/occupancy_probe_stream.sh [frames]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; F=${1:-8192}
cd /tmp; export TMPDIR=/tmp
export ARIA_ORB_HIP_LIBRARY=$R/build/ab/libprobe4w.so
for KB in 13 0 13 0; do
  if [ $KB = 0 ]; then unset ARIA_STREAM_LDS_KB; N=4waves; else export ARIA_STREAM_LDS_KB=$KB; N=3waves; fi
  rm -rf $O/occ_$N
  timeout -k 10 300 rocprofv3 --kernel-trace -d $O/occ_$N -o run -- python3 $R/tools/prof_extract.py --pairs $((F / 2)) --iters 4 --chunk $F > $O/occ_$N.log 2>&1 || { echo "$N failed"; tail -5 $O/occ_$N.log; }
  db=$(find $O/occ_$N -name "*.db" | head -1)
  echo "== probe build, $N per SIMD"; python3 $R/tools/level_times.py $db 640 480 $F | grep "^L0"
  rm -rf $O/occ_$N
done
