#!/bin/bash
# k_fast_blur_stream by waves per SIMD, on the VARIANTS library (aria_slam_amd/csrc: make variants), whose ARIA_STREAM_LDS_KB
# pads the workgroup's LDS: 10 KB per wave (unpadded) = 4 waves per SIMD, 13 KB = 3, 20 KB = 2. Same kernel, same results.
# (Round 4 first ran this on a probe build without pyramid step and with short lists -- profiles/r4_occupancy_probe_stream.txt --
# to learn what a fourth wave would buy before the kernel fitted one.) Usage (GPU box): tools/occupancy_probe_stream.sh [frames]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; F=${1:-8192}
cd /tmp; export TMPDIR=/tmp
export ARIA_ORB_HIP_LIBRARY=$R/aria_slam_amd/libaria_orb_hip_variants.so
for KB in 0 13 20 0 13; do
  if [ $KB = 0 ]; then unset ARIA_STREAM_LDS_KB; N=4waves; elif [ $KB = 13 ]; then export ARIA_STREAM_LDS_KB=13; N=3waves; else export ARIA_STREAM_LDS_KB=$KB; N=2waves; fi
  rm -rf $O/occ_$N
  timeout -k 10 300 rocprofv3 --kernel-trace -d $O/occ_$N -o run -- python3 $R/tools/prof_extract.py --pairs $((F / 2)) --iters 4 --chunk $F > $O/occ_$N.log 2>&1 || { echo "$N failed"; tail -5 $O/occ_$N.log; exit 1; }
  db=$(find $O/occ_$N -name "*.db" | head -1)
  echo "== $N per SIMD"; python3 $R/tools/level_times.py $db 640 480 $F | grep -E "^L0|sum over|of 8000"
  rm -rf $O/occ_$N
done
