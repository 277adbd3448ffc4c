#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; F=8192
cd /tmp; export TMPDIR=/tmp
for S in 0 240 160 120 96 0; do
  if [ $S = 0 ]; then unset ARIA_STREAM_SEG_ROWS; else export ARIA_STREAM_SEG_ROWS=$S; fi; export ARIA_ORB_HIP_LIBRARY=$R/aria_slam_amd/libaria_orb_hip_variants.so
  rm -rf $O/seg_$S
  timeout -k 10 300 rocprofv3 --kernel-trace -d $O/seg_$S -o run -- python3 $R/tools/prof_extract.py --pairs $((F / 2)) --iters 4 --chunk $F > $O/seg_$S.log 2>&1 || { echo "$S failed"; tail -5 $O/seg_$S.log; exit 1; }
  db=$(find $O/seg_$S -name "*.db" | head -1)
  python3 $R/tools/level_times.py $db 640 480 $F > $O/seg_$S.txt
  rm -rf $O/seg_$S
  echo "== seg_rows $S"; cat $O/seg_$S.txt
done
