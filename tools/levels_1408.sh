#!/bin/bash
# Per-level table of the batch FAST/blur launches at 1408x1408 / 4000 kp for each library given ("product" = in-tree).
# Usage: tools/levels_1408.sh <frames per launch> <lib> [<lib> ...]   (writes gpurun_out/l1408_<name>.txt)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; F=$1; shift
cd /tmp; export TMPDIR=/tmp
for L in "$@"; do
  N=$(basename $L .so)
  if [ "$L" = product ]; then unset ARIA_ORB_HIP_LIBRARY; else export ARIA_ORB_HIP_LIBRARY=$R/$L; fi
  rm -rf $O/l1408_$N
  timeout -k 10 300 rocprofv3 --kernel-trace -d $O/l1408_$N -o run -- python3 $R/tools/prof_extract.py --pairs $((F / 2)) --iters 3 --chunk $F --width 1408 --height 1408 --features 4000 > $O/l1408_$N.log 2>&1 || { echo "$N failed"; tail -5 $O/l1408_$N.log; exit 1; }
  python3 $R/tools/level_times.py $(find $O/l1408_$N -name "*.db" | head -1) 1408 1408 $F > $O/l1408_$N.txt
  rm -rf $O/l1408_$N
  echo "== $N"; cat $O/l1408_$N.txt
done
