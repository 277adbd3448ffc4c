#!/bin/bash
# A/B of FAST/blur kernel builds on one GPU box: for every library given (path relative to the repo root; "product" = the
# in-tree product library) the per-level table of the batch launches (rocprofv3 --kernel-trace, tools/level_times.py).
# Usage: tools/ab_levels.sh <frames per launch> <lib> [<lib> ...]     (writes gpurun_out/ab_<name>.txt)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; F=$1; shift
cd /tmp; export TMPDIR=/tmp
for L in "$@"; do
  N=$(basename $L .so)
  if [ "$L" = product ]; then unset ARIA_ORB_HIP_LIBRARY; else export ARIA_ORB_HIP_LIBRARY=$R/$L; fi
  rm -rf $O/ab_$N
  timeout -k 10 300 rocprofv3 --kernel-trace -d $O/ab_$N -o run -- python3 $R/tools/prof_extract.py --pairs $((F / 2)) --iters 4 --chunk $F > $O/ab_$N.log 2>&1 || { echo "$N failed"; tail -5 $O/ab_$N.log; exit 1; }
  db=$(find $O/ab_$N -name "*.db" | head -1)
  python3 $R/tools/level_times.py $db 640 480 $F > $O/ab_$N.txt
  python3 $R/tools/kernel_stats_by_shape.py $db 3 | grep -E "k_describe|k_select<false" | cut -c1-120 >> $O/ab_$N.txt
  rm -rf $O/ab_$N
  echo "== $N"; tail -4 $O/ab_$N.txt
done
