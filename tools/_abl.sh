#!/bin/bash
# timing-only ablation of k_fast_blur_band phases (results are invalid under ARIA_ABLATE != 0)
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for a in 0 1 2 4 8 3 15; do
  ARIA_ABLATE=$a rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/abl_$a -- python3 $R/tools/prof_extract.py --pairs 128 --iters 2 > $R/gpurun_out/abl_$a.log 2>&1
done
echo done
