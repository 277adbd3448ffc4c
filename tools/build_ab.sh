#!/bin/bash
# A/B builds of the product library with extra defines: tools/build_ab.sh <name> [-DX=Y ...]  ->  build/ab/lib<name>.so
# (selected on the GPU box with ARIA_ORB_HIP_LIBRARY=build/ab/lib<name>.so, see tools/ab_levels.sh)
set -e
R=$(cd "$(dirname "$0")/.." && pwd); N=$1; shift
mkdir -p $R/build/ab
cd $R/aria_slam_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math \
  -w -I$R/include -I. "$@" -shared -o $R/build/ab/lib$N.so \
  orb_kernels.hip fast_blur_band.hip fast_blur_stream.hip pyramid_pass.hip orb_api.hip match_hip.hip knn2_mfma.hip runtime_api.hip orb_plan.cpp synth.cpp -lpthread
echo built build/ab/lib$N.so
