#!/bin/bash
cd $GRAFT_REPO_ROOT
for cfg in "8 15 5" "16 15 5" "20 15 5" "24 15 5"; do
  set -- $cfg
  ARIA_BAND_BUDGET_KB=$1 ARIA_BAND_QPCT0=$2 ARIA_BAND_QPCT_STEP=$3 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 2 > gpurun_out/sw.json 2> gpurun_out/sw.err
  python -c "
import json; d=json.load(open('gpurun_out/sw.json')); print('$cfg', d['value'], d['roofline']['stage_us_per_frame']['fast_blur'])"
done
