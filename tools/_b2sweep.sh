#!/bin/bash
# quick sensitivity runs of the band kernels (throughput only)
run() { echo "== $*"; env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --no-verify --no-loop-closure --steps 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], r['frac'], r['avg_launch_ms'], r['stage_us_per_frame'], d['config']['slow_path_blocks'])"; }
for a in "$@"; do run $a; done
