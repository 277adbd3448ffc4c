#!/bin/bash
for L in 1 2 3 4; do echo "== lanes $L"; timeout -k 10 300 python bench.py --no-cpu-baseline --no-loop-closure --steps 3 --lanes $L 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['verified'], r['frac'], r['avg_launch_ms'], r['frames_per_launch'], r['stage_us_per_frame'])"; done
