#!/bin/bash
# scratch: pyramid band height of the single-frame schedule vs host-path time
for bh in 8 16 24 32 48; do
  echo "== ARIA_PYR_BH=$bh"; ARIA_PYR_BH=$bh python tools/host_path_rate.py 2>&1 | grep aria_
done
echo "== default"; python tools/host_path_rate.py 2>&1 | grep aria_
