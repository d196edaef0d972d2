#!/bin/bash
# Same-box A/B of an environment knob on the KITTI 8+2 mapping window (tools/map_bench.py, fused path only):
#   tools/ab_map.sh VAR value1 value2
var=$1; shift
for round in 1 2; do
  for v in "$@"; do
    echo -n "$var=$v  "; env "$var=$v" MAP_BENCH_FUSED_ONLY=1 python3 "$(dirname "$0")/map_bench.py" 2>/dev/null | grep "per iteration" | cut -c1-150
  done
done
