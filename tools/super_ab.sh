#!/bin/bash
# same-box A/B of the two-level grouping (LVDGS_FLAG_SUPER_TILES): tracking it/s with the hint off / on / automatic, and the kernels by time
OUT=${1:-gpurun_out/super_ab}
mkdir -p $OUT
for W in surface_100k_1920x1080 kitti07_geom cfg3_500k_1920x1080; do
  for M in 0 1 auto; do
    LVDGS_SUPER_TILES=$M python3 tools/track_ab.py $W 2>/dev/null | sed "s/^/super=$M /"
  done
done > $OUT/track_ab.txt
cat $OUT/track_ab.txt
for M in 0 1; do
  LVDGS_SUPER_TILES=$M LVDGS_BENCH_WORKLOAD=surface_100k_1920x1080 python3 bench.py --steps 100 --warmup 60 --no-cpu-baseline --no-side 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('super=$M', d['value'], 'it/s', d['steady_state']['ms_per_step'], 'ms; kernels us/step:', d['kernels_us_per_step'])" 
done > $OUT/bench_surface.txt
cat $OUT/bench_surface.txt
