#!/bin/bash
# same-box A/B of preprocess_bwd's Gaussians per wave (LVDGS_PBWD_LANES: experiment switch of the library)
OUT=${OUT:-gpurun_out/pbwd_ab}
mkdir -p $OUT
for rep in 1 2; do
for L in 64 32 16; do
  for W in ${WORKLOADS:-surface_100k_1920x1080 cfg3_500k_1920x1080 kitti07_geom}; do
    LVDGS_PBWD_LANES=$L LVDGS_BENCH_WORKLOAD=$W python3 bench.py --steps 100 --warmup 60 --no-cpu-baseline --no-side 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels_us_per_step']
print('lanes=$L $W', d['value'], 'it/s', d['steady_state']['ms_per_step'], 'ms; preprocess_bwd', k.get('preprocess_bwd'), 'tail', k.get('tracking_tail'), 'pose-only it/s', d['config'].get('pose_only_iters_per_s'))"
  done
done
done | tee $OUT/lanes.txt
