#!/bin/bash
# same-box A/B of preprocess_bwd builds on the opaque-surface workload (libraries built into lvd_gs-slam_amd/lib_<name> with
# make -C lvd_gs-slam_amd/csrc OUT=../lib_<name> EXTRA=-D...): tracking it/s and the kernels by time, HIP events
OUT=${OUT:-gpurun_out/pbwd_ab}
mkdir -p $OUT
for n in "$@"; do
  L=$PWD/lvd_gs-slam_amd/lib_$n/liblvdgs.so; [ "$n" = default ] && L=$PWD/lvd_gs-slam_amd/lib/liblvdgs.so
  for W in ${WORKLOADS:-surface_100k_1920x1080}; do
    LVDGS_LIB=$L LVDGS_BENCH_WORKLOAD=$W python3 bench.py --steps 100 --warmup 60 --no-cpu-baseline --no-side 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$n $W', d['value'], 'it/s', d['steady_state']['ms_per_step'], 'ms; kernels us/step:', d['kernels_us_per_step'])"
  done
done | tee $OUT/ab.txt
