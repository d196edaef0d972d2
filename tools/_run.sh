mkdir -p gpurun_out/r04_b
python3 -m pytest tests/test_gpu_pose_only.py tests/test_gpu_c_abi.py tests/test_gpu_fast_tracking.py -x -q > gpurun_out/r04_b/pytest.txt 2>&1; tail -5 gpurun_out/r04_b/pytest.txt
python3 bench.py > gpurun_out/r04_b/bench_default.json 2> gpurun_out/r04_b/bench_default.err; tail -c 3000 gpurun_out/r04_b/bench_default.json; tail -3 gpurun_out/r04_b/bench_default.err
for round in 1 2; do for v in 5 6 7 8; do
  lib=lvd_gs-slam_amd/lib_p$v/liblvdgs.so; [ $v = 6 ] && lib=lvd_gs-slam_amd/lib/liblvdgs.so
  for w in cfg3_500k_1920x1080 kitti07_geom; do
    LVDGS_LIB=$lib python3 bench.py --workload $w --pose-only --no-side --no-cpu-baseline --steps 200 --warmup 60 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('wgs_pose=$v', '$w', d['value'], d['ms_per_step'], d['kernels_us_per_step'])"
  done
done; done > gpurun_out/r04_b/pose_wgs_ab.txt 2>&1
cat gpurun_out/r04_b/pose_wgs_ab.txt
