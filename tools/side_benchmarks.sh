#!/bin/bash
# The workloads next to the headline one, into gpurun_out/<tag>/: the other bench.py workloads, the mapping step, the
# KITTI 8+2 mapping window (tools/map_bench.py) and the warm track_frame loop (tools/track_bench.py).
#   usage (through gpurun, after tools/profile_round.sh <tag>): tools/side_benchmarks.sh r02_g
TAG=${1:-r02_x}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
for w in kitti07_geom cfg2_100k_640x480 cfg5_2m_1920x1280 surface_100k_1920x1080; do
  python3 bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline > "$OUT/bench_$w.json" 2>/dev/null
done
python3 bench.py --step mapping --steps 50 --warmup 5 --no-cpu-baseline > "$OUT/bench_mapping_cfg3.json" 2>/dev/null
python3 bench.py --step mapping --steps 50 --warmup 5 --no-cpu-baseline --workload kitti07_geom > "$OUT/bench_mapping_kitti07.json" 2>/dev/null
python3 bench.py --step mapping --window weak --steps 100 --warmup 10 --no-cpu-baseline > "$OUT/bench_mapping_single_view_cfg3.json" 2>/dev/null
python3 tools/map_bench.py > "$OUT/map_bench.txt" 2>&1
python3 tools/track_bench.py > "$OUT/track_bench.txt" 2>&1
tail -n 3 "$OUT/map_bench.txt"; tail -n 3 "$OUT/track_bench.txt"
