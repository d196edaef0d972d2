#!/bin/bash
# The evidence of a round's final library in one gpurun call: tools/profile_round.sh <tag> (config-3 kernel trace, PMC passes, default
# bench line), the bench line with the driver's flags, the KITTI-geometry pose-only kernel trace, the KITTI mapping window's traces
# (unmasked, masked), the soak run and the whole GPU test suite.   usage: tools/final_round.sh r05_h
TAG=${1:-r0x_final}
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"
bash "$ROOT/tools/profile_round.sh" "$TAG" > "$OUT/profile_round.log" 2>&1
( cd "$ROOT" && python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_flags.json" 2> "$OUT/bench_driver_flags.err" )
( cd /tmp && export TMPDIR=/tmp && LVDGS_BENCH_WORKLOAD=kitti07_geom rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kitti_stats" -o run -- python3 "$ROOT/bench.py" --pose-only --steps 100 --warmup 60 --no-cpu-baseline --no-side > "$OUT/kitti_stats.log" 2>&1 )
bash "$ROOT/tools/map_trace.sh" "${TAG}_unmasked" > "$OUT/map_trace_unmasked.log" 2>&1
MAP_BENCH_MASKED=1 bash "$ROOT/tools/map_trace.sh" "${TAG}_masked" > "$OUT/map_trace_masked.log" 2>&1
( cd "$ROOT" && python3 tools/soak.py > "$OUT/soak.txt" 2>&1 )
( cd "$ROOT" && python3 -m pytest tests -q -m gpu -x 2>&1 | tail -5 > "$OUT/gpu_suite.txt" )
tail -3 "$OUT/gpu_suite.txt"; cat "$OUT/bench.json" | head -c 600; echo; tail -4 "$OUT/soak.txt"
