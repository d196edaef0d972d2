#!/bin/bash
# GPU-busy time of the KITTI 8+2 mapping iteration: rocprofv3 kernel trace of 28 fused iterations (tools/map_bench.py).
# usage: map_trace.sh [tag] [workload]   (MAP_BENCH_MASKED=1: keyframes with a static mask)
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out/map_trace${1:+_$1}; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export MAP_BENCH_FUSED_ONLY=1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o run -- python3 "$ROOT/tools/map_bench.py" ${2:-kitti07_geom} > "$OUT/log.txt" 2>&1
grep "per iteration" "$OUT/log.txt"
python3 - "$OUT" <<'PY' | tee "$OUT/summary.txt"
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel time per iteration (28 iterations + scene set-up): %.3f ms" % (tot / 28 / 1e6))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:22]:
    print("%-60s calls/it %6.1f  us/it %8.1f" % (r["Name"][:60], int(r["Calls"]) / 28, float(r["TotalDurationNs"]) / 28 / 1e3))
PY
