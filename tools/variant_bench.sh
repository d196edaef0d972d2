#!/bin/bash
# blend_bwd variants (LVDGS_BLEND_BWD=1 single pass, 2 two passes) on several workloads: iterations/s and the top kernels.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for w in "$@"; do
  for v in 1 2; do
    LVDGS_BLEND_BWD=$v python3 "$ROOT/bench.py" --workload "$w" --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | \
      python3 -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels_us_per_step']; print('variant $v', d['config']['workload'], d['value'], {n: k[n] for n in list(k)[:6]})"
  done
done
