#!/bin/bash
# Same-box A/B of an environment knob: tools/ab_env.sh VAR value1 value2 [-- bench.py arguments]
# Runs bench.py alternately (two rounds) with VAR set to each value and prints value, ms per step, the pair count
# and the per-kernel microseconds of each run.  (Boxes differ by a few percent: only numbers of one call compare.)
var=$1; shift
vals=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do vals+=("$1"); shift; done
[ "$1" = "--" ] && shift
for round in 1 2; do
  for v in "${vals[@]}"; do
    env "$var=$v" python bench.py --steps 200 --warmup 20 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('$var=$v', d['value'], 'it/s', d['ms_per_step'], 'ms', 'pairs', d['config'].get('pairs'), json.dumps(d.get('kernels_us_per_step')))"
  done
done
