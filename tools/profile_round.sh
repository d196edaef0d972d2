#!/bin/bash
# Collects the evidence behind bench.py's roofline object on an MI355X box (run through gpurun):
#   1. rocprofv3 --kernel-trace --stats of the default workload  -> gpurun_out/<tag>/stats
#   2. separate --pmc passes (one counter set per run, nothing else traced) -> gpurun_out/<tag>/pmc_*
# then tools/summarize_profiles.py turns the CSVs into profiles/<tag>_kernel_stats_cfg3.csv,
# profiles/<tag>_pmc/*.csv and profiles/traffic.json (run that here, after gpurun merged gpurun_out back).
#   usage: tools/profile_round.sh r01_d
set -u
TAG=${1:-r01_x}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- python3 "$ROOT/bench.py" --steps 100 --warmup 60 --no-cpu-baseline --no-side > "$OUT/stats.log" 2>&1   # (60 iterations of warm-up: the GPU is at its clocks when the trace starts counting, tools/short_region.py)
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$C" -o run -- python3 "$ROOT/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-side > "$OUT/pmc_$C.log" 2>&1
done
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d "$OUT/pmc_SQ" -o run -- python3 "$ROOT/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-side > "$OUT/pmc_SQ.log" 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d "$OUT/pmc_SQ2" -o run -- python3 "$ROOT/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-side > "$OUT/pmc_SQ2.log" 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_GRBM" -o run -- python3 "$ROOT/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-side > "$OUT/pmc_GRBM.log" 2>&1
# the product's tracking backward (pose + exposure gradients only, LVDGS_FLAG_POSE_ONLY): kernel times and instruction counts
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/pose_stats" -o run -- python3 "$ROOT/bench.py" --pose-only --steps 100 --warmup 60 --no-cpu-baseline --no-side > "$OUT/pose_stats.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d "$OUT/posepmc_SQ" -o run -- python3 "$ROOT/bench.py" --pose-only --steps 3 --warmup 2 --no-cpu-baseline --no-side > "$OUT/posepmc_SQ.log" 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d "$OUT/posepmc_SQ2" -o run -- python3 "$ROOT/bench.py" --pose-only --steps 3 --warmup 2 --no-cpu-baseline --no-side > "$OUT/posepmc_SQ2.log" 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d "$OUT/posepmc_$C" -o run -- python3 "$ROOT/bench.py" --pose-only --steps 3 --warmup 2 --no-cpu-baseline --no-side > "$OUT/posepmc_$C.log" 2>&1
done
python3 "$ROOT/bench.py" > "$OUT/bench.json" 2> "$OUT/bench.err"   # (the defaults: 200 steps after 20 of warm-up)
ls -R "$OUT" | head -40
