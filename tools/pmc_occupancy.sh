#!/bin/bash
# Average occupancy and wait shares of the blend kernels on a workload: SQ_WAVE_CYCLES (sum of the waves' lifetimes),
# SQ_BUSY_CYCLES, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY per launch.   usage (through gpurun): tools/pmc_occupancy.sh <workload> [tag]
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=${1:-kitti07_geom}
OUT=$ROOT/gpurun_out/${2:-pmc_occ}_$W
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$OUT/s1" -o run -- python3 "$ROOT/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --workload "$W" > "$OUT/s1.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS --output-format csv -d "$OUT/s2" -o run -- python3 "$ROOT/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --workload "$W" > "$OUT/s2.log" 2>&1
python3 - "$OUT" "$W" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/s*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "blend_" in n:
            agg["blend_bwd" if "bwd" in n else "blend_fwd"][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in sorted(agg.items()):
    print(sys.argv[2], k, {n: round(sum(v) / len(v)) for n, v in sorted(c.items())})
PY
