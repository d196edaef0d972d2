#!/bin/bash
# SQ counter sets for the blend kernels, one rocprofv3 --pmc pass per set; prints per-kernel averages.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/${1:-pmc_blend}
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_TRANS_F32"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d "$OUT/s$i" -o run -- python3 "$ROOT/bench.py" --steps 3 --warmup 2 --no-cpu-baseline > "$OUT/s$i.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/s*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "blend_" in n:
            agg["bwd" if "bwd" in n else "fwd"][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in sorted(agg.items()):
    print(k, {n: round(sum(v) / len(v)) for n, v in sorted(c.items())})
PY
