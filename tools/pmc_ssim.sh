#!/bin/bash
# SQ counter sets of the fused L1 + SSIM kernel (tools/ssim_bench.py: one 1080p image), one rocprofv3 --pmc pass per set.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/${1:-pmc_ssim}
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_ANY" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_TRANS_F32"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d "$OUT/s$i" -o run -- python3 "$ROOT/tools/ssim_bench.py" > "$OUT/s$i.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/s*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "ssim_l1_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print({n: round(sum(v) / len(v)) for n, v in sorted(agg.items())})
PY
