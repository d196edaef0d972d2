#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the blend kernels only (two rocprofv3 --pmc passes), printed as MB per launch.
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out/pmc_quick; rm -rf "$OUT"; mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $C --output-format csv -d "$OUT/$C" -o run -- python3 "$ROOT/bench.py" --steps 3 --warmup 2 --no-cpu-baseline > "$OUT/$C.log" 2>&1; done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        for k in ("blend_bwd", "blend_fwd", "preprocess_bwd"):
            if k in n: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in sorted(agg.items()):
    f, w = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"]), sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"])
    print(k, "fetch %.0f KB raw, write %.0f KB -> %.0f MB" % (f, w, (2 * f + w) * 1024 / 1e6))
PY
