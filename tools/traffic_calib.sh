#!/bin/bash
# tools/traffic_calib (known byte counts per launch) under rocprofv3's FETCH_SIZE and WRITE_SIZE passes and a kernel trace:
# what the counters report per byte moved, per access pattern.  Output: gpurun_out/traffic_calib/summary.json (+ the raw csv).
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out/traffic_calib; rm -rf "$OUT"; mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
BIN=$ROOT/tools/traffic_calib
[ -x "$BIN" ] || hipcc --offload-arch=gfx950 -O3 -w -o "$BIN" "$ROOT/tools/traffic_calib.hip" || exit 1
"$BIN" 3 > "$OUT/bytes.json" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o run -- "$BIN" 3 > "$OUT/trace.log" 2>&1
for C in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $C --output-format csv -d "$OUT/$C" -o run -- "$BIN" 3 > "$OUT/$C.log" 2>&1; done
python3 - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
known = json.load(open(out + "/bytes.json"))
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    rows = sorted((r for r in csv.DictReader(open(f)) if "calib_" in r["Kernel_Name"]), key=lambda r: int(r["Dispatch_Id"]))
    seen = collections.Counter()
    for r in rows:
        n = r["Kernel_Name"].split("calib_")[1].split("_kernel")[0]
        if n == "tile4" and "<true>" in r["Kernel_Name"]: n = "tile4xcd"
        if n == "gather44":
            seen[r["Counter_Name"]] += 1
            if seen[r["Counter_Name"]] % 2 == 0: n = "gather44x4"
        per[n][r["Counter_Name"]].append(float(r["Counter_Value"]) * 1024)
dur = collections.defaultdict(list)
for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    rows = sorted((r for r in csv.DictReader(open(f)) if "calib_" in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
    k = 0
    for r in rows:
        n = r["Kernel_Name"].split("calib_")[1].split("_kernel")[0]
        if n == "tile4" and "<true>" in r["Kernel_Name"]: n = "tile4xcd"
        if n == "gather44":
            k += 1
            if k % 2 == 0: n = "gather44x4"
        dur[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
res = {}
for n in known["launch_order"]:
    c = per.get(n, {})
    f = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"]) if c.get("FETCH_SIZE") else None
    w = sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"]) if c.get("WRITE_SIZE") else None
    rb, wb = known["read_bytes"].get(n), known["write_bytes"].get(n)
    e = {"FETCH_SIZE_bytes": f, "WRITE_SIZE_bytes": w, "known_read_bytes": rb, "known_write_bytes": wb, "us": round(sum(dur[n]) / max(1, len(dur[n])), 2)}
    if f is not None and rb: e["FETCH_SIZE_per_known_read_byte"] = round(f / rb, 4)
    if w is not None and wb: e["WRITE_SIZE_per_known_write_byte"] = round(w / wb, 4)
    if n == "or1" and w is not None: e["WRITE_SIZE_bytes_per_atomic"] = round(w / (known["write_bytes"]["or1_as_dwords"] / 4), 3)
    if n == "store1" and w is not None: e["WRITE_SIZE_bytes_per_byte_store"] = round(w / known["write_bytes"]["store1"], 3)
    res[n] = e
json.dump({"known": known, "measured": res}, open(out + "/summary.json", "w"), indent=1)
for n, e in res.items(): print(n, e)
PY
