#!/usr/bin/env python3
"""Turns the rocprofv3 outputs of tools/profile_round.sh (merged back under gpurun_out/<tag>/) into the
committed summaries: profiles/<tag>_kernel_stats_cfg3.csv, profiles/<tag>_pmc/<counter>_lvdgs_kernels.csv,
profiles/<tag>_bench_cfg3.json and profiles/traffic.json.

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KiB and on gfx950 FETCH_SIZE
reports half the bytes of 16-byte-per-lane reads, the access width of every bandwidth-relevant kernel here
(MI355X_MICROARCH.md, HBM / rocprofv3 section).  Counters come from separate --pmc passes and are averaged over
the launches of a kernel in the timed steps."""
import csv
import glob
import json
import os
import re
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# kernel symbol -> the name its launch is timed under in bench.py (ProfScope), where they differ
ALIASES = {"blend_bwd2": "blend_bwd", "blend_bwd3": "blend_bwd", "blend_fwd2": "blend_fwd", "blend_fwd2_deep": "blend_fwd", "preprocess_count": "preprocess_fwd",
           "preprocess_bwd_pose": "preprocess_bwd", "scatter_pairs": "group_scatter", "tile_depth_sort_wave": "tile_sort"}


def short(name):
    m = re.search(r"(\w+)_kernel", name)
    n = m.group(1) if m else name
    return ALIASES.get(n, n)


def main(tag, workload="cfg3_500k_1920x1080", prefix="pmc_", label=None):
    """prefix / label: the counter passes of another variant of the step (profile_round.sh: "posepmc_" = the pose-only backward)
    go to profiles/<tag>_<label>_pmc and to traffic.json under "<workload> (<label>)"."""
    src = os.path.join(ROOT, "gpurun_out", tag)
    if tag.startswith("-") or not os.path.isdir(src):
        raise SystemExit(f"usage: summarize_profiles.py <tag>   ({src} does not exist)")
    dst = os.path.join(ROOT, "profiles")
    suffix = "" if label is None else "_" + label
    stats = glob.glob(os.path.join(src, "stats" if label is None else label + "_stats", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], os.path.join(dst, f"{tag}{suffix}_kernel_stats_cfg3.csv"))
    bench = os.path.join(src, "bench.json")
    if label is None and os.path.exists(bench) and os.path.getsize(bench):
        shutil.copy(bench, os.path.join(dst, f"{tag}_bench_cfg3.json"))
    os.makedirs(os.path.join(dst, f"{tag}{suffix}_pmc"), exist_ok=True)
    per_kernel = defaultdict(lambda: defaultdict(list))
    for d in sorted(glob.glob(os.path.join(src, prefix + "*"))):
        if not os.path.isdir(d):
            continue
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            continue
        rows = [r for r in csv.DictReader(open(files[0])) if "lvdgs" in r["Kernel_Name"]]
        # bench.py's timed step launches the fused-loss instantiation of the backward blend; the run also holds the
        # autograd comparison's launches of the plain one (which reads gradient images): keep the timed kernel's only
        # (the first template argument: true / false until round 4, the LOSS_* code since -- 1 = fused loss, 0 = gradient images)
        fused = ("blend_bwd3_kernel<true", "blend_bwd3_kernel<1")
        plain = ("blend_bwd3_kernel<false", "blend_bwd3_kernel<0")
        if any(any(f in r["Kernel_Name"] for f in fused) for r in rows):
            rows = [r for r in rows if not any(f in r["Kernel_Name"] for f in plain)]
        name = os.path.basename(d)[len(prefix):].lower()
        with open(os.path.join(dst, f"{tag}{suffix}_pmc", f"{name}_lvdgs_kernels.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "SGPR_Count",
                        "Counter_Name", "Counter_Value"])
            for r in rows:
                w.writerow([r["Dispatch_Id"], short(r["Kernel_Name"]) + "_kernel", r["Grid_Size"], r["Workgroup_Size"], r["LDS_Block_Size"],
                            r["VGPR_Count"], r["SGPR_Count"], r["Counter_Name"], r["Counter_Value"]])
                per_kernel[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    note = ("(2*FETCH_SIZE + WRITE_SIZE)*1024: gfx950 FETCH_SIZE reports half the bytes of 16-byte-per-lane reads "
            "(MI355X_MICROARCH.md, HBM section) -- and of every other pattern of these kernels: it counts requests for 128-byte lines at 64 bytes "
            "each whatever the width per lane (calibrated on known byte counts, profiles/experiments/r05_traffic_calib); separate --pmc passes, "
            "averaged over launches")
    tpath = os.path.join(dst, "traffic.json")
    traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
    entry = {}
    for k, c in sorted(per_kernel.items()):
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            f, w = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"]), sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"])
            entry[k] = {"FETCH_SIZE_KB_raw": round(f, 1), "WRITE_SIZE_KB": round(w, 1),
                        "hbm_bytes_per_launch": int((2 * f + w) * 1024), "note": note}
            if "SQ_INSTS_VALU" in c:
                entry[k]["valu_wave_instructions_per_launch"] = int(sum(c["SQ_INSTS_VALU"]) / len(c["SQ_INSTS_VALU"]))
            for name, key in (("SQ_INSTS_SALU", "salu_wave_instructions_per_launch"), ("SQ_INSTS_LDS", "lds_wave_instructions_per_launch"),
                              ("SQ_WAVES", "waves_per_launch")):
                if name in c:
                    entry[k][key] = int(sum(c[name]) / len(c[name]))
    if entry:
        entry["_source"] = f"gpurun_out/{tag} (tools/profile_round.sh), summarised by tools/summarize_profiles.py"
        traffic[workload if label is None else f"{workload} ({label})"] = entry
        json.dump(traffic, open(tpath, "w"), indent=1, sort_keys=True)
    print("kernels with traffic:", [k for k in entry if not k.startswith("_")])


if __name__ == "__main__":
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01_x"
    main(tag)
    if glob.glob(os.path.join(ROOT, "gpurun_out", tag, "posepmc_*")):
        main(tag, prefix="posepmc_", label="pose")
