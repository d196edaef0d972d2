#!/usr/bin/env python3
"""Per-survivor instruction budget of blend_bwd3, taken from the DISASSEMBLY (not from the source): compiles csrc/blend.hip to
gfx950 assembly, cuts the kernel into the regions of its inner loop by what the code between two labels contains, and counts
instructions by kind.  Writes a markdown table (stdout).  Needs no GPU.

    python3 tools/isa_budget.py > profiles/r04_blend_bwd_isa_budget.md

Regions:  pixel pass, straight-line  = the code between two labels that holds eight v_exp_f32 (a full batch of eight survivors);
          pixel pass, rolled         = the loop body with one v_exp_f32 and one M store (the last, partial batch of a round);
          splat pass                 = the code with the DPP subtracts / multiply-adds (one per batch, full or not);
          quadrant test              = the code with v_log_f32 (reaches_rect: 64 staged entries per execution and wave);
          everything else            = prologue (fused loss), staging, flush, loop control."""
import os, re, subprocess, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "lvd_gs-slam_amd", "csrc", "blend.hip")
asm = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize",
                      "-S", "--cuda-device-only", SRC, "-o", "-"], capture_output=True, text=True, check=True).stdout

KINDS = ["valu", "valu_trans", "valu_dpp", "valu_xlane", "salu", "lds", "vmem", "branch", "wait_nop"]
def kind(op):
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier")): return "wait_nop"
    if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc")): return "branch"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op.startswith(("v_exp", "v_rcp", "v_log", "v_sqrt", "v_rsq", "v_sin", "v_cos")): return "valu_trans"
    if op.startswith(("v_permlane", "v_readlane", "v_writelane", "v_readfirstlane")): return "valu_xlane"
    if op.startswith("v_"): return "valu"
    return None

def kernel_text(mangled_part):
    m = re.search(r"^(_ZN5lvdgs12_GLOBAL__N_1" + mangled_part + r"\w*):[^\n]*\n(.*?)\n\s*s_endpgm", asm, re.S | re.M)
    return m.group(1), m.group(2)

def budget(name_re, title):
    name, text = kernel_text(name_re)
    segs, cur = [], []
    for line in text.split("\n"):
        t = line.strip()
        if not t or t.startswith(";"): continue
        if re.match(r"^\.LBB\d+_\d+:", t):
            segs.append(cur); cur = []; continue
        op = t.split()[0]
        if op.startswith("."): continue
        k = kind(op)
        if k is None: continue
        dpp = "row_ror" in t or "row_shr" in t or "row_bcast" in t or "quad_perm" in t or "row_mirror" in t or "row_half_mirror" in t or "_dpp" in op
        cur.append(("valu_dpp" if (k == "valu" and dpp) else k, op))
    segs.append(cur)
    regions = collections.OrderedDict((r, collections.Counter()) for r in ("pixel pass, straight-line batch of 8", "pixel pass, rolled (per entry)",
                                                                          "splat pass (per batch)", "quadrant test (per 64 staged entries and wave)", "everything else"))
    for s in segs:
        ops = [o for _, o in s]
        n_exp = sum(o.startswith("v_exp_f32") for o in ops)
        has_dppmac = any(k == "valu_dpp" and o.startswith(("v_fmac", "v_subrev")) for k, o in s)
        if n_exp >= 8: r = "pixel pass, straight-line batch of 8"
        elif n_exp == 1 and any(o.startswith("ds_write") for o in ops) and any(o.startswith("v_rcp") for o in ops): r = "pixel pass, rolled (per entry)"
        elif has_dppmac: r = "splat pass (per batch)"
        elif any(o.startswith("v_log_f32") for o in ops): r = "quadrant test (per 64 staged entries and wave)"
        else: r = "everything else"
        for k, _ in s: regions[r][k] += 1
    print(f"### {title}\n\n`{name}`\n")
    print("| region | " + " | ".join(KINDS) + " | all vector | per survivor (vector) |")
    print("|---|" + "---|" * (len(KINDS) + 2))
    for r, c in regions.items():
        vec = c["valu"] + c["valu_trans"] + c["valu_dpp"] + c["valu_xlane"]
        per = ""
        if r.startswith("pixel pass, straight"): per = f"{vec / 8:.2f}"
        elif r.startswith("pixel pass, rolled"): per = f"{vec:.0f}"
        elif r.startswith("splat"): per = f"{vec / 8:.2f} at a full batch, {vec / 8 / FILL:.2f} at the measured fill of {FILL:.3f}"
        print(f"| {r} | " + " | ".join(str(c[k]) for k in KINDS) + f" | {vec} | {per} |")
    return regions

FILL = 0.9045   # tools/fill_diag.py on the GPU, config 3: survivors / (8 x batches)
print("# blend_bwd3: instructions per surviving (quadrant, Gaussian), from the gfx950 disassembly\n")
print("Static counts of the code regions (one execution each); `tools/isa_budget.py` regenerates this file.  Issue cost relative to a plain "
      "vector instruction (tools/valu_clock.hip, 8 waves/SIMD): transcendental 3.25, DPP 1.5, v_permlane*_swap 2.7.\n")
full = budget(r"17blend_bwd3_kernelILb1ELb0ELb0EEE", "Full backward, fused loss, no depth gradient (the bench's headline step)")
print()
pose = budget(r"17blend_bwd3_kernelILb1ELb0ELb1EEE", "Pose-only backward (LVDGS_FLAG_POSE_ONLY), fused loss, no depth gradient (the tracking loop's step)")

# ---- reconciliation with the counters (profiles/traffic.json: SQ_INSTS_VALU of separate rocprofv3 --pmc passes) ----
import json
tj = os.path.join(ROOT, "profiles", "traffic.json")
SURVIVORS = 3_526_611   # tools/fill_diag.py, config 3 (quadrant-test survivors of one backward pass; 487 352 batches, 395 105 of them full)
if os.path.exists(tj):
    t = json.load(open(tj))
    rows = [("full", t.get("cfg3_500k_1920x1080", {}).get("blend_bwd"), full), ("pose-only", t.get("cfg3_500k_1920x1080 (pose)", {}).get("blend_bwd"), pose)]
    print("\n### Against the counters (config 3: %d surviving (quadrant, Gaussian) combinations, splat fill %.4f measured on the GPU)\n" % (SURVIVORS, FILL))
    print("| form | SQ_INSTS_VALU per launch | per survivor | pixel + splat from the table | the rest (quadrant tests, staging, flush, prologue) | SQ_INSTS_SALU | SQ_INSTS_LDS |")
    print("|---|---|---|---|---|---|---|")
    for name, ent, reg in rows:
        if not ent: continue
        vec = lambda r: sum(reg[r][k] for k in ("valu", "valu_trans", "valu_dpp", "valu_xlane"))
        two = vec("pixel pass, straight-line batch of 8") / 8 + vec("splat pass (per batch)") / 8 / FILL
        v = ent["valu_wave_instructions_per_launch"]
        print(f"| {name} | {v / 1e6:.1f} M | {v / SURVIVORS:.1f} | {two:.1f} | {v / SURVIVORS - two:.1f} | {ent.get('salu_wave_instructions_per_launch', 0) / 1e6:.1f} M | {ent.get('lds_wave_instructions_per_launch', 0) / 1e6:.1f} M |")
    print("\nThe splat pass runs at 90.5 % batch fill (DESIGN.md of round 3 said 71 %: never measured; `tools/fill_diag.py` counts survivors and batches in a "
          "`-DLVDGS_DIAG_FILL` build, `tools/carry_model.py` reproduces the figure from the oracle's lists and prices a carry of partial batches across "
          "64-entry rounds at 0.905 -> 0.955 fill, i.e. 1.0 of 54.7 instructions per survivor).")
