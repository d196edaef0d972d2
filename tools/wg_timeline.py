#!/usr/bin/env python3
"""Summary of a per-workgroup record of blend_bwd written by tools/phase_diag.py (gpurun_out/wg_diag_<workload>.npz, -DLVDGS_DIAG_PHASES build):
the kernel's span, the workgroups alive every 5 us, lifetimes, the work (survivors of the quadrant test) and the end time of every CU.
    python3 tools/wg_timeline.py gpurun_out/wg_diag_kitti07_geom.npz"""
import sys
import numpy as np
d = np.load(sys.argv[1]); wg = d["wg"]; sv = d["sv"].astype(np.int64)
t0 = wg[:, 0].astype(np.int64); t1 = wg[:, 1].astype(np.int64)
act = t0 > 0
t0, t1, sv, hw = t0[act], t1[act], sv[act], wg[act, 3]
base = t0.min()
life = (t1 - t0) / 100.0
print(f"workgroups that walked entries: {act.sum()}; kernel span {(t1.max() - base) / 100:.1f} us; last start {(t0.max() - base) / 100:.1f} us")
print(f"lifetime of a workgroup, us: max {life.max():.1f}  p90 {np.percentile(life, 90):.1f}  median {np.median(life):.1f}")
ts = np.arange(0, int(t1.max() - base), 500)
print("workgroups alive every 5 us:", [int(((t0 - base <= t) & (t1 - base > t)).sum()) for t in ts])
lo = (hw & np.uint64(0xffffffff)).astype(np.int64); xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xf
key = xcc * 1000 + ((lo >> 13) & 7) * 100 + ((lo >> 8) & 0xf)
ids, inv = np.unique(key, return_inverse=True)
S = np.bincount(inv, weights=sv); C = np.bincount(inv)
E = np.array([(t1[inv == i].max() - base) / 100 for i in range(len(ids))])
print(f"CUs seen: {len(ids)}; workgroups per CU min / mean / max {C.min()} / {C.mean():.2f} / {C.max()}; survivors per CU {S.min():.0f} / {S.mean():.0f} / {S.max():.0f}; "
      f"a CU's last workgroup ends at, us: {E.min():.1f} / {E.mean():.1f} / {E.max():.1f}; correlation(survivors, end) {np.corrcoef(S, E)[0, 1]:.2f}")
print(f"survivors (wave x entry) in total: {sv.sum()}")
