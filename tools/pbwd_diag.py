#!/usr/bin/env python3
"""Where do the waves of preprocess_bwd's compacted sweep (large-footprint Gaussians: sum_region_compacted) spend their time?
Needs a -DLVDGS_DIAG_PBWD build of the library:
    make -C lvd_gs-slam_amd/csrc OUT=../lib_pbwd EXTRA=-DLVDGS_DIAG_PBWD
    LVDGS_LIB=$PWD/lvd_gs-slam_amd/lib_pbwd/liblvdgs.so python3 tools/pbwd_diag.py [workload ...]
Prints per workload the constant-rate clock ticks (s_memtime, 10 ns) a wave spends per phase, averaged over waves and launches."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from lvdgs import _lib
from lvdgs.fast_tracking import TrackingSession
from types import SimpleNamespace
L = _lib.lib()
assert hasattr(L, "lvdgs_diag_pbwd"), "not a -DLVDGS_DIAG_PBWD build"
dev = torch.device("cuda", 0)
for w in (sys.argv[1:] or ["surface_100k_1920x1080"]):
    model, cam, g, (N, W, H) = bench.build_scene(w, 0, dev)
    s = TrackingSession(cam, model, bench.CONFIG, SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False), torch.zeros(3, device=dev), gaussian_gradients=True)
    for _ in range(60):
        s.step()
    torch.cuda.synchronize()
    import numpy as np
    rows = np.zeros((8192, 12), dtype=np.uint64)
    L.lvdgs_diag_pbwd(rows.ctypes.data_as(C.c_void_p), 1)
    steps = 50
    for _ in range(steps):
        s.step()
    torch.cuda.synchronize()
    L.lvdgs_diag_pbwd(rows.ctypes.data_as(C.c_void_p), 1)
    v = [int(x) for x in rows.sum(axis=0)]
    waves = max(v[0], 1)
    per_wave = rows[rows[:, 0] > 0, 1].astype(np.float64) / steps
    live = rows[rows[:, 0] > 0].astype(np.float64) / steps
    order = np.argsort(live[:, 1])
    for name, sel in (("slowest 5 % of the waves", order[-max(len(order) // 20, 1):]), ("median 10 %", order[len(order) * 45 // 100: len(order) * 55 // 100])):
        m = live[sel].mean(axis=0)
        print(f"   {name}: sweep {m[1]:.0f} ticks = flags->list {m[2]:.0f} + gather {m[3]:.0f} + sums {m[4]:.0f} + rest {m[1] - m[2] - m[3] - m[4]:.0f}; segments {m[5]:.1f}, passes {m[6]:.1f}, records {m[7]:.0f}; longest lane's trips {m[8]:.0f}, Gaussians summed by the wave {m[9]:.1f} in {m[10]:.0f} ticks")
    print(f"   whole sweep per wave and launch: min {per_wave.min():.0f}, median {np.median(per_wave):.0f}, max {per_wave.max():.0f} ticks")
    print(f"{w}: N {N}, waves on the compacted sweep per launch {v[0] / steps:.0f}; per wave: {v[5] / waves:.2f} segments, {v[6] / waves:.2f} passes, {v[7] / waves:.1f} records")
    print(f"   whole sweep      {v[1] / waves:9.1f} ticks (x 10 ns)")
    for k, name in ((2, "flags -> list"), (3, "gather -> LDS"), (4, "sums")):
        print(f"   {name:16s} {v[k] / waves:9.1f} ticks  {100.0 * v[k] / max(v[1], 1):5.1f} %   per segment {v[k] / max(v[5], 1):7.1f}   per pass {v[k] / max(v[6], 1):7.1f}")
