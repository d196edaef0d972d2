#!/usr/bin/env python3
"""Where does the workgroup with the longest list spend its time in blend_bwd?  Needs a -DLVDGS_DIAG_PHASES build of the library:
    make -C lvd_gs-slam_amd/csrc OUT=../lib_phases EXTRA=-DLVDGS_DIAG_PHASES
    LVDGS_LIB=lvd_gs-slam_amd/lib_phases/liblvdgs.so python3 tools/phase_diag.py [workload ...]
Prints, per workload, the shader-clock ticks (s_memtime: 100 MHz) wave 0 of workgroup 0 spends per phase, averaged over the steps."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from lvdgs import _lib
from lvdgs.fast_tracking import TrackingSession
from types import SimpleNamespace
L = _lib.lib()
assert hasattr(L, "lvdgs_diag_phases"), "not a -DLVDGS_DIAG_PHASES build"
dev = torch.device("cuda", 0)
names = ["prologue", "staging -> barrier", "test + survivor batches", "wait for the other waves", "flush -> barrier"]
for w in (sys.argv[1:] or ["kitti07_geom", "cfg3_500k_1920x1080"]):
    model, cam, g, (N, W, H) = bench.build_scene(w, 0, dev)
    s = TrackingSession(cam, model, bench.CONFIG, SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False), torch.zeros(3, device=dev), gaussian_gradients=True)
    for _ in range(60):
        s.step()
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 8)()
    L.lvdgs_diag_phases(out, 1)
    for _ in range(50):
        s.step()
    torch.cuda.synchronize()
    L.lvdgs_diag_phases(out, 1)
    v = [int(x) for x in out]
    n = max(v[7], 1)
    tot = sum(v[:5])
    print(f"{w}: launches {v[7]}, rounds {v[5] / n:.1f}, survivors of wave 0 {v[6] / n:.0f}; ticks per launch {tot / n:.0f} (x 10 ns)")
    for k in range(5):
        print(f"   {names[k]:28s} {v[k] / n:9.0f} ticks  {100.0 * v[k] / max(tot, 1):5.1f} %   per round {v[k] / max(v[5], 1):7.1f}")
    # every workgroup of the last launch: timeline and placement
    import numpy as np
    wg = np.zeros((8192, 4), dtype=np.uint64); sv = np.zeros(8192, dtype=np.uint32)
    L.lvdgs_diag_workgroups(wg.ctypes.data_as(C.c_void_p), sv.ctypes.data_as(C.c_void_p))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.savez(os.path.join(ROOT, "gpurun_out", f"wg_diag_{w}.npz"), wg=wg, sv=sv)
