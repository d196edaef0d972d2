#!/usr/bin/env python3
"""How full are blend_bwd's splat batches?  Needs a -DLVDGS_DIAG_FILL build of the library:
    make -C lvd_gs-slam_amd/csrc OUT=../lib_diag EXTRA=-DLVDGS_DIAG_FILL
    LVDGS_LIB=lvd_gs-slam_amd/lib_diag/liblvdgs.so python3 tools/fill_diag.py [workload ...]
Prints, per workload, the survivors of the quadrant test, the splat batches (8 slots each) and the fill = survivors / (8 x batches)
of one tracking iteration's backward blend pass."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from lvdgs import _lib
from lvdgs.fast_tracking import TrackingSession
from types import SimpleNamespace
L = _lib.lib()
assert hasattr(L, "lvdgs_diag_fill"), "not a -DLVDGS_DIAG_FILL build"
dev = torch.device("cuda", 0)
for w in (sys.argv[1:] or ["cfg3_500k_1920x1080", "kitti07_geom", "surface_100k_1920x1080"]):
    model, cam, g, (N, W, H) = bench.build_scene(w, 0, dev)
    s = TrackingSession(cam, model, bench.CONFIG, SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False), torch.zeros(3, device=dev))
    s.step(); torch.cuda.synchronize()
    out = (C.c_ulonglong * 4)()
    L.lvdgs_diag_fill(out, 1)
    s.step(); torch.cuda.synchronize()
    L.lvdgs_diag_fill(out, 1)
    surv, batches, full, rounds = (int(x) for x in out)
    print(f"{w}: pairs {s.num_rendered} survivors {surv} ({surv / max(4 * s.num_rendered, 1):.3f} of quadrant x pair) batches {batches} full {full} "
          f"fill {surv / max(8 * batches, 1):.4f} (wave, round) with survivors {rounds} survivors per such round {surv / max(rounds, 1):.2f}")
