#!/usr/bin/env python3
"""Where do the waves of the two-pass blend_bwd spend their cycles?  Needs the diagnostic build:

    make -C lvd_gs-slam_amd/csrc clean && make -C lvd_gs-slam_amd/csrc -j8 EXTRA=-DLVDGS_STAMP && python tools/stamp_blend.py

Prints, per section, the share of the waves' resident cycles (s_memtime stamps summed over all waves)."""
import ctypes as C
import os
import sys
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import lvdgs  # noqa: E402,F401
import bench  # noqa: E402
from lvdgs import _lib, slam_utils  # noqa: E402
from lvdgs.gaussian_renderer import render  # noqa: E402

dev = torch.device("cuda", 0)
model, cam, _, (N, W, H) = bench.build_scene(os.environ.get("LVDGS_BENCH_WORKLOAD", "cfg3_500k_1920x1080"), 0, dev)
pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False)
bg = torch.zeros(3, device=dev)
L = _lib.lib()
out = (C.c_ulonglong * 12)()


def step():
    for p in model.parameters():
        p.grad = None
    pkg = render(cam, model, pipe, bg)
    slam_utils.get_loss_tracking(bench.CONFIG, pkg["render"], pkg["depth"], pkg["opacity"], cam).backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
assert L.lvdgs_debug_stamps(out) == 0
for _ in range(5):
    step()
torch.cuda.synchronize()
assert L.lvdgs_debug_stamps(out) == 0
v = list(out)
names = ["barrier after staging", "pixel pass", "splat pass", "barrier after the passes", "flush", "staging (wait for records, LDS writes)",
         "first barrier", "culling", "prologue (pixel loads, first records)", "-"]
total, waves = v[10], v[11]
print(f"{waves} waves, {total / waves:.0f} cycles per wave resident")
for n, x in zip(names, v[:10]):
    print(f"  {n:40s} {100.0 * x / total:5.1f} %   {x / waves:9.0f} cycles per wave")
print(f"  {'everything else':40s} {100.0 * (total - sum(v[:10])) / total:5.1f} %")
