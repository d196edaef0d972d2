"""Histogram of the tile lists' lengths of a synthetic workload as the HIP path builds them (culled lists): which of the tile sort's
size classes the tiles fall into.     python tools/list_lengths.py surface_100k_1920x1080 [pose_seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import lvdgs  # noqa: E402,F401
import hip_runner as hr  # noqa: E402
from lvdgs import synthetic  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "surface_100k_1920x1080"
pose_seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cfg = synthetic.CONFIGS[name]
g = synthetic.make_workload_gaussians(name, seed=0)
cam = synthetic.make_camera(cfg["W"], cfg["H"], pose_seed=pose_seed, **{k: cfg[k] for k in ("fx", "fy", "cx", "cy") if k in cfg})
f, _ = hr.run_hip(g, cam, cfg["W"], cfg["H"], torch.tensor([0.1, 0.3, 0.2]))
n = f["ranges"][:, 1].astype(np.int64) - f["ranges"][:, 0]
print(f"{name}: {len(n)} tiles, {int(n.sum())} pairs, mean {n.mean():.1f}, median {np.median(n):.0f}, max {n.max()}")
edges = [0, 1, 64, 128, 256, 512, 768, 1024, 1280, 1536, 2048, 4096, 16384, 1 << 30]
for lo, hi in zip(edges[:-1], edges[1:]):
    k = int(((n >= lo) & (n < hi)).sum())
    if k:
        print(f"  [{lo:6d}, {hi:6d}): {k:6d} tiles, {int(n[(n >= lo) & (n < hi)].sum()):9d} pairs")
