#!/usr/bin/env python3
"""Tracking iterations/s (pose-only and full backward) of a few workloads on a TrackingSession: the number behind same-box A/B runs of
library builds (LVDGS_LIB=<another build>) and environment knobs.  usage: python tools/track_ab.py [workload ...]"""
import os
import sys
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from lvdgs.fast_tracking import TrackingSession  # noqa: E402

dev = torch.device("cuda", 0)
pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False)
tag = os.environ.get("LVDGS_LIB", "default library")
for w in (sys.argv[1:] or ["kitti07_geom", "cfg2_100k_640x480", "cfg3_500k_1920x1080"]):
    model, cam, _, _ = bench.build_scene(w, 0, dev)
    for full in (False, True):
        s = TrackingSession(cam, model, bench.CONFIG, pipe, torch.zeros(3, device=dev), gaussian_gradients=full)
        bench.time_session(s, 200, 200)
        rates = [bench.time_session(s, 50, 400) for _ in range(3)]
        print(f"{tag}: {w} {'full' if full else 'pose-only'}: {max(rates):.1f} it/s (runs {rates})", flush=True)
