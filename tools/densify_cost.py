#!/usr/bin/env python3
"""Where the time of a densification event goes: the sequence of tools/sequence.py run to its last keyframe, then GaussianModel.densify_and_prune's
steps one by one with the device drained before and after each (wall ms).  usage: python tools/densify_cost.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

import lvdgs  # noqa: E402,F401
import sequence as tool  # noqa: E402

dev = torch.device("cuda", 0)
rec, seq = tool.run_sequence(dev, frames=30, refine=0)
be, G, window = seq.backend, seq.gaussians, list(seq.current_window)


def timed(name, fn):
    torch.cuda.synchronize()
    t = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    print(f"  {name:34s} {1e3 * (time.perf_counter() - t):8.3f} ms   (N = {G.get_xyz.shape[0]})")
    return out


for rep in range(3):
    for _ in range(20):
        seq._map(window)
    print(f"event {rep}: N = {G.get_xyz.shape[0]}")
    grads = timed("grads = accum / denom", lambda: (lambda g: (g.__setitem__(g.isnan(), 0.0), g)[1])(G.xyz_gradient_accum / G.denom))
    timed("densify_and_clone", lambda: G.densify_and_clone(grads, be.opt_params.densify_grad_threshold, be.gaussian_extent))
    timed("densify_and_split", lambda: G.densify_and_split(grads, be.opt_params.densify_grad_threshold, be.gaussian_extent))
    prune = timed("prune mask", lambda: (G.get_opacity < be.gaussian_th).squeeze(-1) | (G.max_radii2D > be.size_threshold) | (G.get_scaling.max(dim=1).values > 0.1 * be.gaussian_extent))
    timed("prune_points", lambda: G.prune_points(prune))
    timed("next mapping iteration", lambda: seq._map(window))
    timed("the one after", lambda: seq._map(window))
    timed("whole densify_and_prune (next event)", lambda: G.densify_and_prune(be.opt_params.densify_grad_threshold, be.gaussian_th, be.gaussian_extent, be.size_threshold))
    timed("next mapping iteration", lambda: seq._map(window))
