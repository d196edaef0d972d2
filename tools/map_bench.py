#!/usr/bin/env python3
"""One GPU, BASELINE.json configs[3]'s workload: a mapping iteration over a window of 8 keyframes + 2 random older ones
at KITTI-07's geometry (reference configs/mono/KITTI/07.yaml:8-18, base_config.yaml:37,51; utils/slam_backend.py:167-390),
through backend_map.map_window.  Prints iterations/s, renders/s and the GPU-busy share.
usage: python tools/map_bench.py [workload] [n_window] [n_older]"""
import os
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import lvdgs  # noqa: E402,F401
from lvdgs import _lib, backend_map  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "kitti07_geom"
n_window = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n_older = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = torch.device("cuda", 0)
model, _, _, (N, W, H) = bench.build_scene(workload, 0, dev)
# MAP_BENCH_MASKED=1: every keyframe carries a static_mask (the reference's default: utils/slam_frontend.py:1218,1429-1433)
masked = os.environ.get("MAP_BENCH_MASKED", "0") == "1"
be, window = bench.build_window(workload, n_window + n_older, dev, model, n_window=n_window, masked=masked)   # the newest n_window keyframes; 1..n_older are the older ones
# MAP_BENCH_FUSED_ONLY=1 (under rocprofv3 --kernel-trace --stats): 28 fused iterations and nothing else, so that the trace's
# kernel time / 28 is the GPU-busy time of one iteration without this script's own event timing
for fused in ((True,) if os.environ.get("MAP_BENCH_FUSED_ONLY") else (True, False)):
    for _ in range(3):
        backend_map.map_window(be, window, iters=1, fused=fused)
    torch.cuda.synchronize()
    n = 20
    t0 = time.perf_counter()
    backend_map.map_window(be, window, iters=n, fused=fused)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    _lib.profile_reset(); _lib.profile_enable(True)
    backend_map.map_window(be, window, iters=5, fused=fused)
    torch.cuda.synchronize()
    _lib.profile_enable(False)
    k = _lib.profile_read()
    lib_ms = sum(ms for _, ms in k.values()) / 5
    print(f"{workload} N={N} {W}x{H}, window {n_window} + 2 random, keyframes {'WITH' if masked else 'without'} a static mask, fused={fused}: {1e3 * dt:.2f} ms per iteration = {1 / dt:.1f} it/s = "
          f"{(n_window + 2) / dt:.0f} renders+backwards/s; lvdgs kernels {lib_ms:.2f} ms per iteration ({100 * lib_ms / (1e3 * dt):.0f} % of the wall time)")
