#!/usr/bin/env python3
"""The one-GPU 8 + 2 mapping window of a workload, masked and unmasked: ms per iteration (as bench.py's config.side measures it).
usage: python tools/window_bench.py [workload ...]   (LVDGS_MAX_BATCH_TILES / LVDGS_MAP_BATCH / LVDGS_MAP_FWD_BATCH: A/B knobs)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import lvdgs  # noqa: E402,F401
from lvdgs import backend_map  # noqa: E402

dev = torch.device("cuda", 0)
for w in (sys.argv[1:] or ["kitti07_geom", "cfg3_500k_1920x1080"]):
    for masked in (False, True):
        torch.manual_seed(0)
        model, _, _, (N, W, H) = bench.build_scene(w, 0, dev)
        backend, window = bench.build_window(w, 12, dev, model, n_window=8, masked=masked)
        for _ in range(8):
            backend_map.map_window(backend, window, iters=1)
        torch.cuda.synchronize()
        iters = 40 if N <= 200_000 else (25 if N <= 500_000 else 12)
        best = 1e9
        for _ in range(3):
            t = time.perf_counter()
            for _ in range(iters):
                backend_map.map_window(backend, window, iters=1)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t) / iters)
        print(f"{w} masked={masked} batch={getattr(backend, '_lvdgs_window_batch', None) is not None}: {1e3 * best:.3f} ms per iteration", flush=True)
        del backend, model
        torch.cuda.empty_cache()
