#!/usr/bin/env python3
"""Why do 20 timed steps after a synchronisation read ~10 % slower per step than 200?  Per-step GPU time (event after
every TrackingSession.step) of the steps that follow a torch.cuda.synchronize(), after a long warm-up.
usage: python tools/short_region.py [workload] [steps]"""
import os
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import lvdgs  # noqa: E402,F401
from lvdgs.fast_tracking import TrackingSession  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "cfg3_500k_1920x1080"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 60
dev = torch.device("cuda", 0)
model, cam, _, (N, W, H) = bench.build_scene(workload, 0, dev)
pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False)
s = TrackingSession(cam, model, bench.CONFIG, pipe, torch.zeros(3, device=dev))
for _ in range(500):
    s.step()
for rep in range(3):
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
    host = []
    t0 = time.perf_counter()
    ev[0].record()
    for k in range(K):
        s.step()
        ev[k + 1].record()
        host.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    d = [ev[k].elapsed_time(ev[k + 1]) for k in range(K)]
    print(f"rep {rep}: wall {1e3 * wall / K:.4f} ms/step; GPU ms per step: " + " ".join(f"{x:.3f}" for x in d[:12]) + " ... " + " ".join(f"{x:.3f}" for x in d[-4:]))
    print("        host enqueue done at (ms): " + " ".join(f"{1e3 * x:.2f}" for x in host[:8]) + " ... " + f"{1e3 * host[-1]:.2f}")
