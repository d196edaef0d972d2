#!/usr/bin/env python3
"""How much do the views of a mapping iteration gain from running on several HIP streams at once?

A view is ten launches, most of them too small to fill 256 CUs (grouping scans, the tile sort, the tails) or with a
ragged end (blend kernels on a 1848-tile KITTI frame); on one stream they run one after the other.  This tool runs V views
(fast_mapping.MapViewPass: render + mapping loss + backward) on S = 1, 2, 3 streams -- every stream with its own model
copy, buffers and gradient sets, so nothing is shared but the GPU -- and prints the time per view.
usage: python tools/stream_overlap.py [workload] [views]"""
import os
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import lvdgs  # noqa: E402,F401
from lvdgs.fast_mapping import MapViewPass  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "kitti07_geom"
V = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dev = torch.device("cuda", 0)
lanes = []
for k in range(3):
    model, _, _, (N, W, H) = bench.build_scene(workload, 0, dev)
    be, window = bench.build_window(workload, V, dev, model)
    lanes.append((be, MapViewPass(dev), torch.cuda.Stream(dev)))


def run(S, reps):
    main = torch.cuda.current_stream(dev)
    for _ in range(reps):
        start = torch.cuda.Event()
        start.record(main)
        for s in range(S):
            be, vp, stream = lanes[s]
            stream.wait_event(start)
            with torch.cuda.stream(stream):
                for p in be.gaussians.parameters():
                    p.grad = None
                for v in range(s, V, S):
                    vp.run(be, be.viewpoints[v + 1])
        for s in range(S):
            main.wait_stream(lanes[s][2])


for S in (1, 2, 3, 1, 2, 3):
    run(S, 2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(S, 10)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"{workload} N={N} {W}x{H}: {V} views on {S} stream(s): {1e3 * dt:.3f} ms per iteration, {1e6 * dt / V:.1f} us per view")
