#!/usr/bin/env python3
"""Which PyTorch operators a fused mapping-window iteration still issues (torch.profiler, CPU side with stacks): the launches
that are not the library's.  usage: python tools/map_ops.py [workload]  (MAP_BENCH_MASKED=1: keyframes with a static mask)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402
import lvdgs  # noqa: E402,F401
from lvdgs import backend_map  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "kitti07_geom"
dev = torch.device("cuda", 0)
model, _, _, (N, W, H) = bench.build_scene(workload, 0, dev)
be, window = bench.build_window(workload, 12, dev, model, n_window=8, masked=os.environ.get("MAP_BENCH_MASKED", "0") == "1")
for _ in range(5):
    backend_map.map_window(be, window, iters=1)
torch.cuda.synchronize()
ITERS = 10
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(ITERS):
        backend_map.map_window(be, window, iters=1)
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith("aten::") and e.key not in ("aten::empty", "aten::slice", "aten::as_strided", "aten::view", "aten::select")]
rows.sort(key=lambda e: -e.count)
for e in rows[:60]:
    print(f"{e.key:28s} calls/it {e.count / ITERS:6.1f}  gpu us/it {e.self_device_time_total / ITERS:8.1f}  cpu us/it {e.self_cpu_time_total / ITERS:8.1f}  {str(e.input_shapes)[:150]}")
