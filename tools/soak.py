#!/usr/bin/env python3
"""Soak run: many tracking iterations and many mapping-window iterations back to back, checking that nothing drifts that
should not -- device memory in use, finite losses and gradients, the library's error state -- and that the iteration
time at the end equals the time at the start.  usage: python tools/soak.py [tracking_steps] [mapping_iterations]"""
import os
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import lvdgs  # noqa: E402,F401
from lvdgs import backend_map  # noqa: E402
from lvdgs.fast_tracking import TrackingSession  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
dev = torch.device("cuda", 0)
pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False)

import gc  # noqa: E402
from lvdgs import _lib  # noqa: E402
_lib.set_gc_policy("process")   # (a benchmark process: one collect + freeze for its life, as every round up to 5 did implicitly; the library's default is scoped)
gc.collect(); gc.freeze()
model, cam, _, (N, W, H) = bench.build_scene("kitti07_geom", 0, dev)
# converged_threshold < 0: the sticky "converged" flag is never raised, so EVERY iteration's pose step is applied (round 3's soak
# converged after a few dozen iterations and soaked render + backward with a no-op optimiser step behind them)
s = TrackingSession(cam, model, bench.CONFIG, pipe, torch.zeros(3, device=dev), converged_threshold=-1.0)
for _ in range(50):
    s.step()
torch.cuda.synchronize()
mem0 = torch.cuda.memory_allocated(dev)
marks = []
t0 = time.perf_counter()
for k in range(T):
    s.step()
    if (k + 1) % (T // 6) == 0:
        torch.cuda.synchronize()
        marks.append((k + 1, time.perf_counter() - t0, float(s.loss), bool(torch.isfinite(s.d_tau).all())))
mem1 = torch.cuda.memory_allocated(dev)
prev = (0, 0.0)
for k, t, loss, ok in marks:
    print(f"tracking: iterations {prev[0]:6d}..{k:6d}: {1e3 * (t - prev[1]) / (k - prev[0]):.4f} ms each, loss {loss:.6f}, pose gradient finite: {ok}")
    prev = (k, t)
applied = s.finish()
print(f"tracking: device memory in use {mem0 / 2**20:.1f} MiB -> {mem1 / 2**20:.1f} MiB; pose steps applied {applied} of {T + 50} enqueued "
      f"(pose-only backward: {s.pose_only}); |T| {float(s.T.norm()):.4f}")
assert mem1 == mem0 and all(ok for *_, ok in marks) and applied == T + 50 and bool(torch.isfinite(s.R).all())

model, _, _, _ = bench.build_scene("kitti07_geom", 0, dev)
# (every keyframe with a static mask -- the reference's default -- unless SOAK_NO_MASKS=1)
be, window = bench.build_window("kitti07_geom", 12, dev, model, n_window=8, masked=os.environ.get("SOAK_NO_MASKS", "0") != "1")
torch.manual_seed(0)
for _ in range(20):
    backend_map.map_window(be, window, iters=1)
torch.cuda.synchronize()
mem0 = torch.cuda.memory_allocated(dev)
prev_t, t0 = 0.0, time.perf_counter()
st = {}
for k in range(M):
    backend_map.map_window(be, window, iters=1, stats=st if (k + 1) % (M // 5) == 0 else None)
    if (k + 1) % (M // 5) == 0:
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
        loss = float(st["losses"][-1])
        finite = all(bool(torch.isfinite(p).all()) for p in be.gaussians.parameters())
        print(f"mapping: iterations ..{k + 1:5d}: {1e3 * (t - prev_t) / (M // 5):.3f} ms each, loss {loss:.6f}, parameters finite: {finite}, "
              f"Gaussians {be.gaussians.get_xyz.shape[0]}")
        assert finite
        prev_t = t
mem1 = torch.cuda.memory_allocated(dev)
print(f"mapping: device memory in use {mem0 / 2**20:.1f} MiB -> {mem1 / 2**20:.1f} MiB")
assert abs(mem1 - mem0) < 64 * 2**20
print("soak ok")
