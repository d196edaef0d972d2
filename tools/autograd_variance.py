"""Is the autograd-API tracking step's rate stable?  Ten rounds of 50 iterations in one process, with the caching
allocator's device-allocation count and the rasterizer's overflow re-runs per round.
    python tools/autograd_variance.py [workload] [freeze]
(Found with it: one stall of about 40 ms some 50-100 iterations in -- a full garbage collection over the interpreter's
heap with torch imported; with `freeze` it does not happen.)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
import lvdgs  # noqa: F401
from lvdgs import rasterizer, slam_utils
from lvdgs.gaussian_renderer import render

workload = sys.argv[1] if len(sys.argv) > 1 else "cfg3_500k_1920x1080"
dev = torch.device("cuda", 0)
from types import SimpleNamespace
model, cam, _, (N, W, H) = bench.build_scene(workload, 0, dev)
pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False)
bg = torch.zeros(3, device=dev)
params = model.parameters()
pose_params = [cam.cam_rot_delta, cam.cam_trans_delta, cam.exposure_a, cam.exposure_b]
rasterizer.KEEP_DEBUG_STATE = True


def step():
    for p in params + pose_params:
        p.grad = None
    pkg = render(cam, model, pipe, bg)
    slam_utils.get_loss_tracking(bench.CONFIG, pkg["render"], pkg["depth"], pkg["opacity"], cam).backward()
    return bool(rasterizer._DEBUG_LAST.get("overflowed"))


for _ in range(3):
    step()
if len(sys.argv) > 2 and sys.argv[2] == "freeze":   # everything built so far leaves the collector's generations
    import gc
    gc.collect()
    gc.freeze()
import gc as _gc
_events = []
def _on_gc(phase, info):
    if phase == "start":
        _events.append([info["generation"], time.perf_counter(), None])
    else:
        _events[-1][2] = time.perf_counter() - _events[-1][1]
_gc.callbacks.append(_on_gc)

for r in range(10):
    torch.cuda.synchronize()
    a0 = torch.cuda.memory_stats()["num_device_alloc"]
    del _events[:]
    t0 = time.perf_counter()
    over, per = 0, []
    for _ in range(50):
        t = time.perf_counter()
        over += step()
        per.append(time.perf_counter() - t)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    slow = sorted(range(50), key=lambda i: -per[i])[:2]
    gcs = ", ".join(f"gen{g} {1e3 * d:.1f} ms" for g, _, d in _events if d and d > 1e-3)
    print(f"round {r}: {50 / dt:7.1f} it/s (host loop alone {50 / t_host:7.1f}) overflow re-runs {over} "
          f"device allocations {torch.cuda.memory_stats()['num_device_alloc'] - a0}; slowest iterations "
          + ", ".join(f"#{i} {1e3 * per[i]:.1f} ms" for i in slow) + f"; collections over 1 ms: {gcs or 'none'}")
