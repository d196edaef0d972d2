"""Host time of the three phases of a tracking iteration on a scene small enough that the GPU never holds the host up
(other than the pair-count read inside render())."""
import os
import sys
import time
from types import SimpleNamespace

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench  # noqa: E402
import lvdgs  # noqa: E402,F401
from lvdgs import slam_utils, synthetic  # noqa: E402
from lvdgs.gaussian_renderer import render  # noqa: E402

dev = torch.device("cuda", 0)
synthetic.CONFIGS["tmp"] = dict(N=20000, W=320, H=240)
model, cam, g, _ = bench.build_scene("tmp", 0, dev)
pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False)
bg = torch.zeros(3, device=dev)
params = model.parameters()
pose = [cam.cam_rot_delta, cam.cam_trans_delta, cam.exposure_a, cam.exposure_b]
acc = [0.0, 0.0, 0.0, 0.0]
n = 400
for it in range(n + 20):
    t0 = time.perf_counter()
    for p in params + pose:
        p.grad = None
    t1 = time.perf_counter()
    pkg = render(cam, model, pipe, bg)
    t2 = time.perf_counter()
    loss = slam_utils.get_loss_tracking(bench.CONFIG, pkg["render"], pkg["depth"], pkg["opacity"], cam)
    t3 = time.perf_counter()
    loss.backward()
    t4 = time.perf_counter()
    if it >= 20:
        for k, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
            acc[k] += d
torch.cuda.synchronize()
print("host us per step: zero grads %.1f | render %.1f | loss %.1f | backward %.1f | total %.1f"
      % tuple(1e6 * a / n for a in acc + [sum(acc)]))

# ---- time spent inside the C library calls themselves (launch enqueues + the pair-count wait) ----
from lvdgs import _lib  # noqa: E402
L = _lib.lib()
tot = {"lvdgs_forward": 0.0, "lvdgs_backward": 0.0, "lvdgs_photometric_loss_forward": 0.0, "lvdgs_photometric_loss_backward": 0.0}


class Timed:
    def __init__(self, name, fn):
        self.name, self.fn = name, fn

    def __call__(self, *a):
        t = time.perf_counter()
        r = self.fn(*a)
        tot[self.name] += time.perf_counter() - t
        return r


class Proxy:
    def __getattr__(self, k):
        f = getattr(L, k)
        return Timed(k, f) if k in tot else f


_lib._lib = Proxy()
for it in range(n):
    for p in params + pose:
        p.grad = None
    pkg = render(cam, model, pipe, bg)
    loss = slam_utils.get_loss_tracking(bench.CONFIG, pkg["render"], pkg["depth"], pkg["opacity"], cam)
    loss.backward()
torch.cuda.synchronize()
print("inside the library, us per step:", {k: round(1e6 * v / n, 1) for k, v in tot.items()})
