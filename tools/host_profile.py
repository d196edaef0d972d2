"""Where the host time of one tracking iteration goes (small scene, so that the GPU is never the bottleneck)."""
import cProfile
import os
import pstats
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


def step():
    for p in params + pose:
        p.grad = None
    pkg = render(cam, model, pipe, bg)
    loss = slam_utils.get_loss_tracking(bench.CONFIG, pkg["render"], pkg["depth"], pkg["opacity"], cam)
    loss.backward()


for _ in range(20):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300):
    step()
torch.cuda.synchronize()
print("per step: %.1f us" % ((time.perf_counter() - t0) / 300 * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
