import sys, time, os
sys.path.insert(0, os.getcwd())
import torch, lvdgs
from types import SimpleNamespace
from lvdgs import synthetic, slam_utils
from lvdgs.gaussian_model import GaussianModel
from lvdgs.gaussian_renderer import render
import bench
dev = torch.device("cuda", 0)
for (N, W, H) in [(2000, 64, 64), (500000, 1920, 1080)]:
    synthetic.CONFIGS["tmp"] = dict(N=N, W=W, H=H)
    model, cam, g, _ = bench.build_scene("tmp", 0, dev)
    pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False)
    bg = torch.zeros(3, device=dev)
    params = model.parameters(); pose = [cam.cam_rot_delta, cam.cam_trans_delta, cam.exposure_a, cam.exposure_b]
    def step():
        for p in params + pose: p.grad = None
        pkg = render(cam, model, pipe, bg)
        loss = slam_utils.get_loss_tracking(bench.CONFIG, pkg["render"], pkg["depth"], pkg["opacity"], cam)
        loss.backward()
    for _ in range(10): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(N, W, H, "enqueue per step %.1f us, total per step %.1f us" % ((t1 - t0) / 200 * 1e6, (t2 - t0) / 200 * 1e6))
