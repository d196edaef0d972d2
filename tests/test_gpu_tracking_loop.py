"""The pose-optimisation loop of FrontEnd.tracking (utils/slam_frontend.py:1467-1533) end to end on the GPU:
render -> get_loss_tracking -> backward -> Adam on (cam_rot_delta, cam_trans_delta, exposure) -> update_pose.
A perturbed camera must walk back to the pose the target image was rendered from; that only happens if the
rasterizer's dL/dtau has the sign, scale and (rho, theta) ordering update_pose (pinned by the reference's
golden vectors) expects."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CONFIG = {"Training": {"monocular": True, "rgb_boundary_threshold": 0.01, "alpha": 0.98,
                       "lr": {"cam_rot_delta": 0.003, "cam_trans_delta": 0.001}, "edge_threshold": 1.1},
          "Dataset": {"type": "KITTI"}}


def _scene(W, H, n=6000):
    from lvdgs import synthetic
    from lvdgs.gaussian_model import GaussianModel
    g = synthetic.make_gaussians(n, W, H, seed=11, r_min=4.0, r_max=14.0, z_min=2.0, z_max=8.0)
    with torch.no_grad():
        g["opacities"].clamp_(min=0.5)
    return GaussianModel.from_activated(g["means3D"], g["scales"], g["rotations"], g["opacities"], shs=g["shs"])


def _camera(W, H, image, w2c):
    from lvdgs.camera_utils import Camera
    from lvdgs.graphics_utils import focal2fov, getProjectionMatrix2
    fx = fy = float(W)
    proj = getProjectionMatrix2(znear=0.01, zfar=100.0, fx=fx, fy=fy, cx=W / 2.0, cy=H / 2.0, W=W, H=H).transpose(0, 1)
    cam = Camera(1, image, None, None, torch.eye(4), proj.cuda(), fx, fy, W / 2.0, H / 2.0, focal2fov(fx, W), focal2fov(fy, H), H, W,
                 device="cuda")
    cam.update_RT(w2c[:3, :3].cuda(), w2c[:3, 3].cuda())
    cam.grad_mask = torch.ones(1, H, W, dtype=torch.bool, device="cuda")
    return cam


def _pose_error(cam, w2c_true):
    from lvdgs.graphics_utils import getWorld2View2
    cur = getWorld2View2(cam.R, cam.T).double().cpu()
    rel = cur @ torch.linalg.inv(w2c_true.double())
    ang = float(torch.acos(((torch.trace(rel[:3, :3]) - 1) / 2).clamp(-1, 1)))
    return float(rel[:3, 3].norm()), ang


@pytest.mark.parametrize("tau", [[0.03, -0.02, 0.04, 0.0, 0.0, 0.0], [0.0, 0.0, 0.0, 0.01, -0.015, 0.008],
                                 [0.02, 0.03, -0.03, -0.008, 0.01, 0.012]])
def test_tracking_recovers_a_perturbed_pose(tau):
    from lvdgs.gaussian_renderer import render
    from lvdgs.pose_utils import SE3_exp, update_pose
    from lvdgs.slam_utils import get_loss_tracking, get_median_depth
    W, H = 320, 200
    model = _scene(W, H)
    pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False)
    bg = torch.zeros(3, device="cuda")
    w2c_true = SE3_exp(torch.tensor([0.1, -0.05, 0.2, 0.02, 0.03, -0.01]))
    with torch.no_grad():
        target = render(_camera(W, H, torch.zeros(3, H, W, device="cuda"), w2c_true), model, pipe, bg)["render"].clamp(0, 1)
    w2c_start = SE3_exp(torch.tensor(tau)) @ w2c_true
    cam = _camera(W, H, target, w2c_start)
    t0, a0 = _pose_error(cam, w2c_true)
    opt = torch.optim.Adam([
        {"params": [cam.cam_rot_delta], "lr": CONFIG["Training"]["lr"]["cam_rot_delta"]},
        {"params": [cam.cam_trans_delta], "lr": CONFIG["Training"]["lr"]["cam_trans_delta"]},
        {"params": [cam.exposure_a], "lr": 0.01}, {"params": [cam.exposure_b], "lr": 0.01}])
    losses = []
    for it in range(150):
        pkg = render(cam, model, pipe, bg)
        opt.zero_grad()
        loss = get_loss_tracking(CONFIG, pkg["render"], pkg["depth"], pkg["opacity"], cam)
        loss.backward()
        with torch.no_grad():
            opt.step()
            converged = update_pose(cam)
        losses.append(float(loss.detach()))
        assert float(cam.cam_rot_delta.detach().abs().sum()) == 0.0 and float(cam.cam_trans_delta.detach().abs().sum()) == 0.0  # pose_utils.py:85-86
        if converged:
            break
    t1, a1 = _pose_error(cam, w2c_true)
    assert losses[-1] < 0.35 * losses[0], (losses[0], losses[-1])
    # Adam takes lr-sized steps (0.003 rad, 0.001 m) until update_pose reports convergence: allow that much residue
    assert t1 < 0.3 * t0 + 3e-3 and a1 < 0.3 * a0 + 5e-3, ((t0, a0), (t1, a1))
    assert abs(float(cam.exposure_a.detach())) < 0.05 and abs(float(cam.exposure_b.detach())) < 0.05
    med = get_median_depth(pkg["depth"], pkg["opacity"])
    assert 2.0 < float(med) < 8.0 and np.isfinite(losses).all()
