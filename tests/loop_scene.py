"""The toy scene the loop fixtures are generated on and replayed on (tests/golden/make_loop_golden.py,
tests/test_loop_golden.py, tests/test_gpu_loop_golden.py, tests/test_backend_map_gloo.py).

Deterministic and built on the CPU, whatever the device it is then moved to: a seeded set of "true" Gaussians, seven
keyframe cameras looking at it from slightly different poses, ground-truth images / depths rendered from the true map
with the dense float64 renderer (tests/dense_render.py), and the map to be optimised: the true one, perturbed.
Keyframe 2 carries a ``static_mask`` (the masked branch of the mapping loss, reference utils/slam_backend.py:196-261).
"""
import math
from types import SimpleNamespace

import numpy as np
import torch

from lvdgs import synthetic
from lvdgs.gaussian_model import GaussianModel
from lvdgs.graphics_utils import focal2fov, getProjectionMatrix2
from lvdgs.pose_utils import SE3_exp

W, H, N_TRUE = 64, 48, 140
N_CAMERAS = 7
WINDOW = [6, 5, 4, 3]       # newest first, as the front end orders its window; keyframes 0..2 are the "older" ones
MAP_ITERS = 6

# small iteration counts / intervals so that densification, pruning and an opacity reset all happen inside the fixture
CONFIG_OVERRIDES = {
    "Training": {"init_itr_num": 12, "init_gaussian_update": 5, "init_gaussian_reset": 9, "init_gaussian_th": 0.005,
                 "init_gaussian_extent": 30, "gaussian_update_every": 4, "gaussian_update_offset": 2, "gaussian_reset": 5,
                 "gaussian_th": 0.4, "gaussian_extent": 1.0, "size_threshold": 30, "window_size": 4, "pose_window": 3,
                 "prune_mode": "slam", "prune_num": 1, "tracking_itr_num": 15, "depth_lambda": 0.1},
    "Dataset": {"depth_loss": True},
    "Results": {"save_dir": "/tmp/lvdgs_loop_golden", "save_results": False, "save_trj": False, "save_trj_kf_intv": 5,
                "use_gui": False},
}

import json as _json
import os as _os

# optimiser settings: the opt_params block of the merged KITTI-07 config (configs/mono/KITTI/base_config.yaml:58-76)
OPT = _json.load(open(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "golden", "config_07.json")))["opt_params"]


def loop_config():
    """The merged KITTI-07 config with the fixture's small iteration counts."""
    cfg = _json.load(open(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "golden", "config_07.json")))
    cfg["Training"]["monocular"] = cfg["Dataset"]["sensor_type"] == "monocular"  # set by the absent slam.py entry point
    for sec, kv in CONFIG_OVERRIDES.items():
        cfg.setdefault(sec, {}).update(kv)
    return cfg


def _camera(camera_cls, uid, image, mono_depth, pose, device):
    fx = fy = float(W)
    cx, cy = W / 2.0, H / 2.0
    proj = getProjectionMatrix2(znear=0.01, zfar=100.0, fx=fx, fy=fy, cx=cx, cy=cy, W=W, H=H).transpose(0, 1)
    cam = camera_cls(uid, image.to(device), None, mono_depth, torch.eye(4), proj.to(device), fx, fy, cx, cy,
                     focal2fov(fx, W), focal2fov(fy, H), H, W, device=device)
    cam.update_RT(pose[:3, :3].to(device), pose[:3, 3].to(device))
    cam.grad_mask = torch.ones(1, H, W, dtype=torch.bool, device=device)
    return cam


def make_keyframe_optimizer(viewpoints, window, cfg):
    """The Adam the back end builds for a new keyframe (reference utils/slam_backend.py:545-598)."""
    opt_params = []
    frames_to_optimize = cfg["Training"]["pose_window"]
    for cam_idx in range(len(window)):
        if window[cam_idx] == 0:
            continue
        vp = viewpoints[window[cam_idx]]
        if cam_idx < frames_to_optimize:
            opt_params.append({"params": [vp.cam_rot_delta], "lr": cfg["Training"]["lr"]["cam_rot_delta"] * 0.5, "name": f"rot_{vp.uid}"})
            opt_params.append({"params": [vp.cam_trans_delta], "lr": cfg["Training"]["lr"]["cam_trans_delta"] * 0.5, "name": f"trans_{vp.uid}"})
        opt_params.append({"params": [vp.exposure_a], "lr": 0.01, "name": f"exposure_a_{vp.uid}"})
        opt_params.append({"params": [vp.exposure_b], "lr": 0.01, "name": f"exposure_b_{vp.uid}"})
    return torch.optim.Adam(opt_params)


def build_scene(device="cpu", camera_cls=None, n_cameras=N_CAMERAS, window=None):
    """``n_cameras`` / ``window``: a larger set of keyframes than the fixtures' seven (the eight-rank test of the reference's
    8 + 2 window); the defaults are the scene the golden fixtures were generated on, draw for draw."""
    from dense_render import dense_render
    N_CAMERAS = n_cameras   # (shadows the module's constant below)
    if camera_cls is None:
        from lvdgs.camera_utils import Camera as camera_cls
    g = synthetic.make_gaussians(N_TRUE, W, H, seed=21, r_min=3.0, r_max=9.0, z_min=2.0, z_max=6.0)
    with torch.no_grad():
        g["opacities"].clamp_(min=0.5)
    truth = GaussianModel.from_activated(g["means3D"], g["scales"], g["rotations"], g["opacities"], shs=g["shs"], device="cpu")
    pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False)
    bg = torch.zeros(3)
    gen = torch.Generator().manual_seed(77)
    poses = [torch.eye(4)] + [SE3_exp(torch.randn(6, generator=gen) * torch.tensor([0.05, 0.05, 0.05, 0.02, 0.02, 0.02]))
                              for _ in range(N_CAMERAS)]
    from lvdgs.camera_utils import Camera as _OwnCamera
    cams = []
    for i in range(N_CAMERAS + 1):   # the last one is the frame to be tracked
        probe = _camera(_OwnCamera, i, torch.zeros(3, H, W), None, poses[i], "cpu")
        with torch.no_grad():
            pkg = dense_render(probe, truth, pipe, bg)
        img = pkg["render"].clamp(0, 1).contiguous()
        opac = pkg["opacity"][0]
        depth = torch.where(opac > 0.5, pkg["depth"][0] / opac.clamp(min=1e-3), torch.zeros_like(opac))
        cams.append((img, depth.numpy().astype(np.float32)))
    # the estimated poses start slightly off the true ones (so the pose gradients are alive)
    off = [SE3_exp(torch.randn(6, generator=gen) * 0.004) @ poses[i] for i in range(N_CAMERAS + 1)]
    off[0] = poses[0]
    cameras = [_camera(camera_cls, i, cams[i][0], cams[i][1], off[i], device) for i in range(N_CAMERAS)]
    mask = torch.ones(H, W, dtype=torch.bool)
    mask[10:30, 20:44] = False
    cameras[4].static_mask = mask.to(device)
    track_camera = _camera(camera_cls, N_CAMERAS, cams[N_CAMERAS][0], cams[N_CAMERAS][1], poses[N_CAMERAS], device)
    # the map under optimisation: the truth, perturbed
    noise = lambda *s, k: torch.randn(*s, generator=gen) * k
    model = GaussianModel.from_activated(
        g["means3D"] + noise(N_TRUE, 3, k=0.02), g["scales"] * torch.exp(noise(N_TRUE, 3, k=0.15)),
        g["rotations"] + noise(N_TRUE, 4, k=0.05), (g["opacities"] * 0.95).clamp(0.3, 0.95),
        shs=g["shs"] + noise(*g["shs"].shape, k=0.1), device=device)
    model.unique_kfIDs = (torch.arange(N_TRUE) % N_CAMERAS).to(torch.int32)   # which keyframe seeded each Gaussian
    model.config = {"Dataset": {"sensor_type": "monocular"}}
    model.init_lr(6.0)
    model.training_setup(OPT)
    return dict(gaussians=model, cameras=cameras, track_camera=track_camera, track_mono_depth=cams[N_CAMERAS][1],
                background=bg.to(device), pipe=pipe, window=list(WINDOW if window is None else window), map_iters=MAP_ITERS,
                make_keyframe_optimizer=make_keyframe_optimizer, opt=dict(OPT))
