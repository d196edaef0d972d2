#!/usr/bin/env python3
"""Generate golden input/output vectors from the reference's importable Python.

Run in the authoring container only (needs /root/reference, never on the GPU box):

    python -B tests/golden/make_golden.py

It imports utils.pose_utils / utils.slam_utils / utils.config_utils from the
reference checkout (read-only, `-B` so no bytecode is written there), feeds them
seeded inputs and stores inputs + outputs as small .npz/.json files next to this
script.  Only data is written: no reference source travels.

Fixtures:
  se3_exp.npz        utils/pose_utils.py:23-68   SO3_exp / V / SE3_exp, both angle branches
  update_pose.npz    utils/pose_utils.py:70-87   update_pose with a stub camera
  loss_tracking.npz  utils/slam_utils.py:42-79   get_loss_tracking (+ grads)
  loss_mapping.npz   utils/slam_utils.py:82-121  get_loss_mapping (+ grads)
  median_depth.npz   utils/slam_utils.py:124-134 get_median_depth
  config_07.json     utils/config_utils.py:4-50  merged KITTI-07 config
"""
import json
import os
import sys

import numpy as np
import torch

REF = os.environ.get("LVDGS_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)

import utils.config_utils as ref_cfg  # noqa: E402
import utils.pose_utils as ref_pose  # noqa: E402
import utils.slam_utils as ref_slam  # noqa: E402


class _CudaIsCpu(torch.Tensor):
    """Tensor whose .cuda() returns itself (the reference hard-codes .cuda(),
    utils/slam_utils.py:54,96,111)."""

    @staticmethod
    def wrap(t):
        t = t.as_subclass(_CudaIsCpu)
        return t

    def cuda(self, *a, **k):
        return self.as_subclass(torch.Tensor)


class _StubCam:
    pass


def gen_se3():
    g = torch.Generator().manual_seed(1)
    taus = []
    for i in range(64):
        rho = torch.randn(3, generator=g, dtype=torch.float64)
        th = torch.randn(3, generator=g, dtype=torch.float64)
        th = th / th.norm()
        # magnitudes spanning both branches of pose_utils.py:30,46
        mag = [0.0, 1e-9, 1e-7, 5e-6, 0.99e-5, 1.01e-5, 1e-4, 1e-3, 1e-2, 0.1, 0.5,
               1.0, 2.0, 3.0, 3.14, 6.0][i % 16]
        taus.append(torch.cat([rho * (0.1 if i < 32 else 2.0), th * mag]))
    taus = torch.stack(taus)
    out = {"tau": taus.numpy()}
    for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
        so3, vv, se3 = [], [], []
        for t in taus.to(dt):
            so3.append(ref_pose.SO3_exp(t[3:]).numpy())
            vv.append(ref_pose.V(t[3:]).numpy())
            se3.append(ref_pose.SE3_exp(t).numpy())
        out["so3_" + tag] = np.stack(so3)
        out["v_" + tag] = np.stack(vv)
        out["se3_" + tag] = np.stack(se3)
    np.savez(os.path.join(HERE, "se3_exp.npz"), **out)


def gen_update_pose():
    g = torch.Generator().manual_seed(2)
    Rs, Ts, rhos, thetas, Rn, Tn, conv = [], [], [], [], [], [], []
    for i in range(24):
        T0 = ref_pose.SE3_exp(torch.randn(6, generator=g))
        cam = _StubCam()
        cam.R, cam.T = T0[:3, :3].clone(), T0[:3, 3].clone()
        scale = [1e-2, 1e-3, 2e-5, 1e-6][i % 4]
        cam.cam_trans_delta = torch.nn.Parameter(torch.randn(3, generator=g) * scale)
        cam.cam_rot_delta = torch.nn.Parameter(torch.randn(3, generator=g) * scale)

        def update_RT(R, t, cam=cam):
            cam.R, cam.T = R, t

        cam.update_RT = update_RT
        Rs.append(cam.R.numpy().copy()); Ts.append(cam.T.numpy().copy())
        rhos.append(cam.cam_trans_delta.detach().numpy().copy())
        thetas.append(cam.cam_rot_delta.detach().numpy().copy())
        c = ref_pose.update_pose(cam)
        assert float(cam.cam_rot_delta.detach().abs().sum()) == 0.0
        assert float(cam.cam_trans_delta.detach().abs().sum()) == 0.0
        Rn.append(cam.R.detach().numpy().copy()); Tn.append(cam.T.detach().numpy().copy()); conv.append(bool(c))
    np.savez(os.path.join(HERE, "update_pose.npz"), R=np.stack(Rs), T=np.stack(Ts),
             rho=np.stack(rhos), theta=np.stack(thetas), R_new=np.stack(Rn),
             T_new=np.stack(Tn), converged=np.array(conv))


def _loss_inputs(seed, H=24, W=40):
    g = torch.Generator().manual_seed(seed)
    image = torch.rand(3, H, W, generator=g)
    depth = torch.rand(1, H, W, generator=g) * 10
    opacity = torch.rand(1, H, W, generator=g)
    gt = torch.rand(3, H, W, generator=g)
    gt[:, :3, :5] = 0.0  # exercise rgb_boundary_threshold
    grad_mask = torch.rand(1, H, W, generator=g) > 0.4
    mono = (torch.rand(H, W, generator=g) * 10).numpy().astype(np.float32)
    mono[5:8, 7:12] = 0.0  # exercise gt_depth > 0.01
    a = torch.tensor([0.13]); b = torch.tensor([-0.04])
    return image, depth, opacity, gt, grad_mask, mono, a, b


def _cfg(monocular, depth_loss, alpha=0.98):
    return {"Training": {"monocular": monocular, "rgb_boundary_threshold": 0.01, "alpha": alpha},
            "Dataset": {"depth_loss": depth_loss}}


def gen_losses():
    image, depth, opacity, gt, grad_mask, mono, a, b = _loss_inputs(3)
    base = dict(image=image.numpy(), depth=depth.numpy(), opacity=opacity.numpy(), gt=gt.numpy(),
                grad_mask=grad_mask.numpy(), mono_depth=mono, exposure_a=a.numpy(), exposure_b=b.numpy())

    def run(fn, cfg, **kw):
        img = image.clone().requires_grad_(True)
        dep = depth.clone().requires_grad_(True)
        opa = opacity.clone().requires_grad_(True)
        vp = _StubCam()
        vp.original_image = _CudaIsCpu.wrap(gt.clone())
        vp.grad_mask = grad_mask
        vp.mono_depth = mono
        vp.exposure_a = torch.nn.Parameter(a.clone())
        vp.exposure_b = torch.nn.Parameter(b.clone())
        if fn == "tracking":
            loss = ref_slam.get_loss_tracking(cfg, img, dep, opa, vp)
        else:
            loss = ref_slam.get_loss_mapping(cfg, img, vp, depth=dep, **kw)
        loss.backward()
        z = lambda t: (t.grad if t.grad is not None else torch.zeros_like(t)).numpy()
        return dict(loss=loss.detach().numpy(), d_image=z(img), d_depth=z(dep), d_opacity=z(opa),
                    d_a=z(vp.exposure_a), d_b=z(vp.exposure_b))

    trk = dict(base)
    for name, cfg in (("mono_depthloss", _cfg(True, True)), ("mono", _cfg(True, False)),
                      ("rgbd", _cfg(False, False))):
        for k, v in run("tracking", cfg).items():
            trk[f"{name}.{k}"] = v
    np.savez(os.path.join(HERE, "loss_tracking.npz"), **trk)

    mp = dict(base)
    for name, cfg, kw in (("mono_monodepth", _cfg(True, True), dict(monodepth=True)),
                          ("mono_nodepth", _cfg(True, True), dict(monodepth=False)),
                          ("init", _cfg(True, True), dict(initialization=True)),
                          ("rgbd", _cfg(False, False), dict())):
        for k, v in run("mapping", cfg, **kw).items():
            mp[f"{name}.{k}"] = v
    np.savez(os.path.join(HERE, "loss_mapping.npz"), **mp)


def gen_median():
    g = torch.Generator().manual_seed(4)
    depth = torch.rand(1, 30, 44, generator=g) * 20 - 2
    opacity = torch.rand(1, 30, 44, generator=g) * 0.1 + 0.9
    med = ref_slam.get_median_depth(depth, opacity)
    med2, std2, valid2 = ref_slam.get_median_depth(depth, opacity, return_std=True)
    np.savez(os.path.join(HERE, "median_depth.npz"), depth=depth.numpy(), opacity=opacity.numpy(),
             median=med.numpy(), std=std2.numpy(), valid=valid2.numpy())


def gen_config():
    cwd = os.getcwd()
    os.chdir(REF)  # inherit_from is relative (configs/mono/KITTI/07.yaml:1)
    try:
        cfg = ref_cfg.load_config("configs/mono/KITTI/07.yaml")
    finally:
        os.chdir(cwd)
    with open(os.path.join(HERE, "config_07.json"), "w") as f:
        json.dump(cfg, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    gen_se3(); gen_update_pose(); gen_losses(); gen_median(); gen_config()
    print("golden fixtures written to", HERE)
