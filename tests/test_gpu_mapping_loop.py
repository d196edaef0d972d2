"""Keyframe seeding and the map-initialisation loop on the GPU: GaussianModel + render + losses + densification
working together the way utils/slam_backend.py:75-149 drives them."""
import math
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
pytestmark = pytest.mark.gpu

CONFIG = {
    "Dataset": {"sensor_type": "monocular", "pcd_downsample": 16, "pcd_downsample_init": 8, "point_size": 0.05,
                "adaptive_pointsize": False},
    "Training": {"monocular": True, "rgb_boundary_threshold": 0.01, "alpha": 0.98},
}
OPT = dict(position_lr_init=0.0016, position_lr_final=0.00016, position_lr_delay_mult=0.01, position_lr_max_steps=30000,
           feature_lr=0.0025, opacity_lr=0.05, scaling_lr=0.001, rotation_lr=0.001, percent_dense=0.01,
           densify_grad_threshold=0.0002, lambda_dssim=0.2)


def _camera(W, H, image, pose=None, mono_depth=None):
    from lvdgs.camera_utils import Camera
    from lvdgs.graphics_utils import focal2fov, getProjectionMatrix2
    fx = fy = float(W)
    cx, cy = W / 2.0, H / 2.0
    proj = getProjectionMatrix2(znear=0.01, zfar=100.0, fx=fx, fy=fy, cx=cx, cy=cy, W=W, H=H).transpose(0, 1)
    cam = Camera(0, image.cuda(), None, mono_depth, torch.eye(4), proj.cuda(), fx, fy, cx, cy, focal2fov(fx, W), focal2fov(fy, H),
                 H, W, device="cuda")
    if pose is not None:
        cam.update_RT(pose[:3, :3].cuda(), pose[:3, 3].cuda())
    return cam


def test_keyframe_seeding_backprojects_the_depth_map():
    import aux_oracle
    from lvdgs.gaussian_model import GaussianModel
    from lvdgs.pose_utils import SE3_exp
    from lvdgs.sh_utils import SH2RGB
    W, H = 160, 96
    g = torch.Generator().manual_seed(0)
    image = torch.rand(3, H, W, generator=g)
    depth = 2.0 + torch.rand(H, W, generator=g) * 3.0
    depth[:10] = 0.0        # invalid
    depth[10:14] = 150.0    # beyond the 100 m truncation
    pose = SE3_exp(torch.tensor([0.3, -0.2, 0.5, 0.1, -0.05, 0.2]))
    cam = _camera(W, H, image, pose)
    with torch.no_grad():
        cam.exposure_a.fill_(0.1)
        cam.exposure_b.fill_(-0.02)
    m = GaussianModel(0, config=CONFIG)
    m.init_lr(6.0)
    m.training_setup(OPT)
    m.extend_from_pcd_seq(cam, kf_id=3, init=True, scale=2.0, depthmap=depth.numpy())
    n = m.get_xyz.shape[0]
    assert n == (H - 14) * W // 8
    # every seed re-projects onto a distinct pixel centre at that pixel's depth
    pc = m.get_xyz.detach() @ cam.R.t() + cam.T
    u = (pc[:, 0] / pc[:, 2] * cam.fx + cam.cx).cpu()
    v = (pc[:, 1] / pc[:, 2] * cam.fy + cam.cy).cpu()
    ui, vi = u.round().long(), v.round().long()
    assert float((u - ui).abs().max()) < 1e-2 and float((v - vi).abs().max()) < 1e-2
    assert int(vi.min()) >= 14 and len(set(zip(ui.tolist(), vi.tolist()))) == n
    assert torch.allclose(pc[:, 2].cpu(), depth[vi, ui], rtol=1e-5)
    # colours: exposure-corrected image through 8 bits, stored as the SH DC term
    ab = (math.exp(0.1) * image - 0.02).clamp(0, 1)
    want_rgb = (ab * 255).to(torch.uint8).float() / 255.0
    assert torch.allclose(SH2RGB(m.get_features.detach()[:, 0]).cpu(), want_rgb[:, vi, ui].t(), atol=1e-6)
    # isotropic scale from the 3-nearest-neighbour distance (checked against the brute-force oracle)
    d2 = aux_oracle.dist2_knn3(m.get_xyz.detach().cpu().numpy().astype(np.float64))
    want_scale = np.log(np.sqrt(np.maximum(d2, 1e-7) * 0.05))
    np.testing.assert_allclose(m._scaling.detach().cpu().numpy(), np.repeat(want_scale[:, None], 3, 1), rtol=1e-4, atol=1e-5)
    assert torch.allclose(m.get_opacity, torch.full((n, 1), 0.5, device="cuda"))
    assert torch.equal(m.get_rotation.detach().cpu(), torch.tensor([[1.0, 0, 0, 0]]).expand(n, 4))
    assert m.unique_kfIDs.tolist() == [3] * n and m.n_obs.tolist() == [0] * n
    assert all(gp["params"][0].shape[0] == n for gp in m.optimizer.param_groups)


def test_monocular_seeding_without_depth_is_a_slab_at_the_given_scale():
    from lvdgs.gaussian_model import GaussianModel
    W, H = 96, 64
    cam = _camera(W, H, torch.rand(3, H, W))
    m = GaussianModel(0, config=CONFIG)
    m.extend_from_pcd_seq(cam, kf_id=0, init=True, scale=3.0)
    z = m.get_xyz.detach()[:, 2]
    assert m.get_xyz.shape[0] == W * H // 8
    assert abs(float(z.mean()) - 3.0 * (1 - 0.025)) < 0.02 and 0.1 < float(z.std()) < 0.2  # (1 + (n - 0.5) * 0.05) * 3


def _scene_and_target(W, H, n=3000):
    """A target image / depth rendered from a seeded scene, to be re-learnt from a seeded map."""
    from lvdgs import synthetic
    from lvdgs.gaussian_model import GaussianModel
    from lvdgs.gaussian_renderer import render
    from types import SimpleNamespace
    g = synthetic.make_gaussians(n, W, H, seed=5, r_min=3.0, r_max=10.0, z_min=2.0, z_max=6.0)
    with torch.no_grad():
        g["opacities"].clamp_(min=0.6)
    truth = GaussianModel.from_activated(g["means3D"], g["scales"], g["rotations"], g["opacities"], shs=g["shs"])
    pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False)
    bg = torch.zeros(3, device="cuda")
    cam = _camera(W, H, torch.zeros(3, H, W))
    with torch.no_grad():
        pkg = render(cam, truth, pipe, bg)
    return pkg["render"].clamp(0, 1), pkg["depth"][0], pkg["opacity"][0], pipe, bg


def test_initialize_map_loop_learns_the_target():
    """The body of BackEnd.initialize_map (utils/slam_backend.py:95-149) with a seeded map: loss falls, bookkeeping
    stays consistent through densify / prune / opacity reset."""
    from lvdgs.gaussian_model import GaussianModel
    from lvdgs.gaussian_renderer import render
    from lvdgs.slam_utils import get_loss_mapping
    W, H = 240, 160
    target, depth, opac, pipe, bg = _scene_and_target(W, H)
    seed_depth = torch.where(opac > 0.5, depth / opac.clamp(min=1e-3), torch.zeros_like(depth))
    cam = _camera(W, H, target, mono_depth=seed_depth.cpu().numpy())
    m = GaussianModel(0, config=CONFIG)
    m.init_lr(6.0)
    m.training_setup(OPT)
    m.extend_from_pcd_seq(cam, kf_id=0, init=True, scale=2.0, depthmap=seed_depth.cpu().numpy())
    n0 = m.get_xyz.shape[0]
    losses = []
    for it in range(1, 121):
        pkg = render(cam, m, pipe, bg)
        loss = get_loss_mapping(CONFIG, pkg["render"], cam, depth=pkg["depth"], initialization=True)
        loss.backward()
        with torch.no_grad():
            vis, radii = pkg["visibility_filter"], pkg["radii"]
            m.max_radii2D[vis] = torch.max(m.max_radii2D[vis], radii[vis])
            m.add_densification_stats(pkg["viewspace_points"], vis)
            if it % 30 == 0:
                m.densify_and_prune(OPT["densify_grad_threshold"], 0.005, 30.0, None)
            if it == 70:
                m.reset_opacity()
            m.optimizer.step()
            m.optimizer.zero_grad(set_to_none=True)
        losses.append(float(loss.detach()))
        n = m.get_xyz.shape[0]
        assert all(gp["params"][0].shape[0] == n for gp in m.optimizer.param_groups)
        assert m.max_radii2D.shape[0] == n and m.denom.shape[0] == n and m.unique_kfIDs.shape[0] == n and m.n_obs.shape[0] == n
    assert all(np.isfinite(losses))
    assert np.mean(losses[60:69]) < 0.6 * losses[0], (losses[0], losses[60:69])  # before the opacity reset
    assert losses[-1] < 0.7 * losses[0]
    assert m.get_xyz.shape[0] != n0  # densification / pruning changed the map
    assert (pkg["n_touched"] > 0).sum() > 0


def test_masked_l1_dssim_mapping_step_drives_every_parameter():
    """One iteration of the static-mask branch of BackEnd.map (utils/slam_backend.py:199-215, 303-306)."""
    from lvdgs.gaussian_model import GaussianModel
    from lvdgs.gaussian_renderer import render
    from lvdgs.loss_utils import l1_dssim_loss
    W, H = 240, 160
    target, depth, opac, pipe, bg = _scene_and_target(W, H)
    cam = _camera(W, H, target)
    static = torch.ones(H, W, dtype=torch.bool, device="cuda")
    static[40:100, 60:140] = False
    m = GaussianModel(0, config=CONFIG)
    m.init_lr(6.0)
    m.training_setup(OPT)
    m.extend_from_pcd_seq(cam, kf_id=0, init=True, scale=3.0)
    pkg = render(cam, m, pipe, bg)
    loss = l1_dssim_loss(pkg["render"], cam.original_image, OPT["lambda_dssim"], static, bg)
    scaling = m.get_scaling
    loss = loss + 10 * torch.abs(scaling - scaling.mean(dim=1).view(-1, 1)).mean()
    loss.backward()
    for name, p in zip(("xyz", "f_dc", "scaling", "rotation", "opacity"),
                       (m._xyz, m._features_dc, m._scaling, m._rotation, m._opacity)):
        assert p.grad is not None and torch.isfinite(p.grad).all() and float(p.grad.abs().sum()) > 0, name
    # Gaussians that only cover the dynamic rectangle get no photometric gradient
    pc = m.get_xyz.detach()
    u, v = pc[:, 0] / pc[:, 2] * cam.fx + cam.cx, pc[:, 1] / pc[:, 2] * cam.fy + cam.cy
    deep_inside = (u > 85) & (u < 115) & (v > 62) & (v < 78) & (pkg["radii"] < 8) & (pkg["radii"] > 0)
    assert int(deep_inside.sum()) > 0
    assert float(m._features_dc.grad[deep_inside].abs().max()) == 0.0
