"""Pins the C oracle (oracle/lvdgs_oracle.c) against an independent dense float64 autograd
formulation (tests/ref_torch.py) and finite differences.  There are no reference golden vectors
for the rasterizer (parity unpinned), so this is the internal anchor."""
import math
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import oracle as orc  # noqa: E402
import ref_torch  # noqa: E402
from lvdgs import synthetic  # noqa: E402
from lvdgs.graphics_utils import getProjectionMatrix2  # noqa: E402

torch.set_default_dtype(torch.float32)


def _scene(N, W, H, seed, sh_degree=0, pose_seed=3, big=False):
    g = synthetic.make_gaussians(N, W, H, seed=seed, sh_degree=sh_degree, r_min=1.0, r_max=10.0 if big else 5.0,
                                 z_min=1.0, z_max=6.0)
    cam = synthetic.make_camera(W, H, pose_seed=pose_seed, fx=W * 0.9, fy=W * 0.8, cx=W * 0.52, cy=H * 0.47)
    return {k: v.double() for k, v in g.items()}, cam


def _dense(g, cam, W, H, bg, tau, use_sh, sh_degree=0, cov_precomp=None, req=()):
    R, T = cam.R.double(), cam.T.double()
    view, proj, campos = ref_torch.camera_matrices(R, T, tau, cam.projection_matrix.double())
    leaves = {k: g[k].clone().requires_grad_(k in req) for k in g}
    kw = dict(scales=leaves["scales"], rotations=leaves["rotations"])
    if cov_precomp is not None:
        kw = dict(cov3D_precomp=cov_precomp)
    out = ref_torch.render_dense(
        leaves["means3D"], leaves["opacities"], H, W, cam.tanfovx, cam.tanfovy, bg, view, proj,
        campos, shs=leaves["shs"] if use_sh else None,
        colors_precomp=None if use_sh else leaves["colors"], sh_degree=sh_degree, **kw)
    return out, leaves


def _oracle_fwd(g, cam, W, H, bg, use_sh, sh_degree=0, cov_precomp=None, prec="f64"):
    o = orc.Oracle(prec)
    kw = dict(scales=g["scales"].numpy(), rotations=g["rotations"].numpy())
    if cov_precomp is not None:
        kw = dict(cov3D_precomp=cov_precomp.numpy())
    out = o.forward(means3D=g["means3D"].numpy(), opacities=g["opacities"].numpy(), W=W, H=H,
                    tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, viewmatrix=cam.world_view_transform.double().numpy(),
                    projmatrix=(cam.world_view_transform.double() @ cam.projection_matrix.double()).numpy(),
                    projmatrix_raw=cam.projection_matrix.double().numpy(),
                    campos=torch.linalg.inv(cam.world_view_transform.double())[3, :3].numpy(),
                    bg=bg.numpy(), shs=g["shs"].numpy() if use_sh else None,
                    colors_precomp=None if use_sh else g["colors"].numpy(), sh_degree=sh_degree, **kw)
    return o, out


@pytest.mark.parametrize("seed,W,H,N", [(0, 48, 32, 120), (1, 40, 40, 200), (2, 70, 35, 60)])
def test_forward_matches_dense_autograd_formulation(seed, W, H, N):
    g, cam = _scene(N, W, H, seed)
    bg = torch.tensor([0.1, 0.2, 0.3], dtype=torch.float64)
    tau = torch.zeros(6, dtype=torch.float64)
    ref, _ = _dense(g, cam, W, H, bg, tau, use_sh=False)
    o, out = _oracle_fwd(g, cam, W, H, bg, use_sh=False)
    assert out["num_rendered"] > 0
    np.testing.assert_array_equal(out["radii"], ref["radii"].numpy())
    np.testing.assert_array_equal(out["tiles_touched"], ref["tiles_touched"].numpy())
    np.testing.assert_allclose(out["color"], ref["color"].detach().numpy(), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(out["depth"], ref["depth"].detach().numpy(), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(out["opacity"], ref["opacity"].detach().numpy(), rtol=1e-10, atol=1e-12)
    np.testing.assert_array_equal(out["n_touched"], ref["n_touched"].numpy())
    np.testing.assert_array_equal(out["n_contrib"], ref["n_contrib"].numpy())
    # the sorted list is (tile, depth, index) ordered and the ranges partition it
    keys = out["keys_sorted"]
    assert np.all(keys[1:] >= keys[:-1])
    assert int((out["ranges"][:, 1] - out["ranges"][:, 0]).sum()) == out["num_rendered"]
    o.free()


def _loss_weights(W, H, seed):
    gc, gd, go = synthetic.make_image_grads(W, H, seed)
    return gc.double(), gd.double(), go.double()


@pytest.mark.parametrize("seed,W,H,N,big", [(0, 48, 32, 120, False), (3, 40, 40, 150, True)])
def test_backward_matches_autograd_incl_pose(seed, W, H, N, big):
    g, cam = _scene(N, W, H, seed, big=big)
    bg = torch.tensor([0.3, 0.1, 0.7], dtype=torch.float64)
    tau = torch.zeros(6, dtype=torch.float64, requires_grad=True)
    ref, lv = _dense(g, cam, W, H, bg, tau, use_sh=False, req=("means3D", "scales", "rotations", "opacities", "colors"))
    gc, gd, go = _loss_weights(W, H, seed)
    loss = (ref["color"] * gc).sum() + (ref["depth"] * gd).sum() + (ref["opacity"] * go).sum()
    loss.backward()
    o, out = _oracle_fwd(g, cam, W, H, bg, use_sh=False)
    gr = o.backward(gc.numpy(), gd.numpy(), go.numpy())
    pairs = [("means3D", lv["means3D"].grad), ("scales", lv["scales"].grad), ("rotations", lv["rotations"].grad),
             ("opacities", lv["opacities"].grad), ("colors", lv["colors"].grad), ("tau", tau.grad)]
    for name, r in pairs:
        a, b = gr[name], r.numpy().reshape(gr[name].shape)
        scale = np.abs(b).max()
        np.testing.assert_allclose(a, b, rtol=1e-7, atol=1e-9 * scale, err_msg=name)
    # viewspace gradient = d loss / d NDC xy
    pix_grad = ref["means2D_pix"].grad.numpy()
    np.testing.assert_allclose(gr["means2D"][:, 0], pix_grad[:, 0] * 0.5 * W, rtol=1e-7, atol=1e-9 * np.abs(pix_grad).max() * W)
    np.testing.assert_allclose(gr["means2D"][:, 1], pix_grad[:, 1] * 0.5 * H, rtol=1e-7, atol=1e-9 * np.abs(pix_grad).max() * H)
    assert np.all(gr["means2D"][:, 2] == 0)
    # culled Gaussians get exactly zero gradient
    dead = out["radii"] == 0
    assert not np.any(gr["means3D"][dead]) and not np.any(gr["scales"][dead])
    o.free()


@pytest.mark.parametrize("deg", [1, 2, 3])
def test_backward_sh_degrees(deg):
    W, H, N = 40, 32, 90
    g, cam = _scene(N, W, H, seed=10 + deg, sh_degree=deg)
    g["shs"][:, 0] -= 1.2  # push some channels below zero so the clamp mask is exercised
    bg = torch.zeros(3, dtype=torch.float64)
    # tau takes part in the graph: the camera centre in the SH view direction depends on the pose (C = -R^T T), so the
    # colours -- and with them dL/dtau -- do too once the degree is above 0
    tau = torch.zeros(6, dtype=torch.float64, requires_grad=True)
    ref, lv = _dense(g, cam, W, H, bg, tau, use_sh=True, sh_degree=deg, req=("means3D", "shs", "opacities"))
    gc, gd, go = _loss_weights(W, H, deg)
    ((ref["color"] * gc).sum() + (ref["depth"] * gd).sum()).backward()
    o, out = _oracle_fwd(g, cam, W, H, bg, use_sh=True, sh_degree=deg)
    assert out["clamped"].any()
    np.testing.assert_allclose(out["color"], ref["color"].detach().numpy(), rtol=1e-10, atol=1e-12)
    gr = o.backward(gc.numpy(), gd.numpy(), None)
    for name in ("means3D", "shs", "opacities"):
        b = lv[name].grad.numpy().reshape(gr[name].shape)
        np.testing.assert_allclose(gr[name], b, rtol=1e-7, atol=1e-9 * np.abs(b).max(), err_msg=name)
    t = tau.grad.numpy()
    np.testing.assert_allclose(gr["tau"], t, rtol=1e-7, atol=1e-9 * np.abs(t).max(), err_msg="tau (incl. the view-direction path)")
    o.free()


def test_backward_cov3d_precomp():
    W, H, N = 40, 32, 80
    g, cam = _scene(N, W, H, seed=21)
    R = ref_torch.quat_to_rot(g["rotations"])
    M = R @ torch.diag_embed(g["scales"])
    S = M @ M.transpose(1, 2)
    cov = torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], 1).contiguous()
    cov_leaf = cov.clone().requires_grad_(True)
    bg = torch.zeros(3, dtype=torch.float64)
    ref, lv = _dense(g, cam, W, H, bg, torch.zeros(6, dtype=torch.float64), use_sh=False, cov_precomp=cov_leaf)
    gc, gd, go = _loss_weights(W, H, 5)
    ((ref["color"] * gc).sum() + (ref["opacity"] * go).sum()).backward()
    o, out = _oracle_fwd(g, cam, W, H, bg, use_sh=False, cov_precomp=cov)
    gr = o.backward(gc.numpy(), None, go.numpy())
    b = cov_leaf.grad.numpy()
    np.testing.assert_allclose(gr["cov3D"], b, rtol=1e-7, atol=1e-9 * np.abs(b).max())
    o.free()


def test_pose_gradient_finite_difference():
    """dL/dtau from the oracle against central differences of the oracle's own forward with the
    camera moved by SE3_exp(eps e_k) (the retraction of reference utils/pose_utils.py:70-87)."""
    from lvdgs.pose_utils import SE3_exp
    W, H, N = 48, 32, 100
    g, cam = _scene(N, W, H, seed=31)
    bg = torch.tensor([0.2, 0.2, 0.2], dtype=torch.float64)
    gc, gd, go = _loss_weights(W, H, 9)
    o, out = _oracle_fwd(g, cam, W, H, bg, use_sh=False)
    gr = o.backward(gc.numpy(), gd.numpy(), go.numpy())
    o.free()

    def loss_at(tau):
        w2c = torch.eye(4, dtype=torch.float64)
        w2c[:3, :3], w2c[:3, 3] = cam.R.double(), cam.T.double()
        view = (SE3_exp(tau) @ w2c).t().contiguous()
        o2 = orc.Oracle("f64")
        r = o2.forward(means3D=g["means3D"].numpy(), opacities=g["opacities"].numpy(), W=W, H=H, tanfovx=cam.tanfovx,
                       tanfovy=cam.tanfovy, viewmatrix=view.numpy(), projmatrix=(view @ cam.projection_matrix.double()).numpy(),
                       projmatrix_raw=cam.projection_matrix.double().numpy(), bg=bg.numpy(),
                       scales=g["scales"].numpy(), rotations=g["rotations"].numpy(), colors_precomp=g["colors"].numpy())
        o2.free()
        return float((r["color"] * gc.numpy()).sum() + (r["depth"] * gd.numpy()).sum() + (r["opacity"] * go.numpy()).sum())

    eps = 1e-6
    fd = np.zeros(6)
    for k in range(6):
        e = torch.zeros(6, dtype=torch.float64); e[k] = eps
        fd[k] = (loss_at(e) - loss_at(-e)) / (2 * eps)
    # thresholds (alpha >= 1/255, rect membership) make the loss only piecewise smooth: loose tolerance
    np.testing.assert_allclose(gr["tau"], fd, rtol=2e-3, atol=2e-3 * np.abs(fd).max())


def test_f32_build_agrees_with_f64_build():
    W, H, N = 96, 64, 600
    g, cam = _scene(N, W, H, seed=40)
    bg = torch.zeros(3, dtype=torch.float64)
    o64, r64 = _oracle_fwd(g, cam, W, H, bg, use_sh=False, prec="f64")
    g32 = {k: v.float() for k, v in g.items()}
    o32 = orc.Oracle("f32")
    r32 = o32.forward(means3D=g32["means3D"].numpy(), opacities=g32["opacities"].numpy(), W=W, H=H,
                      tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, viewmatrix=cam.world_view_transform.numpy(),
                      projmatrix=cam.full_proj_transform.numpy(), projmatrix_raw=cam.projection_matrix.numpy(),
                      campos=cam.camera_center.numpy(), bg=bg.numpy(), scales=g32["scales"].numpy(),
                      rotations=g32["rotations"].numpy(), colors_precomp=g32["colors"].numpy())
    assert (r32["radii"] != r64["radii"]).mean() < 0.01
    np.testing.assert_allclose(r32["color"], r64["color"], rtol=0, atol=2e-3)
    gc, gd, go = synthetic.make_image_grads(W, H, 1)
    b32 = o32.backward(gc.numpy(), gd.numpy(), go.numpy())
    b64 = o64.backward(gc.double().numpy(), gd.double().numpy(), go.double().numpy())
    for k in ("means3D", "scales", "rotations", "opacities", "colors", "tau"):
        s = np.abs(b64[k]).max()
        assert np.abs(b32[k] - b64[k]).max() < 5e-3 * s, k
    o32.free(); o64.free()


def test_empty_and_all_culled_inputs():
    o = orc.Oracle("f32")
    cam = synthetic.make_camera(32, 32)
    g = synthetic.make_gaussians(5, 32, 32, seed=0)
    g["means3D"][:, 2] = -1.0  # behind the camera
    r = o.forward(means3D=g["means3D"].numpy(), opacities=g["opacities"].numpy(), W=32, H=32, tanfovx=cam.tanfovx,
                  tanfovy=cam.tanfovy, viewmatrix=cam.world_view_transform.numpy(), projmatrix=cam.full_proj_transform.numpy(),
                  projmatrix_raw=cam.projection_matrix.numpy(), bg=np.array([0.5, 0.25, 0.125]),
                  scales=g["scales"].numpy(), rotations=g["rotations"].numpy(), colors_precomp=g["colors"].numpy())
    assert r["num_rendered"] == 0 and not r["radii"].any()
    assert np.all(r["color"][0] == 0.5) and np.all(r["color"][2] == 0.125) and not r["opacity"].any()
    b = o.backward(np.ones((3, 32, 32)), np.ones((32, 32)), np.ones((32, 32)))
    assert not b["means3D"].any() and not b["tau"].any()
    assert not orc.mark_visible(g["means3D"].numpy(), cam.world_view_transform.numpy()).any()
    o.free()


def test_openmp_build_equals_the_scalar_oracle():
    """liblvdgs_oracle_f32_omp.so (bench.py's CPU baseline: the same loops over all host cores) gives the scalar build's
    forward bit for bit and its gradients to the order of a few double-precision additions."""
    W, H, N = 96, 64, 900
    g, cam = _scene(N, W, H, seed=3)
    g32 = {k: v.float() for k, v in g.items()}
    bg = np.array([0.1, 0.2, 0.3])
    gc, gd, go = _loss_weights(W, H, 5)
    out = {}
    for prec in ("f32", "f32_omp"):
        o = orc.Oracle(prec)
        f = o.forward(means3D=g32["means3D"].numpy(), opacities=g32["opacities"].numpy(), W=W, H=H, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
                      viewmatrix=cam.world_view_transform.numpy(), projmatrix=(cam.world_view_transform @ cam.projection_matrix).numpy(),
                      projmatrix_raw=cam.projection_matrix.numpy(), campos=torch.linalg.inv(cam.world_view_transform)[3, :3].numpy(), bg=bg,
                      scales=g32["scales"].numpy(), rotations=g32["rotations"].numpy(), colors_precomp=g32["colors"].numpy())
        b = o.backward(gc.numpy(), gd.numpy(), go.numpy())
        out[prec] = (f, b, o.threads)
        o.free()
    (f0, b0, t0), (f1, b1, t1) = out["f32"], out["f32_omp"]
    assert t0 == 1 and t1 >= 1
    for k in ("color", "depth", "opacity", "radii", "n_touched", "ids_sorted", "keys_sorted", "ranges", "n_contrib", "final_T"):
        np.testing.assert_array_equal(f0[k], f1[k], err_msg=k)
    for k in b0:
        np.testing.assert_allclose(b1[k], b0[k], rtol=1e-6, atol=1e-6 * max(np.abs(b0[k]).max(), 1e-30), err_msg=k)
