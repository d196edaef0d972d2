"""HIP vs CPU oracle at the FULL sizes BASELINE.json names -- the same checks as the small parity cases
(integers bit-exact, images and every gradient incl. dL/dtau within 1e-4), not only the size-independent
properties of tests/test_gpu_fullsize.py:

  * configs[2]: 500k Gaussians, 1920x1080 (the benchmark's workload);
  * configs[3] geometry: 200k Gaussians at KITTI-07's 1226x370 with the sequence's intrinsics
    (reference configs/mono/KITTI/07.yaml:8-18);
  * configs[4] shape: 2M Gaussians at 1920x1280 (reference configs/mono/waymo/405841.yaml:15-16).

and of ``render_with_custom_resolution`` (reference utils/init_pose.py:141-158: the tracked frame's map rendered
at MASt3R's raster size, depth back-projected with intrinsics scaled by W1/W and H1/H).

The scalar C oracle needs 4 s (KITTI), 8 s (500k) and about 30 s (2M) per forward + backward on one host core.
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))

import test_gpu_parity as tp  # noqa: E402

pytestmark = pytest.mark.gpu

GRADS = ["means3D", "means2D", "opacities", "scales", "rotations", "colors", "tau"]


def _workload(name, pose_seed):
    from lvdgs import synthetic
    cfg = synthetic.CONFIGS[name]
    N, W, H = cfg["N"], cfg["W"], cfg["H"]
    g = synthetic.make_workload_gaussians(name, seed=0)
    cam = synthetic.make_camera(W, H, pose_seed=pose_seed, **{k: cfg[k] for k in ("fx", "fy", "cx", "cy") if k in cfg})
    return g, cam, N, W, H


@pytest.mark.parametrize("name,pose_seed", [("cfg3_500k_1920x1080", 3), ("kitti07_geom", 5), ("cfg5_2m_1920x1280", None)])
def test_full_size_forward_and_backward_match_oracle(name, pose_seed):
    orc, hr, syn = tp._mods()
    g, cam, N, W, H = _workload(name, pose_seed)
    bg = torch.tensor([0.1, 0.3, 0.2])
    grads = syn.make_image_grads(W, H, 0)
    f_hip, b_hip = hr.run_hip(g, cam, W, H, bg, grads=grads)
    f_ora, b_ora = hr.run_oracle(orc, g, cam, W, H, bg, grads=grads)
    assert f_ora["num_rendered"] > 2 * N
    tp._check_forward(f_hip, f_ora, W, H)
    # every gradient of EVERY Gaussian and the pose to the strict tolerances, on the problem without the fragile pixels
    tp._check_backward(b_hip, b_ora, GRADS, f_ora, W, H, rerun=tp.masked_rerun(hr, orc, g, cam, W, H, bg, grads))
    # The index state at BASELINE's sizes, bit for bit: with LVDGS_FLAG_LIST_ALL_TILES (every tile of the 3-sigma rectangle
    # listed, as the reference does) tiles_touched, the pair count, the sorted (tile, depth, id) list, the tile ranges
    # and n_contrib ARE the oracle's (check_pair_lists / _check_forward compare them with assert_array_equal in this
    # mode), and the images and counters of the two modes are the same bits.
    f_all, _ = hr.run_hip(g, cam, W, H, bg, tile_cull=False)
    assert f_all["num_rendered"] == f_ora["num_rendered"] > f_hip["num_rendered"]
    tp._check_forward(f_all, f_ora, W, H)
    np.testing.assert_array_equal(f_all["tiles_touched"], f_ora["tiles_touched"])
    np.testing.assert_array_equal(f_all["point_list"], f_ora["ids_sorted"])
    np.testing.assert_array_equal(f_all["ranges"], f_ora["ranges"])
    solid = f_ora["fragile"] == 0
    np.testing.assert_array_equal(f_all["n_contrib"][solid], f_ora["n_contrib"][solid])
    frag = ~solid   # (exact lists: positions are the oracle's own; fragile pixels between its two-sided bounds)
    assert ((f_ora["n_contrib_lo"][frag] <= f_all["n_contrib"][frag]) & (f_all["n_contrib"][frag] <= f_ora["n_contrib_hi"][frag])).all()
    for k in ("color", "depth", "opacity", "final_T", "radii", "n_touched"):
        np.testing.assert_array_equal(f_all[k], f_hip[k], err_msg=k)


@pytest.mark.parametrize("name", ["surface_12k_640x480", "surface_100k_1920x1080"])
def test_opaque_surfaces_of_large_gaussians_match_oracle(name):
    """The regime real maps live in (synthetic.make_surface_gaussians): large flat Gaussians on opaque surfaces -- tile
    lists of ~900-1300 entries (beyond one wave's register sort at the long end: the tile sort's queue and its in-launch
    workgroup sort), a third of the rectangles above 64 tiles (culled per block of tiles), pixels that saturate after a
    tenth of their list (early termination, last-contributor culling in the backward pass), a pair count far beyond the
    first capacity guess (the overflow re-run).  Forward and backward against the oracle, and the exact list mode."""
    orc, hr, syn = tp._mods()
    g, cam, N, W, H = _workload(name, 2)
    bg = torch.tensor([0.1, 0.3, 0.2])
    grads = syn.make_image_grads(W, H, 0)
    f_hip, b_hip = hr.run_hip(g, cam, W, H, bg, grads=grads)
    f_ora, b_ora = hr.run_oracle(orc, g, cam, W, H, bg, grads=grads)
    lists = f_ora["ranges"][:, 1].astype(np.int64) - f_ora["ranges"][:, 0]
    assert lists.mean() > 500 and lists.max() > 1024 and (f_ora["tiles_touched"] > 64).mean() > 0.2
    assert f_ora["n_contrib"].mean() < 0.25 * lists.mean()          # pixels saturate early
    assert f_hip["num_rendered"] < 0.9 * f_ora["num_rendered"]      # block + tile culling together (opaque blobs reach most of their square)
    tp._check_forward(f_hip, f_ora, W, H)
    # Every gradient of every Gaussian, and the pose gradient -- here six sums over every Gaussian of terms that are large
    # (opaque, large footprints) and cancel: float32 itself cannot hold them to 1e-5 of the largest component in whatever
    # order they are added, and such a tensor must be no farther from the FLOAT64 oracle than the float32 oracle is (x 1.5, plus
    # the strict tolerance): tp._check_backward with the fragile pixels' image gradients zeroed on both sides.
    tp._check_backward(b_hip, b_ora, GRADS, f_ora, W, H, rerun=tp.masked_rerun(hr, orc, g, cam, W, H, bg, grads))
    f_all, _ = hr.run_hip(g, cam, W, H, bg, tile_cull=False)
    assert f_all["num_rendered"] == f_ora["num_rendered"]
    np.testing.assert_array_equal(f_all["point_list"], f_ora["ids_sorted"])
    np.testing.assert_array_equal(f_all["ranges"], f_ora["ranges"])
    for k in ("color", "depth", "opacity", "final_T", "radii", "n_touched"):
        np.testing.assert_array_equal(f_all[k], f_hip[k], err_msg=k)


def test_long_lists_sort_the_same_on_every_frame_whichever_workgroups_take_them():
    """The tile sort's size classes are chosen by a HINT (the longest list the previous frame on this device queued, read back with
    its pair count): a first frame's lists of more than 1024 entries are sorted by the launch's last workgroups, the following
    frames' by workgroups at its head, four waves to a list (tilesort.hip).  Same order every time: the oracle's, entry for entry
    (exact-list mode), through a frame of SHORT lists in between (which must not disturb the hint's bookkeeping either)."""
    orc, hr, syn = tp._mods()
    g, cam, N, W, H = _workload("surface_12k_640x480", 2)
    bg = torch.tensor([0.1, 0.3, 0.2])
    f_ora, _ = hr.run_oracle(orc, g, cam, W, H, bg)
    lists = f_ora["ranges"][:, 1].astype(np.int64) - f_ora["ranges"][:, 0]
    assert (lists > 1024).sum() > 20 and lists.max() <= 2048
    small, small_cam = tp._scene(syn, 3000, 320, 240, seed=5)
    for frame in range(4):
        f_all, _ = hr.run_hip(g, cam, W, H, bg, tile_cull=False)
        np.testing.assert_array_equal(f_all["point_list"], f_ora["ids_sorted"], err_msg=f"frame {frame}")
        np.testing.assert_array_equal(f_all["ranges"], f_ora["ranges"])
        if frame == 1:
            f_s, _ = hr.run_hip(small, small_cam, 320, 240, bg, tile_cull=False)
            f_so, _ = hr.run_oracle(orc, small, small_cam, 320, 240, bg)
            np.testing.assert_array_equal(f_s["point_list"], f_so["ids_sorted"])


# ---------------------------------------------------------------------------------------------------------------
def _kitti_camera(pose_seed=2):
    """A Camera-like namespace at KITTI-07's geometry with the attributes render() reads, matrices on the GPU."""
    from lvdgs import synthetic
    cfg = synthetic.CONFIGS["kitti07_geom"]
    cam = synthetic.make_camera(cfg["W"], cfg["H"], pose_seed=pose_seed, fx=cfg["fx"], fy=cfg["fy"], cx=cfg["cx"], cy=cfg["cy"])
    gpu = SimpleNamespace(**vars(cam))
    for k in ("world_view_transform", "projection_matrix", "full_proj_transform", "camera_center"):
        setattr(gpu, k, getattr(cam, k).cuda())
    gpu.cam_rot_delta = torch.zeros(3, device="cuda", requires_grad=True)
    gpu.cam_trans_delta = torch.zeros(3, device="cuda", requires_grad=True)
    return cam, gpu


class _ActivatedModel:
    """The accessors render() reads, backed by leaf tensors of the ACTIVATED quantities, so the gradients that land
    on them are the ones the oracle reports (no fused activations, no chain rule through exp / sigmoid)."""
    active_sh_degree = 0
    max_sh_degree = 0

    def __init__(self, g):
        leaf = lambda t: t.cuda().clone().requires_grad_(True)
        self.get_xyz, self.get_scaling, self.get_rotation = leaf(g["means3D"]), leaf(g["scales"]), leaf(g["rotations"])
        self.get_opacity, self.get_features = leaf(g["opacities"]), leaf(g["shs"])


@pytest.mark.parametrize("target", [(512, 160), (512, 144), (1226, 370)])
def test_render_with_custom_resolution_matches_oracle_at_the_target_size(target):
    """The map of a 1226x370 keyframe rendered at MASt3R's raster (long edge 512; 512x144 after its crop is NOT the
    camera's aspect ratio) equals the oracle run at the target size with the camera's fields of view and matrices:
    colour, depth, opacity, radii, n_touched and every gradient."""
    orc, hr, syn = tp._mods()
    from lvdgs import rasterizer
    from lvdgs.gaussian_renderer import render, render_with_custom_resolution
    W1, H1 = target
    cam, gpu = _kitti_camera()
    W0, H0 = cam.image_width, cam.image_height
    N = 30_000
    g = syn.make_gaussians(N, W0, H0, seed=11, r_min=1.0, r_max=20.0)
    model = _ActivatedModel(g)
    pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False)
    bg = torch.tensor([0.2, 0.1, 0.3])
    rasterizer.PROPAGATE_OPACITY_GRAD = True
    try:
        pkg = render_with_custom_resolution(gpu, model, pipe, bg.cuda(), target_width=W1, target_height=H1)
        assert pkg["render"].shape == (3, H1, W1) and pkg["depth"].shape == (1, H1, W1) and pkg["opacity"].shape == (1, H1, W1)
        gc, gd, go = syn.make_image_grads(W1, H1, 3)
        ((pkg["render"] * gc.cuda()).sum() + (pkg["depth"] * gd.cuda()).sum() + (pkg["opacity"] * go.cuda()).sum()).backward()
    finally:
        rasterizer.PROPAGATE_OPACITY_GRAD = False
    # oracle at the target size: same tan(fov/2), same view / projection matrices (init_pose.py:141-146 changes nothing else)
    f_ora, b_ora = hr.run_oracle(orc, g, cam, W1, H1, bg, use_sh=True, sh_degree=0, grads=(gc, gd, go))
    np.testing.assert_array_equal(pkg["radii"].cpu().numpy(), f_ora["radii"])
    np.testing.assert_array_equal(pkg["visibility_filter"].cpu().numpy(), f_ora["radii"] > 0)
    solid = f_ora["fragile"] == 0
    assert solid.mean() > 0.98
    for k, t in (("color", pkg["render"]), ("depth", pkg["depth"]), ("opacity", pkg["opacity"])):
        m = np.broadcast_to(solid, f_ora[k].shape)
        tp._close(np.where(m, t.detach().cpu().numpy(), 0), np.where(m, f_ora[k], 0), what=f"{k} at {W1}x{H1}")
    diff = np.abs(pkg["n_touched"].cpu().numpy().astype(np.int64) - f_ora["n_touched"].astype(np.int64))
    assert diff.sum() <= 4 * int((~solid).sum())
    got = dict(means3D=model.get_xyz.grad, means2D=pkg["viewspace_points"].grad, opacities=model.get_opacity.grad.reshape(-1),
               scales=model.get_scaling.grad, rotations=model.get_rotation.grad, shs=model.get_features.grad,
               tau=torch.cat([gpu.cam_trans_delta.grad, gpu.cam_rot_delta.grad]))
    tp._check_backward({k: t.cpu().numpy() for k, t in got.items()}, b_ora, list(got), f_ora, W1, H1)

    # geometry of the resize (what utils/init_pose.py:149-158 relies on when it scales fx, fy, cx, cy by W1/W, H1/H):
    # pixel centres map affinely, (u + 1/2) * W1/W - 1/2, and depth along a ray does not change
    if (W1, H1) != (W0, H0):
        with torch.no_grad():
            native = render(gpu, model, pipe, bg.cuda())
        sx, sy = W1 / W0, H1 / H0
        # a smooth, opaque scene region: compare depth at target pixel centres with the native depth there
        ys = ((np.arange(H1) + 0.5) / sy - 0.5).round().clip(0, H0 - 1).astype(int)
        xs = ((np.arange(W1) + 0.5) / sx - 0.5).round().clip(0, W0 - 1).astype(int)
        d1, o1 = pkg["depth"][0].detach().cpu().numpy(), pkg["opacity"][0].detach().cpu().numpy()
        d0, o0 = native["depth"][0].cpu().numpy()[np.ix_(ys, xs)], native["opacity"][0].cpu().numpy()[np.ix_(ys, xs)]
        both = (o1 > 0.9) & (o0 > 0.9)
        assert both.mean() > 0.2
        # per-pixel normalised depth agrees up to the smoothing the coarser raster applies (low-pass + sampling)
        r = (d1 / np.maximum(o1, 1e-6))[both] / (d0 / np.maximum(o0, 1e-6))[both]
        assert abs(np.median(r) - 1.0) < 0.05, np.median(r)


def test_custom_resolution_projects_means_affinely():
    """means2D at the target size = (means2D at the native size + 1/2) * (W1/W, H1/H) - 1/2, read from the per-Gaussian
    records of both renders (bit-level agreement is not expected: the NDC -> pixel mapping rounds differently)."""
    orc, hr, syn = tp._mods()
    cam, _ = _kitti_camera(pose_seed=4)
    W0, H0 = cam.image_width, cam.image_height
    g = syn.make_gaussians(5000, W0, H0, seed=12)
    bg = torch.zeros(3)
    f0, _ = hr.run_hip(g, cam, W0, H0, bg)
    f1, _ = hr.run_hip(g, cam, 512, 144, bg)
    vis = (f0["radii"] > 0) & (f1["radii"] > 0)
    assert vis.sum() > 2000
    want = (f0["rec"][vis, 0:2].astype(np.float64) + 0.5) * np.array([512 / W0, 144 / H0]) - 0.5
    np.testing.assert_allclose(f1["rec"][vis, 0:2], want, atol=2e-3)
    np.testing.assert_array_equal(f1["rec"][vis, 9], f0["rec"][vis, 9])  # view depth does not depend on the raster
