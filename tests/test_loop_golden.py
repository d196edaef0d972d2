"""The build's loops against what the REFERENCE's loops did (tests/golden/loops.npz, produced by running
``BackEnd.initialize_map``, ``BackEnd.map`` and ``FrontEnd.tracking`` of /root/reference on the toy scene of
tests/loop_scene.py with the dense CPU renderer -- tests/golden/make_loop_golden.py).

Here the same renderer is used, on the CPU, so everything that can differ is the loop itself: order of bookkeeping,
which losses, optimiser steps, densify / prune / reset schedule, random keyframe choice, pose retraction.  The numbers
must agree to float32 rounding.  (The GPU replay with the HIP rasterizer is tests/test_gpu_loop_golden.py.)"""
import os
import sys
import types

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "oracle"))

GOLD = os.path.join(HERE, "golden", "loops.npz")


def _backend(sc, cfg):
    """A BackEnd-shaped namespace: the attributes reference utils/slam_backend.py:21-72 sets."""
    T = cfg["Training"]
    be = types.SimpleNamespace(
        config=cfg, gaussians=sc["gaussians"], pipeline_params=sc["pipe"], background=sc["background"],
        opt_params=types.SimpleNamespace(**cfg["opt_params"]), monocular=T["monocular"], iteration_count=0, last_sent=0,
        occ_aware_visibility={}, viewpoints={}, current_window=[], initialized=not T["monocular"], keyframe_optimizers=None,
        cameras_extent=6.0, init_itr_num=T["init_itr_num"], init_gaussian_update=T["init_gaussian_update"],
        init_gaussian_reset=T["init_gaussian_reset"], init_gaussian_th=T["init_gaussian_th"],
        init_gaussian_extent=6.0 * T["init_gaussian_extent"], gaussian_update_every=T["gaussian_update_every"],
        gaussian_update_offset=T["gaussian_update_offset"], gaussian_th=T["gaussian_th"], gaussian_extent=6.0 * T["gaussian_extent"],
        gaussian_reset=T["gaussian_reset"], size_threshold=T["size_threshold"], window_size=T["window_size"])
    return be


def _cpu_view_loss(backend, viewpoint, pkg):
    """The window keyframe's loss with the float64 statements of oracle/loss_oracle.py where the product uses its fused
    HIP kernels (which have no CPU path): reference utils/slam_backend.py:196-266."""
    import loss_oracle as lo
    from lvdgs.slam_utils import get_loss_mapping
    if getattr(viewpoint, "static_mask", None) is not None:
        return lo.masked_mapping_loss(pkg["render"], pkg["depth"], viewpoint.original_image, torch.from_numpy(viewpoint.mono_depth),
                                      viewpoint.static_mask, backend.background, backend.opt_params.lambda_dssim,
                                      backend.config["Training"].get("depth_lambda", 0.1)).to(pkg["render"].dtype)
    return get_loss_mapping(backend.config, pkg["render"], viewpoint, depth=pkg["depth"], monodepth=True)


def _record_steps(optimizer, counts, grads):
    """Before every optimiser step: the size of the first parameter and a copy of every group's gradient."""
    def pre(opt, args, kwargs):
        counts.append(int(opt.param_groups[0]["params"][0].shape[0]))
        grads.append({gp["name"]: None if gp["params"][0].grad is None else gp["params"][0].grad.detach().clone().cpu().numpy()
                      for gp in opt.param_groups})
    optimizer.register_step_pre_hook(pre)


def _snap(G):
    return {k: v.detach().numpy() for k, v in G._params_by_name().items()}


def _close(got, want, what, rtol=2e-4, atol_scale=2e-5):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    if want.size == 0:
        return
    tol = rtol * np.abs(want) + atol_scale * max(np.abs(want).max(), 1e-30)
    bad = np.abs(got - want) > tol
    assert not bad.any(), f"{what}: {bad.sum()} / {bad.size} differ, worst {np.abs(got - want).max():.3e} (scale {np.abs(want).max():.3e})"


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def test_initialize_map_replays_the_reference_loop(gold):
    from dense_render import dense_render
    from loop_scene import build_scene, loop_config
    from lvdgs.slam_loops import initialize_map
    cfg = loop_config()
    torch.manual_seed(0)
    sc = build_scene("cpu")
    be = _backend(sc, cfg)
    be.viewpoints[0] = sc["cameras"][0]
    losses, counts, grads = [], [], []
    _record_steps(sc["gaussians"].optimizer, counts, grads)
    initialize_map(be, 0, sc["cameras"][0], render_fn=dense_render, on_iteration=lambda i, loss, pkg: losses.append(float(loss.detach())))
    assert be.iteration_count == int(gold["init_iteration_count"])
    np.testing.assert_array_equal(counts, gold["init_n_per_iter"])          # densify / prune schedule and outcome
    _close(losses, gold["init_losses"], "loss per iteration")
    stepping = [i for i, g in enumerate(grads) if g["xyz"] is not None]
    np.testing.assert_array_equal(stepping, gold["init_stepping_iterations"])
    for n, it in enumerate(stepping[:2]):
        for k in ("xyz", "opacity", "scaling", "f_dc", "rotation"):
            _close(grads[it][k], gold[f"init_grad{n}_{k}"], f"gradient of {k} at stepping iteration {n}")
    for k, v in _snap(be.gaussians).items():
        _close(v, gold["init_end_" + k], "final " + k, rtol=1e-3, atol_scale=1e-4)
    _close(be.gaussians.max_radii2D.numpy(), gold["init_end_max_radii2D"], "max_radii2D")
    np.testing.assert_array_equal(be.occ_aware_visibility[0].numpy(), gold["init_occ0"])


def test_map_window_replays_the_reference_map_loop(gold):
    """BackEnd.map: 6 iterations on a window of 4 keyframes (one with a static mask) + 2 random older ones, with a
    densification, an opacity reset of the non-visible, pose / exposure steps and retraction; then the pruning pass."""
    from dense_render import dense_render
    from loop_scene import build_scene, loop_config
    from lvdgs.backend_map import map_window
    cfg = loop_config()
    torch.manual_seed(1)
    sc = build_scene("cpu")
    be = _backend(sc, cfg)
    be.initialized = True
    for i, cam in enumerate(sc["cameras"]):
        be.viewpoints[i] = cam
    window = sc["window"]
    be.current_window = window
    be.keyframe_optimizers = sc["make_keyframe_optimizer"](be.viewpoints, window, cfg)
    counts, grads, kf_grads = [], [], []
    _record_steps(sc["gaussians"].optimizer, counts, grads)
    _record_steps(be.keyframe_optimizers, [], kf_grads)
    stats = {}
    map_window(be, window, iters=sc["map_iters"], render_fn=dense_render, view_loss_fn=_cpu_view_loss, stats=stats)
    assert be.iteration_count == int(gold["map_iteration_count"])
    np.testing.assert_array_equal(counts, gold["map_n_per_iter"])
    stepping = [i for i, g in enumerate(grads) if g["xyz"] is not None]
    np.testing.assert_array_equal(stepping, gold["map_stepping_iterations"])
    for n, it in enumerate(stepping[:2]):
        for k in ("xyz", "opacity", "scaling", "f_dc", "rotation"):
            _close(grads[it][k], gold[f"map_grad{n}_{k}"], f"gradient of {k} at stepping iteration {n}")
        for name, g in kf_grads[it].items():
            key = f"map_kfgrad{n}_{name}"
            if g is None:   # e.g. the exposure of the masked keyframe: that branch of the loss does not use it
                assert key not in gold.files, key
            else:
                _close(g, gold[key], f"gradient of {name} at stepping iteration {n}")
    for k, v in _snap(be.gaussians).items():
        _close(v, gold["map_end_" + k], "final " + k, rtol=1e-3, atol_scale=1e-4)
    _close(be.gaussians.max_radii2D.numpy(), gold["map_end_max_radii2D"], "max_radii2D")
    _close(be.gaussians.xyz_gradient_accum.numpy(), gold["map_end_xyz_gradient_accum"], "xyz_gradient_accum", rtol=1e-3, atol_scale=1e-4)
    np.testing.assert_array_equal(be.gaussians.denom.numpy(), gold["map_end_denom"])
    np.testing.assert_array_equal(be.gaussians.unique_kfIDs.numpy(), gold["map_end_unique_kfIDs"])
    for i, cam in enumerate(sc["cameras"]):
        _close(cam.R.numpy(), gold[f"map_end_R_{i}"], f"R of keyframe {i}", rtol=1e-5, atol_scale=1e-6)
        _close(cam.T.numpy(), gold[f"map_end_T_{i}"], f"T of keyframe {i}", rtol=1e-4, atol_scale=1e-5)
        _close([float(cam.exposure_a.detach()), float(cam.exposure_b.detach())], gold[f"map_end_exposure_{i}"], f"exposure of keyframe {i}", rtol=1e-3, atol_scale=1e-3)
    for kf in window:
        np.testing.assert_array_equal(be.occ_aware_visibility[kf].numpy(), gold[f"map_end_occ_{kf}"])
    # the pruning pass after the keyframe's iterations (slam_backend.py:601): full window, prune_mode "slam"
    n_before = be.gaussians.get_xyz.shape[0]
    map_window(be, window, prune=True, render_fn=dense_render, view_loss_fn=_cpu_view_loss)
    np.testing.assert_array_equal([n_before, be.gaussians.get_xyz.shape[0]], gold["prune_n_before_after"])
    np.testing.assert_array_equal(be.gaussians.n_obs.numpy(), gold["prune_end_n_obs"])
    _close(be.gaussians.get_xyz.detach().numpy(), gold["prune_end_xyz"], "positions after pruning", rtol=1e-3, atol_scale=1e-4)


def test_track_frame_replays_the_reference_tracking_loop(gold):
    from dense_render import dense_render
    from loop_scene import build_scene, loop_config
    from lvdgs.slam_loops import track_frame
    cfg = loop_config()
    torch.manual_seed(2)
    sc = build_scene("cpu")
    cam, prev = sc["track_camera"], sc["cameras"][0]
    cam.mono_depth = sc["track_mono_depth"]
    cam.update_RT(prev.R, prev.T)   # MASt3R gave no estimate: start from the previous frame's pose (slam_frontend.py:1460-1462)
    losses, taus = [], []
    import lvdgs.slam_loops as sl
    ref_update = sl.update_pose

    def logged(camera, *a, **k):
        taus.append(torch.cat([camera.cam_trans_delta.detach(), camera.cam_rot_delta.detach()]).numpy().copy())
        return ref_update(camera, *a, **k)
    sl.update_pose = logged
    try:
        pkg, median_depth, n_it = track_frame(cam, sc["gaussians"], cfg, sc["pipe"], sc["background"], render_fn=dense_render,
                                              on_iteration=lambda i, loss, pkg: losses.append(float(loss.detach())))
    finally:
        sl.update_pose = ref_update
    assert n_it == len(gold["track_losses"])
    _close(losses, gold["track_losses"], "tracking loss per iteration")
    _close(np.array(taus), gold["track_taus"], "pose increments per iteration", rtol=1e-3, atol_scale=1e-4)
    _close(cam.R.numpy(), gold["track_end_R"], "final R", rtol=1e-5, atol_scale=1e-6)
    _close(cam.T.numpy(), gold["track_end_T"], "final T", rtol=1e-4, atol_scale=1e-5)
    _close([float(cam.exposure_a.detach()), float(cam.exposure_b.detach())], gold["track_end_exposure"], "exposure", rtol=1e-3, atol_scale=1e-3)
    assert abs(float(median_depth) - float(gold["track_median_depth"])) < 1e-4 * float(gold["track_median_depth"])
    _close(pkg["depth"].detach().numpy(), gold["track_last_depth"], "last rendered depth")
    # ---- what the front end decides next (is_keyframe / add_to_window), on this run's own render outputs ----
    check_keyframe_decisions(gold, cfg, sc, cam, pkg, median_depth, lambda c: dense_render(c, sc["gaussians"], sc["pipe"], sc["background"]), exact=True)


def check_keyframe_decisions(gold, cfg, sc, cam, pkg, median_depth, render_keyframe, exact):
    """``keyframe_utils.is_keyframe`` / ``add_to_window`` fed the product's outputs -- the tracked frame's n_touched > 0
    and the keyframes' occlusion-aware visibility rows, (n_touched > 0).long() of their renders -- against what the
    reference's own FrontEnd.is_keyframe / add_to_window (utils/slam_frontend.py:1579-1674) decided in the fixture."""
    import json
    from lvdgs.keyframe_utils import add_to_window, is_keyframe
    with torch.no_grad():
        cur_vis = (pkg["n_touched"] > 0).cpu()
        occ = {i: (render_keyframe(c)["n_touched"] > 0).long().cpu() for i, c in enumerate(sc["cameras"])}
    if exact:
        np.testing.assert_array_equal(cur_vis.numpy(), gold["kf_cur_visibility"])
        for i, v in occ.items():
            np.testing.assert_array_equal(v.numpy(), gold[f"kf_occ_{i}"])
    else:   # the HIP renderer may put a Gaussian on the other side of the "transmittance > 1/2" line at a pixel or two
        assert (cur_vis.numpy() != gold["kf_cur_visibility"]).mean() < 0.02
        for i, v in occ.items():
            assert (v.numpy() != gold[f"kf_occ_{i}"]).mean() < 0.02, i
    cameras = {i: c for i, c in enumerate(sc["cameras"])}
    cameras[7] = cam
    got = [is_keyframe(cfg, cameras, 7, last, cur_vis, occ, median_depth) for last in range(7)]
    assert got == gold["kf_is_keyframe"].tolist()
    assert any(got) and not all(got)
    for scale, want in zip((0.25, 1.0, 4.0, 16.0), gold["kf_is_keyframe_by_depth_scale"].tolist()):
        assert [is_keyframe(cfg, cameras, 7, last, cur_vis, occ, float(median_depth) * scale) for last in range(7)] == want, scale
    occ_thin = dict(occ)
    occ_thin[4] = occ[4] * (torch.arange(occ[4].numel()) % 7 == 0).long()
    cases = json.loads(str(gold["kf_windows_json"]))
    assert any(c["removed"] is not None for c in cases) and any(c["removed"] is None for c in cases)
    for c in cases:
        new_w, removed = add_to_window(cfg, cameras, 7, cur_vis, occ_thin if c["thinned"] else occ, c["window"], initialized=c["initialized"])
        assert (list(new_w), removed) == (c["new_window"], c["removed"]), c


def _cpu_refine_loss(image, gt_image, lambda_dssim, static_mask, background):
    import loss_oracle as lo
    return lo.l1_dssim_loss(image, gt_image, lambda_dssim, static_mask, background).to(image.dtype)


def test_color_refinement_replays_the_reference_loop(gold):
    """BackEnd.color_refinement (utils/slam_backend.py:393-468), first 8 iterations: which keyframes random.randint
    picks, the masked / unmasked L1 + SSIM loss, the Adam steps and the learning-rate schedule."""
    import random
    from dense_render import dense_render
    from loop_scene import build_scene, loop_config
    from lvdgs.slam_loops import color_refinement
    cfg = loop_config()
    torch.manual_seed(3)
    random.seed(3)
    sc = build_scene("cpu")
    be = _backend(sc, cfg)
    for i, cam in enumerate(sc["cameras"]):
        be.viewpoints[i] = cam
    counts, grads, seen = [], [], []
    _record_steps(sc["gaussians"].optimizer, counts, grads)
    color_refinement(be, iteration_total=8, render_fn=dense_render, loss_fn=_cpu_refine_loss,
                     on_iteration=lambda it, kf, loss: seen.append(kf))
    np.testing.assert_array_equal(seen, gold["refine_keyframes"])
    assert 4 in seen   # the keyframe with a static mask was drawn: both branches of the loss ran
    for k in ("xyz", "opacity", "scaling", "f_dc", "rotation"):
        _close(grads[0][k], gold["refine_grad0_" + k], "first gradient of " + k)
    for k, v in _snap(be.gaussians).items():
        _close(v, gold["refine_end_" + k], "final " + k, rtol=1e-3, atol_scale=1e-4)
    _close(be.gaussians.max_radii2D.numpy(), gold["refine_end_max_radii2D"], "max_radii2D")
    lr = [gp["lr"] for gp in be.gaussians.optimizer.param_groups if gp["name"] == "xyz"]
    np.testing.assert_allclose(lr, gold["refine_end_lr_xyz"], rtol=1e-12)
