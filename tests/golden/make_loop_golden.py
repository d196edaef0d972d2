#!/usr/bin/env python3
"""Run the reference's OWN loops -- ``BackEnd.initialize_map``, ``BackEnd.map`` (utils/slam_backend.py:95-390) and
``FrontEnd.tracking`` (utils/slam_frontend.py:1416-1536) -- on a toy scene and store what they did, iteration by
iteration, as fixtures the build's loops are replayed against.

Run in the authoring container only (needs /root/reference; never on the GPU box):

    python -B tests/golden/make_loop_golden.py

How the reference code is made to run here (CPU, no CUDA, no cv2 / MASt3R / GUI):
  * its imports of the absent ``gaussian_splatting`` package resolve to the drop-in shims (lvd_gs-slam_amd/dropin), as
    in tests/test_dropin.py; the Gaussian map is therefore the build's ``GaussianModel`` on the CPU -- the loops
    around it (losses, bookkeeping order, optimiser steps, pose retraction) are the reference's;
  * ``render`` in the reference modules' namespaces is the dense float64 autograd renderer of tests/dense_render.py
    (same signature, same seven keys); ``ssim`` / ``l1_loss`` are the float64 statements of oracle/loss_oracle.py;
  * ``torch.Tensor.cuda`` is the identity (the loops hard-code ``.cuda()``);
  * out-of-scope imports of utils/slam_frontend.py are empty stand-in modules: ``cv2``, ``gui``, ``utils.init_pose``
    (``get_pose`` returns the identity, i.e. MASt3R's "no estimate" branch :1460-1465; ``get_depth`` returns the
    frame's stored depth), ``utils.depth_utils``, ``utils.eval_utils``.
Nothing of the reference's text is stored: only the inputs (scene, cameras, images) and the numbers the loops produced.

Fixture: tests/golden/loops.npz (+ loops.json with the scalar settings).
"""
import importlib
import json
import math
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("LVDGS_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), ROOT, os.path.join(ROOT, "lvd_gs-slam_amd", "dropin"), REF):
    sys.path.insert(0, p)

import lvdgs  # noqa: E402,F401
import loss_oracle as lo  # noqa: E402
from dense_render import dense_render  # noqa: E402
from loop_scene import build_scene, loop_config  # noqa: E402  (tests/loop_scene.py: shared with the replay tests)


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def load_reference():
    torch.Tensor.cuda = lambda self, *a, **k: self
    _stub("cv2")
    gui = _stub("gui")
    gui.gui_utils = _stub("gui.gui_utils", GaussianPacket=lambda **k: None)
    _stub("utils.init_pose", save_depth_comparison=lambda *a, **k: None,
          get_pose=lambda **k: (np.eye(4), None), get_depth=lambda img1, img2, model: model(img2))
    _stub("utils.depth_utils", process_depth=lambda *a, **k: None)
    _stub("utils.eval_utils", eval_ate=lambda *a, **k: None, save_gaussians=lambda *a, **k: None)
    backend = importlib.import_module("utils.slam_backend")
    frontend = importlib.import_module("utils.slam_frontend")
    for mod in (backend, frontend):
        mod.render = dense_render
    backend.ssim = lambda a, b: lo.ssim(a, b).to(a.dtype)
    backend.l1_loss = lambda a, b: lo.l1_loss(a, b).to(a.dtype)
    return backend, frontend


class Recorder:
    """Per-iteration records taken from hooks around the optimiser steps (gradients as the step sees them)."""

    def __init__(self):
        self.rows = []

    def hook_optimizer(self, opt, tag):
        def pre(optimizer, args, kwargs):
            row = {"tag": tag, "n": int(optimizer.param_groups[0]["params"][0].shape[0])}
            for gp in optimizer.param_groups:
                p = gp["params"][0]
                row["grad_" + gp["name"]] = None if p.grad is None else p.grad.detach().clone().numpy()
            self.rows.append(row)
        opt.register_step_pre_hook(pre)


def snapshot(gaussians):
    return {k: v.detach().clone().numpy() for k, v in gaussians._params_by_name().items()} | {
        "max_radii2D": gaussians.max_radii2D.clone().numpy(), "unique_kfIDs": gaussians.unique_kfIDs.clone().numpy(),
        "n_obs": gaussians.n_obs.clone().numpy(), "xyz_gradient_accum": gaussians.xyz_gradient_accum.clone().numpy(),
        "denom": gaussians.denom.clone().numpy()}


def main():
    backend_mod, frontend_mod = load_reference()
    cfg = loop_config()
    from utils.camera_utils import Camera as RefCamera  # the reference's own Camera class (imports through the shims)
    out = {}

    # ------------------------------------------------------------------ BackEnd.initialize_map
    torch.manual_seed(0)
    sc = build_scene("cpu", RefCamera)
    be = backend_mod.BackEnd(cfg)
    be.gaussians, be.background, be.pipeline_params = sc["gaussians"], sc["background"], sc["pipe"]
    be.opt_params = types.SimpleNamespace(**cfg["opt_params"])
    be.cameras_extent = 6.0
    be.set_hyperparams()
    losses = []
    ref_loss = backend_mod.get_loss_mapping
    backend_mod.get_loss_mapping = lambda *a, **k: (losses.append(float(ref_loss(*a, **k).detach())) or ref_loss(*a, **k))
    rec = Recorder()
    rec.hook_optimizer(be.gaussians.optimizer, "init")
    be.viewpoints[0] = sc["cameras"][0]
    be.initialize_map(0, sc["cameras"][0])
    out["init_losses"] = np.array(losses)
    out["init_n_per_iter"] = np.array([r["n"] for r in rec.rows])
    # an iteration that densifies replaces the parameters before the step (slam_backend.py:131-144): that step sees no
    # gradients.  Keep the gradients of the first two iterations that do step.
    stepping = [i for i, r in enumerate(rec.rows) if r["grad_xyz"] is not None]
    out["init_stepping_iterations"] = np.array(stepping)
    for n, it in enumerate(stepping[:2]):
        for k in ("xyz", "opacity", "scaling", "f_dc", "rotation"):
            out[f"init_grad{n}_{k}"] = rec.rows[it]["grad_" + k]
    for k, v in snapshot(be.gaussians).items():
        out["init_end_" + k] = v
    out["init_occ0"] = be.occ_aware_visibility[0].numpy()
    out["init_iteration_count"] = np.array(be.iteration_count)

    # ------------------------------------------------------------------ BackEnd.map (window of 4 + older keyframes)
    torch.manual_seed(1)
    sc = build_scene("cpu", RefCamera)
    be = backend_mod.BackEnd(cfg)
    be.gaussians, be.background, be.pipeline_params = sc["gaussians"], sc["background"], sc["pipe"]
    be.opt_params = types.SimpleNamespace(**cfg["opt_params"])
    be.cameras_extent = 6.0
    be.set_hyperparams()
    be.initialized = True
    for i, cam in enumerate(sc["cameras"]):
        be.viewpoints[i] = cam
    window = sc["window"]
    be.current_window = window
    be.keyframe_optimizers = sc["make_keyframe_optimizer"](be.viewpoints, window, cfg)
    losses.clear()
    rec = Recorder()
    rec.hook_optimizer(be.gaussians.optimizer, "map")
    kf_rows = []
    be.keyframe_optimizers.register_step_pre_hook(lambda opt, a, k: kf_rows.append(
        {gp["name"]: (None if gp["params"][0].grad is None else gp["params"][0].grad.detach().clone().numpy()) for gp in opt.param_groups}))
    picks = []
    ref_randperm = torch.randperm
    backend_mod.torch.randperm = lambda n, *a, **k: (lambda r: (picks.append(r[:2].tolist()) or r))(ref_randperm(n, *a, **k))
    be.map(window, iters=sc["map_iters"])
    backend_mod.torch.randperm = ref_randperm
    out["map_random_picks"] = np.array(picks)
    out["map_n_per_iter"] = np.array([r["n"] for r in rec.rows])
    stepping = [i for i, r in enumerate(rec.rows) if r["grad_xyz"] is not None]
    out["map_stepping_iterations"] = np.array(stepping)
    for n, it in enumerate(stepping[:2]):
        for k in ("xyz", "opacity", "scaling", "f_dc", "rotation"):
            out[f"map_grad{n}_{k}"] = rec.rows[it]["grad_" + k]
        for name, g in kf_rows[it].items():
            out[f"map_kfgrad{n}_{name}"] = g
    out["map_losses"] = np.array(losses)
    for k, v in snapshot(be.gaussians).items():
        out["map_end_" + k] = v
    for i, cam in enumerate(sc["cameras"]):
        out[f"map_end_R_{i}"], out[f"map_end_T_{i}"] = cam.R.numpy(), cam.T.numpy()
        out[f"map_end_exposure_{i}"] = np.array([float(cam.exposure_a), float(cam.exposure_b)])
    for kf in window:
        out[f"map_end_occ_{kf}"] = be.occ_aware_visibility[kf].numpy()
    out["map_iteration_count"] = np.array(be.iteration_count)
    # the pruning pass that follows every keyframe (slam_backend.py:601): full window -> prune by n_obs
    n_before = be.gaussians.get_xyz.shape[0]
    be.map(window, prune=True)
    out["prune_n_before_after"] = np.array([n_before, be.gaussians.get_xyz.shape[0]])
    out["prune_end_n_obs"] = be.gaussians.n_obs.clone().numpy()
    out["prune_end_xyz"] = be.gaussians.get_xyz.detach().clone().numpy()
    backend_mod.get_loss_mapping = ref_loss

    # ------------------------------------------------------------------ BackEnd.color_refinement (first 8 of its 26000 iterations)
    import itertools
    import random
    torch.manual_seed(3)
    random.seed(3)
    sc = build_scene("cpu", RefCamera)
    be = backend_mod.BackEnd(cfg)
    be.gaussians, be.background, be.pipeline_params = sc["gaussians"], sc["background"], sc["pipe"]
    be.opt_params = types.SimpleNamespace(**cfg["opt_params"])
    for i, cam in enumerate(sc["cameras"]):
        be.viewpoints[i] = cam
    backend_mod.tqdm = lambda it: itertools.islice(it, 8)   # the reference hard-codes iteration_total = 26000 (:399)
    rec = Recorder()
    rec.hook_optimizer(be.gaussians.optimizer, "refine")
    ref_render = backend_mod.render
    seen = []
    backend_mod.render = lambda vp, *a, **k: (seen.append(int(vp.uid)) or ref_render(vp, *a, **k))
    be.color_refinement()
    backend_mod.render = ref_render
    out["refine_keyframes"] = np.array(seen)
    for k in ("xyz", "opacity", "scaling", "f_dc", "rotation"):
        out["refine_grad0_" + k] = rec.rows[0]["grad_" + k]
    for k, v in snapshot(be.gaussians).items():
        out["refine_end_" + k] = v
    out["refine_end_lr_xyz"] = np.array([gp["lr"] for gp in be.gaussians.optimizer.param_groups if gp["name"] == "xyz"])

    # ------------------------------------------------------------------ FrontEnd.tracking
    torch.manual_seed(2)
    sc = build_scene("cpu", RefCamera)
    cfg_t = json.loads(json.dumps(cfg))
    cfg_t["dynamic_filtering"] = {"enabled": False}
    fe = frontend_mod.FrontEnd(cfg_t, model=lambda img: sc["track_mono_depth"])
    fe.set_hyperparams()
    fe.gaussians, fe.background, fe.pipeline_params = sc["gaussians"], sc["background"], sc["pipe"]
    fe.device = "cpu"
    fe.dataset = types.SimpleNamespace(dist_coeffs=None)
    from utils.multiprocessing_utils import FakeQueue
    fe.q_main2vis = FakeQueue()
    fe.cameras = {0: sc["cameras"][0], 1: sc["track_camera"]}
    fe.current_window = [0]
    fe.use_every_n_frames = 1
    tr_losses, taus = [], []
    ref_lt = frontend_mod.get_loss_tracking
    frontend_mod.get_loss_tracking = lambda *a, **k: (tr_losses.append(float(ref_lt(*a, **k).detach())) or ref_lt(*a, **k))
    ref_up = frontend_mod.update_pose

    def logged_update(cam, *a, **k):
        taus.append(torch.cat([cam.cam_trans_delta.detach(), cam.cam_rot_delta.detach()]).numpy().copy())
        return ref_up(cam, *a, **k)
    frontend_mod.update_pose = logged_update
    pkg = fe.tracking(1, sc["track_camera"])
    cam = sc["track_camera"]
    out["track_losses"] = np.array(tr_losses)
    out["track_taus"] = np.array(taus)
    out["track_end_R"], out["track_end_T"] = cam.R.numpy(), cam.T.numpy()
    out["track_end_exposure"] = np.array([float(cam.exposure_a), float(cam.exposure_b)])
    out["track_median_depth"] = np.array(float(fe.median_depth))
    out["track_last_depth"] = pkg["depth"].detach().numpy()

    # ------------------------------------------------------------------ FrontEnd.is_keyframe / add_to_window
    # What the front end decides right after tracking (utils/slam_frontend.py:1579-1674), from the tracked frame's
    # n_touched > 0 and the keyframes' occlusion-aware visibility -- both outputs of render().  The tracked frame joins
    # the scene's seven keyframes as index 7; every keyframe's visibility row is what the back end stores for it
    # ((n_touched > 0).long(), utils/slam_backend.py:311-315) on the same map.
    with torch.no_grad():
        cur_vis = (pkg["n_touched"] > 0)
        occ = {i: (dense_render(c, fe.gaussians, fe.pipeline_params, fe.background)["n_touched"] > 0).long() for i, c in enumerate(sc["cameras"])}
    fe.cameras = {i: c for i, c in enumerate(sc["cameras"])}
    fe.cameras[7] = cam
    fe.initialized = True
    out["kf_cur_visibility"] = cur_vis.numpy()
    for i, v in occ.items():
        out[f"kf_occ_{i}"] = v.numpy()
    out["kf_is_keyframe"] = np.array([bool(fe.is_keyframe(7, last, cur_vis, occ)) for last in range(7)])
    # the same question at other scales of the scene (the thresholds are in units of the median depth)
    decisions = []
    for scale in (0.25, 1.0, 4.0, 16.0):
        keep = fe.median_depth
        fe.median_depth = keep * scale
        decisions.append([bool(fe.is_keyframe(7, last, cur_vis, occ)) for last in range(7)])
        fe.median_depth = keep
    out["kf_is_keyframe_by_depth_scale"] = np.array(decisions)
    windows = [[6, 5, 4, 3], [6, 5, 4], [6, 5, 4, 3, 2, 1], [3, 1, 6, 0, 5], [6]]
    # a keyframe that sees little of what the new frame sees (its row thinned out) is the one the covisibility rule removes
    occ_thin = dict(occ)
    occ_thin[4] = occ[4] * (torch.arange(occ[4].numel()) % 7 == 0).long()
    res = []
    for w in windows:
        for table, init in ((occ, True), (occ_thin, True), (occ_thin, False)):
            fe.initialized = init
            new_w, removed = fe.add_to_window(7, cur_vis, table, list(w))
            res.append((w, init, table is occ_thin, new_w, removed))
    fe.initialized = True
    out["kf_windows_json"] = np.array(json.dumps([dict(window=w, initialized=i, thinned=t, new_window=[int(x) for x in nw],
                                                      removed=None if r is None else int(r)) for w, i, t, nw, r in res]))
    print("is_keyframe", out["kf_is_keyframe"].tolist(), "by scale", out["kf_is_keyframe_by_depth_scale"].tolist())
    print("add_to_window", [(w, nw, r) for w, _, _, nw, r in res][:6])

    np.savez_compressed(os.path.join(HERE, "loops.npz"), **{k: v for k, v in out.items() if v is not None})
    print("wrote loops.npz:", {k: (v.shape if hasattr(v, "shape") else v) for k, v in list(out.items())[:12]}, "...")
    print("init losses", out["init_losses"][:3], "->", out["init_losses"][-1], "N", out["init_n_per_iter"])
    print("map N", out["map_n_per_iter"], "picks", out["map_random_picks"].tolist(), "prune", out["prune_n_before_after"])
    print("track losses", out["track_losses"][:3], "->", out["track_losses"][-1], len(tr_losses), "iterations")


if __name__ == "__main__":
    main()
