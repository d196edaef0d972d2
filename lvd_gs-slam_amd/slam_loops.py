"""The single-view optimisation loops around ``render()``: map initialisation, colour refinement, per-frame pose tracking.

Statements of the loop bodies of the reference's ``BackEnd.initialize_map`` (utils/slam_backend.py:95-149),
``BackEnd.color_refinement`` (:393-468) and ``FrontEnd.tracking`` (utils/slam_frontend.py:1467-1536, everything after the
MASt3R pose initialisation), written
against duck-typed ``backend`` / ``viewpoint`` objects so that the reference's own classes can be passed in.  One view
per iteration: nothing to shard (on several GPUs these run as replicas); the sharded loop is
``backend_map.map_window``.

Both are pinned by ``tests/golden/loops.npz``, produced by running the reference's own methods
(tests/golden/make_loop_golden.py): on the CPU with the same dense renderer (loop logic, exact), on the GPU with the
HIP rasterizer (tests/test_gpu_loop_golden.py).
"""
import torch

from .gaussian_renderer import render
from .pose_utils import update_pose
from .slam_utils import get_loss_mapping, get_loss_tracking, get_median_depth


def _update_max_radii(G, radii, visibility_filter):
    """``G.max_radii2D[vis] = torch.max(G.max_radii2D[vis], radii[vis])`` (reference utils/slam_backend.py:124, :460) without the boolean
    gather and scatter -- each of them a device-to-host synchronisation for the number of selected elements, in loops of 0.3-1 ms per
    iteration.  A Gaussian outside ``vis`` has radius 0 and ``max_radii2D`` is never negative, so the element-wise maximum over ALL
    Gaussians writes the same bits."""
    if radii.shape == G.max_radii2D.shape and not radii.dtype.is_floating_point:
        torch.maximum(G.max_radii2D, radii, out=G.max_radii2D)
    else:
        if visibility_filter is None:
            visibility_filter = radii > 0
        G.max_radii2D[visibility_filter] = torch.max(G.max_radii2D[visibility_filter], radii[visibility_filter])


def initialize_map(backend, cur_frame_idx, viewpoint, render_fn=render, on_iteration=None, fused="auto"):
    """``init_itr_num`` iterations of render -> get_loss_mapping(initialization=True) -> backward -> bookkeeping ->
    Adam step on one view (reference utils/slam_backend.py:95-149).  Returns the last render package.
    ``fused=False``: every iteration through the public autograd API (``render()`` -> loss -> ``backward()``)."""
    G = backend.gaussians
    render_pkg = None
    # on the GPU with the default renderer: render + loss + backward as three C-ABI calls without the autograd engine
    # (fast_mapping.MapViewPass, as in backend_map.map_window), the gradients written where autograd would have put them
    vpass = None
    if fused is not False and render_fn is render and G.get_xyz.is_cuda:
        from .fast_mapping import MapViewPass
        vpass = MapViewPass(G.get_xyz.device)
    from . import _lib
    with _lib.quiet_gc():
        render_pkg, n_touched = _initialize_map_iterations(backend, viewpoint, render_fn, on_iteration, vpass)
    backend.occ_aware_visibility[cur_frame_idx] = (n_touched > 0).long()
    return render_pkg


def _initialize_map_iterations(backend, viewpoint, render_fn, on_iteration, vpass):
    G = backend.gaussians
    render_pkg = n_touched = None
    for mapping_iteration in range(backend.init_itr_num):
        backend.iteration_count += 1
        if vpass is not None and G.get_xyz.shape[0] > 0 and type(vpass).usable(backend, viewpoint):
            render_pkg, loss_init = vpass.run(backend, viewpoint, initialization=True)
        else:
            render_pkg = render_fn(viewpoint, G, backend.pipeline_params, backend.background)
            loss_init = get_loss_mapping(backend.config, render_pkg["render"], viewpoint, depth=render_pkg["depth"], initialization=True)
            loss_init.backward()
        viewspace_point_tensor, visibility_filter = render_pkg["viewspace_points"], render_pkg["visibility_filter"]
        radii, n_touched = render_pkg["radii"], render_pkg["n_touched"]
        if on_iteration is not None:
            on_iteration(mapping_iteration, loss_init, render_pkg)
        with torch.no_grad():
            _update_max_radii(G, radii, visibility_filter)
            G.add_densification_stats(viewspace_point_tensor, visibility_filter)
            if mapping_iteration % backend.init_gaussian_update == 0:
                # replaces every parameter: the step below then finds no gradients (the reference does the same)
                G.densify_and_prune(backend.opt_params.densify_grad_threshold, backend.init_gaussian_th,
                                    backend.init_gaussian_extent, None)
            if backend.iteration_count == backend.init_gaussian_reset or (
                    backend.iteration_count == backend.opt_params.densify_from_iter):
                G.reset_opacity()
            G.optimizer.step()
            G.optimizer.zero_grad(set_to_none=True)
    return render_pkg, n_touched


def color_refinement(backend, iteration_total=26000, render_fn=render, loss_fn=None, on_iteration=None, fused="auto"):
    """The post-SLAM colour refinement (reference utils/slam_backend.py:393-468): ``iteration_total`` iterations of one
    random keyframe (``random.randint`` on the keyframe list, as the reference draws it) -> render ->
    ``(1 - l) L1 + l (1 - SSIM)``, on the static pixels only when the keyframe carries a ``static_mask`` -> backward ->
    ``max_radii2D`` -> Adam step -> position learning-rate schedule.  One view per iteration: replicas only.
    ``loss_fn(image, gt_image, lambda_dssim, static_mask, background)`` defaults to the fused HIP kernel
    (``loss_utils.l1_dssim_loss``: both means and the gradient image in one launch)."""
    import random
    if loss_fn is None:
        from .loss_utils import l1_dssim_loss as loss_fn
    G = backend.gaussians
    # on the GPU with the default renderer and loss: forward, the fused L1 + SSIM kernel (value and gradient image in one
    # launch) and backward as library calls without the autograd engine (fast_mapping.MapViewPass, masked_loss without a depth
    # term), on the keyframe's cached mask bytes
    vpass = None
    if fused is not False and render_fn is render and G.get_xyz.is_cuda:
        from .fast_mapping import MapViewPass
        from .loss_utils import l1_dssim_loss
        if loss_fn is l1_dssim_loss:
            vpass = MapViewPass(G.get_xyz.device)
    from . import _lib
    with _lib.quiet_gc():
        lam = float(backend.opt_params.lambda_dssim)
        takes = {}   # keyframe -> can the fused pass take it (checked once per keyframe: its tensors do not change under this loop)
        for iteration in range(1, iteration_total + 1):
            # (`viewpoint_idx_stack.pop(random.randint(0, len - 1))` of the reference: the same draw, the same keyframe)
            viewpoint_idx_stack = list(backend.viewpoints.keys())
            viewpoint_cam_idx = viewpoint_idx_stack[random.randint(0, len(viewpoint_idx_stack) - 1)]
            viewpoint_cam = backend.viewpoints[viewpoint_cam_idx]
            static_mask = getattr(viewpoint_cam, "static_mask", None)
            ok = takes.get(viewpoint_cam_idx) if vpass is not None else False
            if ok is None:
                ok = takes[viewpoint_cam_idx] = bool(G.get_xyz.shape[0] > 0 and type(vpass).usable(backend, viewpoint_cam, allow_static_mask=True)
                                                     and type(vpass).masked_loss_usable(viewpoint_cam, with_depth=False))
            if ok:
                render_pkg, loss = vpass.run(backend, viewpoint_cam, masked_loss=(lam, None), want_visibility=False)
                visibility_filter, radii = None, render_pkg["radii"]
            else:
                render_pkg = render_fn(viewpoint_cam, G, backend.pipeline_params, backend.background)
                image, visibility_filter, radii = render_pkg["render"], render_pkg["visibility_filter"], render_pkg["radii"]
                gt_image = viewpoint_cam.original_image.to(image.device)
                loss = loss_fn(image, gt_image, backend.opt_params.lambda_dssim, static_mask, backend.background if static_mask is not None else None)
                loss.backward()
            if on_iteration is not None:
                on_iteration(iteration, viewpoint_cam_idx, loss)
            with torch.no_grad():
                _update_max_radii(G, radii, visibility_filter)
                G.optimizer.step()
                G.optimizer.zero_grad(set_to_none=True)
                G.update_learning_rate(iteration)


def make_pose_optimizer(viewpoint, config):
    """Adam over the frame's pose deltas and exposure (reference utils/slam_frontend.py:1467-1490)."""
    lr = config["Training"]["lr"]
    return torch.optim.Adam([
        {"params": [viewpoint.cam_rot_delta], "lr": lr["cam_rot_delta"], "name": "rot_{}".format(viewpoint.uid)},
        {"params": [viewpoint.cam_trans_delta], "lr": lr["cam_trans_delta"], "name": "trans_{}".format(viewpoint.uid)},
        {"params": [viewpoint.exposure_a], "lr": 0.01, "name": "exposure_a_{}".format(viewpoint.uid)},
        {"params": [viewpoint.exposure_b], "lr": 0.01, "name": "exposure_b_{}".format(viewpoint.uid)},
    ])


def _can_fuse(viewpoint, gaussians, pipeline_params):
    dev = gaussians.get_xyz.device
    if dev.type != "cuda" or getattr(pipeline_params, "compute_cov3D_python", False) or getattr(pipeline_params, "convert_SHs_python", False):
        return False
    for name in ("cam_rot_delta", "cam_trans_delta", "exposure_a", "exposure_b"):
        p = getattr(viewpoint, name, None)
        if not torch.is_tensor(p) or p.device != dev or p.dtype is not torch.float32 or not p.is_contiguous():
            return False
    return hasattr(viewpoint, "update_RT")


def track_frame(viewpoint, gaussians, config, pipeline_params, background, tracking_itr_num=None, render_fn=render,
                on_iteration=None, fused="auto"):
    """Pose + exposure optimisation of one frame against the map (reference utils/slam_frontend.py:1467-1536): up to
    ``tracking_itr_num`` iterations of render -> get_loss_tracking -> backward -> Adam step -> ``update_pose``, stopping
    when the pose update falls under 1e-4 (utils/pose_utils.py:82).  The frame's initial ``R, T`` (MASt3R / PnP, out of
    scope here) must already be set.  Returns (last render package, median depth :1535, iterations run).

    ``fused`` ("auto" / True / False): with the HIP renderer and everything on the GPU the loop runs on
    ``fast_tracking.TrackingSession`` -- the same arithmetic as three C-ABI calls per iteration on buffers that live for
    the frame, the Adam step / retraction / camera matrices in one device kernel, no autograd engine and no host
    synchronisation inside the loop.  ``on_iteration`` then gets (iteration, loss as a 0-dim CPU tensor, None) after
    the loop.  ``fused=False`` (or another ``render_fn``) is the PyTorch loop below, statement for statement the
    reference's."""
    if fused is True or (fused == "auto" and render_fn is render and _can_fuse(viewpoint, gaussians, pipeline_params)):
        from .fast_tracking import track_frame_fused
        return track_frame_fused(viewpoint, gaussians, config, pipeline_params, background, tracking_itr_num, on_iteration)
    n_iter = config["Training"]["tracking_itr_num"] if tracking_itr_num is None else tracking_itr_num
    pose_optimizer = make_pose_optimizer(viewpoint, config)
    render_pkg, it = None, 0
    for tracking_itr in range(n_iter):
        render_pkg = render_fn(viewpoint, gaussians, pipeline_params, background)
        image, depth, opacity = render_pkg["render"], render_pkg["depth"], render_pkg["opacity"]
        pose_optimizer.zero_grad()
        loss_tracking = get_loss_tracking(config, image, depth, opacity, viewpoint)
        loss_tracking.backward()
        if on_iteration is not None:
            on_iteration(tracking_itr, loss_tracking, render_pkg)
        with torch.no_grad():
            pose_optimizer.step()
            converged = update_pose(viewpoint)
        it = tracking_itr + 1
        if converged:
            break
    median_depth = get_median_depth(render_pkg["depth"], render_pkg["opacity"])
    return render_pkg, median_depth, it
