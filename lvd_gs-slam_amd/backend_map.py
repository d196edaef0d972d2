"""The back end's mapping iteration, sharded over GPUs: ``map_window(backend, current_window, ...)``.

Statement of ``BackEnd.map`` (reference utils/slam_backend.py:153-390) that runs on a ``BackEnd``-shaped object
(the reference's own class works: ``BackEnd.map = lvdgs.backend_map.map_window``) with the views of one iteration --
the window's keyframes (<= 8, ``Training.window_size``) plus two random older ones (:275) -- dealt to the ranks of a
``torch.distributed`` group.  With one rank (or no process group) it is the reference's loop, step for step; that
path is what ``tests/golden/map_loop.npz`` (produced by running the reference's own ``BackEnd.map``) pins.

Per iteration and rank:
  1. render + loss of the rank's own views (HIP rasterizer, fused losses), the isotropic regulariser on rank 0
     (:303-305), ONE backward;
  2. ONE float32 SUM all-reduce (RCCL) of a flat bucket
        [ Gaussian parameter gradients (N x 14 at SH degree 0) | pose / exposure gradients of the window keyframes
          | sum over views of the screen-space gradient norms (N) | visibility counts (N) | loss ]
     and ONE int32 MAX all-reduce of [ max radii (N) | per-view (n_touched > 0) and visibility bytes ];
  3. the bookkeeping of :309-389 on the reduced values, identically on every rank: occlusion-aware visibility,
     pruning, max_radii2D, densification statistics, densify / prune, opacity reset, the Gaussian Adam step, the
     keyframe Adam step and ``update_pose``.
Every rank sees the same gradients for EVERY parameter (keyframe poses and exposures included), so the replicas stay
bit-identical without parameter broadcasts; densification draws its samples from a generator all ranks seed alike
(``GaussianModel.generator``), the two random views from one keyed on the iteration count.

Not sharded (one view per iteration, reference :95-149, :393-468): ``initialize_map``, ``color_refinement`` --
replicas only.
"""
from typing import Dict, List, Optional, Sequence

import torch
import torch.distributed as dist

from .gaussian_renderer import render
from .loss_utils import masked_mapping_loss
from .pose_utils import update_pose
from .slam_utils import get_loss_mapping

_POSE_FIELDS = ("cam_rot_delta", "cam_trans_delta", "exposure_a", "exposure_b")


def _world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def assign_views(n_window: int, n_random: int, world: int, iteration: int = 0) -> List[int]:
    """Owner rank of every view of one iteration.  Window keyframe i belongs to rank i mod world for as long as the
    window stands.  The random views go to the least-loaded ranks, starting from a rank that rotates with the iteration
    count so that no rank carries the extra view every time.  (10 views on 8 ranks still take two renders on the
    busiest rank: the iteration cannot be more than 5 x faster than on one GPU; 5 or 10 ranks divide it evenly.)"""
    owners = [i % world for i in range(n_window)]
    load = [0] * world
    for o in owners:
        load[o] += 1
    for k in range(n_random):
        order = sorted(range(world), key=lambda r: (load[r], (r - iteration - k) % world))
        owners.append(order[0])
        load[order[0]] += 1
    return owners


def random_view_indices(n_candidates: int, k: int, iteration: int, world: int, seed: int = 0) -> List[int]:
    """``torch.randperm(len(random_viewpoint_stack))[:2]`` (reference :275).  One rank: the global generator, exactly
    like the reference.  Several ranks: a generator keyed on the iteration count, so all ranks draw the same views."""
    if n_candidates <= 0:
        return []
    if world == 1:
        return torch.randperm(n_candidates)[:k].tolist()
    g = torch.Generator().manual_seed(seed * 1_000_003 + iteration)
    return torch.randperm(n_candidates, generator=g)[:k].tolist()


class FlatReducer:
    """A float32 SUM bucket and an int32 MAX bucket, each reduced with one collective per iteration.

    The buckets are re-planned whenever the tensors they are asked to carry change size or identity (densify / prune
    replace every Gaussian parameter), so a stale bucket can never be reduced in place of the live gradients."""

    def __init__(self):
        self.fbuf = None
        self.ibuf = None

    @staticmethod
    def _fit(buf, n, dtype, device):
        if buf is None or buf.numel() < n or buf.device != device:
            return torch.empty(max(n, 1), dtype=dtype, device=device)
        return buf

    def sum_floats(self, tensors: Sequence[Optional[torch.Tensor]], sizes: Sequence[int], device, group=None):
        """Pack (None counts as zeros), all-reduce SUM, return views of the reduced pieces in order.

        A tensor that already IS its slice of the bucket (a ``.grad`` assigned from the previous call's result and
        accumulated into since) stays where it is; one that merely overlaps the bucket is copied out first."""
        total = int(sum(sizes))
        self.fbuf = self._fit(self.fbuf, total, torch.float32, device)
        flat = self.fbuf[:total]
        lo, hi = flat.data_ptr(), flat.data_ptr() + 4 * total
        parts, inside = [], False
        for t, n in zip(tensors, sizes):
            if t is not None:
                t = t.reshape(-1)
                if t.dtype != torch.float32:
                    t = t.float()
                inside = inside or (t.numel() > 0 and lo <= t.data_ptr() < hi)
            parts.append(t)
        if not inside:
            if parts:
                torch.cat([t if t is not None else torch.zeros(n, dtype=torch.float32, device=device)
                           for t, n in zip(parts, sizes)], out=flat)
        else:
            off = 0
            staged = []
            for t, n in zip(parts, sizes):
                in_place = t is not None and t.numel() > 0 and t.data_ptr() == lo + 4 * off and t.is_contiguous()
                if t is not None and not in_place and t.numel() > 0 and lo <= t.data_ptr() < hi:
                    t = t.clone()       # overlaps the bucket somewhere else: take it out before anything is written
                staged.append((off, n, t, in_place))
                off += n
            for off, n, t, in_place in staged:
                if in_place:
                    continue
                if t is None:
                    flat[off:off + n].zero_()
                else:
                    flat[off:off + n].copy_(t)
        _, world = _world(group)
        if world > 1:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        return list(flat.split(list(sizes)))

    def max_ints(self, tensors: Sequence[torch.Tensor], device, group=None):
        """int32 tensors, all-reduce MAX, views of the reduced pieces."""
        sizes = [int(t.numel()) for t in tensors]
        total = int(sum(sizes))
        self.ibuf = self._fit(self.ibuf, total, torch.int32, device)
        flat = self.ibuf[:total]
        if tensors:
            torch.cat([t.reshape(-1) for t in tensors], out=flat)
        _, world = _world(group)
        if world > 1:
            dist.all_reduce(flat, op=dist.ReduceOp.MAX, group=group)
        return list(flat.split(sizes))


def _flag_words(rows: int, n: int, device):
    """rows x n byte flags, stored so that the same memory reads as int32 words (row length padded to 4)."""
    n4 = (n + 3) // 4 * 4
    return torch.zeros(rows, n4, dtype=torch.uint8, device=device)


def view_loss(backend, viewpoint, pkg):
    """Loss of one window keyframe: the static-mask branch when the keyframe carries a mask (:196-261), else
    ``get_loss_mapping`` (:263-266)."""
    image, depth = pkg["render"], pkg["depth"]
    if getattr(viewpoint, "static_mask", None) is not None:
        return masked_mapping_loss(image, depth, viewpoint, backend.background, backend.opt_params.lambda_dssim,
                                   backend.config["Training"].get("depth_lambda", 0.1))
    return get_loss_mapping(backend.config, image, viewpoint, depth=depth, monodepth=True)


def map_window(backend, current_window, prune=False, iters=1, up_pose=True, group=None, reducer: Optional[FlatReducer] = None,
               render_fn=render, view_loss_fn=None, stats: Optional[Dict] = None):
    """``BackEnd.map(current_window, prune, iters, up_pose)`` (reference utils/slam_backend.py:153-390) with the
    iteration's views sharded over ``group``.  Returns ``gaussian_split`` of the last iteration like the reference.
    ``stats`` (optional dict) receives per-iteration records: loss, views of this rank, collective time.
    ``render_fn`` / ``view_loss_fn`` default to the HIP renderer and the fused losses; the CPU tests of the loop logic
    pass the dense renderer and the float64 loss statements instead."""
    if len(current_window) == 0:
        return
    view_loss_fn = view_loss if view_loss_fn is None else view_loss_fn
    rank, world = _world(group)
    reducer = reducer if reducer is not None else getattr(backend, "_lvdgs_reducer", None) or FlatReducer()
    try:
        backend._lvdgs_reducer = reducer
    except Exception:
        pass
    cfg = backend.config
    G = backend.gaussians
    viewpoint_stack = [backend.viewpoints[kf_idx] for kf_idx in current_window]
    frames_to_optimize = cfg["Training"]["pose_window"]
    window_set = set(current_window)
    random_viewpoint_stack = [vp for cam_idx, vp in backend.viewpoints.items() if cam_idx not in window_set]
    n_window = len(current_window)
    # pose / exposure parameters the keyframe optimiser holds: their gradients ride in the float bucket
    kf_params = []
    if backend.keyframe_optimizers is not None:
        for gp in backend.keyframe_optimizers.param_groups:
            kf_params.extend(gp["params"])
    gaussian_split = False

    for _ in range(iters):
        backend.iteration_count += 1
        backend.last_sent += 1
        picks = random_view_indices(len(random_viewpoint_stack), 2, backend.iteration_count, world,
                                    seed=getattr(backend, "shard_seed", 0))
        views = viewpoint_stack + [random_viewpoint_stack[i] for i in picks]
        owners = assign_views(n_window, len(picks), world, backend.iteration_count)
        mine = [i for i, o in enumerate(owners) if o == rank]

        loss_mapping = 0
        pkgs = {}
        for i in mine:
            pkg = render_fn(views[i], G, backend.pipeline_params, backend.background)
            pkgs[i] = pkg
            if i < n_window:
                loss_mapping = loss_mapping + view_loss_fn(backend, views[i], pkg)
            else:
                loss_mapping = loss_mapping + get_loss_mapping(cfg, pkg["render"], views[i], depth=pkg["depth"], monodepth=True)
        if rank == 0:
            scaling = G.get_scaling
            isotropic_loss = torch.abs(scaling - scaling.mean(dim=1).view(-1, 1))
            loss_mapping = loss_mapping + 10 * isotropic_loss.mean()
        if torch.is_tensor(loss_mapping):
            loss_mapping.backward()

        with torch.no_grad():
            N = G.get_xyz.shape[0]
            dev = G.get_xyz.device
            # ---- what this rank's views say, in view order ----
            radii_max = torch.zeros(N, dtype=torch.int32, device=dev)
            norm_sum = torch.zeros(N, dtype=torch.float32, device=dev)
            vis_count = torch.zeros(N, dtype=torch.float32, device=dev)
            flags = _flag_words(n_window + 1, N, dev)   # rows 0..n_window-1: n_touched > 0 ; last row: seen by any view
            for i in mine:
                pkg = pkgs[i]
                vis = pkg["visibility_filter"]
                radii_max = torch.maximum(radii_max, pkg["radii"].to(torch.int32))
                vg = pkg["viewspace_points"].grad
                if vg is not None:
                    norm_sum += torch.where(vis, torch.norm(vg[:, :2], dim=-1), torch.zeros_like(norm_sum))
                vis_count += vis.to(torch.float32)
                flags[n_window, :N] |= vis.to(torch.uint8)
                if i < n_window:
                    flags[i, :N] = (pkg["n_touched"] > 0).to(torch.uint8)
            # ---- two collectives (a pruning pass reduces the flags only, see below) ----
            params = G.parameters()
            t0 = _now(dev) if stats is not None else None
            if not prune and world > 1:
                tensors = [p.grad for p in params] + [p.grad for p in kf_params] + [norm_sum, vis_count,
                           loss_mapping.detach().reshape(1).float() if torch.is_tensor(loss_mapping) else None]
                sizes = [p.numel() for p in params] + [p.numel() for p in kf_params] + [N, N, 1]
                red = reducer.sum_floats(tensors, sizes, dev, group)
                for p, g in zip(params + kf_params, red):
                    p.grad = g.view_as(p)
                norm_sum, vis_count = red[-3], red[-2]
                if stats is not None:
                    stats.setdefault("losses", []).append(red[-1].clone())
            elif stats is not None and torch.is_tensor(loss_mapping):
                stats.setdefault("losses", []).append(loss_mapping.detach().reshape(1).float())
            if world > 1:
                ired = reducer.max_ints([radii_max, flags.view(torch.int32)], dev, group)
                radii_max = ired[0]
                flags = ired[1].view(torch.uint8).view(n_window + 1, -1)
            flags = flags[:, :N]
            if stats is not None:
                stats.setdefault("iterations", []).append(dict(views=list(mine), comm_s=_now(dev) - t0))
            seen_by_any = flags[n_window].bool()

            # ---- bookkeeping of reference :309-389 on the reduced values ----
            backend.occ_aware_visibility = {}
            for idx in range(n_window):
                backend.occ_aware_visibility[current_window[idx]] = flags[idx].long()

            # Only prune on the last iteration and when we have full window (:318-348).  The reference returns from here
            # without an optimizer step and without clearing the gradients, so (unless pruning replaces the parameters)
            # they are still there when the next call's backward accumulates.  To keep that sum right across ranks
            # the gradients of this pass stay LOCAL (unreduced): the next pass reduces old + new together.
            if prune:
                if n_window == cfg["Training"]["window_size"]:
                    prune_mode = cfg["Training"]["prune_mode"]
                    prune_coviz = cfg["Training"]["prune_num"]
                    G.n_obs.fill_(0)
                    for _, visibility in backend.occ_aware_visibility.items():
                        G.n_obs += visibility.cpu()
                    to_prune = None
                    if prune_mode == "odometry":
                        to_prune = G.n_obs < 3
                    if prune_mode == "slam":
                        sorted_window = sorted(current_window, reverse=True)
                        mask = G.unique_kfIDs >= sorted_window[2]
                        if not backend.initialized:
                            mask = G.unique_kfIDs >= 0
                        to_prune = torch.logical_and(G.n_obs <= prune_coviz, mask)
                    if to_prune is not None and backend.monocular:
                        G.prune_points(to_prune.to(dev))
                        keep = ~to_prune.to(dev)
                        for idx in range(n_window):
                            k = current_window[idx]
                            backend.occ_aware_visibility[k] = backend.occ_aware_visibility[k][keep]
                    if not backend.initialized:
                        backend.initialized = True
                return False

            G.max_radii2D = torch.max(G.max_radii2D, radii_max.to(G.max_radii2D.dtype))
            G.xyz_gradient_accum += norm_sum[:, None]
            G.denom += vis_count[:, None]

            update_gaussian = backend.iteration_count % backend.gaussian_update_every == backend.gaussian_update_offset
            gaussian_split = False
            if update_gaussian:
                G.densify_and_prune(backend.opt_params.densify_grad_threshold, backend.gaussian_th, backend.gaussian_extent,
                                    backend.size_threshold)
                gaussian_split = True
            if (backend.iteration_count % backend.gaussian_reset) == 0 and (not update_gaussian):
                G.reset_opacity_nonvisible([seen_by_any])
                gaussian_split = True

            G.optimizer.step()
            G.optimizer.zero_grad(set_to_none=True)
            G.update_learning_rate(backend.iteration_count)
            if backend.keyframe_optimizers is not None:
                backend.keyframe_optimizers.step()
                backend.keyframe_optimizers.zero_grad(set_to_none=True)
            for v in views:   # exposure gradients of the random views are never stepped; do not let them pile up
                for name in _POSE_FIELDS:
                    p = getattr(v, name, None)
                    if p is not None and p.grad is not None and not any(p is q for q in kf_params):
                        p.grad = None
            # Pose update (:383-389): every rank holds the same reduced gradients, so every rank moves every keyframe
            if up_pose:
                for cam_idx in range(min(frames_to_optimize, n_window)):
                    viewpoint = viewpoint_stack[cam_idx]
                    if viewpoint.uid == 0:
                        continue
                    update_pose(viewpoint)
    return gaussian_split


def _now(dev):
    import time
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    return time.perf_counter()
