"""Mapping-window keyframe sharding: one keyframe per GPU, one gradient all-reduce per iteration.

The reference's mapping iteration renders every keyframe of the window (<= 8, plus 2 random older
ones) against the same Gaussians, sums the losses, and calls backward once
(utils/slam_backend.py:180-306).  The views are independent until that sum, so they shard:
Gaussians are replicated, view ``i`` goes to rank ``i mod p``, every rank runs forward + backward
for its own views, and the N x 14 float parameter gradient is summed across ranks with ONE
bucketed all-reduce (RCCL over xGMI; 28 MB at 500k Gaussians).  Every rank then applies the same
optimizer step, so the replicas stay bit-identical without broadcasting parameters.

What else the reference derives from per-view outputs is reduced here too:
  * ``n_touched > 0`` per window keyframe (occlusion-aware visibility, slam_backend.py:311-315)
    -> all-gather of one bool vector per view;
  * ``max_radii2D`` update (slam_backend.py:350-354)            -> element-wise MAX all-reduce;
  * densification statistics: sum of ||viewspace grad|| and visibility counts
    (slam_backend.py:355-357)                                   -> SUM all-reduce;
  * pose / exposure parameters belong to one keyframe: the owner steps them and broadcasts
    the 6 + 2 floats (slam_backend.py:381-389).
The view-independent isotropic regulariser (slam_backend.py:303-305) is added on rank 0 only, and
the two random keyframes (slam_backend.py:275) are drawn from a generator all ranks seed alike.

No collective is issued inside a render; tracking / initialisation / refinement have one view per
iteration and stay replicas.
"""
from typing import Callable, Dict, List, Optional, Sequence

import torch
import torch.distributed as dist


def _world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


class GradientBucket:
    """One flat float32 buffer for all parameter gradients: a single all-reduce per iteration."""

    def __init__(self, params: Sequence[torch.Tensor]):
        self.params = list(params)
        self.sizes = [p.numel() for p in self.params]
        self.flat = torch.zeros(sum(self.sizes), dtype=torch.float32, device=self.params[0].device) if self.params else None

    def rebuild_if_needed(self):
        if [p.numel() for p in self.params] != self.sizes or (self.params and self.flat.device != self.params[0].device):
            self.__init__(self.params)

    def all_reduce(self, group=None):
        """Sum gradients over ranks (missing gradients count as zero).

        Pack = one ``torch.cat`` into the flat buffer, one all-reduce, and no unpack: afterwards every
        ``p.grad`` is a view into the flat buffer (valid until the next call)."""
        _, world = _world(group)
        if world == 1 or not self.params:
            return
        self.rebuild_if_needed()
        parts = [(p.grad.reshape(-1) if p.grad is not None else torch.zeros(n, dtype=torch.float32, device=self.flat.device))
                 for p, n in zip(self.params, self.sizes)]
        torch.cat(parts, out=self.flat)
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        for p, g in zip(self.params, self.flat.split(self.sizes)):
            p.grad = g.view_as(p)


def owner_of(view_index: int, world: int) -> int:
    return view_index % world


def shared_random_views(num_candidates: int, k: int, iteration: int, seed: int = 0) -> List[int]:
    """The reference's ``torch.randperm(len(stack))[:2]`` with a generator every rank seeds alike."""
    if num_candidates <= 0:
        return []
    g = torch.Generator().manual_seed(seed * 1_000_003 + iteration)
    return torch.randperm(num_candidates, generator=g)[:k].tolist()


def sharded_map_iteration(render_fn: Callable, loss_fn: Callable, views: Sequence, gaussians, bucket: GradientBucket,
                          extra_loss_fn: Optional[Callable] = None, window_size: Optional[int] = None,
                          group=None) -> Dict:
    """One mapping iteration over ``views`` sharded across the ranks of ``group``.

    ``render_fn(view) -> render package dict``; ``loss_fn(view, pkg) -> scalar``;
    ``extra_loss_fn() -> scalar`` is the view-independent term (rank 0 only).  After the call
    every rank holds the summed parameter gradients of all views in ``p.grad`` and the merged
    bookkeeping below.  ``views[:window_size]`` are the window keyframes (their ``n_touched`` is
    gathered); the rest are the random older ones.
    """
    rank, world = _world(group)
    n_views = len(views)
    window_size = n_views if window_size is None else window_size
    mine = [i for i in range(n_views) if owner_of(i, world) == rank]
    loss = None
    pkgs = {}
    for i in mine:
        pkg = render_fn(views[i])
        pkgs[i] = pkg
        li = loss_fn(views[i], pkg)
        loss = li if loss is None else loss + li
    if extra_loss_fn is not None and rank == 0:
        le = extra_loss_fn()
        loss = le if loss is None else loss + le
    if loss is not None:
        loss.backward()
    bucket.all_reduce(group)

    # ---- bookkeeping the backend derives from the per-view outputs ----
    N = gaussians.get_xyz.shape[0]
    dev = gaussians.get_xyz.device
    radii_max = torch.zeros(N, dtype=torch.int32, device=dev)
    grad_norm_sum = torch.zeros(N, dtype=torch.float32, device=dev)
    vis_count = torch.zeros(N, dtype=torch.float32, device=dev)
    touched = torch.zeros(max(window_size, 1), N, dtype=torch.uint8, device=dev)
    for i, pkg in pkgs.items():
        vis = pkg["visibility_filter"]
        radii_max = torch.maximum(radii_max, torch.where(vis, pkg["radii"].to(torch.int32), torch.zeros_like(radii_max)))
        vg = pkg["viewspace_points"].grad
        if vg is not None:
            grad_norm_sum += torch.where(vis, vg[:, :2].norm(dim=-1), torch.zeros_like(grad_norm_sum))
        vis_count += vis.to(torch.float32)
        if i < window_size:
            touched[i] = (pkg["n_touched"] > 0).to(torch.uint8)
    if world > 1:
        dist.all_reduce(radii_max, op=dist.ReduceOp.MAX, group=group)
        dist.all_reduce(grad_norm_sum, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(vis_count, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(touched, op=dist.ReduceOp.MAX, group=group)  # each row is written by exactly one rank
    loss_value = torch.zeros((), device=dev) if loss is None else loss.detach().to(dev).float()
    if world > 1:
        dist.all_reduce(loss_value, op=dist.ReduceOp.SUM, group=group)
    return {"loss": loss_value, "radii_max": radii_max, "viewspace_grad_norm_sum": grad_norm_sum,
            "visibility_count": vis_count, "n_touched_gt0": touched[:window_size].bool(), "my_views": mine}


def broadcast_keyframe_params(views: Sequence, names=("cam_rot_delta", "cam_trans_delta", "exposure_a", "exposure_b", "R", "T"),
                              group=None):
    """After the owners stepped their keyframes' pose / exposure, make every replica agree."""
    rank, world = _world(group)
    if world == 1:
        return
    for i, v in enumerate(views):
        src = owner_of(i, world)
        for n in names:
            t = getattr(v, n, None)
            if t is None:
                continue
            data = t.data if isinstance(t, torch.nn.Parameter) else t
            buf = data.contiguous()
            dist.broadcast(buf, src=dist.get_global_rank(group, src) if group is not None else src, group=group)
            if buf.data_ptr() != data.data_ptr():
                data.copy_(buf)
