"""The back end's mapping iteration, sharded over GPUs: ``map_window(backend, current_window, ...)``.

Statement of ``BackEnd.map`` (reference utils/slam_backend.py:153-390) that runs on a ``BackEnd``-shaped object
(the reference's own class works: ``BackEnd.map = lvdgs.backend_map.map_window``) with the views of one iteration --
the window's keyframes (<= 8, ``Training.window_size``) plus two random older ones (:275) -- dealt to the ranks of a
``torch.distributed`` group.  With one rank (or no process group) it is the reference's loop, step for step; that
path is what ``tests/golden/map_loop.npz`` (produced by running the reference's own ``BackEnd.map``) pins.

Views are dealt whole as far as they divide evenly; the rest are cut into BANDS of tile rows (``plan_pieces``): a view's
tiles are independent and its loss is a sum over pixels, so a band's forward + backward (``lvdgs_args.tile_row_begin /
_end``) yields that band's share of every gradient and the shares add up in the all-reduce.  Ten views on eight ranks
are 1.25 views of blend work per rank instead of two renders on the busiest one.

Per iteration and rank:
  1. render + loss + backward of the rank's own views -- three C-ABI calls per view on the GPU (fast_mapping.MapViewPass:
     no autograd engine, gradients written where autograd would have put them), the autograd path for views with a
     static mask and off the GPU -- and the isotropic regulariser on rank 0 (:303-305);
  2. ONE float32 SUM all-reduce (RCCL) of a flat bucket
        [ Gaussian parameter gradients (N x 14 at SH degree 0) | pose / exposure gradients of the window keyframes
          | sum over views of the screen-space gradient norms (N) | visibility counts (N)
          | screen-space gradients (N x 2) of the views that were split into bands | loss ],
     one int32 MAX all-reduce of the max radii (N) and one uint8 MAX all-reduce of the per-view (n_touched > 0) bytes
     (a byte-wise OR: several ranks hold flags of a view that was split);
  3. the bookkeeping of :309-389 on the reduced values, identically on every rank: occlusion-aware visibility,
     pruning, max_radii2D, densification statistics, densify / prune, opacity reset, the Gaussian Adam step, the
     keyframe Adam step and ``update_pose``.
Every rank sees the same gradients for EVERY parameter (keyframe poses and exposures included), so the replicas stay
bit-identical without parameter broadcasts; densification draws its samples from a generator all ranks seed alike
(``GaussianModel.generator``), the two random views from one keyed on the iteration count.

Not sharded (one view per iteration, reference :95-149, :393-468): ``initialize_map``, ``color_refinement`` --
replicas only.
"""
import ctypes as C
import os
from typing import Dict, List, Optional, Sequence

import torch
import torch.distributed as dist

from . import _lib

from .fast_mapping import MapViewPass, MapWindowBatch, _PARAM_FIELDS
from .gaussian_renderer import render
from .loss_utils import masked_mapping_loss
from .pose_utils import update_pose
from .slam_utils import get_loss_mapping

_POSE_FIELDS = ("cam_rot_delta", "cam_trans_delta", "exposure_a", "exposure_b")


def _world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


# "leftover": views are dealt whole as far as they divide evenly, the remaining ones are cut into bands; "all": every
# splittable view is cut into one band per rank (tests; also the finest balance); "none": whole views only (round 2).
SPLIT_POLICY = "leftover"


def plan_pieces(rows: Sequence[int], world: int, iteration: int = 0, splittable: Optional[Sequence[bool]] = None,
                policy: Optional[str] = None):
    """The iteration's work as pieces ``(view, row0, row1, rank)``: tile rows [row0, row1) of view ``view`` (``rows[v]``
    = tile rows of view v) go to ``rank``.  Whole views first -- view i of the first ``(V // world) * world`` to rank
    i mod world, so a window keyframe stays with its rank for as long as the window stands -- then the remaining
    V mod world views as bands: their rows, laid end to end, are cut into ``world`` equal shares (to whole tile rows),
    starting at a rank that rotates with the iteration count.  A view that cannot be rendered in bands (``splittable[v]``
    false: it goes through autograd with a loss that is no sum over pixels) is dealt whole to the least loaded rank.
    Every rank ends with the same number of rows (+- one tile row per cut) as long as the views are of one size."""
    policy = SPLIT_POLICY if policy is None else policy
    V = len(rows)
    splittable = [True] * V if splittable is None else list(splittable)
    if world <= 1:
        return [(v, 0, int(rows[v]), 0) for v in range(V)]
    pieces, load = [], [0] * world          # load: tile rows per rank
    n_whole = V if policy == "none" else ((V // world) * world if policy == "leftover" else 0)
    split = []
    for v in range(V):
        if v < n_whole and policy != "none":
            pieces.append((v, 0, int(rows[v]), v % world)); load[v % world] += int(rows[v])
        elif policy != "none" and splittable[v] and rows[v] > 0:
            split.append(v)
        else:   # whole, to the least loaded rank (rotating among equals)
            k = len(pieces)
            r = min(range(world), key=lambda q: (load[q], (q - iteration - k) % world))
            pieces.append((v, 0, int(rows[v]), r)); load[r] += int(rows[v])
    if split and policy == "all":
        for v in split:
            n = int(rows[v])
            cuts = [round(n * k / world) for k in range(world + 1)]
            for k in range(world):
                if cuts[k + 1] > cuts[k]:
                    pieces.append((v, cuts[k], cuts[k + 1], (k + v + iteration) % world))
    elif split:
        # the split views' rows laid end to end, cut so that every rank ends with the same number of rows: a rank that
        # already carries more (an unsplittable view dealt whole) gets a shorter share
        total = sum(int(rows[v]) for v in split)
        order = sorted(range(world), key=lambda q: (load[q], (q - iteration) % world))
        level = (total + sum(load)) / world
        want = [max(0.0, level - load[q]) for q in order]
        scale = total / max(sum(want), 1e-9)
        cuts, acc = [0], 0.0
        for w in want:
            acc += w * scale
            cuts.append(min(total, round(acc)))
        cuts[-1] = total
        start = 0
        for v in split:
            n = int(rows[v])
            for k in range(world):
                lo, hi = max(cuts[k], start), min(cuts[k + 1], start + n)
                if hi > lo:
                    pieces.append((v, lo - start, hi - start, order[k]))
            start += n
    return pieces


def assign_views(n_window: int, n_random: int, world: int, iteration: int = 0) -> List[int]:
    """Owner rank of every view when views are dealt whole (``plan_pieces(..., policy="none")``)."""
    return [p[3] for p in plan_pieces([1] * (n_window + n_random), world, iteration, policy="none")]


class _BandView:
    """A viewpoint whose target image and mono depth are zero outside pixel rows [y0, y1): every pixel outside fails the
    loss's own validity masks (reference utils/slam_utils.py:97-98, :112: ``gt.sum(0) > rgb_boundary_threshold``,
    ``gt_depth > 0.01``), so ``get_loss_mapping`` of the WHOLE rendered image is the band's share of the view's loss --
    the means keep the whole image's pixel count.  For pieces that go through autograd (CPU tests, views MapViewPass
    does not take); everything else is the viewpoint's own attribute (pose and exposure tensors included)."""

    def __init__(self, viewpoint, y0, y1):
        object.__setattr__(self, "_vp", viewpoint)
        img = viewpoint.original_image
        band = torch.zeros_like(img)
        band[..., y0:y1, :] = img[..., y0:y1, :]
        md = viewpoint.mono_depth
        mdt = md if torch.is_tensor(md) else torch.from_numpy(md)
        mband = torch.zeros_like(mdt)
        mband[..., y0:y1, :] = mdt[..., y0:y1, :]
        object.__setattr__(self, "original_image", band)
        object.__setattr__(self, "mono_depth", mband if torch.is_tensor(md) else mband.numpy())

    def __getattr__(self, name):
        return getattr(object.__getattribute__(self, "_vp"), name)

    def __setattr__(self, name, value):
        if name in ("_lvdgs_mono_depth",):
            object.__setattr__(self, name, value)
        else:
            setattr(object.__getattribute__(self, "_vp"), name, value)


def _tile_rows(viewpoint):
    return (int(viewpoint.image_height) + 15) // 16


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

    def plan_floats(self, sizes: Sequence[int], device):
        """Views of the float bucket laid out for ``sizes``: whoever produces a piece can write it where ``sum_floats``
        will look for it (a piece found in place is not copied)."""
        total = int(sum(sizes))
        self.fbuf = self._fit(self.fbuf, total, torch.float32, device)
        return list(self.fbuf[:total].split([int(n) for n in sizes]))

    def plan_ints(self, total: int, device):
        """The first ``total`` int32 words of the int bucket (uninitialised), for a producer to lay its pieces out in."""
        self.ibuf = self._fit(self.ibuf, int(total), torch.int32, device)
        return self.ibuf[:int(total)]

    def sum_floats(self, tensors: Sequence[Optional[torch.Tensor]], sizes: Sequence[int], device, group=None, scatter=None, scatter_len=0):
        """Pack (None counts as zeros), all-reduce SUM, return views of the reduced pieces in order.
        ``scatter`` (a ShardedAdam) with ``scatter_len``: the first ``scatter_len`` floats are reduce-SCATTERED instead -- this
        rank's summed share is left in ``self.share``, the views of that part are NOT reduced -- and the rest all-reduced.

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
        self.share = None
        if world > 1 and scatter is not None and scatter_len > 0:
            self.share = scatter.reduce_scatter(flat[:scatter_len])
            if total > scatter_len:
                dist.all_reduce(flat[scatter_len:], op=dist.ReduceOp.SUM, group=group)
        elif world > 1:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        return list(flat.split(list(sizes)))

    def max_ints(self, tensors: Sequence[torch.Tensor], device, group=None, async_op=False):
        """int32 tensors, all-reduce MAX, views of the reduced pieces (``async_op``: and the collective's work handle, to be
        waited for before the views are read -- None on one rank)."""
        sizes = [int(t.numel()) for t in tensors]
        total = int(sum(sizes))
        self.ibuf = self._fit(self.ibuf, total, torch.int32, device)
        flat = self.ibuf[:total]
        offs = [sum(sizes[:i]) for i in range(len(sizes))]
        in_place = bool(tensors) and all(t.is_contiguous() and t.data_ptr() == flat.data_ptr() + 4 * o for t, o in zip(tensors, offs))
        if tensors and not in_place:   # (pieces a producer laid out with plan_ints are where they belong already)
            torch.cat([t.reshape(-1) for t in tensors], out=flat)
        _, world = _world(group)
        work = None
        if world > 1:
            work = dist.all_reduce(flat, op=dist.ReduceOp.MAX, group=group, async_op=async_op)
        return (list(flat.split(sizes)), work) if async_op else list(flat.split(sizes))


FLAGS_AS_INT32 = False   # set by ``collective_preflight`` when the communicator cannot MAX-reduce uint8


class _WidenedFlagsWork:
    """Work handle of a uint8 MAX all-reduce carried out on an int32 copy: ``wait()`` waits for the collective, then narrows the
    result back into the caller's bytes."""

    def __init__(self, flags, wide, work):
        self.flags, self.wide, self.work = flags, wide, work

    def wait(self):
        if self.work is not None:
            self.work.wait()
        self.flags.copy_(self.wide)


def _max_bytes(flags: torch.Tensor, group=None, async_op=False):
    """uint8 all-reduce MAX in place: the byte-wise OR of 0 / 1 flags.  (Packed into int32 words and reduced with MAX --
    round 2 -- a word's high byte decided for all four: flags of other ranks were dropped.)  ``async_op``: returns the
    collective's work handle (None when there is nothing to reduce) instead of the flags.  ``FLAGS_AS_INT32``: every flag widened
    to an int32 of its own for the collective (four times the bytes, the same result) -- the fallback for a communicator whose
    uint8 MAX failed the preflight."""
    _, world = _world(group)
    work = None
    if world > 1 and flags.numel():
        if FLAGS_AS_INT32:
            wide = flags.to(torch.int32)
            w = dist.all_reduce(wide, op=dist.ReduceOp.MAX, group=group, async_op=async_op)
            work = _WidenedFlagsWork(flags, wide, w if async_op else None)
            if not async_op:
                work.wait()
        else:
            work = dist.all_reduce(flags, op=dist.ReduceOp.MAX, group=group, async_op=async_op)
    return work if async_op else flags


def collective_preflight(device, group=None, aux_group=None, sharded_adam=False, fail=None):
    """One tiny collective of every (operation, dtype, communicator) a sharded mapping iteration issues, BEFORE anything is timed or
    depends on it: float32 SUM all-reduce (the gradient bucket), int32 MAX all-reduce (radii), uint8 MAX all-reduce (visibility flags) on
    the main group and on the auxiliary one, ``reduce_scatter_tensor`` / ``all_gather_into_tensor`` (the sharded Adam) on the main group.
    Each runs in a ``try`` of its own and is checked against the value it must produce; the verdict is MIN-all-reduced over the ranks, so
    every rank takes the same fallback:

      * uint8 MAX fails                -> ``FLAGS_AS_INT32`` (the flags travel as int32);
      * anything on ``aux_group`` fails -> ``use_aux_group`` False (the two MAX collectives go to the main group: same results, no overlap);
      * reduce-scatter / all-gather    -> ``use_sharded_adam`` False (the replicated Adam step);
      * float32 SUM or int32 MAX on the main group -> ``fatal``: the iteration cannot run, the caller must stop with the reason.

    ``fail``: operations to fail on purpose, comma separated (also ``LVDGS_PREFLIGHT_FAIL``) -- the tests' way to walk the fallbacks:
    ``f32_sum, i32_max, u8_max, aux_i32_max, aux_u8_max, reduce_scatter, all_gather``.  Returns a dict for the benchmark's JSON line."""
    global FLAGS_AS_INT32
    rank, world = _world(group)
    fail = set(x.strip() for x in (fail if fail is not None else os.environ.get("LVDGS_PREFLIGHT_FAIL", "")).split(",") if x.strip())
    report = dict(world=world, ops={}, notes=[], use_aux_group=aux_group is not None, use_sharded_adam=bool(sharded_adam), fatal=None,
                  flags_as_int32=False)
    if world <= 1:
        return report
    ok_buf = torch.ones(1, dtype=torch.float32, device=device)

    def agreed(ok):
        """MIN over the ranks (on the main group with the one operation everything else depends on anyway)."""
        ok_buf.fill_(1.0 if ok else 0.0)
        try:
            dist.all_reduce(ok_buf, op=dist.ReduceOp.MIN, group=group)
            return bool(ok_buf.item() > 0.5)
        except Exception:   # noqa: BLE001
            return False

    def attempt(name, fn):
        err = None
        try:
            if name in fail:
                raise RuntimeError("failed on purpose (LVDGS_PREFLIGHT_FAIL)")
            fn()
            if device.type == "cuda":
                torch.cuda.synchronize(device)
        except Exception as e:   # noqa: BLE001
            err = f"{type(e).__name__}: {e}"
        ok = agreed(err is None)
        report["ops"][name] = "ok" if ok else (err or "failed on another rank")
        return ok

    def reduce_check(dtype, op, g, want):
        def fn():
            t = torch.full((64,), rank + 1, dtype=dtype, device=device)
            dist.all_reduce(t, op=op, group=g)
            if not bool((t == want).all()):
                raise RuntimeError(f"wrong result {t[:2].tolist()} (expected {want})")
        return fn

    total = world * (world + 1) // 2
    if not attempt("f32_sum", reduce_check(torch.float32, dist.ReduceOp.SUM, group, total)):
        report["fatal"] = "float32 SUM all-reduce on the main communicator: " + report["ops"]["f32_sum"]
    if not attempt("i32_max", reduce_check(torch.int32, dist.ReduceOp.MAX, group, world)):
        report["fatal"] = report["fatal"] or "int32 MAX all-reduce on the main communicator: " + report["ops"]["i32_max"]
    u8 = attempt("u8_max", reduce_check(torch.uint8, dist.ReduceOp.MAX, group, world))
    if aux_group is not None:
        aux_ok = attempt("aux_i32_max", reduce_check(torch.int32, dist.ReduceOp.MAX, aux_group, world))
        aux_u8 = attempt("aux_u8_max", reduce_check(torch.uint8, dist.ReduceOp.MAX, aux_group, world))
        if not aux_ok:
            report["use_aux_group"] = False
            report["notes"].append("the auxiliary communicator failed its int32 MAX: the MAX collectives run on the main communicator")
        u8 = u8 and aux_u8
    if not u8:
        FLAGS_AS_INT32 = report["flags_as_int32"] = True
        report["notes"].append("uint8 MAX all-reduce failed: the visibility flags travel as int32")
    if sharded_adam:
        def rs():
            src = torch.arange(world * 8, dtype=torch.float32, device=device) * (rank + 1)
            out = torch.empty(8, dtype=torch.float32, device=device)
            dist.reduce_scatter_tensor(out, src, op=dist.ReduceOp.SUM, group=group)
            want = torch.arange(rank * 8, rank * 8 + 8, dtype=torch.float32, device=device) * total
            if not torch.equal(out, want):
                raise RuntimeError("wrong result")

        def ag():
            mine = torch.full((8,), float(rank), dtype=torch.float32, device=device)
            full = torch.empty(world * 8, dtype=torch.float32, device=device)
            dist.all_gather_into_tensor(full, mine, group=group)
            if not torch.equal(full, torch.arange(world, dtype=torch.float32, device=device).repeat_interleave(8)):
                raise RuntimeError("wrong result")
        ok = attempt("reduce_scatter", rs)
        ok = attempt("all_gather", ag) and ok
        if not ok:
            report["use_sharded_adam"] = False
            report["notes"].append("reduce_scatter_tensor / all_gather_into_tensor failed: the replicated Adam step")
    return report


class ShardedAdam:
    """The map's Adam step with every rank stepping ONE SHARE of the parameters (``map_window(sharded_adam=True)`` /
    ``backend.shard_optimizer``): the float bucket's parameter-gradient part is reduce-scattered instead of all-reduced -- rank r
    receives the sum of share r, a contiguous 1 / world of the bucket, cutting through the parameter tensors wherever it falls
    (Adam is elementwise) -- every rank runs Adam on its share alone, and the stepped shares are all-gathered back into the
    parameter tensors: the collective volume of the all-reduce (reduce-scatter + all-gather IS an all-reduce), 1 / world of the
    Adam arithmetic and moment traffic per rank.  Every parameter element is computed by exactly one rank and copied to the
    others, so the replicas stay bit-identical.  The moments of a rank are current on its share only: ``sync_moments()``
    all-gathers them before anything that edits or reads them per Gaussian (densification, pruning, the opacity reset, a
    pruning pass, a switch back to the replicated step) -- every ~150 iterations in the reference's schedule.
    Priced in DESIGN.md section 4: worth ~35 us of ~1.1 ms at 500 k Gaussians on eight GPUs, ~100 us of ~3 ms at 2 M; OFF by
    default (no multi-GPU box was available to measure it)."""

    def __init__(self, group=None):
        self.group = group
        self.rank, self.world = _world(group)
        self.stale = False          # moments current on the local share only
        self.layout = None          # sizes of the parameter tensors the stale moments' shares were cut from (step())
        self.bufs = {}

    def _buf(self, name, n, dev):
        b = self.bufs.get(name)
        if b is None or b.numel() < n or b.device != dev:
            b = torch.empty(max(n, 1), dtype=torch.float32, device=dev)
            self.bufs[name] = b
        return b[:n]

    def plan(self, params):
        """Flat layout of the parameter tensors (the order of the float bucket) padded to a multiple of the world size."""
        sizes = [int(p.numel()) for p in params]
        total = sum(sizes)
        padded = -(-max(total, 1) // self.world) * self.world
        share = padded // self.world
        return sizes, total, padded, share

    def pieces(self, params):
        """[(tensor index, start in the tensor, start in the share, length)] of this rank's share."""
        sizes, total, padded, share = self.plan(params)
        lo, hi = self.rank * share, min((self.rank + 1) * share, total)
        out, off = [], 0
        for k, n in enumerate(sizes):
            a, b = max(lo, off), min(hi, off + n)
            if b > a:
                out.append((k, a - off, a - lo, b - a))
            off += n
        return out, share

    def reduce_scatter(self, flat_grads):
        """``flat_grads``: the bucket's parameter-gradient part (padded length).  Returns this rank's summed share."""
        share = flat_grads.numel() // self.world
        out = self._buf("grad_share", share, flat_grads.device)
        dist.reduce_scatter_tensor(out, flat_grads, op=dist.ReduceOp.SUM, group=self.group)
        return out

    def _gather_into(self, tensors, name):
        """All-gather every rank's share of ``tensors`` (flat layout of ``plan``) and copy the result into them."""
        sizes, total, padded, share = self.plan(tensors)
        dev = tensors[0].device
        mine = self._buf(name + "_share", share, dev)
        pieces, _ = self.pieces(tensors)
        for k, a, c, n in pieces:
            mine[c:c + n].copy_(tensors[k].detach().reshape(-1)[a:a + n])
        full = self._buf(name + "_full", padded, dev)
        dist.all_gather_into_tensor(full, mine, group=self.group)
        off = 0
        for t, n in zip(tensors, sizes):
            if n:
                t.detach().reshape(-1).copy_(full[off:off + n])
            off += n

    def step(self, optimizer, params, grad_share, skip=()):
        """Adam on this rank's share (the optimiser's own state tensors and hyper-parameters; its step counters advance on
        every rank), then the all-gather of the stepped parameters.  ``skip``: indices of parameter tensors that have no
        gradient this iteration (just replaced by the bookkeeping): not stepped, counters untouched."""
        import math
        by_param = {}
        for gp in optimizer.param_groups:
            for q in gp["params"]:
                by_param[id(q)] = gp
        items = []
        for k, p in enumerate(params):
            gp = by_param[id(p)]
            st = optimizer.state[p]
            if len(st) == 0:
                st["step"] = torch.tensor(0.0)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            if k not in skip:
                st["step"] = st["step"] + 1
            items.append((gp, st))
        pieces, _ = self.pieces(params)
        pieces = [pc for pc in pieces if pc[0] not in skip]
        if len(skip) == len(params):
            return
        dev = params[0].device
        use_kernel = dev.type == "cuda" and all(_on_gpu_f32(p.detach(), optimizer.state[p]["exp_avg"], optimizer.state[p]["exp_avg_sq"]) for p in params if p.numel())
        if use_kernel and pieces:
            arr = (_lib.AdamTensor * 8)()
            b1, b2 = items[pieces[0][0]][0]["betas"]
            eps = float(items[pieces[0][0]][0]["eps"])
            for n_, (k, a, c, n) in enumerate(pieces):
                gp, st = items[k]
                if tuple(gp["betas"]) != (b1, b2) or float(gp["eps"]) != eps or gp.get("weight_decay", 0) or gp.get("amsgrad", False):
                    use_kernel = False
                    break
                t = arr[n_]
                t.param = C.c_void_p(params[k].data_ptr() + 4 * a)
                t.grad = C.c_void_p(grad_share.data_ptr() + 4 * c)
                t.exp_avg = C.c_void_p(st["exp_avg"].data_ptr() + 4 * a)
                t.exp_avg_sq = C.c_void_p(st["exp_avg_sq"].data_ptr() + 4 * a)
                t.numel, t.step, t.lr = n, int(st["step"]), float(gp["lr"])
            if use_kernel:
                with _lib.on_device(dev):
                    _lib.check(_lib.lib().lvdgs_adam_step(arr, len(pieces), float(b1), float(b2), eps, _lib.raw_stream(dev)), "lvdgs_adam_step (share)")
        if not use_kernel:
            for k, a, c, n in pieces:   # torch.optim.Adam's statements (no weight decay / amsgrad) on the slices
                gp, st = items[k]
                b1, b2 = gp["betas"]
                step = float(st["step"])
                g = grad_share[c:c + n]
                pv = params[k].detach().reshape(-1)[a:a + n]
                m, v = st["exp_avg"].reshape(-1)[a:a + n], st["exp_avg_sq"].reshape(-1)[a:a + n]
                m.mul_(b1).add_(g, alpha=1 - b1)
                v.mul_(b2).addcmul_(g, g, value=1 - b2)
                denom = (v.sqrt() / math.sqrt(1 - b2 ** step)).add_(gp["eps"])
                pv.addcdiv_(m, denom, value=-gp["lr"] / (1 - b1 ** step))
        self._gather_into(params, "param")
        self.stale = True
        self.layout = tuple(int(p.numel()) for p in params)

    def sync_moments(self, optimizer, params):
        """Bring every rank's moments up to date (no-op when they are).  The shares are cut from the flat layout of the CURRENT
        parameter tensors: a map whose size changed since the last step (an extension, a pruning) has moved every share
        boundary, and gathering then would overwrite fresh moments with another rank's stale ones -- which is why ``map_window``
        never returns with stale moments (it syncs before anything that resizes the map, and before it returns)."""
        if not self.stale:
            return
        if self.layout is not None and self.layout != tuple(int(p.numel()) for p in params):
            raise RuntimeError("ShardedAdam.sync_moments: the map's size changed while the Adam moments were sharded "
                               f"(stepped on {self.layout}, now {tuple(int(p.numel()) for p in params)}): sync before resizing the map")
        for key in ("exp_avg", "exp_avg_sq"):
            tensors = [optimizer.state[p][key] for p in params]
            self._gather_into(tensors, key)
        self.stale = False


_P = lambda t: None if t is None else C.c_void_p(t.data_ptr())


def _on_gpu_f32(*tensors):
    return all(t is not None and t.is_cuda and t.dtype is torch.float32 and t.is_contiguous() for t in tensors)


class KeyframeStepper:
    """``keyframe_optimizers.step()`` + ``update_pose`` of the window's keyframes (reference utils/slam_backend.py:381-389)
    as ONE launch per keyframe (``lvdgs_pose_step``): Adam on the keyframe's pose deltas and exposure with the learning
    rates of the optimiser's groups, the SE(3) retraction, deltas zeroed -- no ``if angle < 1e-5`` / ``converged``
    round trips to the host.  Built from the ``torch.optim.Adam`` the back end made for the window (:545-598): the
    parameters stay the viewpoints' own tensors, the Adam moments live here (the torch optimiser's state is not used)."""

    def __init__(self, optimizer, viewpoints: Sequence, pose_window: int):
        self.optimizer = optimizer
        self.signature = self._sig(optimizer)
        lr_of = {}
        for gp in optimizer.param_groups:
            for p in gp["params"]:
                lr_of[id(p)] = (float(gp["lr"]), gp.get("betas", (0.9, 0.999)), float(gp.get("eps", 1e-8)))
        self.items = []
        for idx, vp in enumerate(viewpoints):
            has = {n: id(getattr(vp, n, None)) in lr_of for n in _POSE_FIELDS}
            if not any(has.values()):
                continue
            pose = has["cam_rot_delta"] and has["cam_trans_delta"] and idx < pose_window and vp.uid != 0
            if (has["cam_rot_delta"] or has["cam_trans_delta"]) and not pose:
                raise NotImplementedError("pose deltas in the optimiser of a keyframe that update_pose skips")
            dev = vp.exposure_a.device
            a = _lib.PoseStepArgs()
            state = torch.zeros(24, dtype=torch.float32, device=dev)
            keep = [state]
            if pose:
                R = vp.R.detach().to(device=dev, dtype=torch.float32).contiguous().clone()
                T = vp.T.detach().to(device=dev, dtype=torch.float32).contiguous().clone()
                vp.update_RT(R, T)     # the viewpoint's R, T ARE the buffers the kernel advances in place
                keep += [R, T]
                a.R, a.T, a.cam_rot_delta, a.cam_trans_delta = _P(R), _P(T), _P(vp.cam_rot_delta), _P(vp.cam_trans_delta)
                a.lr_rot, a.lr_trans = lr_of[id(vp.cam_rot_delta)][0], lr_of[id(vp.cam_trans_delta)][0]
                # the launch also leaves the matrices the next render reads (Camera._matrices: getWorld2View2, a bmm and
                # torch.inverse -- whose error check waits for the GPU -- per keyframe and iteration otherwise)
                if hasattr(vp, "_derived_key") and torch.is_tensor(getattr(vp, "projection_matrix", None)):
                    P = vp.projection_matrix
                    Pc = P.detach().to(device=dev, dtype=torch.float32).contiguous()
                    view, full, centre = (torch.empty(4, 4, device=dev), torch.empty(4, 4, device=dev), torch.empty(3, device=dev))
                    a.projmatrix_raw, a.viewmatrix, a.projmatrix, a.campos = _P(Pc), _P(view), _P(full), _P(centre)
                    keep += [P, Pc, view, full, centre]
            if has["exposure_a"]:
                a.exposure_a, a.lr_exposure = _P(vp.exposure_a), lr_of[id(vp.exposure_a)][0]
            if has["exposure_b"]:
                a.exposure_b, a.lr_exposure = _P(vp.exposure_b), lr_of[id(vp.exposure_b)][0]
            any_p = next(getattr(vp, n) for n in _POSE_FIELDS if has[n])
            (a.beta1, a.beta2), a.eps = lr_of[id(any_p)][1], lr_of[id(any_p)][2]
            a.state, a.converged_threshold = _P(state), -1.0
            self._import_state(optimizer, vp, has, state)
            self.items.append((vp, a, pose, has, keep))

    _STATE_SLOTS = {"cam_rot_delta": (0, 19), "cam_trans_delta": (6, 20), "exposure_a": (12, 21), "exposure_b": (14, 22)}   # (moments, step count) in `state`

    @classmethod
    def _import_state(cls, optimizer, vp, has, state):
        """Moments and step counts the torch optimiser already holds for this keyframe (iterations that went through
        ``keyframe_optimizers.step()`` -- ``fused=False``, ``up_pose=False`` -- before the stepper took over) continue here;
        a fresh optimiser has none and the state stays zero.  The counterpart of ``export_to_optimizer``."""
        host = None
        for name, (off, cnt) in cls._STATE_SLOTS.items():
            p = getattr(vp, name, None)
            st = optimizer.state.get(p) if (has[name] and p is not None) else None
            if not st or "exp_avg" not in st:
                continue
            host = torch.zeros(24, dtype=torch.float32) if host is None else host
            n = p.numel()
            mv = host[off:off + 2 * n].view(n, 2)
            mv[:, 0] = st["exp_avg"].detach().reshape(-1).float().cpu()
            mv[:, 1] = st["exp_avg_sq"].detach().reshape(-1).float().cpu()
            host[cnt] = float(st["step"])
        if host is not None:
            state.copy_(host.to(state.device))

    @staticmethod
    def _sig(optimizer):
        return tuple((id(p), float(gp["lr"])) for gp in optimizer.param_groups for p in gp["params"])

    @staticmethod
    def usable(optimizer, viewpoints) -> bool:
        if type(optimizer) is not torch.optim.Adam:
            return False
        for gp in optimizer.param_groups:
            if gp.get("weight_decay", 0) or gp.get("amsgrad", False) or gp.get("maximize", False):
                return False
            if not _on_gpu_f32(*gp["params"]):
                return False
        owned = {id(getattr(vp, n, None)) for vp in viewpoints for n in _POSE_FIELDS}
        return all(id(p) in owned for gp in optimizer.param_groups for p in gp["params"]) and all(hasattr(vp, "update_RT") for vp in viewpoints)

    def export_to_optimizer(self):
        """Write the moments and step counts kept here into the torch optimiser's own state, so that a later
        ``keyframe_optimizers.step()`` (``fused=False``, ``up_pose=False``) continues from them instead of from a stale or
        empty state."""
        for vp, a, pose, has, keep in self.items:
            st = keep[0].detach().cpu()
            for name, (off, cnt) in self._STATE_SLOTS.items():
                if not has[name] or float(st[cnt]) == 0.0:
                    continue
                p = getattr(vp, name)
                n = p.numel()
                mv = st[off:off + 2 * n].view(n, 2)
                self.optimizer.state[p] = {"step": torch.tensor(float(st[cnt])), "exp_avg": mv[:, 0].clone().to(p.device).view_as(p),
                                           "exp_avg_sq": mv[:, 1].clone().to(p.device).view_as(p)}

    def step(self):
        """All keyframes' steps in ONE launch (``lvdgs_pose_step_batch``; they touch disjoint cameras)."""
        L = _lib.lib()
        if not self.items:
            return
        if getattr(self, "_batch", None) is None or len(self._batch) != len(self.items):
            self._batch = (_lib.PoseStepArgs * len(self.items))()
        dev = self.items[0][0].exposure_a.device
        for i, (vp, a, pose, has, keep) in enumerate(self.items):
            g = lambda n: getattr(vp, n).grad if has[n] else None
            gr, gt, ga, gb = g("cam_rot_delta"), g("cam_trans_delta"), g("exposure_a"), g("exposure_b")
            for t in (gr, gt, ga, gb):
                if t is not None and not _on_gpu_f32(t):
                    raise ValueError("KeyframeStepper: gradients must be contiguous float32 GPU tensors")
            a.grad_rot, a.grad_trans, a.grad_exposure_a, a.grad_exposure_b = _P(gr), _P(gt), _P(ga), _P(gb)
            if pose and (vp.R is not keep[1] or vp.T is not keep[2]):
                # somebody gave the keyframe a new pose since the last step (update_RT): adopt it into the buffers
                keep[1].copy_(vp.R.detach().to(keep[1]))
                keep[2].copy_(vp.T.detach().to(keep[2]))
                vp.update_RT(keep[1], keep[2])
            if vp.exposure_a.device != dev:
                raise ValueError("KeyframeStepper: keyframes on different devices")
            C.memmove(C.byref(self._batch, i * C.sizeof(_lib.PoseStepArgs)), C.byref(a), C.sizeof(_lib.PoseStepArgs))
        with _lib.on_device(dev):
            _lib.check(L.lvdgs_pose_step_batch(self._batch, len(self.items), _lib.raw_stream(dev)), "lvdgs_pose_step_batch")
        for vp, a, pose, has, keep in self.items:
            if pose and hasattr(vp, "_derived_key"):
                # R / T were advanced in place (their version counters did not move): install the matrices the launch
                # wrote as the camera's cache for exactly these tensors, or drop the stale cache
                if len(keep) > 3 and vp.projection_matrix is keep[3]:
                    R, T, P = keep[1], keep[2], keep[3]
                    vp._derived, vp._derived_key = (keep[5], keep[6], keep[7]), (R, R._version, T, T._version, P, P._version)
                else:
                    vp._derived_key = None


class _ViewStats:
    """radii_max / norm_sum / vis_count / flags of the pieces a rank rendered, filled by one ``lvdgs_view_stats`` launch per
    piece on the GPU (the PyTorch statements otherwise).  ``split``: the views of the iteration that were cut into bands,
    in order: a band's screen-space gradient goes to that view's plane of ``split_xy`` (the norm is of the sum over the
    bands, which only exists after the all-reduce), and the view is counted as seen by the rank holding its first band."""

    def __init__(self, N, n_window, dev, split: Sequence[int] = (), float_pieces=None):
        """``float_pieces`` (sharded runs): norm_sum, vis_count and the split planes are the given slices of the reducer's
        float bucket, so the collective has nothing to pack."""
        self.N, self.n_window, self.dev = N, n_window, dev
        self.split = list(split)
        # one zeroed block (one fill launch per iteration instead of three): radii_max | norm_sum, vis_count, split planes | flags
        n_float = 0 if float_pieces is not None else (2 + 2 * len(self.split)) * N
        words = N + n_float
        block = torch.zeros(4 * words + n_window * N, dtype=torch.uint8, device=dev)
        self.radii_max = block[:4 * N].view(torch.int32)
        self.flags = block[4 * words:].view(n_window, N)   # row i: n_touched > 0 in window view i
        if float_pieces is not None:
            self.norm_sum, self.vis_count, self.split_xy = float_pieces
            self.norm_sum.zero_(); self.vis_count.zero_(); self.split_xy.zero_()
        else:
            both = block[4 * N:4 * words].view(torch.float32)
            self.norm_sum, self.vis_count, self.split_xy = both[:N], both[N:2 * N], both[2 * N:]

    def targets(self, view, row0):
        """(radii_max, norm_sum, vis_count | None, touched_row | None, split plane | None) of one piece, for
        ``MapViewPass.run(stats=...)``: the statistics taken by the launch that finishes the piece's loss."""
        N = self.N
        k = self.split.index(view) if view in self.split else -1
        return (self.radii_max, self.norm_sum, self.vis_count if row0 == 0 else None, self.flags[view] if view < self.n_window else None,
                self.split_xy[2 * N * k:2 * N * (k + 1)] if k >= 0 else None)

    def add(self, view, row0, pkg):
        """One piece of view ``view`` (``row0``: its first tile row; the piece starting at row 0 counts the view as seen)."""
        if getattr(pkg["viewspace_points"], "stats_taken", False):
            return
        N, n_window = self.N, self.n_window
        radii, nt, vg = pkg["radii"], pkg["n_touched"], pkg["viewspace_points"].grad
        k = self.split.index(view) if view in self.split else -1
        plane = self.split_xy[2 * N * k:2 * N * (k + 1)] if k >= 0 else None
        counts = row0 == 0
        fast = (self.dev.type == "cuda" and radii.dtype is torch.int32 and nt.dtype is torch.int32 and radii.is_contiguous()
                and nt.is_contiguous() and (vg is None or _on_gpu_f32(vg)))
        if fast:
            row = self.flags[view] if view < n_window else None
            with _lib.on_device(self.dev):
                _lib.check(_lib.lib().lvdgs_view_stats(N, _P(radii), _P(nt), _P(vg), _P(self.radii_max), _P(self.norm_sum),
                                                       _P(self.vis_count) if counts else None, _P(row), _P(plane),
                                                       _lib.raw_stream(self.dev)), "lvdgs_view_stats")
            return
        vis = pkg["visibility_filter"]
        self.radii_max.copy_(torch.maximum(self.radii_max, radii.to(torch.int32)))
        if vg is not None:
            if plane is not None:
                plane.view(N, 2).copy_(torch.where(vis[:, None], vg[:, :2], torch.zeros_like(vg[:, :2])))
            else:
                self.norm_sum += torch.where(vis, torch.norm(vg[:, :2], dim=-1), torch.zeros_like(self.norm_sum))
        if counts:
            self.vis_count += vis.to(torch.float32)
        if view < n_window:
            self.flags[view] = (nt > 0).to(torch.uint8)

    def apply(self, G, radii_max, norm_sum, vis_count, split_xy):
        """``max_radii2D`` / ``xyz_gradient_accum`` / ``denom`` from the (reduced) statistics (reference :350-357)."""
        N, n_split = self.N, len(self.split)
        fast = (self.dev.type == "cuda" and _on_gpu_f32(G.max_radii2D, G.xyz_gradient_accum, G.denom, norm_sum, vis_count)
                and radii_max.dtype is torch.int32 and radii_max.is_contiguous() and (n_split == 0 or _on_gpu_f32(split_xy))
                and G.max_radii2D.numel() == N and G.xyz_gradient_accum.numel() == N and G.denom.numel() == N)
        if fast:
            with _lib.on_device(self.dev):
                _lib.check(_lib.lib().lvdgs_map_stats_apply(N, _P(radii_max), _P(norm_sum), _P(vis_count), _P(split_xy) if n_split else None,
                                                            n_split, _P(G.max_radii2D), _P(G.xyz_gradient_accum), _P(G.denom),
                                                            _lib.raw_stream(self.dev)), "lvdgs_map_stats_apply")
            return
        G.max_radii2D = torch.maximum(G.max_radii2D, radii_max)   # (int32 promotes to the float32 of max_radii2D)
        if n_split:
            norm_sum = norm_sum + torch.norm(split_xy.view(n_split, N, 2), dim=-1).sum(0)
        torch._foreach_add_([G.xyz_gradient_accum, G.denom], [norm_sum[:, None], vis_count[:, None]])


def _isotropic_term(G, weight=10.0):
    """``weight * mean |s - mean_k s|`` over the activated scales (reference utils/slam_backend.py:303-305) as an
    autograd expression -- the statement the reference makes."""
    scaling = G.get_scaling
    return weight * torch.abs(scaling - scaling.mean(dim=1).view(-1, 1)).mean()


def _isotropic_fused(G, weight=10.0):
    """The same term through ``lvdgs_isotropic_reg``: value returned, gradient ADDED to ``G._scaling.grad`` (two
    launches instead of a dozen).  None when the model is not the standard one on the GPU."""
    raw = getattr(G, "_scaling", None)
    if not (getattr(G, "standard_activations", False) and torch.is_tensor(raw) and _on_gpu_f32(raw) and raw.dim() == 2 and raw.shape[1] == 3):
        return None
    N, dev = raw.shape[0], raw.device
    if raw.grad is None:
        raw.grad = torch.zeros_like(raw)
    elif not _on_gpu_f32(raw.grad):
        return None
    L = _lib.lib()
    scratch = torch.empty(int(L.lvdgs_isotropic_scratch_bytes(N)), dtype=torch.uint8, device=dev)
    loss = torch.empty((), dtype=torch.float32, device=dev)
    with _lib.on_device(dev):
        _lib.check(L.lvdgs_isotropic_reg(N, _P(raw), _P(raw.grad), float(weight), _P(scratch), scratch.numel(), _P(loss),
                                         _lib.raw_stream(dev)), "lvdgs_isotropic_reg")
    return loss


def view_loss(backend, viewpoint, pkg):
    """Loss of one window keyframe: the static-mask branch when the keyframe carries a mask (:196-261), else
    ``get_loss_mapping`` (:263-266)."""
    image, depth = pkg["render"], pkg["depth"]
    if getattr(viewpoint, "static_mask", None) is not None:
        return masked_mapping_loss(image, depth, viewpoint, backend.background, backend.opt_params.lambda_dssim,
                                   backend.config["Training"].get("depth_lambda", 0.1))
    return get_loss_mapping(backend.config, image, viewpoint, depth=depth, monodepth=True)


def map_window(backend, current_window, prune=False, iters=1, up_pose=True, group=None, reducer: Optional[FlatReducer] = None,
               render_fn=render, view_loss_fn=None, stats: Optional[Dict] = None, fused=True, aux_group=None, bands_ok: Optional[bool] = None,
               sharded_adam: Optional[bool] = None):
    """``BackEnd.map(current_window, prune, iters, up_pose)`` (reference utils/slam_backend.py:153-390) with the
    iteration's views sharded over ``group``.  Returns ``gaussian_split`` of the last iteration like the reference.
    ``stats`` (optional dict) receives per-iteration records: loss, views of this rank, collective time.
    ``render_fn`` / ``view_loss_fn`` default to the HIP renderer and the fused losses; the CPU tests of the loop logic
    pass the dense renderer and the float64 loss statements instead.  ``fused`` (GPU only): the isotropic regulariser,
    the per-view statistics and the keyframes' Adam + ``update_pose`` run as single launches (``lvdgs_isotropic_reg``,
    ``lvdgs_view_stats``, ``lvdgs_pose_step``) instead of PyTorch statements; ``fused=False`` keeps the statements.
    ``aux_group``: a second process group over the same ranks (``dist.new_group()``, made once by the caller -- creating a
    group is itself a collective of the whole job): the two small MAX collectives (radii, visibility flags) are started on it
    BEFORE the float SUM is issued on ``group`` and waited for after it -- with RCCL a communicator and stream of their own, so
    that their latency (two dependent ~30 us collectives at eight ranks) lies under the gradient all-reduce instead of behind
    it.  Without it they are started on ``group`` itself (same results, no overlap).  ``backend.shard_aux_group`` is read when
    the argument is None.
    ``bands_ok``: may window views be cut into bands of tile rows (several ranks)?  Only a loss that is a sum over pixels gated
    by the target's own validity masks survives that (``_BandView``): ``get_loss_mapping`` is one, so the default is True for the
    default ``view_loss_fn`` -- keyframes with a static mask excepted, as always -- and False for a caller's own, which then
    says so itself (the CPU tests' float64 statement of the same loss does).
    ``sharded_adam`` (default: ``backend.shard_optimizer``, else False; several ranks only): the Gaussian Adam as reduce-scatter ->
    every rank steps its share -> all-gather (``ShardedAdam``) instead of an all-reduce and the same step on every rank."""
    if len(current_window) == 0:
        return
    with _lib.quiet_gc():   # (no full-heap garbage collection inside a 2 ms loop; the host's collector is as before afterwards)
        return _map_window(backend, current_window, prune, iters, up_pose, group, reducer, render_fn, view_loss_fn, stats, fused, aux_group,
                           bands_ok, sharded_adam)


def _map_window(backend, current_window, prune, iters, up_pose, group, reducer, render_fn, view_loss_fn, stats, fused, aux_group, bands_ok,
                sharded_adam):
    if bands_ok is None:
        bands_ok = view_loss_fn is None or view_loss_fn is view_loss
    if sharded_adam is None:
        sharded_adam = bool(getattr(backend, "shard_optimizer", False))
    view_loss_fn = view_loss if view_loss_fn is None else view_loss_fn
    rank, world = _world(group)
    reducer = reducer if reducer is not None else getattr(backend, "_lvdgs_reducer", None) or FlatReducer()
    try:
        backend._lvdgs_reducer = reducer
    except Exception:
        pass
    cfg = backend.config
    G = backend.gaussians
    sharder = getattr(backend, "_lvdgs_sharder", None)
    if sharded_adam and world > 1 and (sharder is None or sharder.group is not group):
        sharder = ShardedAdam(group)
        try:
            backend._lvdgs_sharder = sharder
        except Exception:
            pass
    if sharder is not None and not (sharded_adam and world > 1):
        sharder.sync_moments(G.optimizer, G.parameters())   # back to the replicated step: every rank needs every moment
        sharder = None
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
        marks = _PhaseMarks(G.get_xyz.device) if stats is not None else None
        vpass = _view_pass(backend) if (fused and render_fn is render and view_loss_fn is view_loss) else None
        # a view renders in bands when its loss is a sum over pixels: get_loss_mapping, i.e. no static mask (L1 + SSIM and a
        # count-normalised depth term are not); window views without MapViewPass take the _BandView route through autograd
        # (the random views always score with get_loss_mapping; a caller's own window loss splits only if the caller says so)
        splittable = [(bands_ok and getattr(v, "static_mask", None) is None) or i >= n_window for i, v in enumerate(views)]
        pieces = plan_pieces([_tile_rows(v) for v in views], world, backend.iteration_count, splittable)
        split = sorted({v for v, r0, r1, _ in pieces if (r0, r1) != (0, _tile_rows(views[v]))})
        mine = [(v, r0, r1) for v, r0, r1, o in pieces if o == rank]

        loss_mapping = 0      # pieces that go through autograd (a graph)
        loss_direct = None    # pieces rendered, scored and differentiated by MapViewPass (values only)
        direct_losses = []    # (summed once: an addition per view is a launch per view)
        pkgs = []
        # sharded: the first piece's backward writes the parameter gradients straight into their slices of the float
        # bucket the all-reduce works on (no packing copy); the statistics' float pieces live there too
        plan = first = None
        N0 = G.get_xyz.shape[0]
        pad = 0   # (sharded Adam: the parameter-gradient part of the bucket is a multiple of the world size)
        if world > 1:
            live = G.parameters()
            if sharder is not None:
                _, total_p, padded_p, _ = sharder.plan(live)
                pad = padded_p - total_p
            plan = reducer.plan_floats([p.numel() for p in live] + [pad] + [p.numel() for p in kf_params] + [N0, N0, 2 * N0 * len(split), 1],
                                       live[0].device)
            if vpass is not None:
                first = {n: plan[k].view_as(p) for k, (n, p) in enumerate(zip(_PARAM_FIELDS, live))}
        with torch.no_grad():
            vs = _ViewStats(N0, n_window, G.get_xyz.device, split, (plan[-4], plan[-3], plan[-2]) if plan is not None else None)
        # A window keyframe with a static mask -- under the reference's default configuration every one of them
        # (utils/slam_frontend.py:1218,1429-1433) -- is scored by L1 + SSIM on the static pixels and the masked depth term (:196-261):
        # (lambda_dssim, depth_lambda) for MapViewPass / MapWindowBatch; every other view by get_loss_mapping (None)
        masked_spec = (float(backend.opt_params.lambda_dssim), float(cfg["Training"].get("depth_lambda", 0.1)))
        masked_of = [masked_spec if (i < n_window and getattr(v, "static_mask", None) is not None and vpass is not None
                                     and MapViewPass.masked_loss_usable(v)) else None for i, v in enumerate(views)]
        # The WHOLE views of this rank (one GPU: the whole window; two GPUs: five views each; four: two each + bands; eight: one +
        # a band) with every stage in one launch for all of them: forward chains, blend passes, static-mask losses
        # (fast_mapping.MapWindowBatch -- a KITTI-size frame alone leaves the chip half empty; LVDGS_MAP_BATCH=0: view by view).
        # The rank's bands, and whole views the batch cannot take, follow view by view and add to the batch's gradients.
        together = []
        if vpass is not None and os.environ.get("LVDGS_MAP_BATCH", "1") != "0":
            together = [(v, r0, r1) for v, r0, r1 in mine if (r0, r1) == (0, _tile_rows(views[v]))
                        and (masked_of[v] is not None or getattr(views[v], "static_mask", None) is None or v >= n_window)]
            if not MapWindowBatch.usable(backend, [views[v] for v, _, _ in together], [masked_of[v] for v, _, _ in together]):
                together = []
        if together:
            batch = getattr(backend, "_lvdgs_window_batch", None)
            if batch is None or batch.passes[0] is not vpass:
                batch = backend._lvdgs_window_batch = MapWindowBatch(vpass)
            for (v, r0, _), (pkg, l) in zip(together, batch.run(backend, [views[v] for v, _, _ in together], first=first,
                                                                 stats=[vs.targets(v, r0) for v, r0, _ in together],
                                                                 masked=[masked_of[v] for v, _, _ in together])):
                pkgs.append((v, r0, pkg))
                direct_losses.append(l)
        for v, r0, r1 in [pc for pc in mine if pc not in together]:
            whole = (r0, r1) == (0, _tile_rows(views[v]))
            masked = v < n_window and getattr(views[v], "static_mask", None) is not None
            if vpass is not None and MapViewPass.usable(backend, views[v], allow_static_mask=True) and (not masked or masked_of[v] is not None):
                # a window keyframe with a static mask: the static-mask loss (its colour-gradient image from the fused L1 + SSIM
                # launch, the depth term's gradient inside the backward blend pass); every other view: get_loss_mapping inside
                # the backward blend pass
                pkg, l = vpass.run(backend, views[v], first=first, band=None if whole else (r0, r1), masked_loss=masked_of[v],
                                   stats=vs.targets(v, r0))
                pkgs.append((v, r0, pkg))
                direct_losses.append(l)
                continue
            pkg = render_fn(views[v], G, backend.pipeline_params, backend.background)
            pkgs.append((v, r0, pkg))
            if not whole and not cfg["Training"]["rgb_boundary_threshold"] >= 0:
                raise ValueError("a view in bands needs rgb_boundary_threshold >= 0: pixels outside the band are masked out by their zeroed target")
            target = views[v] if whole else _BandView(views[v], 16 * r0, min(16 * r1, int(views[v].image_height)))
            if v < n_window:
                loss_mapping = loss_mapping + view_loss_fn(backend, target, pkg)
            else:
                loss_mapping = loss_mapping + get_loss_mapping(cfg, pkg["render"], target, depth=pkg["depth"], monodepth=True)
        # isotropic regulariser (:303-305), rank 0 only: fused kernel after the backward where it applies, else in the graph
        fuse_iso = rank == 0 and fused and G.get_xyz.is_cuda and getattr(G, "standard_activations", False)
        if rank == 0 and not fuse_iso:
            loss_mapping = loss_mapping + _isotropic_term(G)
        if torch.is_tensor(loss_mapping):
            loss_mapping.backward()
            loss_mapping = loss_mapping.detach()
        if direct_losses:
            loss_direct = direct_losses[0] if len(direct_losses) == 1 else torch.stack(direct_losses).sum()
        if loss_direct is not None:
            loss_mapping = loss_mapping + loss_direct
        if fuse_iso:
            iso = _isotropic_fused(G)
            if iso is None:          # not the standard model after all: the autograd statement
                iso = _isotropic_term(G)
                iso.backward()
            loss_mapping = loss_mapping + iso.detach()

        with torch.no_grad():
            N = G.get_xyz.shape[0]
            dev = G.get_xyz.device
            if marks: marks.mark("views")
            # ---- what this rank's pieces say, in order (pieces that went through MapViewPass have said it already) ----
            for v, r0, pkg in pkgs:
                vs.add(v, r0, pkg)
            radii_max, norm_sum, vis_count, split_xy, flags = vs.radii_max, vs.norm_sum, vs.vis_count, vs.split_xy, vs.flags
            # ---- the collectives (a pruning pass reduces the flags only, see below) ----
            params = G.parameters()
            if marks: marks.mark("statistics")
            # the two small MAX collectives first (on the auxiliary communicator when there is one): they are in flight while
            # the float bucket is reduced
            small_works = []
            if world > 1:
                small_group = aux_group if aux_group is not None else (getattr(backend, "shard_aux_group", None) or group)
                (radii_red,), w_radii = reducer.max_ints([radii_max], dev, small_group, async_op=True)
                small_works = [w for w in (w_radii, _max_bytes(flags, small_group, async_op=True)) if w is not None]
            grad_share = None
            if not prune and world > 1:
                tensors = [p.grad for p in params] + [None] + [p.grad for p in kf_params] + [norm_sum, vis_count, split_xy,
                           loss_mapping.detach().reshape(1).float() if torch.is_tensor(loss_mapping) else None]
                sizes = [p.numel() for p in params] + [pad] + [p.numel() for p in kf_params] + [N, N, 2 * N * len(split), 1]
                n_par = sum(p.numel() for p in params)
                red = reducer.sum_floats(tensors, sizes, dev, group, scatter=sharder, scatter_len=(n_par + pad) if sharder is not None else 0)
                grad_share = reducer.share
                for p, g in zip(params, red):
                    p.grad = None if sharder is not None else g.view_as(p)   # (sharded: no rank holds the whole reduced gradient)
                for p, g in zip(kf_params, red[len(params) + 1:]):
                    p.grad = g.view_as(p)
                norm_sum, vis_count, split_xy = red[-4], red[-3], red[-2]
                if stats is not None:
                    stats.setdefault("losses", []).append(red[-1].clone())
            elif stats is not None and torch.is_tensor(loss_mapping):
                stats.setdefault("losses", []).append(loss_mapping.detach().reshape(1).float())
            if world > 1:
                for w in small_works:
                    w.wait()
                radii_max = radii_red
            if marks: marks.mark("collectives")
            if stats is not None:
                stats.setdefault("iterations", []).append(dict(views=[v for v, _, _ in mine], pieces=list(mine), phases=marks))

            # ---- bookkeeping of reference :309-389 on the reduced values ----
            backend.occ_aware_visibility = {}
            touched = flags[:n_window].long()   # one conversion for the window; the dict holds its rows
            for idx in range(n_window):
                backend.occ_aware_visibility[current_window[idx]] = touched[idx]

            # Only prune on the last iteration and when we have full window (:318-348).  The reference returns from here
            # without an optimizer step and without clearing the gradients, so (unless pruning replaces the parameters)
            # they are still there when the next call's backward accumulates.  To keep that sum right across ranks
            # the gradients of this pass stay LOCAL (unreduced): the next pass reduces old + new together.
            if prune:
                if sharder is not None:
                    sharder.sync_moments(G.optimizer, G.parameters())   # pruning edits the moments per Gaussian
                if n_window == cfg["Training"]["window_size"]:
                    prune_mode = cfg["Training"]["prune_mode"]
                    prune_coviz = cfg["Training"]["prune_num"]
                    # (`n_obs += visibility.cpu()` per window keyframe: the same integer sums, made on the device and fetched once --
                    # eight blocking copies fewer in a pass the free-running back end makes every ten iterations)
                    rows = list(backend.occ_aware_visibility.values())
                    summed = torch.stack(rows).sum(0) if all(r.is_cuda for r in rows) else sum(r.cpu() for r in rows)
                    G.n_obs.copy_(summed.cpu().to(G.n_obs.dtype))
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
                        # (the rows that stay, as indices found once: a boolean mask per keyframe is a device count the host waits for each time)
                        keep_idx = torch.nonzero(~to_prune.to(dev)).squeeze(1)
                        for idx in range(n_window):
                            k = current_window[idx]
                            backend.occ_aware_visibility[k] = backend.occ_aware_visibility[k].index_select(0, keep_idx.to(backend.occ_aware_visibility[k].device))
                    if not backend.initialized:
                        backend.initialized = True
                return False

            vs.apply(G, radii_max, norm_sum, vis_count, split_xy)

            update_gaussian = backend.iteration_count % backend.gaussian_update_every == backend.gaussian_update_offset
            gaussian_split = False
            resetting = (backend.iteration_count % backend.gaussian_reset) == 0 and (not update_gaussian)
            if sharder is not None and (update_gaussian or resetting):
                sharder.sync_moments(G.optimizer, G.parameters())   # densification / the reset edit the moments per Gaussian
            if update_gaussian:
                G.densify_and_prune(backend.opt_params.densify_grad_threshold, backend.gaussian_th, backend.gaussian_extent,
                                    backend.size_threshold)
                gaussian_split = True
            if (backend.iteration_count % backend.gaussian_reset) == 0 and (not update_gaussian):
                # seen by any view of the iteration: every view is counted by exactly one rank, a sum that reduces correctly
                G.reset_opacity_nonvisible([vis_count > 0])
                gaussian_split = True

            if stats is not None and callable(stats.get("before_steps")):
                stats["before_steps"](backend)   # tests look at the (reduced) gradients here
            if marks: marks.mark("bookkeeping")
            if sharder is not None and grad_share is not None:
                # A parameter tensor the bookkeeping has just replaced carries no gradient, and torch's Adam passes over it
                # (all of them after a densification -- the reference's iteration then steps nothing --, the opacities after the
                # reset of the non-visible): the same here.
                now = G.parameters()
                replaced = {k for k, (a, b) in enumerate(zip(params, now)) if a is not b}
                if len(now) == len(params) and all(a.numel() == b.numel() for a, b in zip(params, now)):
                    sharder.step(G.optimizer, now, grad_share, skip=replaced)
            else:
                G.optimizer.step()
            G.optimizer.zero_grad(set_to_none=True)
            G.update_learning_rate(backend.iteration_count)
            stepper = None
            if backend.keyframe_optimizers is not None:
                stepper = _keyframe_stepper(backend, viewpoint_stack, frames_to_optimize) if (fused and up_pose) else None
                if stepper is not None:
                    stepper.step()       # Adam + update_pose of every window keyframe, one launch each
                else:
                    stale = getattr(backend.keyframe_optimizers, "_lvdgs_stepper", None)
                    if stale is not None:   # earlier iterations went through the stepper: hand its moments to torch's optimiser
                        stale.export_to_optimizer()
                        backend.keyframe_optimizers._lvdgs_stepper = None
                    backend.keyframe_optimizers.step()
                backend.keyframe_optimizers.zero_grad(set_to_none=True)
            for v in views:   # exposure gradients of the random views are never stepped; do not let them pile up
                for name in _POSE_FIELDS:
                    p = getattr(v, name, None)
                    if p is not None and p.grad is not None and not any(p is q for q in kf_params):
                        p.grad = None
            # Pose update (:383-389): every rank holds the same reduced gradients, so every rank moves every keyframe
            if up_pose and stepper is None:
                for cam_idx in range(min(frames_to_optimize, n_window)):
                    viewpoint = viewpoint_stack[cam_idx]
                    if viewpoint.uid == 0:
                        continue
                    update_pose(viewpoint)
            if marks: marks.mark("optimizer_steps")
    # The moments leave this function whole on every rank: between calls the reference extends the map (extend_from_pcd_seq on
    # every keyframe, utils/slam_backend.py:75-78 -- the share boundaries move) and steps the optimiser outside this loop
    # (initialize_map, color_refinement): an all-gather of 2 x 14 N floats per call of `iters` iterations.
    if sharder is not None:
        sharder.sync_moments(G.optimizer, G.parameters())
    return gaussian_split


def _view_pass(backend):
    """The back end's MapViewPass (its buffers live as long as the back end); None off the GPU."""
    dev = backend.gaussians.get_xyz.device
    if dev.type != "cuda":
        return None
    vp = getattr(backend, "_lvdgs_view_pass", None)
    if vp is None or vp.dev != dev:
        vp = MapViewPass(dev)
        try:
            backend._lvdgs_view_pass = vp
        except Exception:
            pass
    return vp


def _keyframe_stepper(backend, viewpoint_stack, pose_window):
    """The KeyframeStepper of the back end's current keyframe optimiser (rebuilt when the optimiser or its groups change);
    None when that optimiser is not a plain torch.optim.Adam over GPU float32 parameters of the window's viewpoints."""
    opt = backend.keyframe_optimizers
    st = getattr(opt, "_lvdgs_stepper", None)
    if st is not None and st.signature == KeyframeStepper._sig(opt):
        return st
    if not KeyframeStepper.usable(opt, viewpoint_stack):
        return None
    try:
        st = KeyframeStepper(opt, viewpoint_stack, pose_window)
    except NotImplementedError:
        return None
    opt._lvdgs_stepper = st
    return st


class _PhaseMarks:
    """Named points on the device's stream (events on the GPU, wall clock off it); ``seconds()`` waits once, at the end, and
    returns {phase: seconds since the previous mark}.  No device-wide synchronisation around the collectives."""

    def __init__(self, dev):
        self.dev, self.names, self.points = dev, [], []
        self._point()

    def _point(self):
        if self.dev.type == "cuda":
            e = torch.cuda.Event(enable_timing=True)
            e.record(torch.cuda.current_stream(self.dev))
            self.points.append(e)
        else:
            import time
            self.points.append(time.perf_counter())

    def mark(self, name):
        self.names.append(name)
        self._point()

    def seconds(self):
        if self.dev.type == "cuda":
            self.points[-1].synchronize()
            return {n: a.elapsed_time(b) * 1e-3 for n, a, b in zip(self.names, self.points, self.points[1:])}
        return {n: b - a for n, a, b in zip(self.names, self.points, self.points[1:])}
