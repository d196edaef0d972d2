"""Per-frame pose tracking without the autograd engine: ``TrackingSession``.

One iteration of the reference's tracking loop (utils/slam_frontend.py:1492-1533) is

    render -> get_loss_tracking -> backward -> pose_optimizer.step() -> update_pose -> (next render's camera matrices)

Driven through PyTorch that is two ``autograd.Function`` round trips, the autograd engine's thread hand-over, a
``torch.optim.Adam`` step over four tiny tensors and about 60 small launches for the SE(3) retraction and the camera
matrices, with three host synchronisations (``if angle < 1e-5`` twice, ``if converged``): 0.5 ms of host time per
iteration whatever the scene, which is what bounds tracking at SLAM-sized maps (1e5 Gaussians, KITTI frames).

The session does the same arithmetic as three calls into the C ABI on pre-filled argument blocks and buffers that
live for the frame:

    lvdgs_forward -> lvdgs_backward_fused_loss -> lvdgs_tracking_tail

(the backward evaluates the tracking loss's image gradients itself, pixel by pixel, instead of reading gradient images a
separate pass would have written; the tail finishes the loss value and the exposure gradients, reduces the pose gradient
and applies the pose step in one launch: the arithmetic of lvdgs_photometric_loss_value_and_grad / lvdgs_backward /
lvdgs_pose_step, three launches and ~80 MB of traffic per 1080p iteration fewer)

and nothing comes back to the host inside the loop except the pair count ``lvdgs_forward`` has always read (the GPU has
the rest of the iteration queued behind it).  Convergence (``||tau|| < 1e-4``, utils/pose_utils.py:82) is a sticky
flag on the device: the host polls it a couple of iterations late, and iterations enqueued after it was raised leave the
pose untouched (``lvdgs_pose_step``), so the final pose, exposure and iteration count are those of the loop that breaks
at the converged iteration.  (The images left in the session after such a late stop are a re-render at the converged
pose, not the render that preceded the last step; they differ by a pose change below the convergence threshold.)

Gradients w.r.t. the Gaussians: the reference's tracking optimiser holds the pose and the exposure alone
(utils/slam_frontend.py:1468-1490, stepped at :1520); autograd computes the Gaussians' gradients beside them and drops them.
The session does not compute them (``LVDGS_FLAG_POSE_ONLY``: the backward blend pass without the colour / opacity sums,
the per-Gaussian pass without parameter reads it does not need or any N x 14 gradient stores; dL/dtau and the exposure
gradients are bit for bit those of the full backward) unless asked to with ``gaussian_gradients=True`` -- they then land in
scratch the session owns (``d_m3`` ...), the model's ``.grad`` fields are not touched -- or unless the colours depend on
the view direction (active SH degree > 0: the colour gradient then feeds the pose gradient).
"""
import ctypes as C

import torch

from . import _lib
from . import rasterizer as _rz
from .gaussian_renderer import _raw_parameters
from .slam_utils import _gt_image, _mono_depth, get_median_depth

_P = lambda t: None if t is None else C.c_void_p(t.data_ptr())


def _f32c(t, dev):
    t = t.detach()
    if t.dtype is not torch.float32 or t.device != dev or not t.is_contiguous():
        t = t.to(device=dev, dtype=torch.float32).contiguous()
    return t


class TrackingSession:
    """State of one frame's pose optimisation.  ``step()`` enqueues one iteration; ``finish()`` writes the result back
    into the viewpoint (``R``, ``T`` via ``update_RT``; exposure and the zeroed deltas were updated in place)."""

    def __init__(self, viewpoint, gaussians, config, pipeline_params, background, converged_threshold=1e-4, gaussian_gradients=False):
        import math
        dev = gaussians.get_xyz.device
        if dev.type != "cuda":
            raise _lib.LvdgsError("TrackingSession needs the map on the GPU (there is no CPU path)")
        if getattr(pipeline_params, "compute_cov3D_python", False) or getattr(pipeline_params, "convert_SHs_python", False):
            raise NotImplementedError("TrackingSession: pipeline_params with the *_python switches take the autograd path")
        self.L = _lib.lib()
        self.dev, self.vp, self.cfg = dev, viewpoint, config
        T = config["Training"]
        H, W = int(viewpoint.image_height), int(viewpoint.image_width)
        self.H, self.W = H, W
        N = int(gaussians.get_xyz.shape[0])
        self.N = N
        e = lambda *s, dt=torch.float32: torch.empty(*s, dtype=dt, device=dev)
        keep = self._keep = []

        # ---- camera state on the device: the session's own copies, advanced by lvdgs_pose_step ----
        self.R = _f32c(viewpoint.R, dev).clone()
        self.T = _f32c(viewpoint.T, dev).clone()
        self.proj_raw = _f32c(viewpoint.projection_matrix, dev)
        self.view = _f32c(viewpoint.world_view_transform, dev).clone()
        self.proj = _f32c(viewpoint.full_proj_transform, dev).clone()
        self.campos = _f32c(viewpoint.camera_center, dev).clone()
        for name in ("cam_rot_delta", "cam_trans_delta", "exposure_a", "exposure_b"):
            p = getattr(viewpoint, name)
            if p.device != dev or p.dtype is not torch.float32 or not p.is_contiguous():
                raise ValueError(f"TrackingSession: viewpoint.{name} must be a contiguous float32 tensor on {dev}")
        self.bg = _f32c(background, dev)

        # ---- forward / backward argument block (include/lvdgs.h: lvdgs_args) ----
        a = self.a = _lib.Args()
        a.image_height, a.image_width = H, W
        a.tanfovx, a.tanfovy = math.tan(viewpoint.FoVx * 0.5), math.tan(viewpoint.FoVy * 0.5)
        a.scale_modifier, a.sh_degree, a.prefiltered, a.debug = 1.0, int(gaussians.active_sh_degree), 0, 0
        a.bg, a.viewmatrix, a.projmatrix, a.projmatrix_raw, a.campos = _P(self.bg), _P(self.view), _P(self.proj), _P(self.proj_raw), _P(self.campos)
        raw = _raw_parameters(gaussians)
        if raw is not None:
            scales, rotations, opacity = raw
            a.activations = _rz.ACT_EXP_SCALES | _rz.ACT_NORMALIZE_ROTATIONS | _rz.ACT_SIGMOID_OPACITIES
        else:
            scales, rotations, opacity = gaussians.get_scaling, gaussians.get_rotation, gaussians.get_opacity
            a.activations = 0
        a.flags = _lib.FLAG_LIST_ALL_TILES if _rz.LIST_ALL_TILES else 0
        self.pose_only = not gaussian_gradients and int(gaussians.active_sh_degree) == 0
        if self.pose_only:
            a.flags |= _lib.FLAG_POSE_ONLY
        m3, sc, rot, op, shs = (_f32c(t, dev) for t in (gaussians.get_xyz, scales, rotations, opacity, gaussians.get_features))
        keep += [m3, sc, rot, op, shs]
        a.num_gaussians, a.sh_coeffs = N, int(shs.shape[1])
        a.means3D, a.opacities, a.scales, a.rotations, a.shs = _P(m3), _P(op), _P(sc), _P(rot), _P(shs)
        self.color, self.depth, self.opacity = e(3, H, W), e(1, H, W), e(1, H, W)
        self.radii, self.n_touched = e(N, dt=torch.int32), e(N, dt=torch.int32)
        a.radii, a.n_touched = _P(self.radii), _P(self.n_touched)
        a.out_color, a.out_depth, a.out_opacity = _P(self.color), _P(self.depth), _P(self.opacity)
        L = self.L
        bytes_ = lambda n: torch.empty(max(int(n), 256), dtype=torch.uint8, device=dev)
        self.geom, self.image = bytes_(L.lvdgs_geom_bytes(N)), bytes_(L.lvdgs_image_bytes(W, H))
        a.geom_state, a.geom_bytes = _P(self.geom), self.geom.numel()
        a.image_state, a.image_bytes = _P(self.image), self.image.numel()
        key = dev.index if dev.index is not None else torch.cuda.current_device()
        self.cap = max(_rz._PAIR_CAPACITY.get(key, 0), _rz._MIN_PAIR_CAPACITY, _rz._PAIRS_PER_GAUSSIAN_GUESS * N, 1)
        self._size_for_pairs(self.cap)
        # gradient outputs of lvdgs_backward (only dL/dtau is consumed; the Gaussians' go to scratch when they are asked for)
        self.d_tau = e(6)
        self.d_m3 = self.d_m2 = self.d_op = self.d_sc = self.d_rot = self.d_sh = None
        if not self.pose_only:
            self.d_m3, self.d_m2, self.d_op = e(N, 3), e(N, 3), e(*op.shape)
            self.d_sc, self.d_rot, self.d_sh = e(N, 3), e(N, 4), e(*shs.shape)
        a.dL_dmeans3D, a.dL_dmeans2D, a.dL_dopacities = _P(self.d_m3), _P(self.d_m2), _P(self.d_op)
        a.dL_dscales, a.dL_drotations, a.dL_dshs = _P(self.d_sc), _P(self.d_rot), _P(self.d_sh)
        a.dL_dtau = None   # the backward leaves its partial sums, lvdgs_tracking_tail reduces them into self.d_tau

        # ---- tracking loss (reference utils/slam_utils.py:42-79; include/lvdgs.h: lvdgs_loss_args) ----
        la = self.la = _lib.LossArgs()
        la.width, la.height = W, H
        gt = _f32c(_gt_image(viewpoint, self.color), dev)
        gm = viewpoint.grad_mask
        gm = None if gm is None else gm.reshape(-1).to(dev)
        if gm is not None:
            gm = (gm.view(torch.uint8) if gm.dtype == torch.bool else gm.ne(0).view(torch.uint8)).contiguous()
        keep += [gt, gm]
        la.image, la.opacity, la.gt_image, la.grad_mask = _P(self.color), _P(self.opacity), _P(gt), _P(gm)
        la.exposure_a, la.exposure_b = _P(viewpoint.exposure_a), _P(viewpoint.exposure_b)
        la.rgb_boundary_threshold, la.weight_by_opacity = float(T["rgb_boundary_threshold"]), 1
        if T["monocular"]:   # RGB-only whether or not Dataset.depth_loss is set (utils/slam_utils.py:45-49)
            la.weight_rgb, la.weight_depth, la.depth_needs_opaque = 1.0, 0.0, 0
        else:
            alpha = T.get("alpha", 0.95)
            md = _f32c(_mono_depth(viewpoint, self.color), dev)
            keep.append(md)
            la.depth, la.gt_depth = _P(self.depth), _P(md)
            la.weight_rgb, la.weight_depth, la.depth_needs_opaque = float(alpha), float(1 - alpha), 1
        self.loss_scratch = bytes_(L.lvdgs_loss_scratch_bytes(W, H))
        self.loss, self.one = e(()), torch.ones((), dtype=torch.float32, device=dev)
        self.d_a, self.d_b = e(1), e(1)
        la.scratch, la.scratch_bytes, la.loss, la.grad_loss = _P(self.loss_scratch), self.loss_scratch.numel(), _P(self.loss), _P(self.one)
        la.d_image = la.d_depth = la.d_opacity = None   # never materialised: the backward computes them per pixel
        la.d_exposure_a, la.d_exposure_b = _P(self.d_a), _P(self.d_b)
        a.dL_dout_color = a.dL_dout_depth = a.dL_dout_opacity = None

        # ---- pose step (include/lvdgs.h: lvdgs_pose_step_args): torch.optim.Adam defaults, the front end's learning rates ----
        pa = self.pa = _lib.PoseStepArgs()
        self.pose_state = torch.zeros(24, dtype=torch.float32, device=dev)
        pa.R, pa.T, pa.cam_rot_delta, pa.cam_trans_delta = _P(self.R), _P(self.T), _P(viewpoint.cam_rot_delta), _P(viewpoint.cam_trans_delta)
        pa.exposure_a, pa.exposure_b = _P(viewpoint.exposure_a), _P(viewpoint.exposure_b)
        pa.grad_tau, pa.grad_exposure_a, pa.grad_exposure_b, pa.state = _P(self.d_tau), _P(self.d_a), _P(self.d_b), _P(self.pose_state)
        pa.lr_rot, pa.lr_trans, pa.lr_exposure = float(T["lr"]["cam_rot_delta"]), float(T["lr"]["cam_trans_delta"]), 0.01
        pa.beta1, pa.beta2, pa.eps, pa.converged_threshold = 0.9, 0.999, 1e-8, float(converged_threshold)
        pa.projmatrix_raw, pa.viewmatrix, pa.projmatrix, pa.campos = _P(self.proj_raw), _P(self.view), _P(self.proj), _P(self.campos)

        # The converged flag and the count of applied steps, where the host can see them without asking: two pinned words the
        # tail launch stores to (lvdgs_pose_step_args.host_flags).  Where pinned memory is not mapped into the device's
        # address space: a ring of pinned words + events for late, non-blocking copies of the flag (a copy per iteration is a
        # blit kernel per iteration).
        self.host_flags = None
        try:
            flags = torch.zeros(2, dtype=torch.float32).pin_memory()
            dptr = C.c_void_p()
            if self.L.lvdgs_host_device_pointer(C.c_void_p(flags.data_ptr()), C.byref(dptr)) == _lib.OK and dptr.value:
                self.host_flags, pa.host_flags = flags, dptr
        except RuntimeError:
            pass
        self._ring = None if self.host_flags is not None else [(torch.cuda.Event(), torch.zeros(2, dtype=torch.float32).pin_memory()) for _ in range(8)]
        self._asked, self._answered = 0, 0
        self.iterations_enqueued = 0
        self._streams_used = set()
        self.num_rendered = 0

    def _size_for_pairs(self, pairs):
        L, a, dev = self.L, self.a, self.dev
        bytes_ = lambda n: torch.empty(max(int(n), 256), dtype=torch.uint8, device=dev)
        self.binning = bytes_(L.lvdgs_binning_bytes(pairs))
        self.scratch = bytes_(max(L.lvdgs_prepare_scratch_bytes(self.N), L.lvdgs_render_scratch_bytes(self.N, pairs, self.W, self.H),
                                  L.lvdgs_backward_scratch_bytes(self.N, pairs)))
        a.pair_capacity = pairs
        a.binning_state, a.binning_bytes = _P(self.binning), self.binning.numel()
        a.scratch, a.scratch_bytes = _P(self.scratch), self.scratch.numel()

    def step(self, record_loss=None):
        """Enqueue one tracking iteration.  ``record_loss``: a 0-dim device tensor to receive a copy of the loss."""
        L, a = self.L, self.a
        with _lib.on_device(self.dev):
            stream = _lib.raw_stream(self.dev)
            num = C.c_int64(0)
            # forward and backward as one call: the backward evaluates the loss's image gradients as it reads its pixels (the
            # objective is the loss: d/d loss = 1), and on small grids the two blend passes of a tile share a launch
            # (lvdgs_forward_backward_fused_loss); the loss's final reduction, the pose gradient's and the pose step share the
            # iteration's last launch
            # (the two-level grouping hint follows the previous iteration's pair count: rasterizer.super_tiles_flag)
            a.flags = (a.flags & ~_lib.FLAG_SUPER_TILES) | _rz.super_tiles_flag(self.N, self.num_rendered)
            status = L.lvdgs_forward_backward_fused_loss(C.byref(a), C.byref(self.la), int(_rz.PROPAGATE_OPACITY_GRAD), C.byref(num), stream)
            D = int(num.value)
            if status == _lib.E_CAPACITY:   # more pairs than the buffers hold: grow them and redo binning + blend, then the backward
                self._size_for_pairs(D + D // 2)
                a.num_rendered = D
                _lib.check(L.lvdgs_forward_render(C.byref(a), stream), "lvdgs_forward_render")
                _lib.check(L.lvdgs_backward_fused_loss(C.byref(a), C.byref(self.la), int(_rz.PROPAGATE_OPACITY_GRAD), stream), "lvdgs_backward_fused_loss")
            else:
                _lib.check(status, "lvdgs_forward_backward_fused_loss")
            self.num_rendered = a.num_rendered = D
            _lib.check(L.lvdgs_tracking_tail(C.byref(self.la), C.byref(a), C.byref(self.pa), _P(self.d_tau), 1, stream), "lvdgs_tracking_tail")
            if record_loss is not None:
                record_loss.copy_(self.loss)
        self.iterations_enqueued += 1
        self._drained = False
        self._streams_used.add(stream.value)   # (raw handles: close() waits for these streams only)

    def close(self):
        """Wait for everything the session has enqueued.  The tail launch of every step stores into ``host_flags`` -- pinned
        memory PyTorch's host allocator owns and knows nothing of that use: were the tensor released with steps still in
        flight (the design lets the host run ahead of the device), a late store would land in whoever got the block next.
        ``finish()`` synchronises already; this is for a session that is dropped without it (also called on deletion)."""
        if getattr(self, "iterations_enqueued", 0) and not getattr(self, "_drained", False):
            # only the streams the session's steps went to -- not the device: this runs at garbage-collection time on whichever
            # thread drops the session (the back end's mapping thread, a stream capture) and must not stall their streams
            for raw in list(getattr(self, "_streams_used", ())):
                (torch.cuda.ExternalStream(raw, device=self.dev) if raw else torch.cuda.default_stream(self.dev)).synchronize()
            self._drained = True

    def __del__(self):
        try:
            self.close()
        except Exception:   # (interpreter shutdown: nothing left to protect)
            pass

    def converged_lagging(self, lag=2):
        """True once the device's sticky flag has been SEEN set.  Non-blocking.  With ``host_flags`` (pinned words the tail
        launch itself stores to, the normal case) that is a read of host memory: the flag is seen as soon as the step that
        raised it has run, ``lag`` plays no part.  Without them: a copy of the flag is requested at every call, and
        requests made at least ``lag`` calls ago are read if their copy has landed (the oldest one is waited for only when
        the ring of 8 outstanding requests is full)."""
        if self.host_flags is not None:
            return float(self.host_flags[0]) != 0.0
        if self._asked - self._answered == len(self._ring):
            self._ring[self._answered % len(self._ring)][0].synchronize()
        ev, buf = self._ring[self._asked % len(self._ring)]
        buf.copy_(self.pose_state[17:19], non_blocking=True)
        ev.record()
        self._asked += 1
        while self._asked - self._answered > lag:
            ev0, buf0 = self._ring[self._answered % len(self._ring)]
            if not ev0.query():
                break
            self._answered += 1
            if float(buf0[0]) != 0.0:
                return True
        return False

    def converged(self):
        """Has the device's sticky converged flag been raised?  (After ``finish()``: a plain read.)"""
        if self.host_flags is not None:
            return float(self.host_flags[0]) != 0.0
        return float(self.pose_state[17].item()) != 0.0

    def finish(self):
        """Synchronise, write the pose back into the viewpoint, return the number of iterations the reference's loop
        would have run (it breaks at the first converged one)."""
        torch.cuda.synchronize(self.dev)
        self._drained = True
        applied = int(self.host_flags[1]) if self.host_flags is not None else int(self.pose_state[18].item())
        self.vp.update_RT(self.R.clone(), self.T.clone())
        return applied

    def render_package(self):
        """The seven-key dict of the last enqueued iteration's render (views of the session's buffers)."""
        return {"render": self.color, "viewspace_points": None, "visibility_filter": self.radii > 0, "radii": self.radii,
                "depth": self.depth, "opacity": self.opacity, "n_touched": self.n_touched}


def track_frame_fused(viewpoint, gaussians, config, pipeline_params, background, tracking_itr_num=None, on_iteration=None,
                      poll_lag=2):
    """``slam_loops.track_frame`` on a ``TrackingSession``: same arguments, same results (render package of the last
    iteration, median depth, iterations run), no autograd, no host synchronisation inside the loop."""
    n_iter = config["Training"]["tracking_itr_num"] if tracking_itr_num is None else tracking_itr_num
    sess = TrackingSession(viewpoint, gaussians, config, pipeline_params, background)
    losses = torch.zeros(max(n_iter, 1), dtype=torch.float32, device=sess.dev) if on_iteration is not None else None
    with _lib.quiet_gc():   # (no full-heap garbage collection inside a 0.2 ms loop; the host's collector is as before afterwards)
        for it in range(n_iter):
            sess.step(None if losses is None else losses[it])
            if sess.converged_lagging(poll_lag):
                break
        applied = sess.finish()
        # The images a converged frame leaves must not depend on how far the host happened to be ahead of the device: when the flag
        # was seen before anything else was enqueued behind the converging iteration, one more iteration is (its pose step is a no-op
        # once the flag stands) -- the package is then ALWAYS the render at the converged pose.  Replicas of a tracker on several
        # ranks take their keyframe decisions from it and must agree.
        if applied < n_iter and sess.iterations_enqueued == applied and sess.converged():
            sess.step()
            sess.finish()
    if on_iteration is not None:
        for it, v in enumerate(losses[:applied].cpu()):
            on_iteration(it, v, None)
    pkg = sess.render_package()
    return pkg, get_median_depth(pkg["depth"], pkg["opacity"]), applied
