"""One mapping view without the autograd engine: ``MapViewPass``.

Per window keyframe the reference's mapping iteration (utils/slam_backend.py:175-266) does

    render(viewpoint, gaussians, ...)  ->  get_loss_mapping(...)  ->  (summed over the views)  loss.backward()

Through PyTorch that is two ``autograd.Function`` round trips per view, the concatenation of the SH features, the
engine's thread hand-over and a dozen accumulation kernels: ~0.3 ms of host time per view, which is what bounds a
mapping iteration once the kernels are fast (one view per GPU in the sharded window, ten on a single GPU).

The pass does the same arithmetic as three calls into the C ABI on argument blocks and buffers that live as long as the
back end:

    lvdgs_forward -> lvdgs_backward_fused_loss -> lvdgs_tracking_tail (reductions only)

(the backward evaluates the mapping loss's image gradients itself as it reads its pixels)

with the activations fused into the rasterizer (``lvdgs_args.activations``), so ``lvdgs_backward`` writes the gradients
w.r.t. the model's RAW parameters: they go straight into the ``.grad`` fields autograd would have filled (the first view
of an iteration writes them, later views are added with one ``_foreach_add_``), the pose / exposure gradients into the
viewpoint's ``cam_rot_delta.grad`` ... ``exposure_b.grad``.  Models with non-standard activations keep the autograd path
(``usable``).

A keyframe that carries a ``static_mask`` -- every window keyframe of LVD-GS under its default configuration
(utils/slam_frontend.py:1218,1309-1329,1429-1433) -- is scored by L1 + SSIM on the static pixels plus the count-normalised masked
depth term (utils/slam_backend.py:196-261) instead of ``get_loss_mapping``.  Same three-call shape (``masked_loss``):

    lvdgs_forward -> lvdgs_masked_loss_batch (L1 + SSIM value and gradient image, the depth term's sum and count: two launches)
                  -> lvdgs_backward_masked_loss (the depth term's gradient evaluated per pixel in the blend pass) -> lvdgs_map_view_tail

on the viewpoint's cached mask bytes and mono depth (``slam_utils._static_mask_bytes`` / ``_mono_depth``: no upload per iteration)
and scratch the pass owns.
"""
import ctypes as C
import math
import os
from types import SimpleNamespace

import torch

from . import _lib
from . import rasterizer as _rz
from .gaussian_renderer import _raw_parameters
from .slam_utils import _gt_image, _mono_depth, _static_mask_bytes

_P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
_PARAM_FIELDS = ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity")
_VIEW_FIELDS = ("cam_rot_delta", "cam_trans_delta", "exposure_a", "exposure_b")


def _gpu_f32c(t, dev):
    return torch.is_tensor(t) and t.device == dev and t.dtype is torch.float32 and t.is_contiguous()


class MapViewPass:
    """Buffers and argument blocks for render + ``get_loss_mapping`` + backward of one view; re-pointed at every call
    (the model's tensors are replaced by densification, the viewpoint changes from call to call)."""

    def __init__(self, device, own_gradient_buffers=True):
        if device.type != "cuda":
            raise _lib.LvdgsError("MapViewPass needs the map on the GPU (there is no CPU path)")
        self.dev = device
        self.own_gradient_buffers = own_gradient_buffers   # (False: a later view of a MapWindowBatch -- it adds to the first view's)
        self.L = _lib.lib()
        self.a, self.la, self.ml = _lib.Args(), _lib.LossArgs(), _lib.MaskedLossArgs()
        self.N = self.W = self.H = -1
        self.cap = 0
        self.one = torch.ones((), dtype=torch.float32, device=device)
        self._keep = []

    # ---- eligibility -------------------------------------------------------------------------------------------------
    @staticmethod
    def masked_loss_usable(viewpoint, with_depth=True) -> bool:
        """Can the static-mask loss of ``viewpoint`` take the fused route?  Mask, target image and mono depth of the image's size
        (the reference crops all three to their common top-left window when they differ, utils/slam_backend.py:240-246: the
        autograd path does that)."""
        H, W = int(viewpoint.image_height), int(viewpoint.image_width)
        m = getattr(viewpoint, "static_mask", None)
        if m is not None and (not torch.is_tensor(m) or m.numel() != H * W):
            return False
        gt = getattr(viewpoint, "original_image", None)
        if not torch.is_tensor(gt) or tuple(gt.shape) != (3, H, W):
            return False
        md = getattr(viewpoint, "mono_depth", None) if with_depth else None
        return md is None or int(md.size if not torch.is_tensor(md) else md.numel()) == H * W

    @staticmethod
    def usable(backend, viewpoint, allow_static_mask=False) -> bool:
        G = backend.gaussians
        dev = G.get_xyz.device
        pp = backend.pipeline_params
        if dev.type != "cuda" or getattr(pp, "compute_cov3D_python", False) or getattr(pp, "convert_SHs_python", False):
            return False
        if G.get_xyz.shape[0] == 0:   # an empty map renders the background; the autograd path returns None for it
            return False
        if (not allow_static_mask and getattr(viewpoint, "static_mask", None) is not None) or _raw_parameters(G) is None:
            return False
        if not all(_gpu_f32c(getattr(G, n, None), dev) and getattr(G, n).requires_grad for n in _PARAM_FIELDS):
            return False
        if G._features_dc.dim() != 3 or G._features_dc.shape[1] != 1 or G._opacity.numel() != G._xyz.shape[0]:
            return False
        return all(_gpu_f32c(getattr(viewpoint, n, None), dev) for n in _VIEW_FIELDS)

    # ---- buffers -----------------------------------------------------------------------------------------------------
    def _bytes(self, n):
        return torch.empty(max(int(n), 256), dtype=torch.uint8, device=self.dev)

    def _size_for_model(self, N, K):
        e = lambda *s, dt=torch.float32: torch.empty(*s, dtype=dt, device=self.dev)
        self.N, self.K = N, K
        self.geom = self._bytes(self.L.lvdgs_geom_bytes(N))
        # two sets of gradient buffers: `first` becomes the parameters' .grad, `more` receives the later views
        mk = lambda: dict(_xyz=e(N, 3), _features_dc=e(N, 1, 3), _features_rest=e(N, K - 1, 3), _scaling=e(N, 3),
                          _rotation=e(N, 4), _opacity=e(N, 1), sh=(e(N, K, 3) if K > 1 else None))
        self.first, self.more = (mk(), mk()) if self.own_gradient_buffers else (None, None)
        self.shs = e(N, K, 3) if K > 1 else None
        a = self.a
        a.num_gaussians, a.sh_coeffs = N, K
        a.geom_state, a.geom_bytes = _P(self.geom), self.geom.numel()
        self._size_for_pairs(max(self.cap, _rz._MIN_PAIR_CAPACITY, _rz._PAIRS_PER_GAUSSIAN_GUESS * N, 1))

    def _size_for_image(self, W, H):
        e = lambda *s: torch.empty(*s, dtype=torch.float32, device=self.dev)
        self.W, self.H = W, H
        self.image = self._bytes(self.L.lvdgs_image_bytes(W, H))
        self.loss_scratch = self._bytes(self.L.lvdgs_loss_scratch_bytes(W, H))
        a, la = self.a, self.la
        a.image_height, a.image_width = H, W
        a.image_state, a.image_bytes = _P(self.image), self.image.numel()
        a.dL_dout_color = a.dL_dout_depth = a.dL_dout_opacity = None   # the backward evaluates the loss per pixel
        la.width, la.height = W, H
        la.scratch, la.scratch_bytes, la.grad_loss = _P(self.loss_scratch), self.loss_scratch.numel(), _P(self.one)
        la.d_image = la.d_depth = la.d_opacity = la.opacity = la.grad_mask = None
        la.weight_by_opacity = la.depth_needs_opaque = 0
        # the static-mask loss: its partial sums and the colour-gradient image it hands to the backward blend pass
        ml = self.ml
        self.masked_scratch = self._bytes(self.L.lvdgs_masked_loss_scratch_bytes(W, H))
        self.d_image = e(3, H, W)
        ml.width, ml.height = W, H
        ml.scratch, ml.scratch_bytes, ml.d_image = _P(self.masked_scratch), self.masked_scratch.numel(), _P(self.d_image)
        if self.N >= 0:
            self._size_for_pairs(self.cap)

    def _size_for_pairs(self, pairs):
        L, a = self.L, self.a
        self.cap = int(pairs)
        self.binning = self._bytes(L.lvdgs_binning_bytes(self.cap))
        need = max(L.lvdgs_prepare_scratch_bytes(self.N), L.lvdgs_backward_scratch_bytes(self.N, self.cap),
                   L.lvdgs_render_scratch_bytes(self.N, self.cap, self.W, self.H) if self.W > 0 else 0)
        self.scratch = self._bytes(need)
        a.pair_capacity = self.cap
        a.binning_state, a.binning_bytes = _P(self.binning), self.binning.numel()
        a.scratch, a.scratch_bytes = _P(self.scratch), self.scratch.numel()

    def _point_masked_loss(self, backend, viewpoint, masked_loss, color, depth, bg, keep):
        """Fills ``self.ml`` for ``viewpoint`` (``masked_loss = (lambda_dssim, depth_lambda or None)``; None: no depth term -- colour
        refinement); returns the view's 8-float result block (``[0]`` = the loss)."""
        dev, ml = self.dev, self.ml
        lam, dlam = masked_loss
        gt = _gt_image(viewpoint, color)
        gt = gt.detach() if _gpu_f32c(gt, dev) else gt.detach().to(device=dev, dtype=torch.float32).contiguous()
        keep.append(gt)
        mask = _static_mask_bytes(viewpoint, color) if getattr(viewpoint, "static_mask", None) is not None else None
        keep.append(mask)
        ml.image, ml.gt_image, ml.static_mask, ml.bg = _P(color), _P(gt), _P(mask), _P(bg)
        ml.lambda_dssim = float(lam)
        md = getattr(viewpoint, "mono_depth", None) if dlam is not None else None
        if md is not None:
            z = _mono_depth(viewpoint, color)
            z = z if _gpu_f32c(z, dev) else z.to(device=dev, dtype=torch.float32).contiguous()
            keep.append(z)
            ml.depth, ml.gt_depth, ml.depth_lambda = _P(depth), _P(z), float(dlam)
        else:
            ml.depth, ml.gt_depth, ml.depth_lambda = None, None, 0.0
        out = torch.empty(8, dtype=torch.float32, device=dev)
        ml.out = _P(out)
        return out

    # ---- one view ----------------------------------------------------------------------------------------------------
    def run(self, backend, viewpoint, initialization=False, first=None, image_loss=None, band=None, stats=None, masked_loss=None, want_visibility=True):
        """Render ``viewpoint``, evaluate ``get_loss_mapping`` and add its gradients to the model's and the viewpoint's
        ``.grad`` fields (``first``: buffers, by parameter field, for a view that finds no gradients yet to write into --
        the sharded loop passes slices of its all-reduce bucket -- instead of the pass's own).  Returns the render package (fresh tensors, ``viewspace_points`` carries ``.grad``) and the
        loss (0-dim tensor).

        ``image_loss(color, depth) -> (loss, d_color[, d_depth])`` (optional) replaces ``get_loss_mapping``: a loss that
        brings its own gradient images -- colour refinement's ``(1 - l) L1 + l (1 - SSIM)`` from the fused L1 + SSIM
        kernel, the static-mask mapping loss with its masked depth term -- after which the plain ``lvdgs_backward`` runs
        (no exposure terms: those losses do not use the exposure parameters).

        ``band = (row0, row1)``: only tile rows [row0, row1) of the view (``lvdgs_args.tile_row_begin / _end``): the band's
        pixels of the images, the band's share of the loss and of every gradient; ``radii`` are the whole view's,
        ``n_touched`` counts the band's pixels.  The other pixels of the returned images are not written.

        ``masked_loss = (lambda_dssim, depth_lambda or None)``: the view is scored by the static-mask mapping loss (module
        docstring; reference utils/slam_backend.py:196-261, and :420-454 -- colour refinement -- with ``depth_lambda`` None) on its
        ``static_mask`` (every pixel when the viewpoint has none); no exposure gradients, whole views only.

        ``stats = (radii_max, norm_sum, vis_count or None, touched_row or None, split_xy or None)``: the view's statistics
        (``lvdgs_view_stats``) are taken in the launch that finishes its loss (``lvdgs_map_view_tail``) instead of one of
        their own; built-in mapping loss and ``masked_loss``."""
        G, cfg, dev, L = backend.gaussians, backend.config, self.dev, self.L
        T = cfg["Training"]
        N, K = int(G._xyz.shape[0]), 1 + int(G._features_rest.shape[1])
        H, W = int(viewpoint.image_height), int(viewpoint.image_width)
        if N != self.N or K != getattr(self, "K", -1):
            self._size_for_model(N, K)
        if (W, H) != (self.W, self.H):
            self._size_for_image(W, H)
        a, la = self.a, self.la
        e = lambda *s, dt=torch.float32: torch.empty(*s, dtype=dt, device=dev)
        keep = self._keep = []
        f32c = lambda t: t.detach() if _gpu_f32c(t, dev) else t.detach().to(device=dev, dtype=torch.float32).contiguous()

        # ---- inputs ----
        a.tanfovx, a.tanfovy = math.tan(viewpoint.FoVx * 0.5), math.tan(viewpoint.FoVy * 0.5)
        a.scale_modifier, a.sh_degree, a.prefiltered, a.debug = 1.0, int(G.active_sh_degree), 0, 0
        cam = [f32c(t) for t in (backend.background, viewpoint.world_view_transform, viewpoint.full_proj_transform,
                                 viewpoint.projection_matrix, viewpoint.camera_center)]
        keep += cam
        a.bg, a.viewmatrix, a.projmatrix, a.projmatrix_raw, a.campos = (_P(t) for t in cam)
        a.activations = _rz.ACT_EXP_SCALES | _rz.ACT_NORMALIZE_ROTATIONS | _rz.ACT_SIGMOID_OPACITIES
        a.flags = _lib.FLAG_LIST_ALL_TILES if _rz.LIST_ALL_TILES else 0
        # (two-level grouping: hinted by the pair count of the pass's previous view of a map of this size -- rasterizer.super_tiles_flag)
        a.flags |= _rz.super_tiles_flag(N, int(a.num_rendered) if getattr(self, "_pairs_of_n", None) == N else None)
        self._pairs_of_n = N
        a.tile_row_begin, a.tile_row_end = (0, 0) if band is None else (int(band[0]), int(band[1]))
        if band is not None and (image_loss is not None or masked_loss is not None or not 0 <= band[0] < band[1]):
            raise _lib.LvdgsError("MapViewPass: a band needs 0 <= row0 < row1 and the built-in mapping loss")
        if image_loss is not None and masked_loss is not None:
            raise _lib.LvdgsError("MapViewPass: image_loss and masked_loss exclude each other")
        if K > 1:
            torch.cat((G._features_dc.detach(), G._features_rest.detach()), dim=1, out=self.shs)
            shs = self.shs
        else:
            shs = G._features_dc.detach()
        a.means3D, a.opacities, a.scales, a.rotations, a.shs = _P(G._xyz), _P(G._opacity), _P(G._scaling), _P(G._rotation), _P(shs)
        color, depth, opacity = e(3, H, W), e(1, H, W), e(1, H, W)
        radii, n_touched = e(N, dt=torch.int32), e(N, dt=torch.int32)
        a.radii, a.n_touched, a.out_color, a.out_depth, a.out_opacity = _P(radii), _P(n_touched), _P(color), _P(depth), _P(opacity)

        # ---- where the gradients go ----
        # (an empty _features_rest -- SH degree 0 -- gets no gradient from autograd either)
        fields = [n for n in _PARAM_FIELDS if getattr(G, n).numel() > 0]
        has = [getattr(G, n).grad is not None for n in fields]
        if any(has) and not all(has):
            # A parameter without a gradient next to parameters with one: its gradient so far is zero.  (The sharded loop gets here: a pruning
            # pass leaves its gradients in place -- reference utils/slam_backend.py:318-348 returns before the step -- and on the rank that
            # holds the isotropic term but was dealt no view of that pass, the scales alone carry one.)
            for n in fields:
                if getattr(G, n).grad is None:
                    getattr(G, n).grad = torch.zeros_like(getattr(G, n))
            has = [True] * len(fields)
        if has[0] and any(not _gpu_f32c(getattr(G, n).grad, dev) for n in fields):
            raise _lib.LvdgsError("MapViewPass: the model's existing gradients are not contiguous float32 tensors on the GPU")
        # A later view of the iteration ADDS its parameter gradients to the ones that are there, inside the backward's last
        # kernel (LVDGS_FLAG_ACCUMULATE_PARAM_GRADS) -- instead of writing a second set that a multi-tensor add folds in
        # (three passes over N x 14 floats per view).  With SH coefficients beyond degree 0 the kernel's one colour gradient
        # maps onto two parameters: the second set and the add stay.
        accumulate = has[0] and K == 1
        into = ({n: getattr(G, n).grad for n in fields} if accumulate else
                (self.more if has[0] else (self.first if first is None or K > 1 else first)))
        a.flags = (a.flags & ~_lib.FLAG_ACCUMULATE_PARAM_GRADS) | (_lib.FLAG_ACCUMULATE_PARAM_GRADS if accumulate else 0)
        d_sh = into["sh"] if K > 1 else into["_features_dc"]
        a.dL_dmeans3D, a.dL_dopacities, a.dL_dscales = _P(into["_xyz"]), _P(into["_opacity"]), _P(into["_scaling"])
        a.dL_drotations, a.dL_dshs = _P(into["_rotation"]), _P(d_sh)
        d_tau, d_a, d_b, d_m2 = e(6), e(1), e(1), e(N, 3)
        a.dL_dtau, a.dL_dmeans2D = None, _P(d_m2)   # the pose gradient's partial sums are reduced together with the loss's

        loss = e(())
        if masked_loss is not None:
            mout = self._point_masked_loss(backend, viewpoint, masked_loss, color, depth, cam[0], keep)
            loss = mout[0]
        elif image_loss is None:
            # ---- get_loss_mapping (reference utils/slam_utils.py:82-121) ----
            # (monodepth=True at every call site of the mapping loop, so the loss is the rgb-d one whatever Training.monocular says)
            gt = f32c(_gt_image(viewpoint, color))
            keep.append(gt)
            la.image, la.gt_image = _P(color), _P(gt)
            la.rgb_boundary_threshold = float(T["rgb_boundary_threshold"])
            if initialization:
                la.exposure_a = la.exposure_b = la.d_exposure_a = la.d_exposure_b = None
            else:
                la.exposure_a, la.exposure_b, la.d_exposure_a, la.d_exposure_b = _P(viewpoint.exposure_a), _P(viewpoint.exposure_b), _P(d_a), _P(d_b)
            alpha = T.get("alpha", 0.95)
            md = f32c(_mono_depth(viewpoint, color))
            keep.append(md)
            la.depth, la.gt_depth = _P(depth), _P(md)
            la.weight_rgb, la.weight_depth = float(alpha), float(1 - alpha)
            la.loss = _P(loss)

        with _lib.on_device(dev):
            stream = _lib.raw_stream(dev)
            num = C.c_int64(0)
            # the built-in loss: forward and backward as one call (on small frames and bands the two blend passes of a tile share a
            # launch, lvdgs_forward_backward_fused_loss); a loss that needs the whole image first: the forward alone
            together = image_loss is None and masked_loss is None
            status = (L.lvdgs_forward_backward_fused_loss(C.byref(a), C.byref(la), 0, C.byref(num), stream) if together
                      else L.lvdgs_forward(C.byref(a), C.byref(num), stream))
            D = int(num.value)
            if status == _lib.E_CAPACITY:   # more pairs than the buffers hold: grow them and redo binning + blend
                self._size_for_pairs(D + D // 2)
                a.num_rendered = D
                _lib.check(L.lvdgs_forward_render(C.byref(a), stream), "lvdgs_forward_render")
                if together:
                    _lib.check(L.lvdgs_backward_fused_loss(C.byref(a), C.byref(la), 0, stream), "lvdgs_backward_fused_loss")
            else:
                _lib.check(status, "lvdgs_forward_backward_fused_loss" if together else "lvdgs_forward")
            a.num_rendered = D
            if masked_loss is not None:
                ml = self.ml
                views = (C.POINTER(_lib.MaskedLossArgs) * 1)(C.pointer(ml))
                _lib.check(L.lvdgs_masked_loss_batch(views, 1, stream), "lvdgs_masked_loss_batch")
                _lib.check(L.lvdgs_backward_masked_loss(C.byref(a), C.byref(ml), stream), "lvdgs_backward_masked_loss")
                if stats is not None:
                    sa = _lib.ViewStatsArgs()
                    sa.radii_max, sa.norm_sum, sa.vis_count, sa.touched_row, sa.split_xy = (_P(t) for t in stats)
                    _lib.check(L.lvdgs_map_view_tail(None, C.byref(a), _P(d_tau), C.byref(sa), stream), "lvdgs_map_view_tail")
                else:
                    _lib.check(L.lvdgs_tracking_tail(None, C.byref(a), None, _P(d_tau), 1, stream), "lvdgs_tracking_tail")
                initialization = True   # no exposure gradients from this loss
            elif image_loss is None:
                if stats is not None:
                    sa = _lib.ViewStatsArgs()
                    sa.radii_max, sa.norm_sum, sa.vis_count, sa.touched_row, sa.split_xy = (_P(t) for t in stats)
                    _lib.check(L.lvdgs_map_view_tail(C.byref(la), C.byref(a), _P(d_tau), C.byref(sa), stream), "lvdgs_map_view_tail")
                else:
                    _lib.check(L.lvdgs_tracking_tail(C.byref(la), C.byref(a), None, _P(d_tau), 1, stream), "lvdgs_tracking_tail")
        if image_loss is not None:
            res = image_loss(color, depth)
            loss, d_color = res[0], f32c(res[1])
            d_depth = f32c(res[2]) if len(res) > 2 and res[2] is not None else None
            keep += [d_color, d_depth]
            a.dL_dout_color, a.dL_dout_depth, a.dL_dout_opacity, a.dL_dtau = _P(d_color), _P(d_depth), None, _P(d_tau)
            with _lib.on_device(dev):
                _lib.check(L.lvdgs_backward(C.byref(a), _lib.raw_stream(dev)), "lvdgs_backward")
            a.dL_dout_color = a.dL_dout_depth = a.dL_dtau = None
            initialization = True   # no exposure gradients from this loss

        # ---- hand the gradients over exactly where autograd would have put them ----
        if K > 1:
            into["_features_dc"].copy_(d_sh[:, :1])
            into["_features_rest"].copy_(d_sh[:, 1:])
        if accumulate:
            pass   # (added in place by the backward)
        elif has[0]:
            torch._foreach_add_([getattr(G, n).grad for n in fields], [into[n] for n in fields])
        else:
            for n in fields:
                getattr(G, n).grad = into[n]
        pose = (("cam_trans_delta", d_tau[:3]), ("cam_rot_delta", d_tau[3:]))
        expo = () if initialization else (("exposure_a", d_a), ("exposure_b", d_b))
        for name, g in pose + expo:
            p = getattr(viewpoint, name)
            if p.requires_grad:
                g = g.view_as(p)
                p.grad = g if p.grad is None else p.grad + g
        vsp = SimpleNamespace(grad=d_m2, stats_taken=stats is not None and image_loss is None)   # stands in for the leaf autograd would have filled: only .grad is read
        # (visibility_filter: a launch of its own; a caller that had the statistics taken by the tail launch does not read it)
        pkg = {"render": color, "viewspace_points": vsp, "visibility_filter": (radii > 0) if (want_visibility and not vsp.stats_taken) else None, "radii": radii, "depth": depth,
               "opacity": opacity, "n_touched": n_touched}
        return pkg, loss


# Frames of up to 16384 tiles (the counting path's limit: lvdgs_forward_batch).  Round 4 stopped at 4096 -- a 1080p frame fills the
# chip by itself and the batched BLEND passes alone bought nothing there (5.79 -> 5.77 ms) -- but with the forward chains and the
# static-mask losses batched too the window gains at every size (same box, ms per 8 + 2 iteration, unmasked / masked keyframes):
# 500 k / 1080p 5.79 -> 5.42 / 6.47 -> 5.94, 2 M / 1920x1280 12.55 -> 12.14 / 13.06 -> 12.48.  (LVDGS_MAX_BATCH_TILES: A/B knob.)
MAX_BATCH_TILES = int(os.environ.get("LVDGS_MAX_BATCH_TILES", "16384"))
MEMORY_FRACTION = 0.8   # of the device memory still to be had: what a window batch may ask for (MapWindowBatch.usable)


class MapWindowBatch:
    """The views of a mapping window through ``MapViewPass``'s three calls with the two blend passes of ALL views in one launch each:

        once:      lvdgs_forward_batch (LVDGS_FLAG_NO_BLEND: projection + counting, the two scans, the scatter and the per-tile depth
                   sort of ALL views, a launch per stage; LVDGS_MAP_FWD_BATCH=0: lvdgs_forward view by view, as in round 4)
        once:      lvdgs_blend_forward_batch, [lvdgs_masked_loss_batch over the views with a static mask,] lvdgs_blend_backward_window_batch
        once:      lvdgs_gaussian_backward_batch (the per-Gaussian passes of all views, the parameter gradients summed in registers in the
                   window's order and written once), lvdgs_map_view_tail_batch

    Keyframes with a static mask -- the reference's default: all eight of the window -- and views without one (the two random older
    views) share the launches: the backward blend kernel takes each view's pixel gradients from the loss that view is scored by.

    A KITTI-size frame (1848 tiles) leaves the chip half empty and ends in a tail of its heaviest tiles; ten frames fill it.
    Every view keeps buffers of its own between the phases (geometry, pair lists, image state, scratch: ~120 MB per view at
    KITTI's size, ~300 MB at 500 k Gaussians / 1080p, ~1 GB at 2 M: checked against the device's free memory, ``usable``).  Results are ``MapViewPass.run``'s view after view, bit for bit: the same kernels on
    the same data, the parameter gradients added in the same order.

    For whole views scored by the built-in mapping loss or the static-mask loss on a model of SH degree 0 (``usable``); anything
    else goes view by view."""

    def __init__(self, lead: MapViewPass):
        self.passes = [lead]

    @staticmethod
    def usable(backend, viewpoints, masked=None) -> bool:
        """``masked[k]``: view k is scored by the static-mask loss (default: none is)."""
        G = backend.gaussians
        if len(viewpoints) < 2 or int(G._features_rest.shape[1]) != 0 or _rz.LIST_ALL_TILES:
            return False
        size = {(int(v.image_height), int(v.image_width)) for v in viewpoints}
        if len(size) != 1 or any(((h + 15) // 16) * ((w + 15) // 16) > MAX_BATCH_TILES for h, w in size):
            return False
        masked = masked or [None] * len(viewpoints)
        if not all(MapViewPass.usable(backend, v, allow_static_mask=True) and (m is None or MapViewPass.masked_loss_usable(v, m[1] is not None))
                   for v, m in zip(viewpoints, masked)):
            return False
        # every view holds geometry, pair lists, image state and scratch of its own: a large map at a small frame size can ask for
        # more than the device has left (an allocation failure half-way through the batch has no way back to the view-by-view path).
        # The verdict is kept for as long as what it was taken for stands -- map size, frame size, number of views, which of them are
        # masked, the pair capacity the passes have grown to: torch.cuda.mem_get_info is a driver call, and this runs in every
        # iteration of a ~1.6 ms loop
        dev = G.get_xyz.device
        (h, w), = size
        batch = getattr(backend, "_lvdgs_window_batch", None)
        cap = max((int(getattr(p, "cap", 0) or 0) for p in batch.passes), default=0) if batch is not None else 0
        key = (int(G._xyz.shape[0]), w, h, len(viewpoints), tuple(m is not None for m in masked), cap, dev)
        kept = getattr(backend, "_lvdgs_batch_fits", None)
        if kept is not None and kept[0] == key:
            return kept[1]
        need = MapWindowBatch.bytes_per_view(int(G._xyz.shape[0]), w, h, cap) * len(viewpoints)
        free, _ = torch.cuda.mem_get_info(dev)
        cached = torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)   # (what the caching allocator can hand out again)
        held = getattr(batch, "bytes_held", 0) if batch is not None else 0
        fits = need - held <= MEMORY_FRACTION * (free + cached)
        try:
            backend._lvdgs_batch_fits = (key, fits)
        except Exception:
            pass
        return fits

    @staticmethod
    def bytes_per_view(N, W, H, cap=None):
        """Device memory one view of the batch keeps between the phases (state, scratch, images, per-Gaussian outputs)."""
        L = _lib.lib()
        cap = max(int(cap or 0), _rz._MIN_PAIR_CAPACITY, _rz._PAIRS_PER_GAUSSIAN_GUESS * N, 1)
        scratch = max(L.lvdgs_prepare_scratch_bytes(N), L.lvdgs_backward_scratch_bytes(N, cap), L.lvdgs_render_scratch_bytes(N, cap, W, H))
        return int(L.lvdgs_geom_bytes(N) + L.lvdgs_binning_bytes(cap) + L.lvdgs_image_bytes(W, H) + scratch + L.lvdgs_loss_scratch_bytes(W, H)
                   + L.lvdgs_masked_loss_scratch_bytes(W, H) + 4 * (3 + 5) * W * H + 4 * (2 + 3) * N)

    def run(self, backend, viewpoints, initialization=False, first=None, stats=None, masked=None):
        """-> [(pkg, loss)] in the order of ``viewpoints`` (``stats[k]``: ``MapViewPass.run``'s ``stats`` of view k, ``masked[k]``: its
        ``masked_loss`` -- None for a view scored by ``get_loss_mapping``)."""
        lead = self.passes[0]
        dev, L = lead.dev, lead.L
        while len(self.passes) < len(viewpoints):
            self.passes.append(MapViewPass(dev, own_gradient_buffers=False))
        G = backend.gaussians
        self.runs = getattr(self, "runs", 0) + 1
        self.bytes_held = self.bytes_per_view(int(G._xyz.shape[0]), int(viewpoints[0].image_width), int(viewpoints[0].image_height),
                                              max((int(getattr(p, "cap", 0) or 0) for p in self.passes), default=0)) * len(self.passes)
        n = len(viewpoints)
        masked = masked or [None] * n
        ctxs = []
        fwd_batch = os.environ.get("LVDGS_MAP_FWD_BATCH", "1") != "0"
        with _lib.on_device(dev):
            stream = _lib.raw_stream(dev)
            # (two-level grouping -- rasterizer.super_tiles_flag -- for all views alike: hinted by the largest pair count the passes saw last time)
            N_now = int(G._xyz.shape[0])
            last_pairs = max((int(p.a.num_rendered) for p in self.passes[:n]), default=0) if getattr(self, "_pairs_of_n", None) == N_now else None
            self._pairs_of_n = N_now
            sflag = _rz.super_tiles_flag(N_now, last_pairs)
            for k, vp in enumerate(viewpoints):
                ctxs.append(self.passes[k]._begin_for_batch(backend, vp, initialization, first if k == 0 else None,
                                                            None if k == 0 else ctxs[0]["into"], stream, masked[k], forward=not fwd_batch, super_flag=sflag))
            # (the pointer arrays over the passes' argument blocks are made once per window shape: nothing to allocate per iteration)
            key = (n, tuple(m is not None for m in masked))
            ct = getattr(self, "_ct", {}).get(key)
            if ct is None:
                which = [k for k in range(n) if masked[k] is not None]
                ct = dict(views=(C.POINTER(_lib.Args) * n)(*[C.pointer(self.passes[k].a) for k in range(n)]),
                          losses=(C.POINTER(_lib.LossArgs) * n)(*[C.pointer(self.passes[k].la) for k in range(n)]),
                          nums=(C.c_int64 * n)(), which=which,
                          mviews=(C.POINTER(_lib.MaskedLossArgs) * max(len(which), 1))(*[C.pointer(self.passes[k].ml) for k in which]),
                          per_view=(C.POINTER(_lib.MaskedLossArgs) * n)(*[C.pointer(self.passes[k].ml) if masked[k] is not None else None for k in range(n)]),
                          tail_losses=(C.POINTER(_lib.LossArgs) * n)(*[None if masked[k] is not None else C.pointer(self.passes[k].la) for k in range(n)]),
                          taus=(C.c_void_p * n)(), sas=[_lib.ViewStatsArgs() for _ in range(n)])
                ct["sap"] = (C.POINTER(_lib.ViewStatsArgs) * n)(*[C.pointer(sa) for sa in ct["sas"]])
                self._ct = {key: ct}
            views = ct["views"]
            if fwd_batch:
                nums = ct["nums"]
                status = L.lvdgs_forward_batch(views, n, nums, stream)
                if status not in (_lib.OK, _lib.E_CAPACITY):
                    _lib.check(status, "lvdgs_forward_batch")
                for k in range(n):
                    self.passes[k]._after_forward(int(nums[k]), status == _lib.E_CAPACITY and int(nums[k]) > self.passes[k].cap, stream)
            losses = ct["losses"]
            _lib.check(L.lvdgs_blend_forward_batch(views, n, stream), "lvdgs_blend_forward_batch")
            which = ct["which"]
            if which:
                _lib.check(L.lvdgs_masked_loss_batch(ct["mviews"], len(which), stream), "lvdgs_masked_loss_batch")
                _lib.check(L.lvdgs_blend_backward_window_batch(views, losses, ct["per_view"], n, 0, stream), "lvdgs_blend_backward_window_batch")
            else:
                _lib.check(L.lvdgs_blend_backward_fused_loss_batch(views, losses, n, 0, stream), "lvdgs_blend_backward_fused_loss_batch")
            # the per-Gaussian passes of all views in ONE launch, the parameter gradients summed in registers in the window's order
            # (lvdgs_gaussian_backward_batch; LVDGS_MAP_PBWD_BATCH=0: view after view, each adding to the buffers, as until round 5) ...
            one_tail = stats is not None and all(st is not None for st in stats) and os.environ.get("LVDGS_MAP_TAIL_BATCH", "1") != "0"
            one_pass = os.environ.get("LVDGS_MAP_PBWD_BATCH", "1") != "0" and int(G.active_sh_degree) == 0
            if one_pass:
                _lib.check(L.lvdgs_gaussian_backward_batch(views, n, stream), "lvdgs_gaussian_backward_batch")
            for k in range(n):
                self.passes[k]._backward_for_batch(ctxs[k], None if stats is None else stats[k], stream, tail=not one_tail, gaussian_pass=not one_pass)
            if one_tail:   # ... and their tails -- loss, pose gradient, the view's statistics -- in ONE launch, the statistics in view order
                for k in range(n):
                    sa = ct["sas"][k]
                    sa.radii_max, sa.norm_sum, sa.vis_count, sa.touched_row, sa.split_xy = (_P(t) for t in stats[k])
                    ct["taus"][k] = ctxs[k]["d_tau"].data_ptr()
                    ctxs[k]["stats_taken"] = True
                _lib.check(L.lvdgs_map_view_tail_batch(ct["tail_losses"], views, ct["taus"], ct["sap"], n, stream), "lvdgs_map_view_tail_batch")
        return [self.passes[k]._finish_for_batch(backend, viewpoints[k], ctxs[k]) for k in range(n)]


def _after_forward(self, D, overflow, stream):
    """The view's pair count is known: more pairs than the buffers hold -> grow them and redo the binning (still without the blend)."""
    a = self.a
    if overflow:
        self._size_for_pairs(D + D // 2)
        a.num_rendered = D
        _lib.check(self.L.lvdgs_forward_render(C.byref(a), stream), "lvdgs_forward_render")
    a.num_rendered = D


def _begin_for_batch(self, backend, viewpoint, initialization, first, lead_into, stream, masked_loss=None, forward=True, super_flag=0):
    """``MapViewPass.run`` up to the forward call (whole view, built-in or static-mask loss, SH degree 0), with LVDGS_FLAG_NO_BLEND."""
    G, cfg, dev, L = backend.gaussians, backend.config, self.dev, self.L
    T = cfg["Training"]
    N, K = int(G._xyz.shape[0]), 1 + int(G._features_rest.shape[1])
    H, W = int(viewpoint.image_height), int(viewpoint.image_width)
    if N != self.N or K != getattr(self, "K", -1):
        self._size_for_model(N, K)
    if (W, H) != (self.W, self.H):
        self._size_for_image(W, H)
    a, la = self.a, self.la
    e = lambda *s, dt=torch.float32: torch.empty(*s, dtype=dt, device=dev)
    keep = self._keep = []
    f32c = lambda t: t.detach() if _gpu_f32c(t, dev) else t.detach().to(device=dev, dtype=torch.float32).contiguous()
    a.tanfovx, a.tanfovy = math.tan(viewpoint.FoVx * 0.5), math.tan(viewpoint.FoVy * 0.5)
    a.scale_modifier, a.sh_degree, a.prefiltered, a.debug = 1.0, int(G.active_sh_degree), 0, 0
    cam = [f32c(t) for t in (backend.background, viewpoint.world_view_transform, viewpoint.full_proj_transform,
                             viewpoint.projection_matrix, viewpoint.camera_center)]
    keep += cam
    a.bg, a.viewmatrix, a.projmatrix, a.projmatrix_raw, a.campos = (_P(t) for t in cam)
    a.activations = _rz.ACT_EXP_SCALES | _rz.ACT_NORMALIZE_ROTATIONS | _rz.ACT_SIGMOID_OPACITIES
    a.tile_row_begin, a.tile_row_end = 0, 0
    shs = G._features_dc.detach()
    a.means3D, a.opacities, a.scales, a.rotations, a.shs = _P(G._xyz), _P(G._opacity), _P(G._scaling), _P(G._rotation), _P(shs)
    color, depth, opacity = e(3, H, W), e(1, H, W), e(1, H, W)
    radii, n_touched = e(N, dt=torch.int32), e(N, dt=torch.int32)
    a.radii, a.n_touched, a.out_color, a.out_depth, a.out_opacity = _P(radii), _P(n_touched), _P(color), _P(depth), _P(opacity)

    fields = [n for n in _PARAM_FIELDS if getattr(G, n).numel() > 0]
    if lead_into is None:   # the batch's first view: as MapViewPass.run decides
        has = [getattr(G, n).grad is not None for n in fields]
        if any(has) and not all(has):
            # A parameter without a gradient next to parameters with one: its gradient so far is zero.  (The sharded loop gets here: a pruning
            # pass leaves its gradients in place -- reference utils/slam_backend.py:318-348 returns before the step -- and on the rank that
            # holds the isotropic term but was dealt no view of that pass, the scales alone carry one.)
            for n in fields:
                if getattr(G, n).grad is None:
                    getattr(G, n).grad = torch.zeros_like(getattr(G, n))
            has = [True] * len(fields)
        if has[0] and any(not _gpu_f32c(getattr(G, n).grad, dev) for n in fields):
            raise _lib.LvdgsError("MapViewPass: the model's existing gradients are not contiguous float32 tensors on the GPU")
        accumulate, install = has[0], not has[0]
        into = {n: getattr(G, n).grad for n in fields} if accumulate else (self.first if first is None else first)
    else:                   # a later view: added to where the first view's gradients are (being) written
        accumulate, install, into = True, False, lead_into
    a.flags = _lib.FLAG_NO_BLEND | (_lib.FLAG_ACCUMULATE_PARAM_GRADS if accumulate else 0) | int(super_flag)
    a.dL_dmeans3D, a.dL_dopacities, a.dL_dscales = _P(into["_xyz"]), _P(into["_opacity"]), _P(into["_scaling"])
    a.dL_drotations, a.dL_dshs = _P(into["_rotation"]), _P(into["_features_dc"])
    d_tau, d_a, d_b, d_m2 = e(6), e(1), e(1), e(N, 3)
    a.dL_dtau, a.dL_dmeans2D = None, _P(d_m2)
    loss = e(())
    if masked_loss is not None:
        loss = self._point_masked_loss(backend, viewpoint, masked_loss, color, depth, cam[0], keep)[0]
        initialization = True   # no exposure gradients from this loss
    else:
        gt = f32c(_gt_image(viewpoint, color))
        keep.append(gt)
        la.image, la.gt_image = _P(color), _P(gt)
        la.rgb_boundary_threshold = float(T["rgb_boundary_threshold"])
        if initialization:
            la.exposure_a = la.exposure_b = la.d_exposure_a = la.d_exposure_b = None
        else:
            la.exposure_a, la.exposure_b, la.d_exposure_a, la.d_exposure_b = _P(viewpoint.exposure_a), _P(viewpoint.exposure_b), _P(d_a), _P(d_b)
        alpha = T.get("alpha", 0.95)
        md = f32c(_mono_depth(viewpoint, color))
        keep.append(md)
        la.depth, la.gt_depth = _P(depth), _P(md)
        la.weight_rgb, la.weight_depth = float(alpha), float(1 - alpha)
        la.loss = _P(loss)

    if forward:   # (else: lvdgs_forward_batch runs the forward passes of all views of the window together)
        num = C.c_int64(0)
        status = L.lvdgs_forward(C.byref(a), C.byref(num), stream)
        if status != _lib.E_CAPACITY:
            _lib.check(status, "lvdgs_forward")
        self._after_forward(int(num.value), status == _lib.E_CAPACITY, stream)
    return dict(color=color, depth=depth, opacity=opacity, radii=radii, n_touched=n_touched, d_tau=d_tau, d_a=d_a, d_b=d_b, d_m2=d_m2,
                loss=loss, into=into, install=install, fields=fields, initialization=initialization, masked=masked_loss is not None)


def _backward_for_batch(self, ctx, stats, stream, tail=True, gaussian_pass=True):
    L, a, la = self.L, self.a, self.la
    if ctx["masked"]:
        if gaussian_pass:
            _lib.check(L.lvdgs_backward_masked_loss(C.byref(a), C.byref(self.ml), stream), "lvdgs_backward_masked_loss")
        la = None   # (the loss value is finished: the tail reduces the pose gradient and takes the statistics)
    elif gaussian_pass:
        _lib.check(L.lvdgs_backward_fused_loss(C.byref(a), C.byref(la), 0, stream), "lvdgs_backward_fused_loss")
    ctx["stats_taken"] = False
    if not tail:   # (the caller finishes all views in one launch: lvdgs_map_view_tail_batch)
        return
    la_ref = None if la is None else C.byref(la)
    if stats is not None:
        sa = _lib.ViewStatsArgs()
        sa.radii_max, sa.norm_sum, sa.vis_count, sa.touched_row, sa.split_xy = (_P(t) for t in stats)
        _lib.check(L.lvdgs_map_view_tail(la_ref, C.byref(a), _P(ctx["d_tau"]), C.byref(sa), stream), "lvdgs_map_view_tail")
    else:
        _lib.check(L.lvdgs_tracking_tail(la_ref, C.byref(a), None, _P(ctx["d_tau"]), 1, stream), "lvdgs_tracking_tail")
    ctx["stats_taken"] = stats is not None


def _finish_for_batch(self, backend, viewpoint, ctx):
    G = backend.gaussians
    if ctx["install"]:
        for n in ctx["fields"]:
            getattr(G, n).grad = ctx["into"][n]
    d_tau = ctx["d_tau"]
    pose = (("cam_trans_delta", d_tau[:3]), ("cam_rot_delta", d_tau[3:]))
    expo = () if ctx["initialization"] else (("exposure_a", ctx["d_a"]), ("exposure_b", ctx["d_b"]))
    for name, g in pose + expo:
        p = getattr(viewpoint, name)
        if p.requires_grad:
            g = g.view_as(p)
            p.grad = g if p.grad is None else p.grad + g
    vsp = SimpleNamespace(grad=ctx["d_m2"], stats_taken=ctx["stats_taken"])
    radii = ctx["radii"]
    pkg = {"render": ctx["color"], "viewspace_points": vsp, "visibility_filter": (radii > 0) if not vsp.stats_taken else None, "radii": radii,
           "depth": ctx["depth"], "opacity": ctx["opacity"], "n_touched": ctx["n_touched"]}
    return pkg, ctx["loss"]


MapViewPass._begin_for_batch = _begin_for_batch
MapViewPass._after_forward = _after_forward
MapViewPass._backward_for_batch = _backward_for_batch
MapViewPass._finish_for_batch = _finish_for_batch
