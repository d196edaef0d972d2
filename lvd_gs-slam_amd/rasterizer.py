"""Autograd boundary of the rasterizer: ``GaussianRasterizationSettings`` / ``GaussianRasterizer``.

Same names, argument lists and return values as the ``diff_gaussian_rasterization`` module the
reference's ``gaussian_splatting.gaussian_renderer.render`` is written against (reference
README.md:43; the module itself is absent from the reference checkout, so the signature is the
published MonoGS one -- SURVEY.md section 8(b)):

    GaussianRasterizer(raster_settings)(means3D, means2D, opacities, shs=None, colors_precomp=None,
                                        scales=None, rotations=None, cov3D_precomp=None,
                                        theta=None, rho=None)
        -> color (3,H,W), radii (N,) int32, depth (1,H,W), opacity (1,H,W), n_touched (N,) int32

Gradients flow to means3D, means2D (the "viewspace points": d/d NDC xy), shs or colors_precomp,
opacities, scales, rotations or cov3D_precomp, and to ``theta`` / ``rho`` -- the camera's
``cam_rot_delta`` / ``cam_trans_delta`` (reference utils/camera_utils.py:51-56), i.e. the left
SE(3) perturbation that ``utils/pose_utils.py:70-87`` later folds into the pose.

All compute happens in ``lib/liblvdgs.so`` (HIP, gfx950) through the C ABI in
``include/lvdgs.h``; this file only allocates tensors and passes pointers.
"""
import ctypes as C
from typing import NamedTuple

import os

import torch
from torch import nn

from . import _lib

# d(loss)/d(opacity image) is NOT propagated by default: the backward binding this module mirrors
# (`rasterize_gaussians_backward(..., dL_dout_color, dL_dout_depth, ...)`) takes the gradients of
# the colour and depth images only, so the opacity image acts as a detached weight in the tracking loss
# (reference utils/slam_utils.py:60 multiplies the residual by it).  The kernels do implement the path
# (opacity = 1 - final transmittance; checked against the oracle); set this to True to use it.
PROPAGATE_OPACITY_GRAD = False

# True: list every tile of a Gaussian's 3-sigma rectangle (lvdgs_args.flags |= LVDGS_FLAG_LIST_ALL_TILES) -- the reference's
# (Gaussian, tile) pair list, num_rendered and n_contrib bit for bit -- instead of only the tiles the Gaussian can reach with
# alpha >= 1/255.  Images, radii, n_touched are the same either way (a dropped pair contributes to no pixel).
LIST_ALL_TILES = False

# Parity tests set this to read intermediates (state buffers) of the most recent forward.
KEEP_DEBUG_STATE = False
_DEBUG_LAST = {}

# pairs the binning buffers are sized for, per device index (see _RasterizeGaussians.forward)
_PAIR_CAPACITY = {}
_MIN_PAIR_CAPACITY = 1 << 20      # floor of the capacity guess
_PAIRS_PER_GAUSSIAN_GUESS = 8     # first guess before any frame has been seen


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    projmatrix_raw: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool = False
    debug: bool = False


def _f32(t, device):
    if t is None:
        return None
    if not torch.is_tensor(t):
        t = torch.as_tensor(t)
    if t.numel() == 0:
        return None
    if t.dtype is torch.float32 and t.device == device and t.is_contiguous():
        return t.detach()  # the common case: no conversion kernel, no dispatcher round trip through .to()
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _bytes(n, device):
    return torch.empty(max(int(n), 256), dtype=torch.uint8, device=device)


def _fill_settings(a, rs, device, keep):
    a.image_height, a.image_width = int(rs.image_height), int(rs.image_width)
    a.tanfovx, a.tanfovy = float(rs.tanfovx), float(rs.tanfovy)
    a.scale_modifier = float(rs.scale_modifier)
    a.sh_degree, a.prefiltered, a.debug = int(rs.sh_degree), int(bool(rs.prefiltered)), int(bool(rs.debug))
    for name in ("bg", "viewmatrix", "projmatrix", "projmatrix_raw", "campos"):
        t = _f32(getattr(rs, name), device)
        keep.append(t)
        setattr(a, name, _ptr(t))


def _stream(device=None):
    """Raw hipStream_t of PyTorch's current stream on `device` (a direct C call: torch.cuda.current_stream()
    goes through lazy-init and device-count checks that cost tens of microseconds per call)."""
    idx = device.index if (device is not None and device.index is not None) else torch.cuda.current_device()
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(idx))


# Two-level grouping (include/lvdgs.h: LVDGS_FLAG_SUPER_TILES): worth it when a Gaussian is listed on many tiles -- the opaque surfaces of
# large flat Gaussians SLAM maps are made of (70-80 tiles each) -- and a loss on scenes of small blobs (3 tiles each), so the hint is set
# from what the PREVIOUS frame of the same caller looked like: pairs per Gaussian at or above SUPER_TILES_MIN_PAIRS_PER_GAUSSIAN.  Outputs
# are the same bits either way.  LVDGS_SUPER_TILES=0 / 1 in the environment: never / always (A/B measurements, tests).
SUPER_TILES_MIN_PAIRS_PER_GAUSSIAN = 16.0
_SUPER_TILES_ENV = os.environ.get("LVDGS_SUPER_TILES", "auto")


def super_tiles_flag(num_gaussians, last_num_rendered):
    """The flag bit for a frame of ``num_gaussians`` whose predecessor listed ``last_num_rendered`` pairs (None / 0: unknown)."""
    if _SUPER_TILES_ENV == "0" or LIST_ALL_TILES:
        return 0
    if _SUPER_TILES_ENV == "1":
        return _lib.FLAG_SUPER_TILES
    if not last_num_rendered or num_gaussians <= 0:
        return 0
    return _lib.FLAG_SUPER_TILES if last_num_rendered >= SUPER_TILES_MIN_PAIRS_PER_GAUSSIAN * num_gaussians else 0


_LAST_PAIRS = {}   # device index -> (N, D) of the last frame through the autograd API on that device


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, theta, rho,
                raster_settings, activations=0):
        L = _lib.lib()
        rs = raster_settings
        dev = means3D.device
        if dev.type != "cuda":
            raise _lib.LvdgsError("GaussianRasterizer needs tensors on the GPU (there is no CPU path)")
        N = int(means3D.shape[0])
        H, W = int(rs.image_height), int(rs.image_width)
        keep = []
        a = _lib.Args()
        _fill_settings(a, rs, dev, keep)
        m3 = _f32(means3D, dev)
        op = _f32(opacities, dev)
        shs, col = _f32(sh, dev), _f32(colors_precomp, dev)
        sc, rot, cov = _f32(scales, dev), _f32(rotations, dev), _f32(cov3Ds_precomp, dev)
        if N > 0:
            if (shs is None) == (col is None):
                raise ValueError("Please provide exactly one of either SHs or precomputed colors!")
            if ((sc is None or rot is None) and cov is None) or ((sc is not None or rot is not None) and cov is not None):
                raise ValueError("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        a.num_gaussians = N
        a.activations = int(activations)
        a.flags = _lib.FLAG_LIST_ALL_TILES if LIST_ALL_TILES else 0
        _key = dev.index if dev.index is not None else torch.cuda.current_device()
        _last = _LAST_PAIRS.get(_key)
        a.flags |= super_tiles_flag(N, _last[1] if (_last is not None and _last[0] == N) else None)
        a.sh_coeffs = int(shs.shape[1]) if shs is not None else 0
        a.means3D, a.opacities, a.scales, a.rotations = _ptr(m3), _ptr(op), _ptr(sc), _ptr(rot)
        a.cov3D_precomp, a.shs, a.colors_precomp = _ptr(cov), _ptr(shs), _ptr(col)

        radii = torch.empty(N, dtype=torch.int32, device=dev)
        n_touched = torch.empty(N, dtype=torch.int32, device=dev)
        color = torch.empty(3, H, W, dtype=torch.float32, device=dev)
        depth = torch.empty(1, H, W, dtype=torch.float32, device=dev)
        opac = torch.empty(1, H, W, dtype=torch.float32, device=dev)
        geom = _bytes(L.lvdgs_geom_bytes(N), dev)
        image = _bytes(L.lvdgs_image_bytes(W, H), dev)
        a.radii, a.n_touched = _ptr(radii), _ptr(n_touched)
        a.out_color, a.out_depth, a.out_opacity = _ptr(color), _ptr(depth), _ptr(opac)
        a.geom_state, a.geom_bytes = _ptr(geom), geom.numel()
        a.image_state, a.image_bytes = _ptr(image), image.numel()

        # Single-call forward: buffers are sized for a pair capacity remembered per device (grown when a
        # frame gets within 25 % of it), every kernel is enqueued before the host waits for the pair
        # count, and an overflow (rare) re-runs only the binning + blend stage with exact sizes.
        stream = _stream(dev)
        key = dev.index if dev.index is not None else torch.cuda.current_device()
        cap = max(_PAIR_CAPACITY.get(key, 0), _MIN_PAIR_CAPACITY, _PAIRS_PER_GAUSSIAN_GUESS * N, 1) if N > 0 else 0
        binning = _bytes(L.lvdgs_binning_bytes(cap), dev)
        scratch = _bytes(max(L.lvdgs_prepare_scratch_bytes(N), L.lvdgs_render_scratch_bytes(N, cap, W, H)), dev)
        a.pair_capacity = cap
        a.binning_state, a.binning_bytes = _ptr(binning), binning.numel()
        a.scratch, a.scratch_bytes = _ptr(scratch), scratch.numel()
        num = C.c_int64(0)
        status = L.lvdgs_forward(C.byref(a), C.byref(num), stream)
        D = int(num.value)
        binning_pairs = cap
        if status == _lib.E_CAPACITY:
            binning_pairs = D
            binning = _bytes(L.lvdgs_binning_bytes(D), dev)
            scratch = _bytes(L.lvdgs_render_scratch_bytes(N, D, W, H), dev)
            a.num_rendered = D
            a.binning_state, a.binning_bytes = _ptr(binning), binning.numel()
            a.scratch, a.scratch_bytes = _ptr(scratch), scratch.numel()
            _lib.check(L.lvdgs_forward_render(C.byref(a), stream), "lvdgs_forward_render")
        else:
            _lib.check(status, "lvdgs_forward")
        _LAST_PAIRS[_key] = (N, D)
        if N > 0 and 4 * D > 3 * cap:
            _PAIR_CAPACITY[key] = max(cap, D + D // 2)
        else:
            _PAIR_CAPACITY[key] = cap

        if KEEP_DEBUG_STATE:
            _DEBUG_LAST.clear()
            _DEBUG_LAST.update(geom=geom, binning=binning, image=image, num_rendered=D, N=N, W=W, H=H,
                               binning_pairs=binning_pairs, overflowed=status == _lib.E_CAPACITY)
        ctx.raster_settings = rs
        ctx.num_rendered = D
        ctx.activations = int(activations)
        ctx.flags = int(a.flags)
        ctx.pose = (torch.is_tensor(theta) and theta.numel() == 3, torch.is_tensor(rho) and rho.numel() == 3)
        ctx.save_for_backward(m3, op, sc, rot, cov, shs, col, radii, geom, binning, image)
        ctx.mark_non_differentiable(radii, n_touched)
        return color, radii, depth, opac, n_touched

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_depth, grad_opacity, grad_n_touched):
        L = _lib.lib()
        rs = ctx.raster_settings
        m3, op, sc, rot, cov, shs, col, radii, geom, binning, image = ctx.saved_tensors
        dev = m3.device
        N = int(m3.shape[0])
        H, W = int(rs.image_height), int(rs.image_width)
        D = ctx.num_rendered
        keep = []
        a = _lib.Args()
        _fill_settings(a, rs, dev, keep)
        a.num_gaussians = N
        a.sh_coeffs = int(shs.shape[1]) if shs is not None else 0
        a.means3D, a.opacities, a.scales, a.rotations = _ptr(m3), _ptr(op), _ptr(sc), _ptr(rot)
        a.cov3D_precomp, a.shs, a.colors_precomp = _ptr(cov), _ptr(shs), _ptr(col)
        a.radii = _ptr(radii)
        a.activations = ctx.activations
        a.flags = ctx.flags
        a.num_rendered = D
        a.geom_state, a.geom_bytes = _ptr(geom), geom.numel()
        a.binning_state, a.binning_bytes = _ptr(binning), binning.numel()
        a.image_state, a.image_bytes = _ptr(image), image.numel()
        scratch = _bytes(L.lvdgs_backward_scratch_bytes(N, D), dev)
        a.scratch, a.scratch_bytes = _ptr(scratch), scratch.numel()

        g_color = _f32(grad_color, dev) if grad_color is not None else torch.zeros(3, H, W, device=dev)
        g_depth = _f32(grad_depth, dev) if grad_depth is not None else None
        g_opac = _f32(grad_opacity, dev) if (grad_opacity is not None and PROPAGATE_OPACITY_GRAD) else None
        a.dL_dout_color, a.dL_dout_depth, a.dL_dout_opacity = _ptr(g_color), _ptr(g_depth), _ptr(g_opac)

        e = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        d_m3, d_m2, d_op = e(N, 3), e(N, 3), e(*op.shape) if op is not None else e(N, 1)
        d_sc = e(N, 3) if sc is not None else None
        d_rot = e(N, 4) if rot is not None else None
        d_cov = e(N, 6) if cov is not None else None
        d_sh = e(*shs.shape) if shs is not None else None
        d_col = e(N, 3) if col is not None else None
        d_tau = e(6)
        a.dL_dmeans3D, a.dL_dmeans2D, a.dL_dopacities = _ptr(d_m3), _ptr(d_m2), _ptr(d_op)
        a.dL_dscales, a.dL_drotations, a.dL_dcov3D = _ptr(d_sc), _ptr(d_rot), _ptr(d_cov)
        a.dL_dshs, a.dL_dcolors, a.dL_dtau = _ptr(d_sh), _ptr(d_col), _ptr(d_tau)
        _lib.check(L.lvdgs_backward(C.byref(a), _stream(dev)), "lvdgs_backward")
        # (means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, theta, rho, settings)
        d_theta = d_tau[3:] if ctx.pose[0] else None
        d_rho = d_tau[:3] if ctx.pose[1] else None
        return d_m3, d_m2, d_sh, d_col, d_op, d_sc, d_rot, d_cov, d_theta, d_rho, None, None


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, theta, rho,
                        raster_settings, activations=0):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                     theta, rho, raster_settings, activations)


# `activations` bits (include/lvdgs.h): the rasterizer applies the model's activations itself
ACT_EXP_SCALES, ACT_NORMALIZE_ROTATIONS, ACT_SIGMOID_OPACITIES = 1, 2, 4


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        """Boolean mask of Gaussians in front of the view's near plane."""
        with torch.no_grad():
            rs = self.raster_settings
            pos = _f32(positions, positions.device)
            N = int(positions.shape[0])
            out = torch.empty(N, dtype=torch.uint8, device=positions.device)
            view, proj = _f32(rs.viewmatrix, positions.device), _f32(rs.projmatrix, positions.device)
            _lib.check(_lib.lib().lvdgs_mark_visible(N, _ptr(pos), _ptr(view), _ptr(proj), _ptr(out), _stream(positions.device)),
                       "lvdgs_mark_visible")
            return out.bool()

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, theta=None, rho=None, activations=0):
        """`activations` (lvdgs extension, default 0 = upstream behaviour): bit mask saying that `scales` are
        log-scales, `rotations` un-normalised quaternions, `opacities` logits; the kernels then apply
        exp / normalise / sigmoid and return gradients w.r.t. those raw values."""
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or (
                (scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        empty = torch.Tensor([])
        shs = empty if shs is None else shs
        colors_precomp = empty if colors_precomp is None else colors_precomp
        scales = empty if scales is None else scales
        rotations = empty if rotations is None else rotations
        cov3D_precomp = empty if cov3D_precomp is None else cov3D_precomp
        theta = empty if theta is None else theta
        rho = empty if rho is None else rho
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                   theta, rho, self.raster_settings, activations)
