"""``l1_loss`` / ``ssim`` of the mapping and refinement losses, on the fused HIP kernel.

The reference imports both from ``gaussian_splatting.utils.loss_utils`` (``utils/slam_backend.py:11``,
``utils/eval_utils_0806.py:28``; the package is absent from the checkout) and combines them as
``(1 - lambda_dssim) * l1_loss(a, b) + lambda_dssim * (1 - ssim(a, b))`` -- on the static pixels only when
the keyframe carries a ``static_mask`` (``utils/slam_backend.py:199-215``) and in colour refinement
(``:438-454``).  ``ssim`` is the published 11-tap Gaussian-window SSIM (sigma 1.5, zero padding).

``l1_dssim_loss`` evaluates the whole combination (mask overwrite included) in ONE launch that also
produces the gradient image; ``ssim`` and ``l1_loss`` keep the upstream signatures.  GPU float32 only:
there is no PyTorch convolution fallback.

``masked_depth_l1`` is the depth term the same branch adds (``utils/slam_backend.py:216-261``): the mean of
``|depth - mono_depth|`` over ``static_mask & (mono_depth > 0) & (depth > 0)`` -- normalised by the number of
such pixels, not by the image size -- and ``masked_mapping_loss`` is that branch as a whole.
"""
import ctypes as C

import torch

from . import _lib
from .fused_loss import _aligned, _c32, _p, _raw_stream


def _as_planes(img):
    if img.dim() == 3:
        return img.shape[0], img.shape[0], img.shape[1], img.shape[2]
    if img.dim() == 4:
        return img.shape[0] * img.shape[1], img.shape[1], img.shape[2], img.shape[3]
    raise ValueError(f"expected (C,H,W) or (B,C,H,W), got {tuple(img.shape)}")


def _launch(img1, img2, keep, bg, w_l1, w_ssim, want_grad):
    """-> (out[2] = mean|a-b|, mean SSIM ; d_img1 or None) with d_img1 = w_l1 dL1 + w_ssim dSSIM."""
    if not img1.is_cuda:
        raise _lib.LvdgsError("ssim / l1_dssim_loss run on the GPU only (HIP kernel); got a CPU tensor")
    if img1.shape != img2.shape:
        raise ValueError(f"shape mismatch: {tuple(img1.shape)} vs {tuple(img2.shape)}")
    planes, channels, H, W = _as_planes(img1)
    L = _lib.lib()
    dev = img1.device
    x, y = _c32(img1), _c32(img2)
    a = _lib.SsimArgs()
    a.width, a.height, a.planes, a.channels = W, H, planes, channels
    a.img1, a.img2, a.keep_mask, a.bg = _p(x), _p(y), _p(keep), _p(bg)
    a.weight_l1, a.weight_ssim = float(w_l1), float(w_ssim)
    scratch = torch.empty(int(L.lvdgs_ssim_scratch_bytes(W, H, planes)), dtype=torch.uint8, device=dev)
    out = torch.empty(2, dtype=torch.float32, device=dev)
    d = torch.empty_like(x) if want_grad else None
    a.scratch, a.scratch_bytes, a.out, a.d_img1 = _p(scratch), scratch.numel(), _p(out), _p(d)
    _lib.check(L.lvdgs_ssim_l1(C.byref(a), _raw_stream(dev)), "lvdgs_ssim_l1")
    return out, d


def _mask_bytes(mask, H, W):
    if mask is None:
        return None
    m = mask.reshape(-1)
    if m.numel() != H * W:
        raise ValueError(f"mask has {m.numel()} elements, image has {H * W} pixels")
    m = m.view(torch.uint8) if m.dtype == torch.bool else m.ne(0).view(torch.uint8)
    return _aligned(m.contiguous())


class _L1Ssim(torch.autograd.Function):
    """value = w_l1 * mean|a - b| + w_ssim * mean SSIM(a, b); gradient w.r.t. ``img1`` only."""

    @staticmethod
    def forward(ctx, img1, img2, keep, bg, w_l1, w_ssim):
        want = ctx.needs_input_grad[0]
        out, d = _launch(img1, img2, keep, bg, w_l1, w_ssim, want)
        ctx.shape = img1.shape
        if want:
            ctx.save_for_backward(d)
        return w_l1 * out[0] + w_ssim * out[1]

    @staticmethod
    def backward(ctx, g):
        (d,) = ctx.saved_tensors
        return (d * g).view(ctx.shape), None, None, None, None, None


def _check_no_grad_on_second(img2):
    if img2.requires_grad and torch.is_grad_enabled():
        raise NotImplementedError("the fused kernel differentiates w.r.t. the first image only; pass the rendered image "
                                  "first (SSIM and L1 are symmetric) or detach the second")


def l1_loss(network_output, gt):
    """``torch.abs(network_output - gt).mean()`` (gaussian_splatting.utils.loss_utils.l1_loss)."""
    return torch.abs(network_output - gt).mean()


def ssim(img1, img2, window_size=11, size_average=True):
    """Mean SSIM of two (C,H,W) or (B,C,H,W) images (gaussian_splatting.utils.loss_utils.ssim)."""
    if window_size != 11 or not size_average:
        raise NotImplementedError("the HIP kernel implements the configuration the reference calls: window 11, mean")
    _check_no_grad_on_second(img2)
    return _L1Ssim.apply(img1, img2, None, None, 0.0, 1.0)


def l1_dssim_loss(image, gt_image, lambda_dssim, static_mask=None, background=None):
    """``(1 - l) * l1_loss(a, b) + l * (1 - ssim(a, b))`` in one launch.

    With ``static_mask`` (bool (H,W), True = static) both images first get ``background[c]`` written
    into the dynamic pixels, as ``utils/slam_backend.py:199-215`` does with clones and index assignment."""
    _check_no_grad_on_second(gt_image)
    _, _, H, W = _as_planes(image)
    keep = _mask_bytes(static_mask, H, W)
    bg = _c32(background) if (background is not None and keep is not None) else None
    lam = float(lambda_dssim)
    return _L1Ssim.apply(image, gt_image, keep, bg, 1.0 - lam, -lam) + lam


# ---- depth term of the static-mask branch (utils/slam_backend.py:216-261) ------------------------------------
def _squeeze_hw(t):
    """(1,H,W) or (H,W,1) -> (H,W), as the reference does before combining depth, mono depth and mask (:223-236)."""
    if t.dim() == 3 and t.shape[0] == 1:
        return t.squeeze(0)
    if t.dim() == 3 and t.shape[-1] == 1:
        return t.squeeze(-1)
    return t


class _MaskedDepthL1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth, gt_depth, mask_bytes):
        L = _lib.lib()
        H, W = depth.shape
        dev = depth.device
        d, z = _c32(depth), _c32(gt_depth)
        a = _lib.MaskedDepthArgs()
        a.width, a.height = W, H
        a.depth, a.gt_depth, a.static_mask = _p(d), _p(z), _p(mask_bytes)
        scratch = torch.empty(int(L.lvdgs_masked_depth_scratch_bytes(W, H)), dtype=torch.uint8, device=dev)
        out = torch.empty(2, dtype=torch.float32, device=dev)
        a.scratch, a.scratch_bytes, a.out = _p(scratch), scratch.numel(), _p(out)
        with _lib.on_device(dev):
            _lib.check(L.lvdgs_masked_depth_l1_forward(C.byref(a), _raw_stream(dev)), "lvdgs_masked_depth_l1_forward")
        ctx.save_for_backward(d, z, out) if mask_bytes is None else ctx.save_for_backward(d, z, out, mask_bytes)
        return out  # [loss, |M|]; only the first entry carries a gradient

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        d, z, out, *m = ctx.saved_tensors
        H, W = d.shape
        dev = d.device
        a = _lib.MaskedDepthArgs()
        a.width, a.height = W, H
        a.depth, a.gt_depth, a.static_mask = _p(d), _p(z), _p(m[0] if m else None)
        gl = g.detach().to(torch.float32).contiguous()  # the kernel reads element 0: d objective / d loss
        dd = torch.empty_like(d)
        a.out, a.grad_loss, a.d_depth = _p(out), _p(gl), _p(dd)
        with _lib.on_device(dev):
            _lib.check(L.lvdgs_masked_depth_l1_backward(C.byref(a), _raw_stream(dev)), "lvdgs_masked_depth_l1_backward")
        return dd, None, None


def masked_depth_l1(depth, mono_depth, static_mask=None, return_count=False):
    """``|depth - mono_depth|[static_mask & (mono_depth > 0) & (depth > 0)].mean()``; 0 (and zero gradient) when no
    pixel qualifies -- the reference then skips the term (``if depth_mask.any()``, utils/slam_backend.py:250).
    Shapes (1,H,W) / (H,W,1) / (H,W) are accepted and, like the reference, cropped to the common top-left
    (min_h, min_w) window when they differ (:240-246).  One pass for the value and the pixel count, one for the gradient."""
    if not depth.is_cuda:
        raise _lib.LvdgsError("masked_depth_l1 runs on the GPU only (HIP kernel); got a CPU tensor")
    d = _squeeze_hw(depth)
    z = _squeeze_hw(mono_depth if torch.is_tensor(mono_depth) else torch.from_numpy(mono_depth)).to(d.device)
    m = None if static_mask is None else _squeeze_hw(static_mask).to(d.device)
    if d.dim() != 2 or z.dim() != 2 or (m is not None and m.dim() != 2):
        raise ValueError(f"masked_depth_l1: expected 2-D maps after squeezing, got {tuple(depth.shape)}, "
                         f"{tuple(z.shape)}, {None if m is None else tuple(m.shape)}")
    hs = [d.shape[0], z.shape[0]] + ([] if m is None else [m.shape[0]])
    ws = [d.shape[1], z.shape[1]] + ([] if m is None else [m.shape[1]])
    h, w = min(hs), min(ws)
    if (d.shape[0], d.shape[1]) != (h, w):
        d = d[:h, :w].contiguous()
    if (z.shape[0], z.shape[1]) != (h, w):
        z = z[:h, :w]
    if m is not None and (m.shape[0], m.shape[1]) != (h, w):
        m = m[:h, :w]
    out = _MaskedDepthL1.apply(d, z, _mask_bytes(m, h, w))
    return (out[0], out[1].detach()) if return_count else out[0]


def masked_mapping_loss_and_grads(image, depth, viewpoint, background, lambda_dssim, depth_lambda=0.1):
    """``masked_mapping_loss`` without autograd, as SEPARATE launches: (loss, d loss / d image (3,H,W), d loss / d depth (1,H,W) or
    None) from the fused L1 + SSIM launch (value and gradient image together) and the two masked-depth launches, for a caller that
    hands gradient images to the plain ``lvdgs_backward`` (``fast_mapping.MapViewPass.run(image_loss=...)``).  The mapping loop itself
    goes through ``lvdgs_masked_loss_batch`` / ``lvdgs_backward_masked_loss`` (``MapViewPass.run(masked_loss=...)``: no depth-gradient
    image, one launch for all masked views of a window); the tests hold the two against each other."""
    from .slam_utils import _mono_depth, _static_mask_bytes
    L = _lib.lib()
    dev = image.device
    H, W = int(image.shape[-2]), int(image.shape[-1])
    gt = viewpoint.original_image.to(dev)
    keep = _static_mask_bytes(viewpoint, image)   # (cached with the viewpoint: no upload / conversion per iteration)
    if keep.numel() != H * W:
        raise ValueError(f"mask has {keep.numel()} elements, image has {H * W} pixels")
    lam = float(lambda_dssim)
    out, d_image = _launch(image.detach(), gt, keep, _c32(background), 1.0 - lam, -lam, True)
    loss = (1.0 - lam) * out[0] - lam * out[1] + lam
    d_depth = None
    if depth is not None and getattr(viewpoint, "mono_depth", None) is not None:
        d = _c32(_squeeze_hw(depth.detach()))
        z = _c32(_squeeze_hw(_mono_depth(viewpoint, image)))
        if d.shape != z.shape or d.shape != (H, W):
            raise ValueError("masked_mapping_loss_and_grads: depth, mono depth and image must have one size")
        a = _lib.MaskedDepthArgs()
        a.width, a.height = W, H
        a.depth, a.gt_depth, a.static_mask = _p(d), _p(z), _p(keep)
        scratch = torch.empty(int(L.lvdgs_masked_depth_scratch_bytes(W, H)), dtype=torch.uint8, device=dev)
        res = torch.empty(2, dtype=torch.float32, device=dev)
        weight = torch.full((1,), float(depth_lambda), dtype=torch.float32, device=dev)
        d_depth = torch.empty(1, H, W, dtype=torch.float32, device=dev)
        a.scratch, a.scratch_bytes, a.out, a.grad_loss, a.d_depth = _p(scratch), scratch.numel(), _p(res), _p(weight), _p(d_depth)
        with _lib.on_device(dev):
            _lib.check(L.lvdgs_masked_depth_l1_forward(C.byref(a), _raw_stream(dev)), "lvdgs_masked_depth_l1_forward")
            _lib.check(L.lvdgs_masked_depth_l1_backward(C.byref(a), _raw_stream(dev)), "lvdgs_masked_depth_l1_backward")
        loss = loss + float(depth_lambda) * res[0]
    return loss, d_image, d_depth


def masked_mapping_loss(image, depth, viewpoint, background, lambda_dssim, depth_lambda=0.1):
    """The mapping loss of a keyframe that carries a ``static_mask`` (utils/slam_backend.py:199-261):
    ``(1 - l) * L1 + l * (1 - SSIM)`` on the images with the dynamic pixels overwritten by the background colour,
    plus ``depth_lambda * masked_depth_l1`` when the keyframe has a ``mono_depth``.  Two fused launches forward."""
    gt = viewpoint.original_image
    if gt.device != image.device:
        gt = gt.to(image.device)
    mask = viewpoint.static_mask
    if mask.device != image.device:
        mask = mask.to(image.device)
    loss = l1_dssim_loss(image, gt, lambda_dssim, mask, background)
    if depth is not None and getattr(viewpoint, "mono_depth", None) is not None:
        loss = loss + depth_lambda * masked_depth_l1(depth, viewpoint.mono_depth, mask)
    return loss
