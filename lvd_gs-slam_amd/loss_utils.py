"""``l1_loss`` / ``ssim`` of the mapping and refinement losses, on the fused HIP kernel.

The reference imports both from ``gaussian_splatting.utils.loss_utils`` (``utils/slam_backend.py:11``,
``utils/eval_utils_0806.py:28``; the package is absent from the checkout) and combines them as
``(1 - lambda_dssim) * l1_loss(a, b) + lambda_dssim * (1 - ssim(a, b))`` -- on the static pixels only when
the keyframe carries a ``static_mask`` (``utils/slam_backend.py:199-215``) and in colour refinement
(``:438-454``).  ``ssim`` is the published 11-tap Gaussian-window SSIM (sigma 1.5, zero padding).

``l1_dssim_loss`` evaluates the whole combination (mask overwrite included) in ONE launch that also
produces the gradient image; ``ssim`` and ``l1_loss`` keep the upstream signatures.  GPU float32 only:
there is no PyTorch convolution fallback.
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
