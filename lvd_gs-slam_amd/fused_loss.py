"""Fused photometric loss (HIP): one pass for the loss, one for all its gradients.

Used by ``slam_utils.get_loss_tracking*`` / ``get_loss_mapping*`` when the rendered image lives on
the GPU; the formulas are those of the reference's ``utils/slam_utils.py:42-121`` (see the C ABI
comment in ``include/lvdgs.h``).  Replaces ~20 full-frame elementwise kernels per iteration.
"""
import ctypes as C

import torch

from . import _lib


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _aligned(t):
    return t if t.data_ptr() % 16 == 0 else t.clone()  # the kernels use 16-byte loads


def _c32(t):
    if t is None:
        return None
    if t.dtype is torch.float32 and t.is_contiguous():
        return _aligned(t.detach())
    return _aligned(t.detach().to(torch.float32).contiguous())


def _raw_stream(dev):
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(idx))


class _Photometric(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, depth, opacity, exposure_a, exposure_b, gt_image, gt_depth, grad_mask, rgb_thr, w_rgb, w_d,
                weight_by_opacity, depth_needs_opaque):
        L = _lib.lib()
        _, H, W = image.shape
        dev = image.device
        a = _lib.LossArgs()
        t = dict(image=_c32(image), depth=_c32(depth), opacity=_c32(opacity), gt_image=_c32(gt_image), gt_depth=_c32(gt_depth),
                 exposure_a=_c32(exposure_a), exposure_b=_c32(exposure_b), grad_mask=grad_mask)
        scratch = torch.empty(int(L.lvdgs_loss_scratch_bytes(W, H)), dtype=torch.uint8, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        a.width, a.height = W, H
        for k, v in t.items():
            setattr(a, k, _p(v))
        a.rgb_boundary_threshold, a.weight_rgb, a.weight_depth = float(rgb_thr), float(w_rgb), float(w_d)
        a.weight_by_opacity, a.depth_needs_opaque = int(weight_by_opacity), int(depth_needs_opaque)
        a.scratch, a.scratch_bytes, a.loss = _p(scratch), scratch.numel(), _p(loss)
        _lib.check(L.lvdgs_photometric_loss_forward(C.byref(a), _raw_stream(dev)),
                   "lvdgs_photometric_loss_forward")
        ctx.cfg = (float(rgb_thr), float(w_rgb), float(w_d), int(weight_by_opacity), int(depth_needs_opaque), H, W)
        ctx.save_for_backward(*[v for v in t.values() if v is not None])
        ctx.present = [k for k, v in t.items() if v is not None]
        ctx.shapes = {k: (None if v is None else v.shape) for k, v in
                      dict(depth=depth, opacity=opacity, exposure_a=exposure_a, exposure_b=exposure_b).items()}
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        L = _lib.lib()
        rgb_thr, w_rgb, w_d, wbo, dno, H, W = ctx.cfg
        t = dict(zip(ctx.present, ctx.saved_tensors))
        dev = grad_loss.device
        a = _lib.LossArgs()
        a.width, a.height = W, H
        for k, v in t.items():
            setattr(a, k, _p(v))
        a.rgb_boundary_threshold, a.weight_rgb, a.weight_depth = rgb_thr, w_rgb, w_d
        a.weight_by_opacity, a.depth_needs_opaque = wbo, dno
        scratch = torch.empty(int(L.lvdgs_loss_scratch_bytes(W, H)), dtype=torch.uint8, device=dev)
        g = grad_loss.detach().to(torch.float32).contiguous()
        e = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        d_image = e(3, H, W)
        d_depth = e(*ctx.shapes["depth"]) if ("depth" in t and "gt_depth" in t and ctx.needs_input_grad[1]) else None
        d_opac = e(*ctx.shapes["opacity"]) if ("opacity" in t and ctx.needs_input_grad[2]) else None
        d_a = e(*ctx.shapes["exposure_a"]) if "exposure_a" in t else None
        d_b = e(*ctx.shapes["exposure_b"]) if "exposure_b" in t else None
        a.scratch, a.scratch_bytes, a.grad_loss = _p(scratch), scratch.numel(), _p(g)
        a.d_image, a.d_depth, a.d_opacity, a.d_exposure_a, a.d_exposure_b = _p(d_image), _p(d_depth), _p(d_opac), _p(d_a), _p(d_b)
        _lib.check(L.lvdgs_photometric_loss_backward(C.byref(a), _raw_stream(dev)),
                   "lvdgs_photometric_loss_backward")
        # no depth term: the depth gradient is None (autograd's zero), not a zero-filled image
        if d_opac is not None and not wbo:
            d_opac = None
        return d_image, d_depth, d_opac, d_a, d_b, None, None, None, None, None, None, None, None


def photometric_loss(image, gt_image, *, depth=None, opacity=None, exposure_a=None, exposure_b=None, gt_depth=None,
                     grad_mask=None, rgb_boundary_threshold=0.01, weight_rgb=1.0, weight_depth=0.0,
                     weight_by_opacity=False, depth_needs_opaque=False):
    """See include/lvdgs.h (lvdgs_loss_args).  All tensors on the same GPU; returns a 0-dim tensor."""
    if grad_mask is not None:
        gm = grad_mask.reshape(-1)
        gm = gm.view(torch.uint8) if gm.dtype == torch.bool else gm.ne(0).view(torch.uint8)
        grad_mask = _aligned(gm.contiguous())
    if gt_depth is None or depth is None:
        weight_depth, gt_depth = 0.0, None
    return _Photometric.apply(image, depth, opacity, exposure_a, exposure_b, gt_image, gt_depth, grad_mask,
                              rgb_boundary_threshold, weight_rgb, weight_depth, weight_by_opacity, depth_needs_opaque)
