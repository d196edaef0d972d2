"""``cuRoPE2D``: in-place 2-D rotary position embedding of ViT tokens.

Counterpart of croco's ``models/curope`` extension that MASt3R uses (reference README.md:49-50; the
croco / dust3r trees are absent from the checkout).  Interface as published: the module takes tokens
``(B, H, N, D)`` and integer positions ``(B, N, 2)`` (y, x), rotates the first half of ``D`` by y and
the second by x with frequencies ``base ** (-i / (D/4))``, in place, and the backward pass is the
inverse rotation of the incoming gradient.
"""
import ctypes as C

import torch
from torch import nn

from . import _lib


_DTYPES = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}  # LVDGS_F32 / LVDGS_F16 / LVDGS_BF16


def rope_2d(tokens: torch.Tensor, positions: torch.Tensor, base: float, fwd: float) -> None:
    """In place on ``tokens`` (B, N, H, D): float32, float16 or bfloat16 (MASt3R runs under autocast), any strides over
    (B, N, H) as long as the D axis is contiguous -- so the transposed view of a (B, H, N, D) tensor is rotated where it
    lies.  Arithmetic is float32; half-precision elements are rounded once when stored."""
    if tokens.device.type != "cuda":
        raise _lib.LvdgsError("rope_2d needs GPU tensors (there is no CPU path)")
    B, N, H, D = tokens.shape
    assert positions.shape == (B, N, 2) and D % 4 == 0
    if tokens.dtype not in _DTYPES:
        raise TypeError(f"rope_2d: float32, float16 or bfloat16 tokens, got {tokens.dtype}")
    sb, sn, sh, sd = tokens.stride()
    if tokens.numel() and sd != 1 and D > 1:
        raise ValueError("rope_2d: the last (feature) axis of the tokens must be contiguous")
    pos = positions.to(device=tokens.device, dtype=torch.int64).contiguous()
    with _lib.on_device(tokens.device):
        st = _lib.lib().lvdgs_rope2d_strided(C.c_void_p(tokens.data_ptr()), _DTYPES[tokens.dtype], C.c_void_p(pos.data_ptr()),
                                             B, N, H, D, sb, sn, sh, float(base), float(fwd), _lib.raw_stream(tokens.device))
    _lib.check(st, "lvdgs_rope2d_strided")


class cuRoPE2D_func(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tokens, positions, base, F0=1.0):
        ctx.save_for_backward(positions)
        ctx.saved_base, ctx.saved_F0 = base, F0
        rope_2d(tokens, positions, base, F0)
        ctx.mark_dirty(tokens)
        return tokens

    @staticmethod
    def backward(ctx, grad_res):
        (positions,) = ctx.saved_tensors
        if grad_res.stride(-1) != 1:
            grad_res = grad_res.contiguous()
        rope_2d(grad_res, positions, ctx.saved_base, -ctx.saved_F0)
        return grad_res, None, None, None


class cuRoPE2D(nn.Module):
    def __init__(self, freq=100.0, F0=1.0):
        super().__init__()
        self.base = freq
        self.F0 = F0

    def forward(self, tokens, positions):
        """tokens: (B, H, N, D); rotated in place through the (B, N, H, D) view -- no transposed copy -- and returned."""
        cuRoPE2D_func.apply(tokens.transpose(1, 2), positions, self.base, self.F0)
        return tokens
