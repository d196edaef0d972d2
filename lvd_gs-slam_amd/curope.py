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


def rope_2d(tokens: torch.Tensor, positions: torch.Tensor, base: float, fwd: float) -> None:
    """In place on ``tokens`` (B, N, H, D) float32 contiguous."""
    if tokens.device.type != "cuda":
        raise _lib.LvdgsError("rope_2d needs GPU tensors (there is no CPU path)")
    B, N, H, D = tokens.shape
    assert positions.shape == (B, N, 2) and D % 4 == 0
    assert tokens.dtype == torch.float32 and tokens.is_contiguous(), "rope_2d: float32 contiguous (B,N,H,D) tokens"
    pos = positions.to(torch.int64).contiguous()
    st = _lib.lib().lvdgs_rope2d(C.c_void_p(tokens.data_ptr()), C.c_void_p(pos.data_ptr()), B, N, H, D, float(base),
                                 float(fwd), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    _lib.check(st, "lvdgs_rope2d")


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
        grad_res = grad_res.contiguous()
        rope_2d(grad_res, positions, ctx.saved_base, -ctx.saved_F0)
        return grad_res, None, None, None


class cuRoPE2D(nn.Module):
    def __init__(self, freq=100.0, F0=1.0):
        super().__init__()
        self.base = freq
        self.F0 = F0

    def forward(self, tokens, positions):
        """tokens: (B, H, N, D); rotated in place (through the (B, N, H, D) view) and returned."""
        t = tokens.transpose(1, 2)
        if not t.is_contiguous():
            tc = t.contiguous()
            cuRoPE2D_func.apply(tc, positions, self.base, self.F0)
            tokens.copy_(tc.transpose(1, 2))
            return tokens
        cuRoPE2D_func.apply(t, positions, self.base, self.F0)
        return tokens
