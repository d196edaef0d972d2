"""Times the fused L1 + SSIM loss (value + gradient) at 1920x1080 against the same loss written with
PyTorch convolutions on the same GPU.  Run on the GPU box: python tools/ssim_bench.py"""
import math
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import lvdgs  # noqa: E402,F401
from lvdgs import _lib  # noqa: E402
from lvdgs.loss_utils import l1_dssim_loss  # noqa: E402


def torch_l1_dssim(a, b, lam, w2):
    conv = lambda t: F.conv2d(t, w2, padding=5, groups=3)
    a4, b4 = a[None], b[None]
    mu1, mu2 = conv(a4), conv(b4)
    s1, s2, s12 = conv(a4 * a4) - mu1 * mu1, conv(b4 * b4) - mu2 * mu2, conv(a4 * b4) - mu1 * mu2
    m = ((2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4)) / ((mu1 * mu1 + mu2 * mu2 + 1e-4) * (s1 + s2 + 9e-4))
    return (1 - lam) * (a - b).abs().mean() + lam * (1 - m.mean())


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6


def main():
    a = torch.rand(3, 1080, 1920, device="cuda")
    b = torch.rand(3, 1080, 1920, device="cuda")
    x = a.clone().requires_grad_(True)
    g = torch.tensor([math.exp(-((k - 5) ** 2) / 4.5) for k in range(11)], device="cuda")
    g = g / g.sum()
    w2 = (g[:, None] * g[None, :])[None, None].expand(3, 1, 11, 11).contiguous()

    def fused():
        l1_dssim_loss(x, b, 0.2).backward()
        x.grad = None

    def eager():
        torch_l1_dssim(x, b, 0.2, w2).backward()
        x.grad = None

    us_fused, us_torch = timed(fused), timed(eager)
    _lib.profile_enable(True)
    _lib.profile_reset()
    for _ in range(20):
        fused()
    torch.cuda.synchronize()
    kern = {k: round(ms / n * 1000, 1) for k, (n, ms) in _lib.profile_read().items()}
    print({"fused_us": round(us_fused, 1), "torch_us": round(us_torch, 1), "kernels_us": kern})


if __name__ == "__main__":
    main()
