"""The SSIM / L1 oracle: convolution form against the loop form, and the properties the definition implies."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import loss_oracle as lo  # noqa: E402


def _pair(C, H, W, seed, noise=0.1):
    g = torch.Generator().manual_seed(seed)
    a = torch.rand(C, H, W, generator=g)
    b = (a + noise * torch.randn(C, H, W, generator=g)).clamp(0, 1)
    return a, b


def test_convolution_form_matches_loops():
    for (C, H, W, seed) in [(3, 13, 17, 0), (1, 7, 30, 1), (2, 24, 9, 2)]:  # smaller than the window too
        a, b = _pair(C, H, W, seed)
        assert abs(float(lo.ssim(a, b)) - lo.ssim_direct(a.numpy(), b.numpy())) < 1e-12


def test_identity_symmetry_and_range():
    a, b = _pair(3, 40, 56, 3, noise=0.3)
    assert abs(float(lo.ssim(a, a)) - 1.0) < 1e-12
    assert abs(float(lo.ssim(a, b)) - float(lo.ssim(b, a))) < 1e-14
    assert -1.0 <= float(lo.ssim(a, b)) < 1.0
    assert float(lo.ssim(a, b)) < float(lo.ssim(a, (a + b) / 2))


def test_window_is_the_published_one():
    w = lo.window_1d().numpy()
    assert w.shape == (11,) and abs(w.sum() - 1.0) < 1e-6 and np.allclose(w, w[::-1])
    assert abs(w[5] / w[4] - np.exp(1.0 / 4.5)) < 1e-6


def test_masked_combination_matches_the_backend_lines():
    """utils/slam_backend.py:199-215 restated with index assignment, against the oracle's where() form."""
    a, b = _pair(3, 20, 28, 4)
    mask = torch.rand(20, 28, generator=torch.Generator().manual_seed(5)) > 0.3
    bg = torch.tensor([0.2, 0.5, 0.9])
    ma, mb = a.clone().double(), b.clone().double()
    for c in range(3):
        ma[c][~mask] = bg[c].double()
        mb[c][~mask] = bg[c].double()
    lam = 0.2
    want = (1 - lam) * (ma - mb).abs().mean() + lam * (1 - lo.ssim(ma, mb))
    assert abs(float(lo.l1_dssim_loss(a, b, lam, mask, bg)) - float(want)) < 1e-14


def test_gradient_of_the_oracle_against_finite_differences():
    a, b = _pair(1, 12, 14, 6)
    a = a.double().requires_grad_(True)
    lo.ssim(a, b).backward()
    eps = 1e-6
    for (y, x) in [(0, 0), (5, 7), (11, 13), (3, 1)]:
        ap, am = a.detach().clone(), a.detach().clone()
        ap[0, y, x] += eps
        am[0, y, x] -= eps
        fd = (float(lo.ssim(ap, b)) - float(lo.ssim(am, b))) / (2 * eps)
        assert abs(fd - float(a.grad[0, y, x])) < 1e-7


def test_masked_depth_term_against_a_pixel_loop():
    """utils/slam_backend.py:216-261: mean |depth - mono| over static & mono > 0 & depth > 0, written as a loop."""
    g = torch.Generator().manual_seed(9)
    H, W = 9, 13
    depth = torch.rand(H, W, generator=g) * 5
    depth[torch.rand(H, W, generator=g) < 0.2] = 0.0
    mono = torch.rand(H, W, generator=g) * 5 - 0.5
    mask = torch.rand(H, W, generator=g) < 0.6
    total, n = 0.0, 0
    for y in range(H):
        for x in range(W):
            if mask[y, x] and mono[y, x] > 0 and depth[y, x] > 0:
                total += abs(float(depth[y, x]) - float(mono[y, x]))
                n += 1
    got, cnt = lo.masked_depth_l1(depth[None], mono[..., None], mask[None])
    assert cnt == n and abs(float(got) - total / n) < 1e-12
    got, cnt = lo.masked_depth_l1(depth, mono, torch.zeros(H, W, dtype=torch.bool))
    assert cnt == 0 and float(got) == 0.0
    # cropped to the common window
    got, cnt = lo.masked_depth_l1(depth, mono[:5], mask[:, :7])
    ref = (depth[:5, :7] - mono[:5, :7]).abs()[mask[:5, :7] & (mono[:5, :7] > 0) & (depth[:5, :7] > 0)]
    assert cnt == ref.numel() and abs(float(got) - float(ref.double().mean())) < 1e-6
