"""TEST INFRASTRUCTURE -- CPU restatement of the L1 / SSIM image losses (float64).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this file; the product path
(lvd_gs-slam_amd/) never does.

PARITY UNPINNED: ``gaussian_splatting.utils.loss_utils`` is absent from the reference checkout
(imported at utils/slam_backend.py:11, utils/eval_utils_0806.py:28), so there is nothing to import or
run.  This follows the published definition used by 3DGS / MonoGS: 11-tap Gaussian window with sigma
1.5 (normalised, built in float32), 2-D window = outer product, zero padding of 5, one group per
channel, C1 = 0.01^2, C2 = 0.03^2, mean over every pixel and channel.  ``ssim_direct`` is a second,
loop-based statement of the same formula that pins the convolution form on small images.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

C1, C2 = 0.01 ** 2, 0.03 ** 2


def window_1d(size=11, sigma=1.5):
    g = torch.tensor([math.exp(-((x - size // 2) ** 2) / float(2 * sigma ** 2)) for x in range(size)], dtype=torch.float32)
    return (g / g.sum()).double()


def ssim_map(img1, img2, size=11):
    """(B,C,H,W) float64 -> SSIM map (B,C,H,W); differentiable."""
    C = img1.shape[1]
    w1 = window_1d(size)
    w2 = (w1[:, None] * w1[None, :])[None, None].expand(C, 1, size, size).contiguous()
    conv = lambda t: F.conv2d(t, w2, padding=size // 2, groups=C)
    mu1, mu2 = conv(img1), conv(img2)
    s1 = conv(img1 * img1) - mu1 * mu1
    s2 = conv(img2 * img2) - mu2 * mu2
    s12 = conv(img1 * img2) - mu1 * mu2
    return ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s1 + s2 + C2))


def ssim(img1, img2):
    a = img1.double()
    b = img2.double()
    if a.dim() == 3:
        a, b = a[None], b[None]
    return ssim_map(a, b).mean()


def l1_loss(a, b):
    return (a.double() - b.double()).abs().mean()


def masked_pair(image, gt, static_mask, background):
    """utils/slam_backend.py:199-209: clones with background[c] written into the dynamic pixels."""
    a, b = image.double().clone(), gt.double().clone()
    if static_mask is not None:
        dyn = ~static_mask.bool()
        for c in range(a.shape[0]):
            a[c] = torch.where(dyn, background[c].double(), a[c])
            b[c] = torch.where(dyn, background[c].double(), b[c])
    return a, b


def l1_dssim_loss(image, gt, lambda_dssim, static_mask=None, background=None):
    a, b = masked_pair(image, gt, static_mask, background)
    return (1.0 - lambda_dssim) * l1_loss(a, b) + lambda_dssim * (1.0 - ssim(a, b))


def masked_depth_l1(depth, mono_depth, static_mask):
    """utils/slam_backend.py:216-261, statement for statement (float64): squeeze (1,H,W)/(H,W,1) to (H,W) (:223-236),
    crop all three to the common top-left window (:240-246), depth_mask = static & (mono > 0) & (depth > 0) (:249),
    mean of |depth - mono| over the mask if it has any pixel (:250-253), else no term."""
    def squeeze(t):
        if t.dim() == 3 and t.shape[0] == 1:
            return t.squeeze(0)
        if t.dim() == 3 and t.shape[-1] == 1:
            return t.squeeze(-1)
        return t
    depth, mono_depth = squeeze(depth.double()), squeeze(mono_depth.double())
    if static_mask.dim() == 3:
        static_mask = static_mask.squeeze(-1) if static_mask.shape[-1] == 1 else static_mask.squeeze(0)
    assert depth.dim() == 2 and mono_depth.dim() == 2 and static_mask.dim() == 2
    min_h = min(depth.shape[0], mono_depth.shape[0], static_mask.shape[0])
    min_w = min(depth.shape[1], mono_depth.shape[1], static_mask.shape[1])
    depth, mono_depth, static_mask = depth[:min_h, :min_w], mono_depth[:min_h, :min_w], static_mask[:min_h, :min_w]
    depth_mask = static_mask.bool() & (mono_depth > 0) & (depth > 0)
    if depth_mask.any():
        return torch.abs(depth[depth_mask] - mono_depth[depth_mask]).mean(), int(depth_mask.sum())
    return depth.sum() * 0.0, 0


def masked_mapping_loss(image, depth, gt, mono_depth, static_mask, background, lambda_dssim, depth_lambda=0.1):
    """The whole static-mask branch of the mapping loss, utils/slam_backend.py:199-261."""
    loss = l1_dssim_loss(image, gt, lambda_dssim, static_mask, background)
    if depth is not None and mono_depth is not None:
        loss = loss + depth_lambda * masked_depth_l1(depth, mono_depth, static_mask)[0]
    return loss


def ssim_direct(img1, img2, size=11):
    """Plain loops over pixels and taps (numpy float64); small images only."""
    a, b = np.asarray(img1, np.float64), np.asarray(img2, np.float64)
    C, H, W = a.shape
    w = window_1d(size).numpy()
    r = size // 2
    total = 0.0
    for c in range(C):
        for y in range(H):
            for x in range(W):
                m1 = m2 = e11 = e22 = e12 = 0.0
                for dy in range(-r, r + 1):
                    yy = y + dy
                    if yy < 0 or yy >= H:
                        continue
                    for dx in range(-r, r + 1):
                        xx = x + dx
                        if xx < 0 or xx >= W:
                            continue
                        k = w[dy + r] * w[dx + r]
                        p, q = a[c, yy, xx], b[c, yy, xx]
                        m1 += k * p
                        m2 += k * q
                        e11 += k * p * p
                        e22 += k * q * q
                        e12 += k * p * q
                s1, s2, s12 = e11 - m1 * m1, e22 - m2 * m2, e12 - m1 * m2
                total += ((2 * m1 * m2 + C1) * (2 * s12 + C2)) / ((m1 * m1 + m2 * m2 + C1) * (s1 + s2 + C2))
    return total / (C * H * W)
