"""Fused L1 + SSIM kernel against the float64 oracle: values, gradients, masks, ragged and full sizes."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
pytestmark = pytest.mark.gpu

REL = 1e-4  # float tolerance of the north star


def _pair(C, H, W, seed, noise=0.1, batch=None):
    g = torch.Generator().manual_seed(seed)
    shape = (C, H, W) if batch is None else (batch, C, H, W)
    a = torch.rand(*shape, generator=g)
    b = (a + noise * torch.randn(*shape, generator=g)).clamp(0, 1)
    return a, b


def _close(got, want, what):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    scale = max(np.abs(want).max(), 1e-30)
    err = np.abs(got - want).max() / scale
    assert err < REL, f"{what}: max error {err:.3e} of the largest magnitude"


@pytest.mark.parametrize("shape", [(3, 37, 53), (1, 8, 8), (3, 64, 96), (3, 370, 1226), (2, 33, 31)])
def test_ssim_value_and_gradient(shape):
    import loss_oracle as lo
    from lvdgs.loss_utils import ssim
    a, b = _pair(*shape, seed=sum(shape))
    ad = a.double().requires_grad_(True)
    want = lo.ssim(ad, b)
    want.backward()
    ag = a.cuda().requires_grad_(True)
    got = ssim(ag, b.cuda())
    got.backward()
    assert abs(float(got.detach()) - float(want.detach())) < 2e-6
    _close(ag.grad.cpu().numpy(), ad.grad.numpy(), "d ssim / d img1")


def test_batched_input_and_no_grad_path():
    import loss_oracle as lo
    from lvdgs.loss_utils import ssim
    a, b = _pair(3, 45, 70, seed=9, batch=2)
    with torch.no_grad():
        got = ssim(a.cuda(), b.cuda())
    want = lo.ssim_map(a.double(), b.double()).mean()
    assert abs(float(got.detach()) - float(want.detach())) < 2e-6
    assert abs(float(ssim(a.cuda(), a.cuda())) - 1.0) < 1e-6


@pytest.mark.parametrize("masked", [False, True])
def test_l1_dssim_combination(masked):
    """(1 - l) L1 + l (1 - SSIM) with the dynamic pixels overwritten (utils/slam_backend.py:199-215)."""
    import loss_oracle as lo
    from lvdgs.loss_utils import l1_dssim_loss
    H, W, lam = 90, 130, 0.2
    a, b = _pair(3, H, W, seed=11)
    mask = (torch.rand(H, W, generator=torch.Generator().manual_seed(12)) > 0.25) if masked else None
    bg = torch.tensor([0.1, 0.6, 0.3])
    ad = a.double().requires_grad_(True)
    want = lo.l1_dssim_loss(ad, b, lam, mask, bg)
    (want * 1.7).backward()
    ag = a.cuda().requires_grad_(True)
    got = l1_dssim_loss(ag, b.cuda(), lam, None if mask is None else mask.cuda(), bg.cuda())
    (got * 1.7).backward()
    assert abs(float(got.detach()) - float(want.detach())) < 2e-6
    _close(ag.grad.cpu().numpy(), ad.grad.numpy(), "d loss / d image")
    if masked:
        assert float(ag.grad[:, ~mask.cuda()].abs().max()) == 0.0


def test_full_hd_properties_and_determinism():
    """1920x1080: identity gives exactly-one SSIM and zero L1; two runs agree bit for bit."""
    from lvdgs.loss_utils import l1_dssim_loss, ssim
    a, b = _pair(3, 1080, 1920, seed=13)
    a, b = a.cuda(), b.cuda()
    assert abs(float(ssim(a, a)) - 1.0) < 1e-6
    x = a.clone().requires_grad_(True)
    l1 = l1_dssim_loss(x, b, 0.2)
    l1.backward()
    y = a.clone().requires_grad_(True)
    l2 = l1_dssim_loss(y, b, 0.2)
    l2.backward()
    assert float(l1) == float(l2) and torch.equal(x.grad, y.grad)
    # SSIM is symmetric in its arguments
    assert abs(float(ssim(a, b)) - float(ssim(b, a))) < 1e-6


def test_cpu_tensors_are_refused():
    from lvdgs._lib import LvdgsError
    from lvdgs.loss_utils import ssim
    with pytest.raises(LvdgsError):
        ssim(torch.rand(3, 16, 16), torch.rand(3, 16, 16))


def test_frame_metrics_of_the_eval_harness():
    """utils/eval_utils_0806.py:231-306: PSNR on non-black pixels, full-frame SSIM, static-region variants."""
    import loss_oracle as lo
    from lvdgs.eval_utils import frame_metrics
    H, W = 70, 110
    a, b = _pair(3, H, W, seed=21)
    b[:, :5] = 0.0  # black border of the ground truth
    static = torch.rand(H, W, generator=torch.Generator().manual_seed(22)) > 0.3
    bg = torch.tensor([0.0, 0.0, 0.0])
    got = frame_metrics(a.cuda() * 1.2 - 0.1, b.cuda(), static.cuda(), bg.cuda())
    img = (a * 1.2 - 0.1).clamp(0, 1).double()
    gt = b.double()
    basic = gt > 0
    psnr = lambda x, y: float(20 * torch.log10(1.0 / torch.sqrt(((x - y) ** 2).mean())))
    assert abs(got["psnr"] - psnr(img[basic], gt[basic])) < 1e-3
    assert abs(got["ssim"] - float(lo.ssim(img, gt))) < 2e-6
    keep = basic & static[None].expand(3, -1, -1)
    assert abs(got["psnr_static"] - psnr(img[keep], gt[keep])) < 1e-3
    zero = torch.zeros_like(img)
    assert abs(got["ssim_static"] - float(lo.ssim(torch.where(keep, img, zero), torch.where(keep, gt, zero)))) < 2e-6
    assert abs(got["static_ratio"] - float(keep.float().mean())) < 1e-6
    plain = frame_metrics(a.cuda(), b.cuda())
    assert plain["psnr_static"] == plain["psnr"] and plain["ssim_static"] == plain["ssim"]


# ---- depth term of the static-mask mapping loss (utils/slam_backend.py:216-261) ----------------------------------
def _depth_case(H, W, seed, frac_static=0.7, hole=0.15):
    g = torch.Generator().manual_seed(seed)
    depth = torch.rand(H, W, generator=g) * 30 + 0.5
    depth[torch.rand(H, W, generator=g) < hole] = 0.0             # un-rendered pixels (depth 0) are excluded
    mono = depth + torch.randn(H, W, generator=g) * 0.7
    mono[torch.rand(H, W, generator=g) < hole] = 0.0              # invalid mono depth is excluded
    mono[torch.rand(H, W, generator=g) < 0.02] = -1.0
    mask = torch.rand(H, W, generator=g) < frac_static
    return depth, mono, mask


@pytest.mark.parametrize("H,W", [(370, 1226), (1, 1), (37, 53), (64, 96), (7, 3)])
def test_masked_depth_l1_value_count_and_gradient(H, W):
    import loss_oracle as lo
    from lvdgs.loss_utils import masked_depth_l1
    depth, mono, mask = _depth_case(H, W, seed=H * 7 + W)
    dd = depth.double().requires_grad_(True)
    want, n = lo.masked_depth_l1(dd, mono, mask)
    want.backward()
    dg = depth.cuda().requires_grad_(True)
    got, cnt = masked_depth_l1(dg, mono.cuda(), mask.cuda(), return_count=True)
    (3.0 * got).backward()                                          # a non-unit upstream gradient
    assert int(cnt) == n
    assert abs(float(got) - float(want)) <= 1e-5 * max(abs(float(want)), 1e-6)
    # sign(D - Z) / |M| on the mask, exactly zero elsewhere
    np.testing.assert_allclose(dg.grad.cpu().numpy(), 3.0 * dd.grad.numpy(), rtol=1e-6, atol=0)


def test_masked_depth_l1_empty_mask_shapes_and_crop():
    import loss_oracle as lo
    from lvdgs.loss_utils import masked_depth_l1
    H, W = 40, 56
    depth, mono, mask = _depth_case(H, W, seed=5)
    # nothing qualifies: the reference adds no term (slam_backend.py:250) -> value 0, zero gradient, count 0
    dg = depth.cuda().requires_grad_(True)
    got, cnt = masked_depth_l1(dg, mono.cuda(), torch.zeros(H, W, dtype=torch.bool).cuda(), return_count=True)
    got.backward()
    assert float(got) == 0.0 and int(cnt) == 0 and not dg.grad.any()
    got = masked_depth_l1(depth.cuda(), -mono.abs().cuda(), mask.cuda())
    assert float(got) == 0.0
    # (1,H,W) render depth, (H,W,1) mono depth as numpy (the reference calls torch.from_numpy on it), (1,H,W) mask
    want, n = lo.masked_depth_l1(depth[None], mono[..., None], mask[None])
    got, cnt = masked_depth_l1(depth[None].cuda(), mono[..., None].numpy(), mask[None].cuda(), return_count=True)
    assert int(cnt) == n and abs(float(got) - float(want)) <= 1e-5 * float(want)
    # unequal sizes are cropped to the common top-left window (slam_backend.py:240-246)
    want, n = lo.masked_depth_l1(depth, mono[:-3, :-5], mask[:-1])
    x = depth.cuda().requires_grad_(True)
    got, cnt = masked_depth_l1(x, mono[:-3, :-5].cuda(), mask[:-1].cuda(), return_count=True)
    got.backward()
    assert int(cnt) == n and abs(float(got) - float(want)) <= 1e-5 * float(want)
    assert x.grad.shape == (H, W) and not x.grad[-3:].any() and not x.grad[:, -5:].any()
    # no mask: every pixel counts as static
    want, n = lo.masked_depth_l1(depth, mono, torch.ones(H, W, dtype=torch.bool))
    got, cnt = masked_depth_l1(depth.cuda(), mono.cuda(), None, return_count=True)
    assert int(cnt) == n and abs(float(got) - float(want)) <= 1e-5 * float(want)


def test_masked_mapping_loss_is_the_whole_static_mask_branch():
    """(1 - l) L1 + l (1 - SSIM) on the background-overwritten images + depth_lambda * masked depth L1, value and the
    gradients w.r.t. the rendered image AND depth, at KITTI-07's frame size (utils/slam_backend.py:199-261,
    lambda_dssim 0.2 configs/mono/KITTI/base_config.yaml:70, depth_lambda 0.1 slam_backend.py:253)."""
    from types import SimpleNamespace
    import loss_oracle as lo
    from lvdgs.loss_utils import masked_mapping_loss
    H, W = 370, 1226
    a, b = _pair(3, H, W, seed=31)
    depth, mono, mask = _depth_case(H, W, seed=32)
    bg = torch.tensor([0.0, 0.0, 0.0])
    ad, dd = a.double().requires_grad_(True), depth.double().requires_grad_(True)
    want = lo.masked_mapping_loss(ad, dd[None], b, mono, mask, bg, 0.2, 0.1)
    want.backward()
    vp = SimpleNamespace(original_image=b.cuda(), static_mask=mask.cuda(), mono_depth=mono.numpy())
    ag, dg = a.cuda().requires_grad_(True), depth[None].cuda().requires_grad_(True)
    got = masked_mapping_loss(ag, dg, vp, bg.cuda(), 0.2, 0.1)
    got.backward()
    assert abs(float(got) - float(want)) < 3e-6
    _close(ag.grad.cpu().numpy(), ad.grad.numpy(), "d loss / d image")
    np.testing.assert_allclose(dg.grad[0].cpu().numpy(), dd.grad.numpy(), rtol=1e-6, atol=0)
    # a keyframe without mono depth gets the photometric part only
    vp2 = SimpleNamespace(original_image=b.cuda(), static_mask=mask.cuda(), mono_depth=None)
    assert abs(float(masked_mapping_loss(a.cuda(), depth[None].cuda(), vp2, bg.cuda(), 0.2)) -
               float(lo.l1_dssim_loss(a, b, 0.2, mask, bg))) < 3e-6
