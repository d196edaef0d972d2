"""Fused HIP photometric loss against the PyTorch formulas that are pinned to the reference's golden
vectors (tests/test_host_golden.py): same value, same gradients w.r.t. image, depth, opacity and the
exposure parameters, for every tracking / mapping branch."""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cfg(monocular, alpha=0.98):
    return {"Training": {"monocular": monocular, "rgb_boundary_threshold": 0.01, "alpha": alpha}, "Dataset": {"depth_loss": True}}


def _inputs(H, W, seed, mask_float=False):
    g = torch.Generator().manual_seed(seed)
    image = torch.rand(3, H, W, generator=g)
    depth = torch.rand(1, H, W, generator=g) * 10
    opacity = torch.rand(1, H, W, generator=g)
    gt = torch.rand(3, H, W, generator=g)
    gt[:, :5, :7] = 0.0
    image[:, 10:12, 3:9] = gt[:, 10:12, 3:9]  # exact zeros in the residual (sign(0) = 0)
    gm = torch.rand(1, H, W, generator=g) > 0.4
    mono = (torch.rand(H, W, generator=g) * 10).numpy().astype(np.float32)
    mono[3:9, 20:30] = 0.0
    vp = types.SimpleNamespace(original_image=gt.cuda(), grad_mask=(gm.float() if mask_float else gm).cuda(), mono_depth=mono,
                               exposure_a=torch.nn.Parameter(torch.tensor([0.11]).cuda()),
                               exposure_b=torch.nn.Parameter(torch.tensor([-0.03]).cuda()))
    return image, depth, opacity, vp


def _run(fn, fused, image, depth, opacity, vp):
    from lvdgs import slam_utils
    slam_utils.USE_FUSED_LOSS = fused
    try:
        leaves = [t.cuda().clone().requires_grad_(True) for t in (image, depth, opacity)]
        vp.exposure_a.grad = vp.exposure_b.grad = None
        loss = fn(*leaves)
        (loss * 1.7).backward()
        z = lambda t: torch.zeros_like(t) if t.grad is None else t.grad
        return loss.detach().cpu().numpy(), [z(t).cpu().numpy() for t in leaves] + [z(vp.exposure_a).cpu().numpy(), z(vp.exposure_b).cpu().numpy()]
    finally:
        slam_utils.USE_FUSED_LOSS = True


@pytest.mark.parametrize("case", ["track_mono", "track_rgbd", "map_rgbd", "map_rgb", "map_init", "track_mono_floatmask"])
@pytest.mark.parametrize("H,W", [(37, 53), (370, 1226)])
def test_fused_loss_matches_torch_formulas(case, H, W):
    from lvdgs import slam_utils
    image, depth, opacity, vp = _inputs(H, W, seed=len(case) + H, mask_float=case.endswith("floatmask"))
    fns = {
        "track_mono": lambda i, d, o: slam_utils.get_loss_tracking(_cfg(True), i, d, o, vp),
        "track_mono_floatmask": lambda i, d, o: slam_utils.get_loss_tracking(_cfg(True), i, d, o, vp),
        "track_rgbd": lambda i, d, o: slam_utils.get_loss_tracking(_cfg(False, 0.9), i, d, o, vp),
        "map_rgbd": lambda i, d, o: slam_utils.get_loss_mapping(_cfg(True), i, vp, depth=d),
        "map_rgb": lambda i, d, o: slam_utils.get_loss_mapping(_cfg(True), i, vp, depth=d, monodepth=False),
        "map_init": lambda i, d, o: slam_utils.get_loss_mapping(_cfg(True), i, vp, depth=d, initialization=True),
    }
    l_f, g_f = _run(fns[case], True, image, depth, opacity, vp)
    l_t, g_t = _run(fns[case], False, image, depth, opacity, vp)
    np.testing.assert_allclose(l_f, l_t, rtol=2e-5)
    for a, b, name in zip(g_f, g_t, ("image", "depth", "opacity", "a", "b")):
        np.testing.assert_allclose(a, b, rtol=2e-4, atol=1e-9 + 2e-5 * np.abs(b).max(), err_msg=name)


def test_value_and_grad_in_one_pass_equals_the_two_passes():
    """lvdgs_photometric_loss_value_and_grad (what the tracking session calls) against forward + backward: loss, image /
    depth / opacity / exposure gradients, with and without a device scalar for d objective / d loss."""
    import ctypes as C
    from lvdgs import _lib
    H, W = 67, 130   # P = 8710, not a multiple of 4: the scalar path; and a multiple-of-4 size below
    for (H, W) in ((67, 130), (64, 96)):
        g = torch.Generator().manual_seed(H)
        dev = torch.device("cuda", 0)
        mk = lambda *s: torch.rand(*s, generator=g).to(dev)
        image, gt, depth, gtd, opac = mk(3, H, W), mk(3, H, W), mk(1, H, W) * 5, mk(1, H, W) * 5, mk(1, H, W)
        gm = (torch.rand(H * W, generator=g) > 0.4).to(torch.uint8).to(dev)
        ea, eb = torch.tensor([0.1], device=dev), torch.tensor([-0.02], device=dev)
        L = _lib.lib()
        P = lambda t: C.c_void_p(t.data_ptr())
        outs = {}
        for mode, scale in (("two", None), ("one", None), ("one_scaled", 0.37), ("two_scaled", 0.37)):
            a = _lib.LossArgs()
            a.width, a.height = W, H
            a.image, a.depth, a.opacity, a.gt_image, a.gt_depth, a.grad_mask = P(image), P(depth), P(opac), P(gt), P(gtd), P(gm)
            a.exposure_a, a.exposure_b = P(ea), P(eb)
            a.rgb_boundary_threshold, a.weight_rgb, a.weight_depth, a.weight_by_opacity, a.depth_needs_opaque = 0.01, 0.9, 0.1, 1, 1
            scratch = torch.empty(int(L.lvdgs_loss_scratch_bytes(W, H)), dtype=torch.uint8, device=dev)
            loss, gl = torch.zeros((), device=dev), torch.full((), 1.0 if scale is None else scale, device=dev)
            d = [torch.empty(3, H, W, device=dev), torch.empty(1, H, W, device=dev), torch.empty(1, H, W, device=dev),
                 torch.empty(1, device=dev), torch.empty(1, device=dev)]
            a.scratch, a.scratch_bytes, a.loss = P(scratch), scratch.numel(), P(loss)
            a.d_image, a.d_depth, a.d_opacity, a.d_exposure_a, a.d_exposure_b = (P(t) for t in d)
            stream = _lib.raw_stream(dev)
            if mode.startswith("two"):
                a.grad_loss = P(gl)
                _lib.check(L.lvdgs_photometric_loss_forward(C.byref(a), stream), "fwd")
                _lib.check(L.lvdgs_photometric_loss_backward(C.byref(a), stream), "bwd")
            else:
                a.grad_loss = None if scale is None else P(gl)
                _lib.check(L.lvdgs_photometric_loss_value_and_grad(C.byref(a), stream), "both")
            torch.cuda.synchronize()
            outs[mode] = [float(loss)] + [t.cpu().numpy() for t in d]
        for x, y in (("one", "two"), ("one_scaled", "two_scaled")):
            assert outs[x][0] == outs[y][0]
            for u, v in zip(outs[x][1:], outs[y][1:]):
                np.testing.assert_array_equal(u, v)
