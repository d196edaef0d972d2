"""The static-mask mapping loss as the mapping loop runs it (reference utils/slam_backend.py:196-261 -- under LVD-GS's default
configuration the loss of EVERY window keyframe: utils/slam_frontend.py:1218,1429-1433):

* lvdgs_masked_loss_batch + lvdgs_backward_masked_loss (one object: L1 + SSIM value and gradient image, the depth term's sum and
  count, the depth term's gradient evaluated per pixel inside the backward blend pass) against the separate launches it replaces
  (lvdgs_ssim_l1 + lvdgs_masked_depth_l1_forward / _backward + lvdgs_backward on gradient images): the gradient images and every
  Gaussian / pose gradient are the same BITS, the loss value agrees to rounding (the depth term's partial sums are taken per
  32x32 tile instead of per 1024 pixels);
* ... against the PyTorch statements of the reference's branch (the loss value; autograd gradient images);
* the window of 8 masked keyframes + 2 unmasked random views with every blend pass and the masked losses in one launch each
  (fast_mapping.MapWindowBatch) against the same window view by view: the same bits after three Adam iterations.
"""
import ctypes as C
import os
import sys

import pytest
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def _scene(workload="tmp_masked_window", n_kf=3, N=30000, W=400, H=240, masked=True):
    import bench
    from lvdgs import synthetic
    synthetic.CONFIGS.setdefault(workload, dict(N=N, W=W, H=H))
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model, cam, g, _ = bench.build_scene(workload, 0, dev)
    backend, window = bench.build_window(workload, n_kf, dev, model, masked=masked)
    return backend, window, dev


def _grads(backend):
    G = backend.gaussians
    return {n: getattr(G, n).grad.detach().clone() for n in ("_xyz", "_features_dc", "_scaling", "_rotation", "_opacity")}


@pytest.mark.parametrize("with_mask,with_depth", [(True, True), (True, False), (False, True)])
def test_masked_loss_route_equals_the_separate_launches_bit_for_bit(with_mask, with_depth):
    from lvdgs.fast_mapping import MapViewPass
    from lvdgs.loss_utils import masked_mapping_loss_and_grads
    backend, window, dev = _scene()
    G = backend.gaussians
    vp = backend.viewpoints[window[0]]
    if not with_mask:
        vp.static_mask = torch.ones_like(vp.static_mask)
    lam, dlam = 0.2, 0.1
    md = vp.mono_depth
    out = {}
    for route in ("separate", "fused"):
        for p in G.parameters():
            p.grad = None
        for n in ("cam_rot_delta", "cam_trans_delta", "exposure_a", "exposure_b"):
            getattr(vp, n).grad = None
        vpass = MapViewPass(dev)
        caught = {}
        if route == "separate":
            vp.mono_depth = md if with_depth else None

            def image_loss(color, depth):
                res = masked_mapping_loss_and_grads(color, depth, vp, backend.background, lam, dlam)
                caught["d_image"], caught["d_depth"] = res[1].clone(), None if res[2] is None else res[2].clone()
                return res
            pkg, loss = vpass.run(backend, vp, image_loss=image_loss)
            vp.mono_depth = md
        else:
            pkg, loss = vpass.run(backend, vp, masked_loss=(lam, dlam if with_depth else None))
            caught["d_image"] = vpass.d_image.clone()
        torch.cuda.synchronize()
        out[route] = dict(loss=float(loss), grads=_grads(backend), tau=torch.cat([vp.cam_trans_delta.grad.flatten(), vp.cam_rot_delta.grad.flatten()]).clone(),
                          m2=pkg["viewspace_points"].grad.clone(), color=pkg["render"].clone(), **caught)
        assert vp.exposure_a.grad is None and vp.exposure_b.grad is None   # this loss does not read the exposure
    a, b = out["separate"], out["fused"]
    assert torch.equal(a["color"], b["color"])
    assert torch.equal(a["d_image"], b["d_image"])
    assert a["d_image"].abs().sum() > 0
    for n in a["grads"]:
        assert torch.equal(a["grads"][n], b["grads"][n]), n
        assert a["grads"][n].abs().sum() > 0, n
    assert torch.equal(a["tau"], b["tau"]) and torch.equal(a["m2"], b["m2"])
    assert abs(a["loss"] - b["loss"]) <= 2e-6 * abs(a["loss"])


def test_masked_loss_value_and_gradient_images_match_the_reference_statements():
    """lvdgs_masked_loss_batch on two views at once against the PyTorch statements of utils/slam_backend.py:199-261 (clone + index
    assignment of the background colour, l1_loss, 1 - ssim as PyTorch convolutions, the masked depth mean) and their autograd."""
    import torch.nn.functional as F
    from lvdgs import _lib
    from lvdgs.gaussian_renderer import render
    from lvdgs.slam_utils import _mono_depth, _static_mask_bytes
    backend, window, dev = _scene(n_kf=2)
    bg = torch.tensor([0.1, 0.25, 0.4], device=dev)
    lam, dlam = 0.2, 0.1
    L = _lib.lib()
    _P = lambda t: None if t is None else C.c_void_p(t.data_ptr())

    def window11(ch):
        g = torch.tensor([-(x - 5) ** 2 / (2 * 1.5 ** 2) for x in range(11)], dtype=torch.float32).exp()
        g = (g / g.sum()).unsqueeze(1)
        return (g @ g.t()).float()[None, None].expand(ch, 1, 11, 11).contiguous().to(dev)

    def ssim_ref(a, b):
        w = window11(3)
        mu1, mu2 = F.conv2d(a[None], w, padding=5, groups=3), F.conv2d(b[None], w, padding=5, groups=3)
        s1 = F.conv2d(a[None] * a[None], w, padding=5, groups=3) - mu1 * mu1
        s2 = F.conv2d(b[None] * b[None], w, padding=5, groups=3) - mu2 * mu2
        s12 = F.conv2d(a[None] * b[None], w, padding=5, groups=3) - mu1 * mu2
        C1, C2 = 0.01 ** 2, 0.03 ** 2
        return (((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s1 + s2 + C2))).mean()

    views, keep, expect = [], [], []
    for kf in window:
        vp = backend.viewpoints[kf]
        with torch.no_grad():
            pkg = render(vp, backend.gaussians, backend.pipeline_params, bg)
        color, depth = pkg["render"].contiguous(), pkg["depth"].contiguous()
        H, W = color.shape[1:]
        # ---- the reference's statements ----
        image = color.clone().requires_grad_(True)
        dd = depth.clone().requires_grad_(True)
        mask = vp.static_mask
        mi, mg = image.clone(), vp.original_image.clone()
        for c in range(3):
            mi[c][~mask] = bg[c]
            mg[c][~mask] = bg[c]
        loss = (1.0 - lam) * torch.abs(mi - mg).mean() + lam * (1.0 - ssim_ref(mi, mg))
        z = torch.from_numpy(vp.mono_depth).to(dev)
        dmask = mask & (z > 0) & (dd[0] > 0)
        loss = loss + dlam * torch.abs(dd[0][dmask] - z[dmask]).mean()
        loss.backward()
        expect.append((float(loss.detach()), image.grad.clone(), dd.grad.clone(), int(dmask.sum())))
        # ---- the kernel's arguments ----
        a = _lib.MaskedLossArgs()
        scratch = torch.empty(int(L.lvdgs_masked_loss_scratch_bytes(W, H)), dtype=torch.uint8, device=dev)
        d_image, out = torch.empty_like(color), torch.zeros(8, device=dev)
        m8, zt = _static_mask_bytes(vp, color), _mono_depth(vp, color).contiguous()
        a.width, a.height = W, H
        a.image, a.gt_image, a.static_mask, a.bg, a.depth, a.gt_depth = _P(color), _P(vp.original_image), _P(m8), _P(bg), _P(depth), _P(zt)
        a.lambda_dssim, a.depth_lambda = lam, dlam
        a.scratch, a.scratch_bytes, a.d_image, a.out = _P(scratch), scratch.numel(), _P(d_image), _P(out)
        views.append(a)
        keep += [color, depth, scratch, d_image, out, m8, zt]
    arr = (C.POINTER(_lib.MaskedLossArgs) * 2)(*[C.pointer(v) for v in views])
    _lib.check(L.lvdgs_masked_loss_batch(arr, 2, _lib.raw_stream(dev)), "lvdgs_masked_loss_batch")
    torch.cuda.synchronize()
    for k in range(2):
        d_image, out = keep[7 * k + 3], keep[7 * k + 4]
        loss, gi, gd, count = expect[k]
        assert abs(float(out[0]) - loss) <= 2e-5 * abs(loss)
        assert int(out[4]) == count and count > 1000
        scale = gi.abs().max()
        assert (d_image - gi).abs().max() <= 2e-4 * scale, float((d_image - gi).abs().max() / scale)
        assert torch.linalg.norm(d_image - gi) <= 6e-5 * torch.linalg.norm(gi)   # (2.0e-5 achieved: f32 convolutions on the other side)
        # (the depth term's gradient is evaluated inside the backward blend pass: dlam * sign / count on M; held against gd by the
        # bit-for-bit test above through the separate launches, whose d_depth image is this formula)
        assert abs(float(gd.abs().max()) - dlam / count) <= 1e-6 * dlam / count


def _run_window(batch, iters, masked=True, workload="tmp_masked_window_batch", n_window=8):
    import bench
    from lvdgs import backend_map, synthetic
    synthetic.CONFIGS.setdefault(workload, dict(N=30000, W=400, H=240))
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model, cam, g, _ = bench.build_scene(workload, 0, dev)
    backend, window = bench.build_window(workload, 12, dev, model, n_window=n_window, masked=masked)
    before = os.environ.get("LVDGS_MAP_BATCH")
    os.environ["LVDGS_MAP_BATCH"] = "1" if batch else "0"
    from lvdgs.fast_mapping import MapViewPass
    calls = []
    run = MapViewPass.run
    MapViewPass.run = lambda self, *a, **k: (calls.append(k.get("masked_loss") is not None) or run(self, *a, **k))
    try:
        st = {}
        for _ in range(iters):
            backend_map.map_window(backend, window, iters=1, stats=st)
    finally:
        MapViewPass.run = run
        if before is None:
            os.environ.pop("LVDGS_MAP_BATCH", None)
        else:
            os.environ["LVDGS_MAP_BATCH"] = before
    torch.cuda.synchronize()
    G = backend.gaussians
    used = getattr(backend, "_lvdgs_window_batch", None) is not None
    params = [p.detach().clone() for p in G.parameters()]
    poses = [torch.cat([vp.cam_rot_delta.detach().flatten(), vp.cam_trans_delta.detach().flatten(), vp.exposure_a.detach().flatten(),
                        vp.exposure_b.detach().flatten(), vp.R.detach().flatten().to(dev), vp.T.detach().flatten().to(dev)]).clone()
             for vp in backend.viewpoints.values()]
    stats = [G.max_radii2D.clone(), G.xyz_gradient_accum.clone(), G.denom.clone()]
    losses = [float(x) for x in st["losses"]]
    vis = {k: v.clone() for k, v in backend.occ_aware_visibility.items()}
    return used, params, poses, stats, losses, vis, calls, backend


def test_window_of_masked_keyframes_in_one_launch_each_is_the_window_view_by_view_bit_for_bit():
    """8 keyframes with a static mask + 2 random older views scored by get_loss_mapping -- the reference's default window."""
    used_b, params_b, poses_b, stats_b, losses_b, vis_b, calls_b, be = _run_window(True, 3)
    used_s, params_s, poses_s, stats_s, losses_s, vis_s, calls_s, _ = _run_window(False, 3)
    assert used_b and not used_s, "the batch path did not run (or ran when switched off)"
    assert calls_b == [] and calls_s == ([True] * 8 + [False] * 2) * 3   # view by view: eight masked, two get_loss_mapping
    for a, b in zip(params_b, params_s):
        assert torch.equal(a, b)
    for a, b in zip(poses_b, poses_s):
        assert torch.equal(a, b)
    for a, b in zip(stats_b, stats_s):
        assert torch.equal(a, b)
    assert losses_b == losses_s and all(0.0 < v < 10.0 for v in losses_b)
    for k in vis_b:
        assert torch.equal(vis_b[k], vis_s[k])
    # the masked keyframes' exposure parameters get no gradient and never move (the branch does not read them); the pose deltas do
    window = be.current_window
    for kf in window:
        vp = be.viewpoints[kf]
        assert float(vp.exposure_a.detach()) == 0.0 and float(vp.exposure_b.detach()) == 0.0


def test_masked_window_differs_from_the_unmasked_one_and_ignores_the_dynamic_pixels():
    """The mask matters (the two windows' maps differ after an iteration), and what lies under it does not: scribbling over the
    dynamic pixels of every keyframe's target image and mono depth leaves the masked window's result unchanged bit for bit."""
    _, params_m, *_ = _run_window(True, 1)
    _, params_u, *_ = _run_window(True, 1, masked=False)
    assert any(not torch.equal(a, b) for a, b in zip(params_m, params_u))

    import bench
    from lvdgs import backend_map, synthetic
    dev = torch.device("cuda", 0)
    res = []
    for scribble in (False, True):
        torch.manual_seed(0)
        model, *_ = bench.build_scene("tmp_masked_window_batch", 0, dev)
        backend, window = bench.build_window("tmp_masked_window_batch", 12, dev, model, n_window=8, masked=True)
        if scribble:
            for kf in window:
                vp = backend.viewpoints[kf]
                dyn = ~vp.static_mask
                vp.original_image = vp.original_image.clone()
                vp.original_image[:, dyn] = 0.77
                md = torch.from_numpy(vp.mono_depth).clone()
                md[dyn.cpu()] = 123.0
                vp.mono_depth = md.numpy()
        backend.shard_seed = 0
        # (only the window's keyframes were edited: the two random older views are scored on their whole images)
        backend_map.map_window(backend, window, iters=1)
        torch.cuda.synchronize()
        res.append([p.detach().clone() for p in backend.gaussians.parameters()])
    for a, b in zip(*res):
        assert torch.equal(a, b)


def test_colour_refinement_takes_the_masked_route_and_matches_autograd():
    """slam_loops.color_refinement (reference utils/slam_backend.py:393-468) on keyframes with and without a static mask: the
    MapViewPass route (masked_loss without a depth term) against render() -> l1_dssim_loss -> backward()."""
    from lvdgs import slam_loops
    from lvdgs.fast_mapping import MapViewPass
    import random
    out = {}
    for fast in (True, False):
        backend, window, dev = _scene(workload="tmp_masked_refine", n_kf=4)
        backend.viewpoints[window[1]].static_mask = None   # one keyframe without a mask
        calls = []
        run = MapViewPass.run
        MapViewPass.run = lambda self, *a, **k: (calls.append(k.get("masked_loss")) or run(self, *a, **k))
        losses = []
        random.seed(5)
        try:
            if fast:
                slam_loops.color_refinement(backend, iteration_total=6, on_iteration=lambda it, kf, loss: losses.append(float(loss.detach())))
            else:
                from lvdgs.loss_utils import l1_dssim_loss
                slam_loops.color_refinement(backend, iteration_total=6, loss_fn=lambda *a: l1_dssim_loss(*a),
                                            on_iteration=lambda it, kf, loss: losses.append(float(loss.detach())))
        finally:
            MapViewPass.run = run
        torch.cuda.synchronize()
        assert len(calls) == (6 if fast else 0) and all(c is not None and c[1] is None for c in calls)
        out[fast] = (losses, [p.detach().clone() for p in backend.gaussians.parameters()])
    for a, b in zip(out[True][0], out[False][0]):
        assert abs(a - b) <= 1e-5 * abs(b)
    for a, b in zip(out[True][1], out[False][1]):
        if a.numel() == 0:   # (no SH coefficients beyond degree 0)
            continue
        d = (a - b).abs()
        assert (d > 1e-4 * b.abs() + 1e-5).float().mean() < 3e-3
