"""BASELINE.json configs[3] and configs[4] as workloads of the mapping loop on ONE GPU (the 8-GPU runs are the driver's):

* configs[3]: KITTI-07 geometry (reference configs/mono/KITTI/07.yaml:8-18: 1226x370, the sequence's intrinsics), 200k
  Gaussians, a window of 8 keyframes + 2 random older ones (base_config.yaml:37, utils/slam_backend.py:275), through
  backend_map.map_window -- the whole iteration, fused steps against the PyTorch statements;
* configs[4]'s shape: 2M Gaussians at 1920x1280 (configs/mono/waymo/405841.yaml:15-16) with keyframes that carry
  dynamic-object masks (the static_mask branch of the mapping loss, utils/slam_backend.py:196-261).
No dataset ships with the reference, so frames are synthetic; what is checked is that the loop runs at these sizes, that
its two implementations agree, and the invariants of an iteration."""
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu


def _backend(workload, n_keyframes, window, masks=False):
    import bench
    dev = torch.device("cuda", 0)
    model, _, _, (N, W, H) = bench.build_scene(workload, 0, dev)
    be, _ = bench.build_window(workload, n_keyframes, dev, model)
    be.current_window = window
    groups = [gp for gp in be.keyframe_optimizers.param_groups if any(gp["name"].endswith(f"_{kf}") for kf in window)]
    be.keyframe_optimizers = torch.optim.Adam(groups)
    if masks:
        g = torch.Generator().manual_seed(9)
        for kf in window:
            m = torch.ones(H, W, dtype=torch.bool)
            y0, x0 = int(torch.randint(0, H - 300, (1,), generator=g)), int(torch.randint(0, W - 400, (1,), generator=g))
            m[y0:y0 + 300, x0:x0 + 400] = False          # a "vehicle"
            be.viewpoints[kf].static_mask = m.to(dev)
    return be, N


# bounds of the end-to-end comparison below: relative L2 error and largest element error (of the tensor's scale) per gradient
# (five times what the run achieves -- 9.2e-7 / 9.9e-7 at worst, the quaternion gradient; round 3, with the L1 loss's coin tosses
# in: 2e-4 and 1e-2)
MAPPING_VIEW_REL_L2 = {n: 5e-6 for n in ("means3D", "opacities", "scales", "rotations", "shs", "tau")}
MAPPING_VIEW_MAX = {n: 5e-6 for n in ("means3D", "opacities", "scales", "rotations", "shs", "tau")}


def test_a_mapping_view_at_kitti_size_matches_the_cpu_chain_end_to_end():
    """One view of the KITTI-sized window through MapViewPass (lvdgs_forward -> lvdgs_backward_fused_loss -> the view's tail)
    against the chain it stands for, on the CPU: the C oracle's forward -> get_loss_mapping as PyTorch statements ->
    autograd of the loss w.r.t. the images -> the oracle's backward -> the chain rule through the parameters' activations.
    (The window tests below compare two implementations of the product; this one has the oracle on the other side.)"""
    import math
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import bench
    import test_gpu_parity as tp
    from lvdgs.fast_mapping import MapViewPass
    from lvdgs.slam_utils import get_loss_mapping
    orc, _, _ = tp._mods()
    dev = torch.device("cuda", 0)
    model, _, _, (N, W, H) = bench.build_scene("kitti07_geom", 0, dev)
    be, window = bench.build_window("kitti07_geom", 2, dev, model)
    G, view = be.gaussians, be.viewpoints[window[0]]
    with torch.no_grad():   # a view with exposure parameters that matter
        view.exposure_a.fill_(0.03); view.exposure_b.fill_(-0.02)
    # The loss is an L1: where a rendered value sits within rounding of its target, two pipelines that agree to 1e-5 can still take
    # different signs, and that pixel's whole contribution flips.  The targets are moved off such pixels (by 1e-3 where a
    # residual is below 2e-4, a few dozen values per frame), so that what is compared is the arithmetic, not a coin toss.
    from lvdgs.gaussian_renderer import render
    with torch.no_grad():
        pre = render(view, G, be.pipeline_params, be.background)
        shown = torch.exp(view.exposure_a) * pre["render"] + view.exposure_b
        r = shown - view.original_image
        near = r.abs() < 2e-4
        view.original_image = torch.where(near, shown - 1e-3 * torch.where(r >= 0, 1.0, -1.0), view.original_image).contiguous()
        md = torch.from_numpy(view.mono_depth).to(dev)
        rd = pre["depth"][0] - md
        neard = rd.abs() < 2e-4 * md.clamp_min(1.0)
        view.mono_depth = torch.where(neard, pre["depth"][0] - 1e-3 * md.clamp_min(1.0) * torch.where(rd >= 0, 1.0, -1.0), md).cpu().numpy()
        moved = int(near.sum()) + int(neard.sum())
    for p in G.parameters():
        p.grad = None
    assert MapViewPass.usable(be, view)
    pkg, loss = MapViewPass(dev).run(be, view)
    torch.cuda.synchronize()

    cpu = lambda t: t.detach().cpu().contiguous()
    o = orc.Oracle("f32")
    f_ora = o.forward(means3D=cpu(G.get_xyz).numpy(), opacities=cpu(G.get_opacity).numpy(), W=W, H=H,
                      tanfovx=math.tan(view.FoVx * 0.5), tanfovy=math.tan(view.FoVy * 0.5),
                      viewmatrix=cpu(view.world_view_transform).numpy(), projmatrix=cpu(view.full_proj_transform).numpy(),
                      projmatrix_raw=cpu(view.projection_matrix).numpy(), campos=cpu(view.camera_center).numpy(), bg=np.zeros(3, np.float32),
                      scales=cpu(G.get_scaling).numpy(), rotations=cpu(G.get_rotation).numpy(), shs=cpu(G.get_features).numpy(), sh_degree=0)
    color = torch.from_numpy(np.ascontiguousarray(f_ora["color"])).reshape(3, H, W).requires_grad_(True)
    depth = torch.from_numpy(np.ascontiguousarray(f_ora["depth"])).reshape(1, H, W).requires_grad_(True)
    cpu_view = SimpleNamespace(original_image=cpu(view.original_image), mono_depth=view.mono_depth,
                               exposure_a=cpu(view.exposure_a).requires_grad_(True), exposure_b=cpu(view.exposure_b).requires_grad_(True))
    loss_cpu = get_loss_mapping(be.config, color, cpu_view, depth=depth, monodepth=True)
    loss_cpu.backward()
    b_ora = o.backward(color.grad.numpy(), depth.grad.numpy(), None)
    o.free()
    assert abs(float(loss) - float(loss_cpu.detach())) <= 1e-5 * abs(float(loss_cpu.detach())), (float(loss), float(loss_cpu.detach()))

    # the oracle's gradients (w.r.t. activated values) carried to the raw parameters
    s = cpu(G.get_scaling).numpy().astype(np.float64)
    op = cpu(G.get_opacity).numpy().astype(np.float64)
    raw_q = cpu(G._rotation).numpy().astype(np.float64)
    qn = np.linalg.norm(raw_q, axis=1, keepdims=True)
    q = raw_q / qn
    g_q = b_ora["rotations"].astype(np.float64)
    ref = {"means3D": b_ora["means3D"], "scales": b_ora["scales"] * s, "opacities": b_ora["opacities"].reshape(op.shape) * op * (1.0 - op),
           "rotations": (g_q - q * (q * g_q).sum(1, keepdims=True)) / qn, "shs": b_ora["shs"], "tau": b_ora["tau"]}
    got = {"means3D": cpu(G._xyz.grad).numpy(), "scales": cpu(G._scaling.grad).numpy(), "opacities": cpu(G._opacity.grad).numpy(),
           "rotations": cpu(G._rotation.grad).numpy(), "shs": cpu(G._features_dc.grad).numpy(),
           "tau": np.concatenate([cpu(view.cam_trans_delta.grad).numpy().reshape(-1), cpu(view.cam_rot_delta.grad).numpy().reshape(-1)])}
    ref = {k: np.asarray(v, np.float32) for k, v in ref.items()}
    # (no residual of the loss sits within rounding of zero -- see above -- so what is left between the two chains is the rasterizer's
    # own float32 latitude: the fragile pixels of tests/test_gpu_parity.py.  Asserted at five times what the run achieves,
    # profiles/r04_parity_report.txt.)
    assert moved < 20000, moved   # (of 1.8 M residuals)
    for n in ("means3D", "opacities", "scales", "rotations", "shs", "tau"):
        st = tp.parity_stats_record("mapping view, grad " + n, got[n].reshape(ref[n].shape), ref[n])
        scale = max(float(np.abs(ref[n]).max()), 1e-30)
        assert st["rel_l2"] <= MAPPING_VIEW_REL_L2[n], (n, st)
        assert float(np.abs(got[n].reshape(ref[n].shape) - ref[n]).max()) <= MAPPING_VIEW_MAX[n] * scale, (n, st)
    for n in ("exposure_a", "exposure_b"):
        a, b = float(getattr(view, n).grad), float(getattr(cpu_view, n).grad)
        assert abs(a - b) <= 2e-4 * max(abs(b), 1e-6), (n, a, b)


def test_kitti07_window_of_8_plus_2_random_keyframes():
    from lvdgs.backend_map import map_window
    window = list(range(12, 4, -1))          # 8 newest of 12 keyframes; 1..4 are the older ones
    out = {}
    for fused in (True, False):
        torch.manual_seed(0)
        be, N = _backend("kitti07_geom", 12, window)
        poses0 = {kf: (be.viewpoints[kf].R.clone(), be.viewpoints[kf].T.clone()) for kf in be.viewpoints}
        stats = {}
        map_window(be, window, iters=2, stats=stats, fused=fused)
        torch.cuda.synchronize()
        assert [len(r["views"]) for r in stats["iterations"]] == [10, 10]
        losses = [float(x) for x in stats["losses"]]
        assert all(np.isfinite(losses)) and be.gaussians.get_xyz.shape[0] == N and be.iteration_count == 2
        moved = [kf for kf in be.viewpoints if not torch.equal(be.viewpoints[kf].R, poses0[kf][0].to(be.viewpoints[kf].R))]
        assert sorted(moved) == sorted(window[:3])                    # pose_window = 3 (base_config.yaml:38)
        for kf in window:
            occ = be.occ_aware_visibility[kf]
            assert occ.shape == (N,) and occ.dtype == torch.int64 and 0 < int(occ.sum()) <= N
        assert float(be.gaussians.max_radii2D.max()) > 0 and float(be.gaussians.denom.max()) <= 20
        out[fused] = (losses, be.gaussians.get_xyz.detach().cpu().numpy(), {kf: be.viewpoints[kf].T.cpu().numpy() for kf in window[:3]})
    np.testing.assert_allclose(out[True][0], out[False][0], rtol=2e-5)
    # Adam moves an element whose gradient is at rounding level by up to its learning rate (1e-2) in either direction:
    # a handful of the 600k coordinates may differ by that much, the map as a whole may not
    d = np.abs(out[True][1] - out[False][1])
    assert (d > 1e-4 * np.abs(out[False][1]) + 1e-5).mean() < 3e-4 and d.max() < 2e-2
    assert np.linalg.norm(d) <= 3e-6 * np.linalg.norm(out[False][1])   # (targets = the map's own views + noise: many gradients at rounding level)
    for kf in out[True][2]:
        np.testing.assert_allclose(out[True][2][kf], out[False][2][kf], atol=5e-6)


def test_2m_gaussians_with_dynamic_object_masks():
    """Keyframes that carry a static mask: fused=True renders, scores and differentiates them without autograd (the
    static-mask loss of lvdgs_masked_loss_batch into lvdgs_backward_masked_loss, fast_mapping.MapViewPass with
    ``masked_loss``); fused=False is render() -> loss_utils.masked_mapping_loss -> backward() through the autograd
    engine.  Same losses, same map after two iterations."""
    from lvdgs.backend_map import map_window
    from lvdgs.fast_mapping import MapViewPass, MapWindowBatch
    window = [2, 1]
    out = {}
    for fused in (True, False):
        torch.manual_seed(0)
        be, N = _backend("cfg5_2m_1920x1280", 2, window, masks=True)
        assert N == 2_000_000
        calls = []
        run, brun = MapViewPass.run, MapWindowBatch.run
        MapViewPass.run = lambda self, *a, **k: (calls.append(k.get("masked_loss") is not None) or run(self, *a, **k))
        MapWindowBatch.run = lambda self, *a, **k: (calls.extend(m is not None for m in k.get("masked")) or brun(self, *a, **k))
        stats = {}
        try:
            map_window(be, window, iters=2, stats=stats, fused=fused)
        finally:
            MapViewPass.run, MapWindowBatch.run = run, brun
        torch.cuda.synchronize()
        # two masked keyframes x two iterations, none through autograd (9600 tiles: the two views go through the window batch)
        assert calls == ([True] * 4 if fused else []) and (getattr(be, "_lvdgs_window_batch", None) is not None) == fused
        losses = [float(x) for x in stats["losses"]]
        assert all(np.isfinite(losses)) and all(0.0 < v < 10.0 for v in losses)
        # masked keyframes take the L1 + SSIM branch: their exposure parameters get no gradient and never move
        for kf in window:
            vp = be.viewpoints[kf]
            assert float(vp.exposure_a.detach()) == 0.0 and float(vp.exposure_b.detach()) == 0.0
            assert int(be.occ_aware_visibility[kf].sum()) > 50_000
        assert be.gaussians.get_xyz.shape[0] == N and torch.isfinite(be.gaussians.get_xyz).all()
        out[fused] = (losses, be.gaussians.get_xyz.detach().cpu().numpy(), be.gaussians._opacity.detach().cpu().numpy(),
                      {kf: be.viewpoints[kf].T.cpu().numpy() for kf in window})
    np.testing.assert_allclose(out[True][0], out[False][0], rtol=5e-5)
    # (Adam moves an element whose gradient is at rounding level by up to its learning rate per step in either direction:
    # positions 1.6e-3 x the scene's extent, opacity logits 5e-2 -- a few per thousand may differ by that much)
    for k, frac, worst in ((1, 3e-4, 3e-2), (2, 3e-3, 0.25)):
        d = np.abs(out[True][k] - out[False][k])
        assert (d > 1e-4 * np.abs(out[False][k]) + 1e-5).mean() < frac and d.max() < worst, k
    for kf in window:
        np.testing.assert_allclose(out[True][3][kf], out[False][3][kf], atol=2e-5)
