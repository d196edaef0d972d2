"""The device-side pose optimiser step and the autograd-free tracking session against the PyTorch statements of the
same loop (torch.optim.Adam + pose_utils.update_pose + Camera's derived matrices; slam_loops.track_frame(fused=False))."""
import ctypes as C
import math
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
pytestmark = pytest.mark.gpu


def _camera(W=64, H=48):
    from lvdgs.camera_utils import Camera
    from lvdgs.graphics_utils import focal2fov, getProjectionMatrix2
    proj = getProjectionMatrix2(znear=0.01, zfar=100.0, fx=70.0, fy=72.0, cx=31.0, cy=25.0, W=W, H=H).transpose(0, 1).cuda()
    return Camera(3, torch.rand(3, H, W).cuda(), None, None, torch.eye(4), proj, 70.0, 72.0, 31.0, 25.0, focal2fov(70.0, W),
                  focal2fov(72.0, H), H, W, device="cuda")


@pytest.mark.parametrize("grad_scale", [1.0, 1e-3, 1e-9])
def test_pose_step_equals_adam_plus_update_pose(grad_scale):
    """Twelve steps with seeded gradients: parameters, Adam moments, R, T, the three derived matrices and the
    converged flag follow torch.optim.Adam.step() + update_pose() (reference utils/slam_frontend.py:1518-1521,
    utils/pose_utils.py:70-87).  grad_scale 1e-9 keeps the rotation below the 1e-5 rad series threshold and converges."""
    from lvdgs import _lib
    from lvdgs.pose_utils import SE3_exp, update_pose
    g = torch.Generator().manual_seed(5)
    cam_t, cam_k = _camera(), _camera()
    pose0 = SE3_exp(torch.tensor([0.3, -0.1, 0.2, 0.2, -0.4, 0.1]))
    for cam in (cam_t, cam_k):
        cam.update_RT(pose0[:3, :3].cuda(), pose0[:3, 3].cuda())
    opt = torch.optim.Adam([{"params": [cam_t.cam_rot_delta], "lr": 0.003}, {"params": [cam_t.cam_trans_delta], "lr": 0.001},
                            {"params": [cam_t.exposure_a], "lr": 0.01}, {"params": [cam_t.exposure_b], "lr": 0.01}])
    L = _lib.lib()
    R, T = cam_k.R.clone().contiguous(), cam_k.T.clone().contiguous()
    view, proj, campos = torch.empty(4, 4, device="cuda"), torch.empty(4, 4, device="cuda"), torch.empty(3, device="cuda")
    state = torch.zeros(24, device="cuda")
    gtau, ga, gb = torch.empty(6, device="cuda"), torch.empty(1, device="cuda"), torch.empty(1, device="cuda")
    pa = _lib.PoseStepArgs()
    P = lambda t: C.c_void_p(t.data_ptr())
    pa.R, pa.T, pa.cam_rot_delta, pa.cam_trans_delta = P(R), P(T), P(cam_k.cam_rot_delta), P(cam_k.cam_trans_delta)
    pa.exposure_a, pa.exposure_b, pa.grad_tau, pa.grad_exposure_a, pa.grad_exposure_b = P(cam_k.exposure_a), P(cam_k.exposure_b), P(gtau), P(ga), P(gb)
    pa.state, pa.lr_rot, pa.lr_trans, pa.lr_exposure = P(state), 0.003, 0.001, 0.01
    pa.beta1, pa.beta2, pa.eps, pa.converged_threshold = 0.9, 0.999, 1e-8, 1e-4
    praw = cam_k.projection_matrix.contiguous()   # the Camera keeps the transposed VIEW it was given: make the memory row-major
    pa.projmatrix_raw, pa.viewmatrix, pa.projmatrix, pa.campos = P(praw), P(view), P(proj), P(campos)
    stream = _lib.raw_stream(torch.device("cuda", 0))
    first_converged = None
    for it in range(12):
        grads = torch.randn(8, generator=g) * grad_scale * (1.0 if it < 8 else 0.0)   # later steps: Adam's momentum alone
        cam_t.cam_trans_delta.grad, cam_t.cam_rot_delta.grad = grads[:3].cuda(), grads[3:6].cuda()
        cam_t.exposure_a.grad, cam_t.exposure_b.grad = grads[6:7].cuda(), grads[7:8].cuda()
        gtau.copy_(grads[:6]); ga.copy_(grads[6:7]); gb.copy_(grads[7:8])
        if first_converged is None:
            with torch.no_grad():
                opt.step()
                if bool(update_pose(cam_t)):
                    first_converged = it
        _lib.check(L.lvdgs_pose_step(C.byref(pa), stream), "lvdgs_pose_step")
        torch.cuda.synchronize()
        assert (float(state[17]) != 0.0) == (first_converged is not None), it
        np.testing.assert_allclose(R.cpu().numpy(), cam_t.R.cpu().numpy(), atol=2e-6, err_msg=f"R at {it}")
        np.testing.assert_allclose(T.cpu().numpy(), cam_t.T.cpu().numpy(), atol=2e-6, err_msg=f"T at {it}")
        np.testing.assert_allclose(float(cam_k.exposure_a.detach()), float(cam_t.exposure_a.detach()), atol=1e-7, rtol=1e-5)
        np.testing.assert_allclose(float(cam_k.exposure_b.detach()), float(cam_t.exposure_b.detach()), atol=1e-7, rtol=1e-5)
        assert not cam_k.cam_rot_delta.detach().any() and not cam_k.cam_trans_delta.detach().any()
        np.testing.assert_allclose(view.cpu().numpy(), cam_t.world_view_transform.cpu().numpy(), atol=2e-6)
        np.testing.assert_allclose(proj.cpu().numpy(), cam_t.full_proj_transform.cpu().numpy(), rtol=1e-5, atol=2e-5)
        np.testing.assert_allclose(campos.cpu().numpy(), cam_t.camera_center.cpu().numpy(), atol=5e-6)
    if grad_scale == 1e-9:
        assert first_converged is not None and int(state[18]) == first_converged + 1   # steps after convergence are no-ops
    else:
        assert int(state[18]) == (12 if first_converged is None else first_converged + 1)
    st = opt.state[cam_t.cam_rot_delta]
    np.testing.assert_allclose(state[0:6:2].cpu().numpy(), st["exp_avg"].cpu().numpy(), rtol=1e-5, atol=1e-12)
    np.testing.assert_allclose(state[1:6:2].cpu().numpy(), st["exp_avg_sq"].cpu().numpy(), rtol=1e-5, atol=1e-20)


@pytest.mark.parametrize("monocular,propagate_opacity", [(True, False), (False, False), (True, True)])
def test_fused_tracking_equals_the_autograd_loop(monocular, propagate_opacity):
    """slam_loops.track_frame on the TrackingSession against the same loop through render() / autograd /
    torch.optim.Adam / update_pose: per-iteration losses, final pose, exposure, median depth, last images -- with the
    RGB-only tracking loss (monocular, utils/slam_utils.py:45-49), the RGB-D one (:65-79), and with the gradient of the
    opacity image switched on."""
    sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
    from loop_scene import build_scene, loop_config
    from lvdgs import rasterizer
    from lvdgs.slam_loops import track_frame
    cfg = loop_config()
    cfg["Training"]["monocular"] = monocular
    rasterizer.PROPAGATE_OPACITY_GRAD = propagate_opacity
    try:
        _compare_fused_and_autograd_tracking(cfg, build_scene, track_frame)
    finally:
        rasterizer.PROPAGATE_OPACITY_GRAD = False


def _compare_fused_and_autograd_tracking(cfg, build_scene, track_frame):
    out = {}
    for fused in (False, True):
        torch.manual_seed(2)
        sc = build_scene("cuda")
        cam, prev = sc["track_camera"], sc["cameras"][0]
        cam.mono_depth = sc["track_mono_depth"]
        cam.update_RT(prev.R, prev.T)
        losses = []
        pkg, med, n_it = track_frame(cam, sc["gaussians"], cfg, sc["pipe"], sc["background"], tracking_itr_num=25, fused=fused,
                                     on_iteration=lambda i, loss, p: losses.append(float(loss.detach())))
        out[fused] = dict(losses=np.array(losses), R=cam.R.cpu().numpy(), T=cam.T.cpu().numpy(), n_it=n_it, med=float(med),
                          exp=[float(cam.exposure_a.detach()), float(cam.exposure_b.detach())],
                          depth=pkg["depth"].detach().cpu().numpy(), image=pkg["render"].detach().cpu().numpy(),
                          deltas=torch.cat([cam.cam_rot_delta.detach(), cam.cam_trans_delta.detach()]).cpu().numpy())
        assert sc["gaussians"]._xyz.grad is None or not fused   # the session leaves the map's .grad fields alone
    a, b = out[True], out[False]
    assert a["n_it"] == b["n_it"] == 25 and len(a["losses"]) == 25
    np.testing.assert_allclose(a["losses"], b["losses"], rtol=2e-5)
    np.testing.assert_allclose(a["R"], b["R"], atol=2e-6)
    np.testing.assert_allclose(a["T"], b["T"], atol=2e-6)
    np.testing.assert_allclose(a["exp"], b["exp"], rtol=1e-3, atol=2e-5)   # 25 Adam steps of 1e-2 each
    assert abs(a["med"] - b["med"]) <= 1e-5 * abs(b["med"])
    np.testing.assert_allclose(a["depth"], b["depth"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(a["image"], b["image"], rtol=1e-4, atol=1e-6)
    assert not a["deltas"].any() and not b["deltas"].any()


def test_fused_tracking_stops_at_convergence_like_the_loop_that_breaks():
    """Start at the optimum with a vanishing learning rate: the pose update is below 1e-4 at once.  The reference loop
    breaks after its first iteration; the session, polled two iterations late, must report one applied iteration and the
    same pose, however many iterations the host had already enqueued."""
    sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
    from loop_scene import build_scene, loop_config
    from lvdgs.slam_loops import track_frame
    cfg = loop_config()
    cfg["Training"]["lr"] = {"cam_rot_delta": 1e-6, "cam_trans_delta": 1e-6}
    res = {}
    for fused in (False, True):
        sc = build_scene("cuda")
        cam = sc["track_camera"]          # already at its true pose
        cam.mono_depth = sc["track_mono_depth"]
        _, _, n_it = track_frame(cam, sc["gaussians"], cfg, sc["pipe"], sc["background"], tracking_itr_num=10, fused=fused)
        res[fused] = (n_it, cam.R.cpu().numpy(), cam.T.cpu().numpy())
    assert res[True][0] == res[False][0] == 1
    np.testing.assert_allclose(res[True][1], res[False][1], atol=1e-7)
    np.testing.assert_allclose(res[True][2], res[False][2], atol=1e-7)


def test_fused_adam_equals_torch_adam_through_densification_style_state_edits():
    """gaussian_model.FusedAdam (one HIP launch per step) against torch.optim.Adam on the map's six groups: ten steps, a
    parameter replaced with its moments kept (what prune / densify do), a group whose gradient is missing, sizes that are
    not multiples of four."""
    from lvdgs.gaussian_model import FusedAdam
    g = torch.Generator().manual_seed(3)
    N = 10_007
    shapes = [(N, 3), (N, 1, 3), (N, 0, 3), (N, 1), (N, 3), (N, 4)]
    lrs = [1.6e-3 * 6, 2.5e-3, 2.5e-3 / 20, 0.05, 1e-3 * 6, 1e-3]
    init = [torch.randn(*s, generator=g) for s in shapes]
    opts = {}
    for kind, cls in (("fused", FusedAdam), ("torch", torch.optim.Adam)):
        ps = [torch.nn.Parameter(t.clone().cuda()) for t in init]
        opts[kind] = (cls([{"params": [p], "lr": lr, "name": str(i)} for i, (p, lr) in enumerate(zip(ps, lrs))], lr=0.0, eps=1e-15), ps)
    for it in range(10):
        grads = [torch.randn(*s, generator=g) * (10.0 ** -(it % 4)) for s in shapes]
        for kind, (opt, ps) in opts.items():
            for i, (p, gr) in enumerate(zip([gp["params"][0] for gp in opt.param_groups], grads)):
                p.grad = None if (i == 3 and it == 4) else gr.cuda()       # one group without a gradient once
            opt.step()
            opt.zero_grad(set_to_none=True)
            if it == 5:   # keep the first 9000 rows of group 0 with their moments (prune_points does this to every group)
                gp = opt.param_groups[0]
                old = gp["params"][0]
                st = opt.state.pop(old)
                new = torch.nn.Parameter(old.data[:9000].clone())
                st["exp_avg"], st["exp_avg_sq"] = st["exp_avg"][:9000].clone(), st["exp_avg_sq"][:9000].clone()
                opt.state[new] = st
                gp["params"][0] = new
                shapes[0] = (9000, 3) if kind == "torch" else shapes[0]
        a = [gp["params"][0].detach().cpu().numpy() for gp in opts["fused"][0].param_groups]
        b = [gp["params"][0].detach().cpu().numpy() for gp in opts["torch"][0].param_groups]
        for i, (x, y) in enumerate(zip(a, b)):
            np.testing.assert_allclose(x, y, rtol=2e-6, atol=2e-6, err_msg=f"group {i} after step {it}")  # parameters are O(1): a few ulps
    fo, to = opts["fused"][0], opts["torch"][0]
    for gf, gt in zip(fo.param_groups, to.param_groups):
        sf, st = fo.state[gf["params"][0]], to.state[gt["params"][0]]
        if "exp_avg" in st:
            assert int(sf["step"]) == int(st["step"])
            np.testing.assert_allclose(sf["exp_avg"].cpu().numpy(), st["exp_avg"].cpu().numpy(), rtol=1e-5, atol=1e-7)
            np.testing.assert_allclose(sf["exp_avg_sq"].cpu().numpy(), st["exp_avg_sq"].cpu().numpy(), rtol=1e-5, atol=1e-9)


def test_map_window_with_fused_steps_equals_the_pytorch_statements():
    """backend_map.map_window(fused=True) -- isotropic regulariser, per-view statistics and the keyframes' Adam +
    update_pose as single launches -- against fused=False (autograd term, PyTorch bookkeeping, torch.optim.Adam +
    pose_utils.update_pose) on the toy scene: six iterations through a densification and an opacity reset."""
    sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
    import test_loop_golden as tl
    from loop_scene import build_scene, loop_config
    from lvdgs.backend_map import map_window
    cfg = loop_config()
    res = {}
    for fused in (False, True):
        torch.manual_seed(1)
        sc = build_scene("cuda")
        be = tl._backend(sc, cfg)
        be.initialized = True
        for i, cam in enumerate(sc["cameras"]):
            be.viewpoints[i] = cam
        window = sc["window"]
        be.current_window = window
        be.keyframe_optimizers = sc["make_keyframe_optimizer"](be.viewpoints, window, cfg)
        counts = []
        tl._record_steps(sc["gaussians"].optimizer, counts, [])
        stats = {}
        map_window(be, window, iters=sc["map_iters"], stats=stats, fused=fused)
        G = be.gaussians
        res[fused] = dict(counts=counts, losses=[float(x) for x in stats["losses"]],
                          params={k: v.detach().cpu().numpy() for k, v in G._params_by_name().items()},
                          radii=G.max_radii2D.cpu().numpy(), accum=G.xyz_gradient_accum.cpu().numpy(), denom=G.denom.cpu().numpy(),
                          poses=[(c.R.cpu().numpy(), c.T.cpu().numpy(), float(c.exposure_a.detach()), float(c.exposure_b.detach()))
                                 for c in sc["cameras"]],
                          occ={kf: be.occ_aware_visibility[kf].cpu().numpy() for kf in window},
                          deltas=[torch.cat([c.cam_rot_delta.detach(), c.cam_trans_delta.detach()]).cpu().numpy() for c in sc["cameras"]])
        assert hasattr(be.keyframe_optimizers, "_lvdgs_stepper") == fused
    a, b = res[True], res[False]
    assert a["counts"] == b["counts"] and len(set(a["counts"])) > 1
    np.testing.assert_allclose(a["losses"], b["losses"], rtol=2e-5)
    for k in a["params"]:
        if a["params"][k].size:
            np.testing.assert_allclose(a["params"][k], b["params"][k], rtol=2e-4, atol=2e-5, err_msg=k)
    np.testing.assert_array_equal(a["radii"], b["radii"])
    np.testing.assert_array_equal(a["denom"], b["denom"])
    np.testing.assert_allclose(a["accum"], b["accum"], rtol=1e-4, atol=1e-7)
    for (Ra, Ta, ea, fa), (Rb, Tb, eb, fb) in zip(a["poses"], b["poses"]):
        np.testing.assert_allclose(Ra, Rb, atol=5e-6)
        np.testing.assert_allclose(Ta, Tb, atol=5e-6)
        assert abs(ea - eb) < 1e-5 and abs(fa - fb) < 1e-5
    for kf in a["occ"]:
        np.testing.assert_array_equal(a["occ"][kf], b["occ"][kf])
    # the three keyframes of the pose window were retracted (deltas zero), the others never moved
    assert all(not d.any() for d in a["deltas"])


def test_the_keyframe_stepper_continues_from_the_torch_optimisers_state_and_hands_it_back():
    """Iterations may change hands between ``keyframe_optimizers.step()`` (fused=False) and the device-side stepper: two
    iterations through torch's Adam, two through the stepper (which IMPORTS the moments and step counts it finds), two through
    torch's again (the stepper EXPORTS) must leave the keyframes where six iterations of either alone leave them."""
    sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
    import test_loop_golden as tl
    from loop_scene import build_scene, loop_config
    from lvdgs.backend_map import map_window
    cfg = loop_config()
    cfg["Training"].update(gaussian_update_every=1000, gaussian_update_offset=999, gaussian_reset=1000)   # the map's size stays
    res = {}
    for plan in ((False,) * 6, (False, False, True, True, False, False)):
        torch.manual_seed(1)
        sc = build_scene("cuda")
        be = tl._backend(sc, cfg)
        be.initialized = True
        for i, cam in enumerate(sc["cameras"]):
            be.viewpoints[i] = cam
        window = sc["window"]
        be.current_window = window
        be.keyframe_optimizers = sc["make_keyframe_optimizer"](be.viewpoints, window, cfg)
        for fused in plan:
            map_window(be, window, iters=1, fused=fused)
        res[plan] = [(c.R.cpu().numpy(), c.T.cpu().numpy(), float(c.exposure_a.detach()), float(c.exposure_b.detach())) for c in sc["cameras"]]
    (a, b) = res.values()
    moved = 0
    for (Ra, Ta, ea, fa), (Rb, Tb, eb, fb) in zip(a, b):
        np.testing.assert_allclose(Ra, Rb, atol=5e-6)
        np.testing.assert_allclose(Ta, Tb, atol=5e-6)
        assert abs(ea - eb) < 1e-5 and abs(fa - fb) < 1e-5
        moved += int(abs(ea) > 1e-3)
    assert moved >= 3   # (Adam's sixth step from imported moments differs from a first step by far more than the tolerance)


TRACKING_ITERATION_BOUND = 8e-6   # (achieved: 9.9e-7 / 1.65e-6 at worst, the quaternion gradient) relative L2 and largest element error (of the tensor's scale) of every gradient below


def test_a_tracking_iteration_at_kitti_size_matches_the_cpu_chain_end_to_end():
    """One TrackingSession.step (lvdgs_forward -> lvdgs_backward_fused_loss -> lvdgs_tracking_tail) at KITTI-07's geometry
    against the chain it stands for, on the CPU: the C oracle's forward -> get_loss_tracking as PyTorch statements (opacity
    image detached, edge mask, exposure) -> autograd of the loss w.r.t. the image -> the oracle's backward: the loss, the
    pose gradient the optimiser step consumed, the exposure gradients and every Gaussian gradient."""
    import math
    from types import SimpleNamespace
    sys.path.insert(0, os.path.join(HERE, ".."))
    sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
    import bench
    import test_gpu_parity as tp
    from lvdgs.fast_tracking import TrackingSession
    from lvdgs.slam_utils import get_loss_tracking
    orc, _, _ = tp._mods()
    dev = torch.device("cuda", torch.cuda.current_device())
    model, cam, g, (N, W, H) = bench.build_scene("kitti07_geom", 2, dev)   # (rank 2: a camera off the origin)
    with torch.no_grad():
        cam.exposure_a.fill_(0.04); cam.exposure_b.fill_(-0.03)
    pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False)
    # The loss is an L1: where a rendered value sits within rounding of its target, two pipelines that agree to 1e-5 can take
    # different signs and that pixel's whole contribution flips.  The target is moved off such pixels (by 1e-3 where a residual
    # is below 2e-4), so that what is compared is the arithmetic, not a coin toss.
    from lvdgs.gaussian_renderer import render
    with torch.no_grad():
        pre = render(cam, model, pipe, torch.zeros(3, device=dev))
        shown = torch.exp(cam.exposure_a) * pre["render"] + cam.exposure_b
        r = shown - cam.original_image
        near = r.abs() < 2e-4
        cam.original_image = torch.where(near, shown - 1e-3 * torch.where(r >= 0, 1.0, -1.0), cam.original_image).contiguous()
        assert int(near.sum()) < 5000
    s = TrackingSession(cam, model, bench.CONFIG, pipe, torch.zeros(3, device=dev), gaussian_gradients=True)
    cpu = lambda t: t.detach().cpu().contiguous().clone()
    view, proj, proj_raw, campos = cpu(s.view), cpu(s.proj), cpu(s.proj_raw), cpu(s.campos)   # the camera the step renders from
    exp_a, exp_b = cpu(cam.exposure_a), cpu(cam.exposure_b)
    s.step()
    torch.cuda.synchronize()

    G = model
    o = orc.Oracle("f32")
    f_ora = o.forward(means3D=cpu(G.get_xyz).numpy(), opacities=cpu(G.get_opacity).numpy(), W=W, H=H,
                      tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5), viewmatrix=view.numpy(), projmatrix=proj.numpy(),
                      projmatrix_raw=proj_raw.numpy(), campos=campos.numpy(), bg=np.zeros(3, np.float32), scales=cpu(G.get_scaling).numpy(),
                      rotations=cpu(G.get_rotation).numpy(), shs=cpu(G.get_features).numpy(), sh_degree=0)
    color = torch.from_numpy(np.ascontiguousarray(f_ora["color"])).reshape(3, H, W).requires_grad_(True)
    depth = torch.from_numpy(np.ascontiguousarray(f_ora["depth"])).reshape(1, H, W)
    opacity = torch.from_numpy(np.ascontiguousarray(f_ora["opacity"])).reshape(1, H, W)
    cpu_view = SimpleNamespace(original_image=cpu(cam.original_image), grad_mask=cpu(cam.grad_mask), mono_depth=cam.mono_depth,
                               exposure_a=exp_a.requires_grad_(True), exposure_b=exp_b.requires_grad_(True))
    loss_cpu = get_loss_tracking(bench.CONFIG, color, depth, opacity, cpu_view)
    loss_cpu.backward()
    b_ora = o.backward(color.grad.numpy(), None, None)
    o.free()
    assert abs(float(s.loss) - float(loss_cpu.detach())) <= 1e-5 * abs(float(loss_cpu.detach())), (float(s.loss), float(loss_cpu.detach()))

    sc = cpu(G.get_scaling).numpy().astype(np.float64)
    op = cpu(G.get_opacity).numpy().astype(np.float64)
    raw_q = cpu(G._rotation).numpy().astype(np.float64)
    qn = np.linalg.norm(raw_q, axis=1, keepdims=True)
    q = raw_q / qn
    g_q = b_ora["rotations"].astype(np.float64)
    ref = {"means3D": b_ora["means3D"], "scales": b_ora["scales"] * sc, "opacities": b_ora["opacities"].reshape(op.shape) * op * (1.0 - op),
           "rotations": (g_q - q * (q * g_q).sum(1, keepdims=True)) / qn, "shs": b_ora["shs"], "tau": b_ora["tau"]}
    got = {"means3D": cpu(s.d_m3).numpy(), "scales": cpu(s.d_sc).numpy(), "opacities": cpu(s.d_op).numpy(), "rotations": cpu(s.d_rot).numpy(),
           "shs": cpu(s.d_sh).numpy(), "tau": cpu(s.d_tau).numpy()}
    # (no residual of the loss within rounding of zero -- above -- so what is left is the rasterizer's float32 latitude; asserted at
    # five times what the run achieves, profiles/r04_parity_report.txt; round 3, with the coin tosses in: 2e-4 and 1e-2)
    for n in ("means3D", "opacities", "scales", "rotations", "shs", "tau"):
        r = np.asarray(ref[n], np.float32)
        a = got[n].reshape(r.shape)
        st = tp.parity_stats_record("tracking iteration, grad " + n, a, r)
        scale = max(float(np.abs(r).max()), 1e-30)
        assert st["rel_l2"] <= TRACKING_ITERATION_BOUND, (n, st)
        assert float(np.abs(a - r).max()) <= TRACKING_ITERATION_BOUND * scale, (n, st)
    for got_e, ref_e, n in ((s.d_a, cpu_view.exposure_a.grad, "exposure_a"), (s.d_b, cpu_view.exposure_b.grad, "exposure_b")):
        a, b = float(got_e), float(ref_e)
        assert abs(a - b) <= 2e-4 * max(abs(b), 1e-6), (n, a, b)


def test_map_view_pass_equals_render_loss_backward_through_autograd():
    """fast_mapping.MapViewPass.run against render() -> get_loss_mapping() -> backward() on the same view: the loss, the
    images, the gradients of the six parameter tensors, of the pose deltas and of the exposure, and the screen-space
    gradient -- first as the only view (gradients assigned), then as a second view (gradients added)."""
    sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
    import test_loop_golden as tl
    from loop_scene import build_scene, loop_config
    from lvdgs.fast_mapping import MapViewPass
    from lvdgs.gaussian_renderer import render
    from lvdgs.slam_utils import get_loss_mapping
    cfg = loop_config()
    torch.manual_seed(3)
    sc = build_scene("cuda")
    be = tl._backend(sc, cfg)
    G = be.gaussians
    cams = sc["cameras"]
    names = ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity")
    view_names = ("cam_rot_delta", "cam_trans_delta", "exposure_a", "exposure_b")

    def grad_of(c, n):
        g = getattr(c, n).grad
        return None if g is None else g.clone()

    def clear():
        for n in names:
            getattr(G, n).grad = None
        for c in cams:
            for n in view_names:
                getattr(c, n).grad = None

    def through_autograd(views):
        clear()
        loss, pkgs = 0, []
        for c in views:
            pkg = render(c, G, be.pipeline_params, be.background)
            loss = loss + get_loss_mapping(cfg, pkg["render"], c, depth=pkg["depth"], monodepth=True)
            pkgs.append(pkg)
        loss.backward()
        return (float(loss.detach()), [grad_of(G, n) for n in names],
                [[grad_of(c, n) for n in view_names] for c in views], pkgs)

    def through_pass(views):
        clear()
        assert all(MapViewPass.usable(be, c) for c in views)
        vp, loss, pkgs = MapViewPass(torch.device("cuda", torch.cuda.current_device())), 0.0, []
        for c in views:
            pkg, l = vp.run(be, c)
            loss = loss + float(l)
            pkgs.append(pkg)
        return (loss, [grad_of(G, n) for n in names],
                [[grad_of(c, n) for n in view_names] for c in views], pkgs)

    for views in ([cams[1]], [cams[1], cams[2]]):
        la, ga, va, pa = through_autograd(views)
        lb, gb, vb, pb = through_pass(views)
        assert abs(la - lb) <= 1e-6 * abs(la)
        for n, x, y in zip(names, ga, gb):
            assert (x is None) == (y is None), n
            if x is not None:
                assert x.shape == y.shape, n
                # one view: the same kernels on the same inputs; two views: a sum of two terms in the other order
                torch.testing.assert_close(y, x, rtol=0 if len(views) == 1 else 1e-6, atol=0 if len(views) == 1 else 1e-9, msg=n)
        for xs, ys in zip(va, vb):
            for n, x, y in zip(view_names, xs, ys):
                assert (x is None) == (y is None), n   # a parameter autograd leaves without a gradient stays without one
                if x is not None:
                    # the pose gradient comes out of the same backward: same bits; the exposure gradients are sums
                    # over tiles in the pass, over 1024-pixel blocks in the loss kernel: equal to rounding
                    tol = dict(rtol=2e-6, atol=1e-10) if n.startswith("exposure") else dict(rtol=0, atol=0)
                    torch.testing.assert_close(y, x, msg=n, **tol)
        for p, q in zip(pa, pb):
            for k in ("render", "depth", "opacity", "radii", "n_touched"):
                assert torch.equal(p[k], q[k]), k
            assert torch.equal(p["viewspace_points"].grad, q["viewspace_points"].grad)


def test_map_view_pass_with_higher_order_sh_splits_the_colour_gradient():
    """SH degree 1 (four coefficients per channel): MapViewPass concatenates _features_dc / _features_rest for the
    rasterizer and splits the SH gradient back into the two parameters; against autograd's cat / backward."""
    sys.path.insert(0, os.path.join(HERE, ".."))
    import bench
    from lvdgs import synthetic
    from lvdgs.fast_mapping import MapViewPass
    from lvdgs.gaussian_model import GaussianModel
    from lvdgs.gaussian_renderer import render
    from lvdgs.slam_utils import get_loss_mapping
    dev = torch.device("cuda", torch.cuda.current_device())
    cfg = synthetic.CONFIGS["cfg1_10k_640x480"]
    g = synthetic.make_gaussians(cfg["N"], cfg["W"], cfg["H"], seed=5, sh_degree=1)
    model = GaussianModel.from_activated(g["means3D"], g["scales"], g["rotations"], g["opacities"], shs=g["shs"], sh_degree=1, device=dev)
    be, window = bench.build_window("cfg1_10k_640x480", 2, dev, model)
    G = be.gaussians
    assert G._features_rest.shape[1] == 3
    names = ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity")
    views = [be.viewpoints[k] for k in window]

    def clear():
        for n in names:
            getattr(G, n).grad = None

    clear()
    loss = 0
    for c in views:
        pkg = render(c, G, be.pipeline_params, be.background)
        loss = loss + get_loss_mapping(be.config, pkg["render"], c, depth=pkg["depth"], monodepth=True)
    loss.backward()
    ref = [getattr(G, n).grad.clone() for n in names]
    clear()
    vp = MapViewPass(dev)
    total = 0.0
    for c in views:
        assert MapViewPass.usable(be, c)
        _, l = vp.run(be, c)
        total += float(l)
    assert abs(total - float(loss.detach())) <= 1e-6 * abs(total)
    for n, r in zip(names, ref):
        got = getattr(G, n).grad
        assert got.shape == r.shape and got.is_contiguous(), n
        torch.testing.assert_close(got, r, rtol=1e-6, atol=1e-9, msg=n)


def test_tracking_tail_and_fused_loss_backward_equal_the_separate_launches():
    """lvdgs_photometric_loss_partials -> lvdgs_backward(dL_dtau = NULL) -> lvdgs_tracking_tail against
    lvdgs_photometric_loss_value_and_grad -> lvdgs_backward -> lvdgs_pose_step from the same state: loss, exposure and pose
    gradients, pose, deltas, exposure, Adam state and the derived matrices must be the same BITS (same additions in the
    same order); and lvdgs_backward_fused_loss -> lvdgs_tracking_tail against both."""
    import ctypes as C
    sys.path.insert(0, os.path.join(HERE, ".."))
    import bench
    from lvdgs import _lib
    from lvdgs.fast_tracking import TrackingSession, _P
    dev = torch.device("cuda", torch.cuda.current_device())
    model, cam, g, (N, W, H) = bench.build_scene("cfg1_10k_640x480", 3, dev)
    cfg = {k: (dict(v) if isinstance(v, dict) else v) for k, v in bench.CONFIG.items()}
    cfg["Training"]["monocular"] = False   # depth term on: all four partial sums are in play
    pipe = type("P", (), dict(convert_SHs_python=False, compute_cov3D_python=False))()
    sess = TrackingSession(cam, model, cfg, pipe, torch.zeros(3, device=dev), gaussian_gradients=True)
    sess.step(); sess.step()   # a state with non-zero Adam moments
    torch.cuda.synchronize()
    L, a, la, pa = sess.L, sess.a, sess.la, sess.pa
    # the session never materialises the loss's gradient images (its backward evaluates them per pixel); the two unfused
    # paths compared here need them
    H, W = sess.H, sess.W
    d_image, d_depth, d_opac = (torch.empty(3, H, W, device=dev), torch.empty(1, H, W, device=dev), torch.empty(1, H, W, device=dev))
    la.d_image, la.d_depth, la.d_opacity = _P(d_image), _P(d_depth), _P(d_opac)
    a.dL_dout_color, a.dL_dout_depth, a.dL_dout_opacity = _P(d_image), _P(d_depth), None
    state = [sess.R, sess.T, cam.cam_rot_delta, cam.cam_trans_delta, cam.exposure_a, cam.exposure_b, sess.pose_state, sess.view, sess.proj, sess.campos]
    outs = [sess.loss, sess.d_a, sess.d_b, sess.d_tau]
    grads = [sess.d_m3, sess.d_m2, sess.d_op, sess.d_sc, sess.d_rot, sess.d_sh]
    snap = [t.detach().clone() for t in state]

    def run(mode):
        with torch.no_grad():
            for t, s0 in zip(state, snap):
                t.copy_(s0)
            for t in outs + grads:
                t.fill_(float("nan"))
        with _lib.on_device(dev):
            stream = _lib.raw_stream(dev)
            num = C.c_int64(0)
            _lib.check(L.lvdgs_forward(C.byref(a), C.byref(num), stream), "forward")
            a.num_rendered = int(num.value)
            if mode == "fused loss":
                a.dL_dtau = None
                _lib.check(L.lvdgs_backward_fused_loss(C.byref(a), C.byref(la), 0, stream), "backward_fused_loss")
                _lib.check(L.lvdgs_tracking_tail(C.byref(la), C.byref(a), C.byref(pa), _P(sess.d_tau), 1, stream), "tail")
            elif mode == "tail":
                a.dL_dtau = None
                _lib.check(L.lvdgs_photometric_loss_partials(C.byref(la), stream), "partials")
                _lib.check(L.lvdgs_backward(C.byref(a), stream), "backward")
                _lib.check(L.lvdgs_tracking_tail(C.byref(la), C.byref(a), C.byref(pa), _P(sess.d_tau), 0, stream), "tail")
            else:
                a.dL_dtau = _P(sess.d_tau)
                _lib.check(L.lvdgs_photometric_loss_value_and_grad(C.byref(la), stream), "value_and_grad")
                _lib.check(L.lvdgs_backward(C.byref(a), stream), "backward")
                _lib.check(L.lvdgs_pose_step(C.byref(pa), stream), "pose_step")
        torch.cuda.synchronize()
        return [t.detach().clone() for t in state + outs + grads]

    ra, rb, rc = run("separate"), run("tail"), run("fused loss")
    names = ("R T rot_delta trans_delta exposure_a exposure_b adam_state view proj campos loss d_a d_b d_tau "
             "d_means3D d_means2D d_opacity d_scales d_rotations d_sh").split()
    for n, x, y in zip(names, ra, rb):
        assert torch.isfinite(x).all(), n
        assert torch.equal(x, y), (n, x, y)
    # The backward that evaluates the loss itself sees the same per-pixel gradients (same formulas, same operations), so the
    # Gaussian gradients and the pose gradient are the same bits; the loss value and the exposure gradients are sums over
    # tiles instead of over 1024-pixel blocks -- equal to rounding -- and with them the exposure step.
    for n, x, z in zip(names, ra, rc):
        if n.startswith("d_") and n not in ("d_a", "d_b"):
            assert torch.equal(x, z), n
        elif n in ("R", "T", "rot_delta", "trans_delta", "view", "proj", "campos"):
            assert torch.equal(x, z), n
        else:
            torch.testing.assert_close(z, x, rtol=2e-6, atol=1e-9, msg=n)
    assert not torch.equal(ra[0], snap[0])   # the step moved the pose


def test_fused_loss_backward_refuses_mismatched_sizes():
    """lvdgs_backward_fused_loss validates like the calls it replaces: a loss block of another image size is an error with a
    message, not a launch."""
    import ctypes as C
    from lvdgs import _lib
    L = _lib.lib()
    dev = torch.device("cuda", torch.cuda.current_device())
    a, la = _lib.Args(), _lib.LossArgs()
    a.image_width, a.image_height, a.num_gaussians, a.num_rendered = 64, 48, 0, 0
    img = torch.zeros(3, 48, 64, device=dev)
    scratch = torch.empty(int(L.lvdgs_loss_scratch_bytes(64, 48)), dtype=torch.uint8, device=dev)
    la.width, la.height = 32, 48
    la.image, la.gt_image = C.c_void_p(img.data_ptr()), C.c_void_p(img.data_ptr())
    la.scratch, la.scratch_bytes = C.c_void_p(scratch.data_ptr()), scratch.numel()
    la.weight_rgb = 1.0
    with _lib.on_device(dev):
        stream = _lib.raw_stream(dev)
        assert L.lvdgs_backward_fused_loss(C.byref(a), C.byref(la), 0, stream) == _lib.E_INVALID
        assert b"image size" in L.lvdgs_last_error()
    # the scratch the loss kernels ask for holds four sums per tile as well as per 1024 pixels
    for W, H in ((64, 48), (1920, 1080), (17, 300)):
        tiles = ((W + 15) // 16) * ((H + 15) // 16)
        assert L.lvdgs_loss_scratch_bytes(W, H) >= 16 * tiles


def test_a_view_that_sees_nothing_tracks_and_maps_like_the_autograd_path():
    """A camera turned away from every Gaussian lists no (Gaussian, tile) pair.  The reference's loops just get the
    background image: the loss is that image's loss, every Gaussian and pose gradient is zero.  The fused paths (the
    defaults) do the same instead of raising: the backward with the loss inside still evaluates the loss over empty
    lists."""
    sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
    from loop_scene import build_scene, loop_config
    from lvdgs.fast_mapping import MapViewPass
    from lvdgs.gaussian_renderer import render
    from lvdgs.pose_utils import SE3_exp
    from lvdgs.slam_loops import track_frame
    from lvdgs.slam_utils import get_loss_mapping, get_loss_tracking
    from types import SimpleNamespace
    cfg = loop_config()
    away = SE3_exp(torch.tensor([0.0, 0.0, 0.0, 0.0, math.pi, 0.0])).cuda()   # half a turn about y: the scene is behind the camera
    res = {}
    for fused in (False, True):
        torch.manual_seed(2)
        sc = build_scene("cuda")
        cam, prev = sc["track_camera"], sc["cameras"][0]
        cam.mono_depth = sc["track_mono_depth"]
        cam.update_RT(away[:3, :3] @ prev.R, away[:3, :3] @ prev.T)
        with torch.no_grad():
            assert int((render(cam, sc["gaussians"], sc["pipe"], sc["background"])["radii"] > 0).sum()) == 0
        losses = []
        pkg, med, n_it = track_frame(cam, sc["gaussians"], cfg, sc["pipe"], sc["background"], tracking_itr_num=4, fused=fused,
                                     on_iteration=lambda i, loss, p: losses.append(float(loss.detach())))
        res[fused] = (np.array(losses), cam.R.cpu().numpy(), cam.T.cpu().numpy(), float(cam.exposure_a.detach()), float(cam.exposure_b.detach()),
                      pkg["render"].detach().cpu().numpy())
    for x, y in zip(res[True], res[False]):
        np.testing.assert_allclose(x, y, rtol=2e-5, atol=1e-7)
    bgc = sc["background"].cpu().numpy()
    np.testing.assert_allclose(res[True][5], np.broadcast_to(bgc[:, None, None], res[True][5].shape), atol=0)

    # one mapping view of the same kind through MapViewPass against render() -> get_loss_mapping() -> backward()
    torch.manual_seed(2)
    sc = build_scene("cuda")
    G, cam = sc["gaussians"], sc["cameras"][1]
    cam.update_RT(away[:3, :3] @ cam.R, away[:3, :3] @ cam.T)
    backend = SimpleNamespace(gaussians=G, config=cfg, pipeline_params=sc["pipe"], background=sc["background"])
    assert MapViewPass.usable(backend, cam)
    pkg, loss = MapViewPass(G.get_xyz.device).run(backend, cam)
    got = [getattr(G, n).grad.clone() for n in ("_xyz", "_features_dc", "_scaling", "_rotation", "_opacity")]
    got_pose = [cam.cam_rot_delta.grad.clone(), cam.cam_trans_delta.grad.clone(), cam.exposure_a.grad.clone(), cam.exposure_b.grad.clone()]
    for n in ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity"):
        getattr(G, n).grad = None
    for n in ("cam_rot_delta", "cam_trans_delta", "exposure_a", "exposure_b"):
        getattr(cam, n).grad = None
    ref_pkg = render(cam, G, sc["pipe"], sc["background"])
    ref_loss = get_loss_mapping(cfg, ref_pkg["render"], cam, depth=ref_pkg["depth"], monodepth=True)
    ref_loss.backward()
    torch.testing.assert_close(loss, ref_loss.detach(), rtol=2e-6, atol=1e-9)
    for g in got:
        assert not g.any()
    assert not got_pose[0].any() and not got_pose[1].any()
    torch.testing.assert_close(got_pose[2], cam.exposure_a.grad, rtol=2e-5, atol=1e-9)
    torch.testing.assert_close(got_pose[3], cam.exposure_b.grad, rtol=2e-5, atol=1e-9)
    assert int(pkg["n_touched"].sum()) == 0 and not pkg["visibility_filter"].any()


@pytest.mark.parametrize("workload,full", [("kitti07_geom", False), ("kitti07_geom", True), ("surface_12k_640x480", False), ("cfg3_500k_1920x1080", False)])
def test_forward_and_backward_in_one_call_are_the_two_calls_bit_for_bit(workload, full):
    """lvdgs_forward_backward_fused_loss -- on frames of up to 4096 tiles the forward and the backward blend pass of a tile in ONE launch
    (blend_fwd_bwd_kernel) -- against lvdgs_forward followed by lvdgs_backward_fused_loss: images, counters, the image state, every
    gradient, the loss and the stepped pose over three iterations, bit for bit; pose-only and full backward, a scene of deep lists
    (the deep-lists build of the forward pass, queued tile-sort segments), and a frame too large to fuse (the calls in turn inside)."""
    import ctypes as C
    import bench
    from types import SimpleNamespace
    from lvdgs import _lib, rasterizer as _rz
    from lvdgs.fast_tracking import TrackingSession
    dev = torch.device("cuda", 0)
    pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False)
    out = []
    for one_call in (True, False):
        model, cam, _, (N, W, H) = bench.build_scene(workload, 0, dev)
        s = TrackingSession(cam, model, bench.CONFIG, pipe, torch.zeros(3, device=dev), gaussian_gradients=full)
        snaps = []
        for it in range(3):
            if one_call:
                s.step()
            else:   # the two calls, then the tail, as TrackingSession.step did until round 4
                L, a = s.L, s.a
                stream = _lib.raw_stream(dev)
                num = C.c_int64(0)
                _lib.check(L.lvdgs_forward(C.byref(a), C.byref(num), stream), "lvdgs_forward")
                s.num_rendered = a.num_rendered = int(num.value)
                _lib.check(L.lvdgs_backward_fused_loss(C.byref(a), C.byref(s.la), int(_rz.PROPAGATE_OPACITY_GRAD), stream), "lvdgs_backward_fused_loss")
                _lib.check(L.lvdgs_tracking_tail(C.byref(s.la), C.byref(a), C.byref(s.pa), C.c_void_p(s.d_tau.data_ptr()), 1, stream), "lvdgs_tracking_tail")
                s.iterations_enqueued += 1
            torch.cuda.synchronize()
            lay = _lib.StateLayout()
            s.L.lvdgs_state_layout_query(N, int(s.num_rendered), W, H, C.byref(lay))
            T_, P_ = ((W + 15) // 16) * ((H + 15) // 16), W * H
            img = s.image   # (ranges, final_T, n_contrib: the parts of the image state that are results; the rest is the tile sort's queue and padding)
            snap = dict(color=s.color.clone(), depth=s.depth.clone(), opacity=s.opacity.clone(), radii=s.radii.clone(), n_touched=s.n_touched.clone(),
                        ranges=img[lay.img_ranges:lay.img_ranges + 8 * T_].clone(), final_T=img[lay.img_final_T:lay.img_final_T + 4 * P_].clone(),
                        n_contrib=img[lay.img_n_contrib:lay.img_n_contrib + 4 * P_].clone(), d_tau=s.d_tau.clone(), loss=s.loss.clone(), d_a=s.d_a.clone(), d_b=s.d_b.clone(), R=s.R.clone(), T=s.T.clone(),
                        D=torch.tensor(int(s.num_rendered)))
            if full:
                snap.update(d_m3=s.d_m3.clone(), d_m2=s.d_m2.clone(), d_op=s.d_op.clone(), d_sc=s.d_sc.clone(), d_rot=s.d_rot.clone(), d_sh=s.d_sh.clone())
            snaps.append(snap)
        s.finish()
        out.append(snaps)
    for it, (a, b) in enumerate(zip(*out)):
        for k in a:
            assert torch.equal(a[k], b[k]), (it, k)
        assert float(a["d_tau"].abs().sum()) > 0 and int(a["D"]) > 1000
    # a session whose buffers are too small for the frame: the call reports LVDGS_E_CAPACITY with BOTH passes already enqueued -- the
    # backward half must have stood down on the device (its record slots lie beyond the buffer) -- and the step re-runs them with room
    model, cam, _, (N, W, H) = bench.build_scene(workload, 0, dev)
    s = TrackingSession(cam, model, bench.CONFIG, pipe, torch.zeros(3, device=dev), gaussian_gradients=full)
    s._size_for_pairs(1000)
    s.step()
    torch.cuda.synchronize()
    first = out[0][0]
    assert s.a.pair_capacity > 1000 and int(s.num_rendered) == int(first["D"])
    for k, t in (("color", s.color), ("d_tau", s.d_tau), ("loss", s.loss), ("R", s.R), ("T", s.T), ("n_touched", s.n_touched)):
        assert torch.equal(t, first[k]), k
    s.finish()
