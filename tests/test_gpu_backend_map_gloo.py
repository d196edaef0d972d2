"""The sharded mapping loop with the HIP path under it: two processes on ONE GPU (gloo carries the CUDA tensors through the
host), each rendering, scoring and back-propagating its own views with ``fast_mapping.MapViewPass`` -- gradients written
straight into the parameters' ``.grad`` -- then the two collectives and the bookkeeping of ``backend_map.map_window``.

Same checks as the CPU test (tests/test_backend_map_gloo.py), whose renderer is the dense CPU one: (1) the replicas end
bit-identical; (2) they agree with the single-process run on the same GPU to rasterizer / summation-order rounding;
(3) the schedule of Gaussian counts is identical.  A fixed rendezvous port per test process (127.0.0.1)."""
import hashlib
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
ITERS = 4


def _paths():
    for p in (os.path.join(ROOT, "oracle"), ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import lvdgs  # noqa: F401


def _run(group_world, policy="leftover", sharded=False, window=None):
    _paths()
    import test_loop_golden as tl
    from loop_scene import build_scene, loop_config
    from lvdgs import backend_map as bm
    bm.SPLIT_POLICY = policy
    cfg = loop_config()
    sc = build_scene("cuda", window=window)
    be = tl._backend(sc, cfg)
    be.initialized = True
    be.shard_optimizer = sharded   # (the Gaussian Adam as reduce-scatter -> lvdgs_adam_step on this rank's share -> all-gather)
    for i, cam in enumerate(sc["cameras"]):
        be.viewpoints[i] = cam
    window = sc["window"]
    be.current_window = window
    be.keyframe_optimizers = sc["make_keyframe_optimizer"](be.viewpoints, window, cfg)
    counts = []
    if group_world == 1:
        keyed = bm.random_view_indices
        bm.random_view_indices = lambda n, k, it, world, seed=0: keyed(n, k, it, 2, seed)
    stats = {"before_steps": lambda backend: counts.append(int(backend.gaussians.get_xyz.shape[0]))}
    try:
        bm.map_window(be, window, iters=ITERS, stats=stats)
        n_mid = be.gaussians.get_xyz.shape[0]
        bm.map_window(be, window, prune=True)
        bm.map_window(be, window, iters=1, stats=stats)
    finally:
        if group_world == 1:
            bm.random_view_indices = keyed
    torch.cuda.synchronize()
    G = be.gaussians
    cpu = lambda t: t.detach().cpu().numpy().copy()
    out = {k: cpu(v) for k, v in G._params_by_name().items()}
    for gp in G.optimizer.param_groups:
        st = G.optimizer.state.get(gp["params"][0], {})
        if "exp_avg" in st:
            out["m_" + gp["name"]], out["v_" + gp["name"]] = cpu(st["exp_avg"]), cpu(st["exp_avg_sq"])
    out.update(max_radii2D=cpu(G.max_radii2D), accum=cpu(G.xyz_gradient_accum), denom=cpu(G.denom), n_obs=cpu(G.n_obs),
               kf_ids=cpu(G.unique_kfIDs), counts=np.array(counts), n_mid=np.array(n_mid))
    for i, cam in enumerate(sc["cameras"]):
        out[f"R{i}"], out[f"T{i}"] = cpu(cam.R), cpu(cam.T)
        out[f"exp{i}"] = np.array([float(cam.exposure_a.detach()), float(cam.exposure_b.detach())])
    for kf in window:
        out[f"occ{kf}"] = cpu(be.occ_aware_visibility[kf])
    out["views_per_iteration"] = np.array([len(r["views"]) for r in stats["iterations"][:ITERS]])
    out["used_view_pass"] = np.array(int(getattr(be, "_lvdgs_view_pass", None) is not None))
    out["used_window_batch"] = np.array(int(getattr(be, "_lvdgs_window_batch", None) is not None))
    return out


def _digest(res):
    h = hashlib.sha256()
    for k in sorted(res):
        if k != "views_per_iteration":
            h.update(k.encode())
            h.update(np.ascontiguousarray(res[k]).tobytes())
    return h.hexdigest()


def _worker(rank, world, port, q, policy, sharded=False, window=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)
        res = _run(world, policy, sharded, window)
        q.put((rank, _digest(res), res))
    finally:
        dist.destroy_process_group()


_SINGLE = {}


@pytest.mark.parametrize("world,policy,sharded,window", [(2, "leftover", False, None), (2, "all", True, None), (2, "leftover", False, (6, 5, 4))])
def test_ranks_on_one_gpu_stay_bit_identical_and_match_the_single_process_run(world, policy, sharded, window):
    """world 2: whole views (six views, three each: the rank's views through the window batch), and EVERY view cut into two bands of
    tile rows, one per rank (the band path of the rasterizer, lvdgs_args.tile_row_*, the split views' statistics, the byte-wise flag
    OR) -- the latter with the Gaussian Adam sharded (backend_map.ShardedAdam: reduce-scatter, lvdgs_adam_step on this rank's
    sub-ranges, all-gather); and a window of three keyframes (+ 2 random = five views): two whole views per rank through the window
    batch AND a band of the fifth added to their gradients view by view -- the mix a four-GPU job has.
    (Three ranks with a band each were run by the builder too; a third interpreter start costs the suite half a minute.)"""
    window = None if window is None else list(window)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000) + 11 * world
    procs = [ctx.Process(target=_worker, args=(r, world, port + (5 if sharded else 0) + (9 if window else 0), q, policy, sharded, window)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=900) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    _, d0, r0 = results[0]
    assert int(r0["used_view_pass"]) == 1   # the views went through MapViewPass, not autograd
    for _, dk, rk in results[1:]:
        for k in r0:
            if k != "views_per_iteration":
                np.testing.assert_array_equal(r0[k], rk[k], err_msg=k)
        assert d0 == dk
    if window is not None:
        assert all(r["views_per_iteration"].tolist() == [3] * ITERS for _, _, r in results)   # two whole views + a band of the fifth
    elif policy == "leftover":
        assert all(r["views_per_iteration"].tolist() == [3] * ITERS for _, _, r in results)
    else:
        assert all(r["views_per_iteration"].min() >= 5 for _, _, r in results)   # a band of (nearly) every view on every rank
    key = "ref" if window is None else "ref" + str(window)
    if key not in _SINGLE:
        torch.manual_seed(7)
        _SINGLE[key] = _run(1, window=window)
    ref = _SINGLE[key]
    np.testing.assert_array_equal(ref["counts"], r0["counts"])
    assert int(ref["n_mid"]) == int(r0["n_mid"]) and len(set(ref["counts"].tolist())) > 1   # a densification happened
    if window is not None:
        assert int(r0["used_window_batch"]) == 1   # the whole views of a rank that also holds a band went through the batch
    for k in ref:
        if k in ("views_per_iteration", "counts", "n_mid", "used_view_pass", "used_window_batch"):
            continue
        a, b = np.asarray(r0[k], np.float64), np.asarray(ref[k], np.float64)
        assert a.shape == b.shape, k
        if a.size:
            tol = 5e-4 * np.abs(b) + 5e-5 * max(np.abs(b).max(), 1e-30)
            assert (np.abs(a - b) <= tol).all(), (k, np.abs(a - b).max(), np.abs(b).max())


def test_bands_of_a_view_add_up_to_the_view():
    """One view rendered whole and as three bands of tile rows (lvdgs_args.tile_row_begin / _end through MapViewPass):
    the bands' pixels are the whole render's bits, radii are the same, n_touched and every gradient -- Gaussian
    parameters, screen-space gradient, pose, exposure -- and the loss add up to the whole view's (to summation order)."""
    _paths()
    from types import SimpleNamespace
    from lvdgs import synthetic
    from lvdgs.camera_utils import Camera
    from lvdgs.fast_mapping import MapViewPass, _PARAM_FIELDS
    from lvdgs.gaussian_model import GaussianModel
    from lvdgs.graphics_utils import focal2fov, getProjectionMatrix2
    from lvdgs.pose_utils import SE3_exp
    W, H, N = 410, 250, 30_000          # 16 tile rows, the last one partial
    dev = torch.device("cuda", 0)
    g = synthetic.make_gaussians(N, W, H, seed=5, r_min=0.7, r_max=14.0)
    proj = getProjectionMatrix2(znear=0.01, zfar=100.0, fx=float(W), fy=float(W), cx=W / 2.0, cy=H / 2.0, W=W, H=H).transpose(0, 1).contiguous().to(dev)
    gen = torch.Generator().manual_seed(77)
    cam = Camera(1, torch.rand(3, H, W, generator=gen).to(dev), None, (torch.rand(H, W, generator=gen) * 40 + 1).numpy(), torch.eye(4), proj,
                 float(W), float(W), W / 2.0, H / 2.0, focal2fov(float(W), W), focal2fov(float(W), H), H, W, device=dev)
    pose = SE3_exp(torch.randn(6, generator=gen) * 0.03)
    cam.update_RT(pose[:3, :3], pose[:3, 3])
    with torch.no_grad():
        cam.exposure_a.fill_(0.05); cam.exposure_b.fill_(-0.02)
    cfg = {"Training": {"monocular": True, "rgb_boundary_threshold": 0.01, "alpha": 0.9}, "Dataset": {}}
    pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False)

    def run(bands):
        model = GaussianModel.from_activated(g["means3D"], g["scales"], g["rotations"], g["opacities"], shs=g["shs"], sh_degree=0, device=dev)
        for n in _PARAM_FIELDS:
            getattr(model, n).requires_grad_(True)
        be = SimpleNamespace(gaussians=model, config=cfg, pipeline_params=pipe, background=torch.tensor([0.1, 0.2, 0.3], device=dev))
        for n in ("cam_rot_delta", "cam_trans_delta", "exposure_a", "exposure_b"):
            getattr(cam, n).grad = None
        vp = MapViewPass(dev)
        assert MapViewPass.usable(be, cam)
        out = []
        for band in bands:
            pkg, loss = vp.run(be, cam, band=band)
            out.append((band, {k: (v.clone() if torch.is_tensor(v) else v.grad.clone()) for k, v in pkg.items()}, loss.clone()))
        torch.cuda.synchronize()
        grads = {n: getattr(model, n).grad.clone() for n in _PARAM_FIELDS if getattr(model, n).numel()}
        grads.update({n: getattr(cam, n).grad.clone() for n in ("cam_rot_delta", "cam_trans_delta", "exposure_a", "exposure_b")})
        return out, grads

    (_, whole, loss_w), = run([None])[0]
    g_whole = run([None])[1]
    parts, g_parts = run([(0, 5), (5, 11), (11, 16)])
    loss_sum = sum(float(l) for _, _, l in parts)
    assert abs(loss_sum - float(loss_w)) <= 2e-6 * abs(float(loss_w))
    nt = torch.zeros_like(whole["n_touched"])
    vsp = torch.zeros_like(whole["viewspace_points"])
    for (r0, r1), pkg, _ in parts:
        y0, y1 = 16 * r0, min(16 * r1, H)
        for k in ("render", "depth", "opacity"):
            assert torch.equal(pkg[k][:, y0:y1], whole[k][:, y0:y1]), (k, r0, r1)
        assert torch.equal(pkg["radii"], whole["radii"]) and torch.equal(pkg["visibility_filter"], whole["visibility_filter"])
        nt += pkg["n_touched"]
        vsp += pkg["viewspace_points"]
    assert torch.equal(nt, whole["n_touched"])
    close = lambda a, b, what: torch.testing.assert_close(a, b, rtol=2e-4, atol=2e-6 * float(b.abs().max()) + 1e-12, msg=lambda m: f"{what}: {m}")
    close(vsp, whole["viewspace_points"], "screen-space gradient")
    for k in g_whole:
        close(g_parts[k], g_whole[k], k)
    assert float(g_whole["_xyz"].abs().max()) > 0 and float(g_whole["cam_rot_delta"].abs().max()) > 0
