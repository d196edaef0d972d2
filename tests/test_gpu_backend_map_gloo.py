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


def _run(group_world):
    _paths()
    import test_loop_golden as tl
    from loop_scene import build_scene, loop_config
    from lvdgs import backend_map as bm
    cfg = loop_config()
    sc = build_scene("cuda")
    be = tl._backend(sc, cfg)
    be.initialized = True
    for i, cam in enumerate(sc["cameras"]):
        be.viewpoints[i] = cam
    window = sc["window"]
    be.current_window = window
    be.keyframe_optimizers = sc["make_keyframe_optimizer"](be.viewpoints, window, cfg)
    counts = []
    sc["gaussians"].optimizer.register_step_pre_hook(lambda opt, a, k: counts.append(int(opt.param_groups[0]["params"][0].shape[0])))
    if group_world == 1:
        keyed = bm.random_view_indices
        bm.random_view_indices = lambda n, k, it, world, seed=0: keyed(n, k, it, 2, seed)
    stats = {}
    try:
        bm.map_window(be, window, iters=ITERS, stats=stats)
        n_mid = be.gaussians.get_xyz.shape[0]
        bm.map_window(be, window, prune=True)
        bm.map_window(be, window, iters=1)
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
    out["views_per_iteration"] = np.array([len(r["views"]) for r in stats["iterations"]])
    out["used_view_pass"] = np.array(int(getattr(be, "_lvdgs_view_pass", None) is not None))
    return out


def _digest(res):
    h = hashlib.sha256()
    for k in sorted(res):
        if k != "views_per_iteration":
            h.update(k.encode())
            h.update(np.ascontiguousarray(res[k]).tobytes())
    return h.hexdigest()


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)
        res = _run(world)
        q.put((rank, _digest(res), res))
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu_stay_bit_identical_and_match_the_single_process_run():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=900) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, d0, r0), (_, d1, r1) = results
    assert int(r0["used_view_pass"]) == 1   # the views went through MapViewPass, not autograd
    for k in r0:
        if k != "views_per_iteration":
            np.testing.assert_array_equal(r0[k], r1[k], err_msg=k)
    assert d0 == d1
    assert r0["views_per_iteration"].tolist() == [3] * ITERS and r1["views_per_iteration"].tolist() == [3] * ITERS
    torch.manual_seed(7)
    ref = _run(1)
    np.testing.assert_array_equal(ref["counts"], r0["counts"])
    assert int(ref["n_mid"]) == int(r0["n_mid"]) and len(set(ref["counts"].tolist())) > 1   # a densification happened
    for k in ref:
        if k in ("views_per_iteration", "counts", "n_mid", "used_view_pass"):
            continue
        a, b = np.asarray(r0[k], np.float64), np.asarray(ref[k], np.float64)
        assert a.shape == b.shape, k
        if a.size:
            tol = 5e-4 * np.abs(b) + 5e-5 * max(np.abs(b).max(), 1e-30)
            assert (np.abs(a - b) <= tol).all(), (k, np.abs(a - b).max(), np.abs(b).max())
