"""The views of a mapping window with their blend passes in ONE launch each (fast_mapping.MapWindowBatch: LVDGS_FLAG_NO_BLEND,
lvdgs_blend_forward_batch, lvdgs_blend_backward_fused_loss_batch) against the same window view by view: the same kernels on
the same data, the parameter gradients added in the same order -- every gradient, every loss, every statistic and the map after
Adam steps must be the same BITS."""
import os
import sys

import pytest
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def _run(batch, iters, n_window=8, workload="tmp_window_batch", world=12):
    import bench
    from lvdgs import backend_map, synthetic
    synthetic.CONFIGS.setdefault(workload, dict(N=30000, W=400, H=240))
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model, cam, g, _ = bench.build_scene(workload, 0, dev)
    backend, window = bench.build_window(workload, world, dev, model, n_window=n_window)
    before = os.environ.get("LVDGS_MAP_BATCH")
    os.environ["LVDGS_MAP_BATCH"] = "1" if batch else "0"
    try:
        st = {}
        grads = None
        for _ in range(iters):
            backend_map.map_window(backend, window, iters=1, stats=st)
    finally:
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
    losses = [float(r["loss"]) for r in st["iterations"]] if st.get("iterations") and "loss" in st["iterations"][0] else []
    return used, params, poses, stats, losses


@pytest.mark.parametrize("workload", ["tmp_window_batch", "surface_12k_640x480", "tmp_window_batch_5k_tiles"])   # (the second: lists of 500-1500 entries -- the deep-lists blend build, queued tile-sort segments; the third: a frame of more than 4096 tiles -- no tile order, runs of tiles per XCD)
def test_window_batch_is_the_window_view_by_view_bit_for_bit(workload):
    from lvdgs import synthetic
    synthetic.CONFIGS.setdefault("tmp_window_batch_5k_tiles", dict(N=40000, W=1440, H=912))   # 90 x 57 = 5130 tiles
    used_b, params_b, poses_b, stats_b, losses_b = _run(True, 3, workload=workload)
    used_s, params_s, poses_s, stats_s, losses_s = _run(False, 3, workload=workload)
    assert used_b and not used_s, "the batch path did not run (or ran when switched off)"
    for a, b in zip(params_b, params_s):
        assert torch.equal(a, b)
    for a, b in zip(poses_b, poses_s):
        assert torch.equal(a, b)
    for a, b in zip(stats_b, stats_s):
        assert torch.equal(a, b)
    assert losses_b == losses_s


def test_window_batch_with_more_views_than_one_launch_takes():
    """13 keyframes + 2 random views: more than any of the batched launches takes at once (forward chain 10, blend 11, per-Gaussian pass
    and tails 12) -- every stage runs in two groups, the second group's per-Gaussian pass ADDING to what the first wrote."""
    used_b, params_b, poses_b, stats_b, losses_b = _run(True, 2, n_window=13, world=16)
    used_s, params_s, poses_s, stats_s, losses_s = _run(False, 2, n_window=13, world=16)
    assert used_b and not used_s
    for xs, ys in ((params_b, params_s), (poses_b, poses_s), (stats_b, stats_s)):
        for a, b in zip(xs, ys):
            assert torch.equal(a, b)
    assert losses_b == losses_s


def test_gaussian_backward_batch_adds_to_gradients_that_are_there():
    """lvdgs_gaussian_backward_batch with LVDGS_FLAG_ACCUMULATE_PARAM_GRADS on its FIRST view (the model's parameters already carry
    gradients when the window's passes begin: nothing cleared them since the last backward): the launch starts from what the buffers
    hold and adds the views in order -- the bits of the view-after-view launches (LVDGS_MAP_PBWD_BATCH=0), which add in memory."""
    import bench
    from lvdgs import synthetic
    from lvdgs.fast_mapping import MapViewPass, MapWindowBatch
    workload = "tmp_window_batch"
    synthetic.CONFIGS.setdefault(workload, dict(N=30000, W=400, H=240))
    dev = torch.device("cuda", 0)
    out = {}
    for one_pass in ("1", "0"):
        torch.manual_seed(0)
        model, cam, g, _ = bench.build_scene(workload, 0, dev)
        backend, window = bench.build_window(workload, 6, dev, model, n_window=4)
        G = backend.gaussians
        gen = torch.Generator(device=dev).manual_seed(5)
        for p_ in G.parameters():
            p_.grad = torch.randn(p_.shape, device=dev, generator=gen) * 1e-3 if p_.numel() else None
        views = [backend.viewpoints[k] for k in window]
        before = os.environ.get("LVDGS_MAP_PBWD_BATCH")
        os.environ["LVDGS_MAP_PBWD_BATCH"] = one_pass
        try:
            MapWindowBatch(MapViewPass(dev)).run(backend, views)
        finally:
            if before is None:
                os.environ.pop("LVDGS_MAP_PBWD_BATCH", None)
            else:
                os.environ["LVDGS_MAP_PBWD_BATCH"] = before
        torch.cuda.synchronize()
        out[one_pass] = [p_.grad.clone() for p_ in G.parameters() if p_.grad is not None]
    assert len(out["1"]) >= 5 and all(float(t.abs().sum()) > 0 for t in out["1"])
    for a, b in zip(out["1"], out["0"]):
        assert torch.equal(a, b)


def test_window_batch_grows_its_pair_buffers_like_the_single_view_pass():
    """A view whose pair count exceeds its buffers' capacity (the first iterations of a back end, a view that sees far more of the
    map than the others) makes lvdgs_forward return LVDGS_E_CAPACITY; the view's pass grows its own buffers and re-runs the binning
    (lvdgs_forward_render, still without its blend pass).  Forced here by starting every pass with buffers for 1000 pairs."""
    import bench
    from lvdgs import backend_map, fast_mapping, rasterizer, synthetic
    workload = "tmp_window_batch_cap"
    synthetic.CONFIGS[workload] = dict(N=20000, W=320, H=240)
    dev = torch.device("cuda", 0)

    def run(batch):
        torch.manual_seed(0)
        model, cam, g, _ = bench.build_scene(workload, 0, dev)
        backend, window = bench.build_window(workload, 6, dev, model, n_window=4)
        before = (os.environ.get("LVDGS_MAP_BATCH"), rasterizer._MIN_PAIR_CAPACITY, rasterizer._PAIRS_PER_GAUSSIAN_GUESS)
        os.environ["LVDGS_MAP_BATCH"] = "1" if batch else "0"
        rasterizer._MIN_PAIR_CAPACITY, rasterizer._PAIRS_PER_GAUSSIAN_GUESS = 1000, 0
        try:
            for _ in range(2):
                backend_map.map_window(backend, window, iters=1)
        finally:
            rasterizer._MIN_PAIR_CAPACITY, rasterizer._PAIRS_PER_GAUSSIAN_GUESS = before[1], before[2]
            if before[0] is None:
                os.environ.pop("LVDGS_MAP_BATCH", None)
            else:
                os.environ["LVDGS_MAP_BATCH"] = before[0]
        torch.cuda.synchronize()
        b = getattr(backend, "_lvdgs_window_batch", None)
        caps = [p.cap for p in b.passes] if b is not None else [backend._lvdgs_view_pass.cap]
        return [p.detach().clone() for p in backend.gaussians.parameters()], caps

    params_b, caps_b = run(True)
    params_s, caps_s = run(False)
    assert len(caps_b) > 1 and min(caps_b) > 1000 and caps_s[0] > 1000   # every pass had to grow
    for a, b in zip(params_b, params_s):
        assert torch.equal(a, b)


@pytest.mark.parametrize("masked", [False, True])
def test_window_batch_across_densification_pruning_and_an_opacity_reset(masked):
    """The map changes size BETWEEN batched iterations -- densify / prune every 3rd iteration, the opacity reset of the non-visible on
    the 5th, the window's pruning pass at the end (reference utils/slam_backend.py:318-376 at a short period) -- so every per-view
    buffer of the batch, the cached mask / mono-depth bytes and the memory verdict meet a new N several times.  Batched and view by
    view must stay the same bits through all of it."""
    import bench
    from lvdgs import backend_map, synthetic
    workload = "tmp_window_batch"
    synthetic.CONFIGS.setdefault(workload, dict(N=30000, W=400, H=240))
    dev = torch.device("cuda", 0)

    def run(batch):
        torch.manual_seed(0)
        model, cam, g, _ = bench.build_scene(workload, 0, dev)
        backend, window = bench.build_window(workload, 12, dev, model, n_window=8, masked=masked)
        backend.gaussian_update_every, backend.gaussian_update_offset, backend.gaussian_reset = 3, 1, 5
        model.unique_kfIDs = (torch.arange(model.get_xyz.shape[0]) % 12 + 1).to(torch.int32)   # (seeded by keyframes 1..12: the pruning pass looks at the newest three)
        backend.opt_params.densify_grad_threshold = 2e-5
        before = os.environ.get("LVDGS_MAP_BATCH")
        os.environ["LVDGS_MAP_BATCH"] = "1" if batch else "0"
        sizes, runs = [], []
        try:
            for _ in range(8):
                backend_map.map_window(backend, window, iters=1)
                sizes.append(int(backend.gaussians.get_xyz.shape[0]))
                runs.append(getattr(getattr(backend, "_lvdgs_window_batch", None), "runs", 0))
            backend_map.map_window(backend, window, prune=True)
            sizes.append(int(backend.gaussians.get_xyz.shape[0]))
            backend_map.map_window(backend, window, iters=2)
            sizes.append(int(backend.gaussians.get_xyz.shape[0]))
        finally:
            if before is None:
                os.environ.pop("LVDGS_MAP_BATCH", None)
            else:
                os.environ["LVDGS_MAP_BATCH"] = before
        torch.cuda.synchronize()
        G = backend.gaussians
        state = [p.detach().clone() for p in G.parameters()] + [G.max_radii2D.clone(), G.xyz_gradient_accum.clone(), G.denom.clone()]
        state += [G.optimizer.state[p][k].clone() for p in G.parameters() if p in G.optimizer.state for k in ("exp_avg", "exp_avg_sq")]
        occ = [backend.occ_aware_visibility[k].clone() for k in window]
        return sizes, runs, state, occ, G.unique_kfIDs.clone(), G.n_obs.clone()

    sizes_b, runs_b, state_b, occ_b, kf_b, obs_b = run(True)
    sizes_s, runs_s, state_s, occ_s, kf_s, obs_s = run(False)
    assert runs_b == list(range(1, 9)) and runs_s == [0] * 8          # batched in every iteration / never
    assert len(set(sizes_b[:8])) >= 3 and sizes_b[8] < sizes_b[7], sizes_b     # densified / pruned several times, then the pruning pass
    assert sizes_b == sizes_s
    for a, b in zip(state_b + occ_b, state_s + occ_s):
        assert a.shape == b.shape and torch.equal(a, b)
    assert torch.equal(kf_b, kf_s) and torch.equal(obs_b, obs_s)
