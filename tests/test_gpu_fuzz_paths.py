"""Randomised sweep over the launch forms added in round 5, each against the form it replaces, BIT FOR BIT on seeded random scenes
(sizes from one Gaussian to 40 k, frames from 17 x 17 to 700 x 400 pixels -- not multiples of the tile --, a case in five an opaque-surface
scene of large flat Gaussians):

* tracking: lvdgs_forward_backward_fused_loss (the forward and the backward blend pass of a tile in one launch, blend_fwd_bwd_kernel)
  against lvdgs_forward + lvdgs_backward_fused_loss, pose-only or with every Gaussian gradient: images, image state, counters,
  gradients, loss, the stepped pose over two iterations;
* mapping: a window of two to five views through fast_mapping.MapWindowBatch (lvdgs_forward_batch: the forward chains of all views in
  five launches; lvdgs_blend_forward_batch; lvdgs_masked_loss_batch; lvdgs_blend_backward_window_batch; lvdgs_map_view_tail_batch) against
  the same window view by view, keyframes with or without static masks: map, poses, statistics and losses after two Adam iterations.

The oracle comparison of the kernels themselves is tests/test_gpu_fuzz.py; this file pins that the batched / fused launches compute what
the single-view launches compute.  LVDGS_FUZZ_PATH_CASES: number of cases (default 8; a sweep of 600 is in profiles/r05_fuzz_sweeps.txt)."""
import ctypes as C
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def _case(seed):
    rng = np.random.default_rng(52000 + seed)
    c = dict(N=int(rng.choice([1, 7, 64, 65, 500, 3000, 12000, 40000])), W=int(rng.integers(17, 700)), H=int(rng.integers(17, 400)))
    if rng.random() < 0.2:
        c.update(kind="surface", N=int(rng.choice([40, 700, 2500])))
    c.update(full=bool(rng.random() < 0.5), n_window=int(rng.integers(2, 6)), masked=bool(rng.random() < 0.6), seed=seed)
    return c


def _register(c):
    from lvdgs import synthetic
    name = "tmp_fuzz_paths_%d" % c["seed"]
    cfg = dict(N=c["N"], W=c["W"], H=c["H"])
    if c.get("kind"):
        cfg["kind"] = c["kind"]
    synthetic.CONFIGS[name] = cfg
    return name


def _tracking(c, workload):
    import bench
    from lvdgs import _lib, rasterizer as _rz
    from lvdgs.fast_tracking import TrackingSession
    dev = torch.device("cuda", 0)
    pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False)
    full = c["full"]
    out = []
    for one_call in (True, False):
        model, cam, _, (N, W, H) = bench.build_scene(workload, 0, dev)
        s = TrackingSession(cam, model, bench.CONFIG, pipe, torch.zeros(3, device=dev), gaussian_gradients=full)
        snaps = []
        for it in range(2):
            if one_call:
                s.step()
            else:
                L, a = s.L, s.a
                stream = _lib.raw_stream(dev)
                num = C.c_int64(0)
                status = L.lvdgs_forward(C.byref(a), C.byref(num), stream)
                if status == _lib.E_CAPACITY:
                    s._size_for_pairs(int(num.value) + int(num.value) // 2)
                    a = s.a
                    status = L.lvdgs_forward(C.byref(a), C.byref(num), stream)
                _lib.check(status, "lvdgs_forward")
                s.num_rendered = a.num_rendered = int(num.value)
                _lib.check(L.lvdgs_backward_fused_loss(C.byref(a), C.byref(s.la), int(_rz.PROPAGATE_OPACITY_GRAD), stream), "lvdgs_backward_fused_loss")
                _lib.check(L.lvdgs_tracking_tail(C.byref(s.la), C.byref(a), C.byref(s.pa), C.c_void_p(s.d_tau.data_ptr()), 1, stream), "lvdgs_tracking_tail")
                s.iterations_enqueued += 1
            torch.cuda.synchronize()
            lay = _lib.StateLayout()
            s.L.lvdgs_state_layout_query(N, max(int(s.num_rendered), 1), W, H, C.byref(lay))
            T_, P_ = ((W + 15) // 16) * ((H + 15) // 16), W * H
            img = s.image
            snap = dict(color=s.color.clone(), depth=s.depth.clone(), opacity=s.opacity.clone(), radii=s.radii.clone(), n_touched=s.n_touched.clone(),
                        ranges=img[lay.img_ranges:lay.img_ranges + 8 * T_].clone(), final_T=img[lay.img_final_T:lay.img_final_T + 4 * P_].clone(),
                        n_contrib=img[lay.img_n_contrib:lay.img_n_contrib + 4 * P_].clone(), d_tau=s.d_tau.clone(), loss=s.loss.clone(),
                        d_a=s.d_a.clone(), d_b=s.d_b.clone(), R=s.R.clone(), T=s.T.clone(), D=torch.tensor(int(s.num_rendered)))
            if full:
                snap.update(d_m3=s.d_m3.clone(), d_m2=s.d_m2.clone(), d_op=s.d_op.clone(), d_sc=s.d_sc.clone(), d_rot=s.d_rot.clone(), d_sh=s.d_sh.clone())
            snaps.append(snap)
        s.finish()
        out.append(snaps)
    for it, (a, b) in enumerate(zip(*out)):
        for k in a:
            assert torch.equal(a[k], b[k]), (c, "tracking", it, k)


def _window(c, workload, batch):
    import bench
    from lvdgs import backend_map
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model, cam, g, _ = bench.build_scene(workload, 0, dev)
    backend, window = bench.build_window(workload, c["n_window"] + 1, dev, model, n_window=c["n_window"], masked=c["masked"])
    before = os.environ.get("LVDGS_MAP_BATCH")
    os.environ["LVDGS_MAP_BATCH"] = "1" if batch else "0"
    try:
        st = {}
        for _ in range(2):
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


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("LVDGS_FUZZ_PATH_CASES", "8")))))
def test_round5_launch_forms_equal_the_forms_they_replace(seed):
    c = _case(seed)
    workload = _register(c)
    try:
        _tracking(c, workload)
        used_b, params_b, poses_b, stats_b, losses_b = _window(c, workload, True)
        used_s, params_s, poses_s, stats_s, losses_s = _window(c, workload, False)
        assert used_b and not used_s, (c, "the batch path did not run (or ran when switched off)")
        for what, xs, ys in (("params", params_b, params_s), ("poses", poses_b, poses_s), ("stats", stats_b, stats_s)):
            for k, (a, b) in enumerate(zip(xs, ys)):
                assert torch.equal(a, b), (c, "window", what, k)
        assert losses_b == losses_s, (c, losses_b, losses_s)
    finally:
        from lvdgs import synthetic
        synthetic.CONFIGS.pop(workload, None)
