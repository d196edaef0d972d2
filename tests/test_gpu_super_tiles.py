"""Two-level grouping (LVDGS_FLAG_SUPER_TILES; csrc/binning.hip): (Gaussian, super-tile) pairs scattered and depth-sorted per 64 x 64-pixel
super-tile, the tiles' lists read off the sorted super lists.  A hint: the sorted pair list, the tile ranges, every image, every
gradient must be the SAME BITS with and without it -- on opaque surfaces (where it pays), on small blobs (where it does not), on a frame
whose size is no multiple of the super-tile, on rectangles of more than 64 tiles and of more than 64 super-tiles, when the pair
capacity overflows, through the autograd API, the tracking session and the mapping window.

In the backward the same flag selects preprocess_bwd's helper-wave kernels (csrc/preprocess.hip: the second half of a large-footprint wave's
pair records summed by a wave of its own): the gradient comparisons below are also "helper waves == one wave, part after part"."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
pytestmark = pytest.mark.gpu


class _super_tiles:
    """rasterizer.super_tiles_flag forced on / off for the block (what LVDGS_SUPER_TILES=1 / 0 in the environment does)."""

    def __init__(self, on):
        self.on = on

    def __enter__(self):
        from lvdgs import rasterizer
        self.before = rasterizer._SUPER_TILES_ENV
        rasterizer._SUPER_TILES_ENV = "1" if self.on else "0"

    def __exit__(self, *exc):
        from lvdgs import rasterizer
        rasterizer._SUPER_TILES_ENV = self.before
        return False


CASES = {
    "surface_12k_640x480": dict(kind="surface", N=12000, W=640, H=480),
    "surface_odd_frame": dict(kind="surface", N=6000, W=610, H=370),            # 39 x 24 tiles: partial super-tiles on both edges
    "blobs_40k_800x600": dict(kind="blobs", N=40000, W=800, H=600),
    "huge_footprints": dict(kind="surface", N=1500, W=1920, H=1080, r_min=150.0, r_max=420.0),   # rectangles of > 64 super-tiles
}


def _scene(c):
    from lvdgs import synthetic
    if c["kind"] == "surface":
        return synthetic.make_surface_gaussians(c["N"], c["W"], c["H"], seed=3, **{k: c[k] for k in ("r_min", "r_max") if k in c})
    return synthetic.make_gaussians(c["N"], c["W"], c["H"], seed=3)


@pytest.mark.parametrize("case", list(CASES))
def test_super_tiles_leave_every_output_bit_identical(case):
    import hip_runner
    from lvdgs import synthetic
    c = CASES[case]
    W, H = c["W"], c["H"]
    g = _scene(c)
    cam = synthetic.make_camera(W, H)
    grads = synthetic.make_image_grads(W, H, 1)
    out = {}
    for on in (False, True):
        with _super_tiles(on):
            out[on] = hip_runner.run_hip(g, cam, W, H, torch.zeros(3), grads=grads)
    (f0, b0), (f1, b1) = out[False], out[True]
    assert f1["num_rendered"] == f0["num_rendered"] > 0   # (huge_footprints overflows the first guess of the pair capacity in whichever run comes first: the re-run's results are what is compared)
    for k in ("point_list", "ranges", "n_contrib", "final_T", "color", "depth", "opacity", "radii", "n_touched", "slot_base", "tiles_touched"):
        assert np.array_equal(f1[k], f0[k]), k
    for k, v in b0.items():
        assert np.array_equal(b1[k], v), k


def test_super_tiles_when_the_pair_capacity_overflows():
    """The first frame runs with room for 1000 pairs: LVDGS_E_CAPACITY, the buffers are grown and binning + blend re-run
    (lvdgs_forward_render) -- both through the two-level grouping."""
    import hip_runner
    from lvdgs import rasterizer, synthetic
    c = CASES["surface_12k_640x480"]
    W, H = c["W"], c["H"]
    g, cam = _scene(c), synthetic.make_camera(c["W"], c["H"])
    with _super_tiles(False):
        want, _ = hip_runner.run_hip(g, cam, W, H, torch.zeros(3))
    before = (rasterizer._MIN_PAIR_CAPACITY, rasterizer._PAIRS_PER_GAUSSIAN_GUESS, dict(rasterizer._PAIR_CAPACITY))
    rasterizer._MIN_PAIR_CAPACITY, rasterizer._PAIRS_PER_GAUSSIAN_GUESS = 1000, 0
    rasterizer._PAIR_CAPACITY.clear()
    try:
        with _super_tiles(True):
            got, _ = hip_runner.run_hip(g, cam, W, H, torch.zeros(3))
    finally:
        rasterizer._MIN_PAIR_CAPACITY, rasterizer._PAIRS_PER_GAUSSIAN_GUESS = before[0], before[1]
        rasterizer._PAIR_CAPACITY.clear(); rasterizer._PAIR_CAPACITY.update(before[2])
    assert got["overflowed"] and got["num_rendered"] == want["num_rendered"]
    for k in ("point_list", "ranges", "n_contrib", "color", "depth", "opacity", "n_touched"):
        assert np.array_equal(got[k], want[k]), k


def test_tracking_session_and_mapping_window_with_super_tiles():
    """The fused tracking iteration (lvdgs_forward_backward_fused_loss) and a mapping window's view-by-view passes on the opaque-surface
    scene: poses, exposures and the map after a few optimiser steps are the same bits with the hint forced on and forced off; and the
    automatic setting (pairs per Gaussian of the previous frame) switches it on for this scene by itself."""
    import bench
    from lvdgs import _lib, backend_map, synthetic
    from lvdgs.fast_tracking import TrackingSession
    from types import SimpleNamespace
    dev = torch.device("cuda", 0)
    pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False)
    workload = "surface_12k_640x480"
    res = {}
    for on in (False, True):
        with _super_tiles(on):
            torch.manual_seed(0)
            model, cam, _, _ = bench.build_scene(workload, 0, dev)
            s = TrackingSession(cam, model, bench.CONFIG, pipe, torch.zeros(3, device=dev), gaussian_gradients=True, converged_threshold=-1.0)
            for _ in range(4):
                s.step()
            s.finish()
            track = [t.detach().clone() for t in (s.R, s.T, cam.exposure_a, cam.exposure_b, s.color, s.depth, s.d_m3, s.d_sc, s.loss)]
            maps = []
            for batch in ("0", "1"):      # the mapping window view by view (lvdgs_forward per view) and batched (lvdgs_forward_batch)
                before = os.environ.get("LVDGS_MAP_BATCH")
                os.environ["LVDGS_MAP_BATCH"] = batch
                try:
                    torch.manual_seed(0)
                    model2, _, _, _ = bench.build_scene(workload, 0, dev)
                    be, window = bench.build_window(workload, 6, dev, model2, n_window=4, masked=True)
                    for _ in range(3):
                        backend_map.map_window(be, window, iters=1)
                finally:
                    if before is None:
                        os.environ.pop("LVDGS_MAP_BATCH", None)
                    else:
                        os.environ["LVDGS_MAP_BATCH"] = before
                torch.cuda.synchronize()
                maps += [p.detach().clone() for p in be.gaussians.parameters()]
                if batch == "1":
                    wb = be._lvdgs_window_batch
                    assert all(bool(p_.a.flags & _lib.FLAG_SUPER_TILES) is on for p_ in wb.passes[:6])
            res[on] = track + maps
            flag_in_session = bool(s.a.flags & _lib.FLAG_SUPER_TILES)
            assert flag_in_session is on
    for a, b in zip(res[False], res[True]):
        assert torch.equal(a, b)
    # automatic: the second step of a session on this scene carries the hint, on small blobs it does not
    from lvdgs import rasterizer
    assert rasterizer._SUPER_TILES_ENV == "auto"
    for workload, want in (("surface_12k_640x480", True), ("cfg2_100k_640x480", False)):
        model, cam, _, _ = bench.build_scene(workload, 0, dev)
        s = TrackingSession(cam, model, bench.CONFIG, pipe, torch.zeros(3, device=dev))
        s.step(); s.step()
        s.finish()
        assert bool(s.a.flags & _lib.FLAG_SUPER_TILES) is want, (workload, s.num_rendered, s.N)
