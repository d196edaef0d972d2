"""The loops as ONE sequence on the CPU at toy size (lvdgs.slam_sequence.SlamSequence with the dense float64 renderer, the loss oracle
and the brute-force neighbour search in the places where the product has HIP kernels only): map initialisation on frame 0 from an
empty map, then frame by frame tracking -> keyframe test -> seeding -> mapping bursts with densification / pruning / opacity resets
-> pruning pass, the back end's free-running iterations, colour refinement, ATE and PSNR.  The order of the calls is the
reference's FrontEnd.run / BackEnd.run (utils/slam_frontend.py:1740-1899, utils/slam_backend.py:485-609); every loop by itself is
pinned by tests/test_loop_golden.py.  The GPU suite runs the same sequence on the HIP path and compares (tests/test_gpu_sequence.py)."""
import gc
import random

import pytest
import torch


@pytest.fixture(scope="module")
def toy_run():
    import sequence_scene as ss
    from lvdgs import simple_knn
    from lvdgs.slam_sequence import SlamSequence
    torch.manual_seed(0)
    random.seed(0)
    cfg, ds, hooks, knn, psnr_only = ss.toy_sequence_on_cpu()
    real_knn, simple_knn.distCUDA2 = simple_knn.distCUDA2, knn
    try:
        seq = SlamSequence(cfg, ds, ss.empty_map(cfg, "cpu"), ss.PIPE, torch.zeros(3), idle_map_iters=2, **hooks)
        frozen_inside = []
        seq.on_event = lambda e, s: frozen_inside.append(gc.get_freeze_count() > 0)
        seq.run()
        before = seq.eval_rendering(psnr_only)
        ate = seq.eval_ate()
        seq.refine(12)
        after = seq.eval_rendering(psnr_only)
    finally:
        simple_knn.distCUDA2 = real_knn
    return dict(seq=seq, ds=ds, cfg=cfg, before=before, after=after, ate=ate, frozen_inside=frozen_inside)


def test_the_sequence_runs_from_an_empty_map_to_a_trajectory_and_a_map(toy_run):
    seq, s = toy_run["seq"], toy_run["seq"].summary()
    assert s["frames"] == len(toy_run["ds"]) and s["keyframes"] >= 4
    assert seq.kf_indices[0] == 0 and seq.kf_indices == sorted(seq.kf_indices)
    # the window filled (monocular: that is what initialises the system, utils/slam_backend.py:341-342) and then slid
    T = toy_run["cfg"]["Training"]
    assert max(len(w) for w in seq.window_log) == T["window_size"] and seq.backend.initialized and seq.initialized
    assert seq.window_log[-1][0] == seq.kf_indices[-1] and 0 not in seq.window_log[-1]
    # the map changed size under the loops: seeded by every keyframe, densified and pruned in between
    assert s["size_changes_by_densification"] >= 2 and s["size_changes_by_pruning"] >= 1
    assert s["gaussians_first"] < s["gaussians_last"] <= s["gaussians_max"]
    events = [e for e, _ in seq.gaussian_counts]
    assert events[:2] == ["seed", "initialize_map"] and events.count("seed") == s["keyframes"]
    # iteration accounting: every tracked frame ran the tracking loop, every keyframe its burst + pruning pass
    assert s["init_iterations"] == T["init_itr_num"]
    assert 0 < s["tracking_iterations"] <= (s["frames"] - 1) * T["tracking_itr_num"]
    assert s["mapping_iterations"] >= (s["keyframes"] - 1) * T["mapping_itr_nosingle"] and s["prune_passes"] >= s["keyframes"] - 1
    assert seq.backend.iteration_count == s["init_iterations"] + s["mapping_iterations"] + s["prune_passes"]
    # non-keyframes were cleaned, keyframes keep their images
    non_kf = [i for i in seq.cameras if i not in seq.kf_indices]
    assert non_kf and all(seq.cameras[i].original_image is None for i in non_kf)
    assert all(seq.cameras[i].original_image is not None for i in seq.kf_indices)


def test_the_sequence_tracks_and_maps(toy_run):
    """Not a tuning exercise: bounds a broken chain (poses not handed on, a map that is not the one tracked against, seeds in the wrong
    frame) misses by an order of magnitude.  The camera travels ~0.45 units; ATE is after Umeyama alignment with scale (monocular),
    the un-aligned camera-centre error is held against the error of not tracking at all (every pose left at frame 0's)."""
    seq = toy_run["seq"]
    assert toy_run["ate"] is not None and toy_run["ate"] < 0.03, toy_run["ate"]
    err = seq.pose_errors()
    frozen = {i: float((cam.R_gt.double().T @ cam.T_gt.double()).norm()) for i, cam in seq.cameras.items()}   # |camera centre|, frame 0 at the origin
    assert sum(err.values()) < 0.3 * sum(frozen.values()), (err, frozen)
    assert max(err.values()) < 0.1, err
    assert all(r["median_depth"] == r["median_depth"] for r in seq.frame_log)      # (no NaN: the map the tracker sees is opaque)
    assert toy_run["before"]["psnr"] > 15.0 and toy_run["after"]["psnr"] > toy_run["before"]["psnr"] - 0.5, (toy_run["before"], toy_run["after"])
    # the front end's copy of the map is the back end's at the last push
    assert seq.frontend_gaussians is not seq.gaussians
    assert torch.equal(seq.frontend_gaussians.get_xyz, seq.gaussians.get_xyz.detach()) and not seq.frontend_gaussians.get_xyz.requires_grad


def test_the_loops_leave_the_hosts_garbage_collector_as_they_found_it(toy_run):
    """The scoped freeze (``_lib.quiet_gc``): frozen while a product loop runs, nothing left frozen once it has returned."""
    assert toy_run["frozen_inside"] and all(toy_run["frozen_inside"])
    assert gc.get_freeze_count() == 0


def test_expand_dynamic_mask_is_a_square_dilation():
    from lvdgs.slam_sequence import expand_dynamic_mask
    m = torch.zeros(12, 16, dtype=torch.bool)
    m[5, 7] = True
    m[0, 0] = True
    out = expand_dynamic_mask(m, 5)
    want = torch.zeros_like(m)
    want[3:8, 5:10] = True
    want[0:3, 0:3] = True
    assert torch.equal(out, want)


def test_a_keyframe_with_dynamic_objects_seeds_from_the_static_pixels_only():
    """``add_new_keyframe`` (utils/slam_frontend.py:1268-1382) with a detector mask: the masks are stored on the viewpoint, the depth
    map is zero on the dynamic pixels and on a margin around them (9 x 9 dilation on frame 0, 7 x 7 afterwards), the mono depth elsewhere.
    (The sequences with dynamic objects run in the GPU suite: on the dense CPU renderer they take minutes.)"""
    import numpy as np
    import sequence_scene as ss
    from lvdgs.slam_sequence import SlamSequence, expand_dynamic_mask
    cfg, ds, hooks, knn, _ = ss.toy_sequence_on_cpu(dynamic_objects=True, n_frames=2)
    seq = SlamSequence(cfg, ds, ss.empty_map(cfg, "cpu"), ss.PIPE, torch.zeros(3), **hooks)
    for idx, k in ((0, 9), (1, 7)):
        seq.cameras[idx] = vp = seq.new_viewpoint(idx)
        depth = seq.add_new_keyframe(idx, init=idx == 0)
        static = ds.static_mask(idx)
        assert torch.equal(vp.static_mask, static) and torch.equal(vp.dynamic_mask, ~static)
        grown = expand_dynamic_mask(~static, k)
        assert torch.equal(vp.expanded_dynamic_mask, grown) and torch.equal(vp.expanded_static_mask, ~grown)
        assert int(grown.sum()) > int((~static).sum()) > 0
        valid = (vp.original_image.sum(0) > cfg["Training"]["rgb_boundary_threshold"]) & ~grown
        np.testing.assert_array_equal(depth, np.where(valid.numpy(), ds.mono_depths[idx], 0.0).astype(np.float32))
    assert seq.kf_indices == [0, 1] and float(seq.theta) >= 0.0
