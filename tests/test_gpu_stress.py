"""Sizes beyond the benchmark: BASELINE.json configs[4]'s 2M Gaussians at 1920x1280, and a scene whose
per-tile lists are thousands of entries long.  No oracle at these sizes: determinism, finiteness and the
list invariants are checked."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _invariants(f, N):
    r = f["ranges"].astype(np.int64)
    assert int((r[:, 1] - r[:, 0]).sum()) == f["num_rendered"] == int(f["tiles_touched"].sum())
    depth_bits = f["rec"][:, 9].view(np.uint32)[f["point_list"]].astype(np.uint64)
    keys = (f["tile_keys"].astype(np.uint64) << np.uint64(32)) | depth_bits
    assert np.all(keys[1:] >= keys[:-1])
    same = keys[1:] == keys[:-1]
    assert np.all(f["point_list"][1:][same] > f["point_list"][:-1][same])
    np.testing.assert_array_equal(np.bincount(f["point_list"], minlength=N), f["tiles_touched"])


def test_two_million_gaussians_1920x1280():
    import hip_runner
    from lvdgs import synthetic
    N, W, H = 2_000_000, 1920, 1280
    g = synthetic.make_gaussians(N, W, H, seed=1)
    cam = synthetic.make_camera(W, H, pose_seed=2)
    grads = synthetic.make_image_grads(W, H, 0)
    bg = torch.zeros(3)
    f, b = hip_runner.run_hip(g, cam, W, H, bg, grads=grads)
    f2, b2 = hip_runner.run_hip(g, cam, W, H, bg, grads=grads)
    assert f["num_rendered"] > 5_000_000
    _invariants(f, N)
    for k in ("color", "depth", "opacity", "point_list", "n_touched", "n_contrib"):
        np.testing.assert_array_equal(f[k], f2[k])
    for k in b:
        assert np.isfinite(b[k]).all(), k
        np.testing.assert_array_equal(b[k], b2[k])


@pytest.mark.parametrize("N,longest", [(60_000, 10_000), (140_000, 16_384)])
def test_very_long_tile_lists(N, longest):
    """Segments of more than 2048 entries sort on LDS with 1024 threads, beyond 16384 in place on global memory."""
    import hip_runner
    from lvdgs import synthetic
    W, H = 64, 64
    g = synthetic.make_gaussians(N, W, H, seed=2, r_min=4.0, r_max=9.0, z_min=1.0, z_max=3.0)
    g["opacities"][:] = 0.02  # faint: nothing saturates, every list is walked to its end
    cam = synthetic.make_camera(W, H)
    # (every tile of the rectangles listed: at opacity 0.02 tile culling would shorten the lists below the sizes under test)
    f, b = hip_runner.run_hip(g, cam, W, H, torch.zeros(3), grads=synthetic.make_image_grads(W, H, 1), tile_cull=False)
    assert (f["ranges"][:, 1] - f["ranges"][:, 0]).max() > longest
    _invariants(f, N)
    assert f["n_contrib"].max() > 5_000
    for k in b:
        assert np.isfinite(b[k]).all(), k


@pytest.mark.parametrize("kind", ["one_depth", "two_depths", "few_visible"])
def test_depth_order_with_degenerate_depth_distributions(kind):
    """The depth order must stay exact when thousands of Gaussians share one depth -- ids break the
    ties -- and when almost everything is culled."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
    import hip_runner
    import oracle as orc
    from lvdgs import synthetic
    N, W, H = 40_000, 160, 96
    g = synthetic.make_gaussians(N, W, H, seed=3, r_min=0.5, r_max=1.5)
    cam = synthetic.make_camera(W, H)
    z = g["means3D"][:, 2].clone()
    if kind == "one_depth":
        znew = torch.full_like(z, 2.0)
    elif kind == "two_depths":
        znew = torch.where(torch.arange(N) % 3 == 0, torch.full_like(z, 1.5), torch.full_like(z, 7.25))
    else:
        znew = torch.where(torch.arange(N) % 97 == 0, z, -z)  # ~1 % in front of the camera
    g["means3D"][:, :2] *= (znew / z).abs()[:, None]
    g["means3D"][:, 2] = znew
    bg = torch.zeros(3)
    f_hip, _ = hip_runner.run_hip(g, cam, W, H, bg)
    f_ora, _ = hip_runner.run_oracle(orc, g, cam, W, H, bg)
    np.testing.assert_array_equal(f_hip["radii"], f_ora["radii"])
    hip_runner.check_pair_lists(f_hip, f_ora, W, H)   # the oracle's lists less pairs that reach no pixel, same order
    f_all, _ = hip_runner.run_hip(g, cam, W, H, bg, tile_cull=False)
    np.testing.assert_array_equal(f_all["point_list"], f_ora["ids_sorted"])
    np.testing.assert_array_equal(f_all["ranges"], f_ora["ranges"])
    # inside every tile: ascending depth, ascending id among equal depths
    pl, rg, d = f_hip["point_list"], f_hip["ranges"], f_hip["rec"][:, 9]
    for b, e in rg[rg[:, 1] > rg[:, 0]][:: max(1, len(rg) // 50)]:
        ids, dd = pl[b:e].astype(np.int64), d[pl[b:e]]
        assert np.all(dd[1:] >= dd[:-1]) and np.all(ids[1:][dd[1:] == dd[:-1]] > ids[:-1][dd[1:] == dd[:-1]])


def test_image_with_more_tiles_than_lds_counters_takes_the_radix_grouping():
    """2576x1712 is 161 x 107 = 17227 tiles, above the 16384 LDS counters of the counting path."""
    import hip_runner
    from lvdgs import synthetic
    N, W, H = 150_000, 2576, 1712
    g = synthetic.make_gaussians(N, W, H, seed=4)
    cam = synthetic.make_camera(W, H)
    grads = synthetic.make_image_grads(W, H, 2)
    f, b = hip_runner.run_hip(g, cam, W, H, torch.zeros(3), grads=grads)
    f2, b2 = hip_runner.run_hip(g, cam, W, H, torch.zeros(3), grads=grads)
    _invariants(f, N)
    for k in ("color", "depth", "opacity", "point_list", "ranges", "n_touched", "n_contrib"):
        np.testing.assert_array_equal(f[k], f2[k])
    for k in b:
        assert np.isfinite(b[k]).all(), k
        np.testing.assert_array_equal(b[k], b2[k])


def test_radix_grouping_gives_the_same_lists_as_the_counting_path():
    """LVDGS_FORCE_RADIX_GROUPING (read once per process) selects the radix path at any image size: run the oracle
    parity cases in a fresh interpreter with it set."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LVDGS_FORCE_RADIX_GROUPING="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_parity.py"), "-q", "-m", "gpu", "-x",
                        "-k", "test_forward_and_backward_match_oracle or overflow or more_than_64_tiles", "-p", "no:cacheprovider", "--durations=8"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=600)
    print(r.stdout[-1500:])   # (shown with -s: where the child's time went)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
