"""LVDGS_FLAG_POSE_ONLY: the tracking backward sized to what tracking consumes.  The reference's tracking optimiser holds the
camera pose and the exposure alone (utils/slam_frontend.py:1468-1490, stepped at :1520): the pose-only backward must give
those gradients bit for bit as the full backward does, at the sizes of BASELINE.json's configurations."""
import os
import sys
from types import SimpleNamespace

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))


@pytest.mark.parametrize("workload,monocular", [("cfg3_500k_1920x1080", True), ("kitti07_geom", True), ("kitti07_geom", False),
                                                ("surface_100k_1920x1080", True), ("cfg2_100k_640x480", False)])
def test_pose_and_exposure_gradients_equal_the_full_backwards_bit_for_bit(workload, monocular):
    import bench
    from lvdgs.fast_tracking import TrackingSession
    dev = torch.device("cuda", torch.cuda.current_device())
    cfg = {k: (dict(v) if isinstance(v, dict) else v) for k, v in bench.CONFIG.items()}
    cfg["Training"]["monocular"] = monocular   # False: the RGB-D tracking loss, i.e. the kernels with the depth-gradient terms
    pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False)
    res = {}
    for full in (True, False):
        model, cam, g, (N, W, H) = bench.build_scene(workload, 2, dev)
        with torch.no_grad():
            cam.exposure_a.fill_(0.03); cam.exposure_b.fill_(-0.02)
        s = TrackingSession(cam, model, cfg, pipe, torch.zeros(3, device=dev), gaussian_gradients=full)
        assert s.pose_only == (not full)
        taus, rest = [], []
        for _ in range(3):   # three iterations: the pose moves, so later iterations test other views
            s.step()
            taus.append(s.d_tau.clone())
            rest.append(torch.cat([s.loss.reshape(1), s.d_a.reshape(1), s.d_b.reshape(1)]).clone())
        n = s.finish()
        res[full] = (torch.stack(taus), torch.stack(rest), s.R.clone(), s.T.clone(), cam.exposure_a.detach().clone(), cam.exposure_b.detach().clone(), n,
                     s.color.clone(), s.n_touched.clone())
        del s, model
    a, b = res[True], res[False]
    assert bool(a[0].abs().sum() > 0)
    for x, y, what in zip(a, b, ("dL/dtau", "loss, d exposure", "R", "T", "exposure_a", "exposure_b", "iterations", "image", "n_touched")):
        assert (x == y) if isinstance(x, int) else torch.equal(x, y), what


def test_a_map_with_view_dependent_colours_keeps_the_full_backward():
    """Active SH degree > 0: the colour moves with the camera centre and its gradient feeds dL/dtau, so the session does not
    ask for the pose-only passes (the library would refuse them)."""
    import bench
    from lvdgs.fast_tracking import TrackingSession
    from lvdgs import synthetic
    from lvdgs.gaussian_model import GaussianModel
    dev = torch.device("cuda", torch.cuda.current_device())
    _, cam, _, (N, W, H) = bench.build_scene("cfg1_10k_640x480", 1, dev)
    g = synthetic.make_gaussians(N, W, H, seed=0, sh_degree=1)
    model = GaussianModel.from_activated(g["means3D"], g["scales"], g["rotations"], g["opacities"], shs=g["shs"], sh_degree=1, device=dev)
    s = TrackingSession(cam, model, bench.CONFIG, SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False), torch.zeros(3, device=dev))
    assert not s.pose_only and s.d_sh is not None
    s.step()
    assert s.finish() == 1 and bool(torch.isfinite(s.d_tau).all())
