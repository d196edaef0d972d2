"""Map bookkeeping of GaussianModel on CPU tensors: optimizer-state surgery, densification rules, opacity
resets, learning-rate schedule and the PLY round trip (members used at utils/slam_backend.py:76-145, 303-380)."""
import math

import numpy as np
import pytest
import torch

from lvdgs.gaussian_model import GaussianModel, build_rotation, get_expon_lr_func, inverse_sigmoid

OPT = dict(position_lr_init=0.0016, position_lr_final=0.00016, position_lr_delay_mult=0.01, position_lr_max_steps=30000,
           feature_lr=0.0025, opacity_lr=0.05, scaling_lr=0.001, rotation_lr=0.001, percent_dense=0.01)


def _model(n=50, seed=0, scale=0.05):
    g = torch.Generator().manual_seed(seed)
    m = GaussianModel.from_activated(
        means3D=torch.randn(n, 3, generator=g), scales=scale * (0.5 + torch.rand(n, 3, generator=g)),
        rotations=torch.nn.functional.normalize(torch.randn(n, 4, generator=g)), opacities=torch.rand(n, 1, generator=g),
        colors=torch.rand(n, 3, generator=g), device="cpu")
    m.unique_kfIDs = torch.arange(n, dtype=torch.int32) % 5
    m.n_obs = torch.arange(n, dtype=torch.int32) % 3
    m.init_lr(6.0)
    m.training_setup(OPT)
    return m


def _one_adam_step(m, seed=1):
    g = torch.Generator().manual_seed(seed)
    for p in m.parameters():
        p.grad = torch.randn(p.shape, generator=g)
    m.optimizer.step()


def test_param_groups_and_learning_rates():
    m = _model()
    lrs = {g["name"]: g["lr"] for g in m.optimizer.param_groups}
    assert lrs == {"xyz": pytest.approx(0.0016 * 6.0), "f_dc": 0.0025, "f_rest": 0.0025 / 20, "opacity": 0.05,
                   "scaling": pytest.approx(0.001 * 6.0), "rotation": 0.001}
    assert m.optimizer.defaults["eps"] == 1e-15
    assert m.update_learning_rate(0) == pytest.approx(0.0016 * 6.0)  # the delay multiplier only bites with delay steps (none here)
    f = get_expon_lr_func(1e-2, 1e-4, max_steps=100)
    assert f(0) == pytest.approx(1e-2) and f(100) == pytest.approx(1e-4) and f(50) == pytest.approx(1e-3) and f(500) == pytest.approx(1e-4)
    assert m.update_learning_rate(30000) == pytest.approx(0.00016 * 6.0)
    assert [g["lr"] for g in m.optimizer.param_groups if g["name"] == "xyz"][0] == pytest.approx(0.00016 * 6.0)


def test_prune_keeps_rows_and_adam_moments_aligned():
    m = _model()
    _one_adam_step(m)
    before = {g["name"]: (g["params"][0].detach().clone(), m.optimizer.state[g["params"][0]]["exp_avg"].clone(),
                          m.optimizer.state[g["params"][0]]["exp_avg_sq"].clone()) for g in m.optimizer.param_groups}
    ids, obs = m.unique_kfIDs.clone(), m.n_obs.clone()
    mask = torch.zeros(50, dtype=torch.bool)
    mask[::3] = True
    m.max_radii2D = torch.arange(50.0)
    m.prune_points(mask)
    keep = ~mask
    assert m.get_xyz.shape[0] == int(keep.sum())
    for g in m.optimizer.param_groups:
        p = g["params"][0]
        val, avg, sq = before[g["name"]]
        assert torch.equal(p.detach(), val[keep]) and p.requires_grad
        assert torch.equal(m.optimizer.state[p]["exp_avg"], avg[keep]) and torch.equal(m.optimizer.state[p]["exp_avg_sq"], sq[keep])
    assert m._xyz is [g for g in m.optimizer.param_groups if g["name"] == "xyz"][0]["params"][0]
    assert torch.equal(m.unique_kfIDs, ids[keep]) and torch.equal(m.n_obs, obs[keep]) and torch.equal(m.max_radii2D, torch.arange(50.0)[keep])
    assert m.unique_kfIDs.device.type == "cpu" and m.n_obs.device.type == "cpu"
    assert len(m.optimizer.state) == 6
    m.prune_points(m.unique_kfIDs >= 0)  # BackEnd.reset (slam_backend.py:89)
    assert m.get_xyz.shape[0] == 0 and m.get_features.shape[0] == 0


def test_densify_clone_split_and_prune_rules():
    m = _model(n=40, scale=0.05)
    extent = 2.0  # percent_dense * extent = 0.02: scales in [0.025, 0.075] are all "large" -> split
    with torch.no_grad():
        m._scaling[:10] = math.log(0.005)  # the first ten are small -> clone
        m._opacity[:] = 2.0
        m._opacity[35:] = inverse_sigmoid(torch.tensor(0.001))  # transparent -> pruned
    m.xyz_gradient_accum = torch.zeros(40, 1)
    m.denom = torch.ones(40, 1)
    m.xyz_gradient_accum[[0, 1, 20, 21, 22]] = 1.0   # above threshold
    m.denom[5] = 0                                    # 0/0 -> NaN -> treated as 0
    xyz0, sc0, ids0 = m.get_xyz.detach().clone(), m.get_scaling.detach().clone(), m.unique_kfIDs.clone()
    m.generator = torch.Generator().manual_seed(7)
    m.densify_and_prune(max_grad=0.5, min_opacity=0.005, extent=extent, max_screen_size=None)
    # 40 + 2 clones + 3*2 split children - 3 split parents - 5 transparent = 40
    assert m.get_xyz.shape[0] == 40
    xyz, sc = m.get_xyz.detach(), m.get_scaling.detach()
    for i in (0, 1):  # clones are exact copies
        assert int((xyz == xyz0[i]).all(dim=1).sum()) == 2
    for i in (20, 21, 22):  # parents are gone, children are 1.6x smaller
        assert int((xyz == xyz0[i]).all(dim=1).sum()) == 0
        assert int(torch.isclose(sc, sc0[i] / 1.6, rtol=1e-5).all(dim=1).sum()) == 2
    assert sorted(m.unique_kfIDs.tolist()) == sorted(
        [int(ids0[i]) for i in range(40) if i not in (20, 21, 22) and i < 35] + [int(ids0[0]), int(ids0[1])]
        + [int(ids0[i]) for i in (20, 21, 22) for _ in range(2)])
    assert m.xyz_gradient_accum.shape == (40, 1) and float(m.xyz_gradient_accum.abs().sum()) == 0.0
    assert m.max_radii2D.shape == (40,) and m.denom.shape == (40, 1)
    for g in m.optimizer.param_groups:  # every group was rebuilt consistently
        assert g["params"][0].shape[0] == 40


def test_screen_size_and_world_size_pruning():
    m = _model(n=20, scale=0.01)
    m.xyz_gradient_accum, m.denom = torch.zeros(20, 1), torch.ones(20, 1)
    with torch.no_grad():
        m._opacity[:] = 3.0
        m._scaling[3] = math.log(0.5)  # > 0.1 * extent
    m.max_radii2D = torch.zeros(20)
    m.max_radii2D[7] = 25.0
    m.densify_and_prune(max_grad=1.0, min_opacity=0.005, extent=1.0, max_screen_size=20)
    # the world-size rule removes #3; the screen-size rule reads max_radii2D AFTER the clone / split step reset it
    # to zero (published behaviour), so #7 survives
    assert m.get_xyz.shape[0] == 19


def test_split_children_follow_the_parent_covariance():
    m = _model(n=1)
    with torch.no_grad():
        m._scaling[:] = torch.log(torch.tensor([[1.0, 0.001, 0.001]]))
        m._rotation[:] = torch.tensor([[math.cos(math.pi / 8), 0.0, 0.0, math.sin(math.pi / 8)]])  # 45 deg about z
    m.generator = torch.Generator().manual_seed(3)
    centre = m.get_xyz.detach().clone()[0]
    m.densify_and_split(torch.ones(1, 1), 0.5, scene_extent=1.0, N=2)
    d = m.get_xyz.detach() - centre
    axis = build_rotation(torch.tensor([[math.cos(math.pi / 8), 0.0, 0.0, math.sin(math.pi / 8)]]))[0][:, 0]
    assert d.shape == (2, 3)
    assert torch.all((d - (d @ axis)[:, None] * axis[None, :]).norm(dim=1) < 0.02)  # offsets lie along the long axis


def test_densification_stats_accumulate_screen_gradient_norms():
    m = _model(n=6)
    vs = torch.zeros(6, 3, requires_grad=True)
    vs.grad = torch.tensor([[3.0, 4.0, 9.0]] * 6)
    vis = torch.tensor([True, False, True, True, False, False])
    m.add_densification_stats(vs, vis)
    m.add_densification_stats(vs, vis)
    assert torch.equal(m.xyz_gradient_accum.squeeze(1), torch.tensor([10.0, 0, 10, 10, 0, 0]))
    assert torch.equal(m.denom.squeeze(1), torch.tensor([2.0, 0, 2, 2, 0, 0]))


def test_opacity_resets():
    import lvdgs.gaussian_model as gm
    m = _model(n=10)
    _one_adam_step(m)
    with torch.no_grad():
        m._opacity[:] = 2.0
    m.reset_opacity()
    assert torch.allclose(m.get_opacity, torch.full((10, 1), 0.01), atol=1e-7)
    st = m.optimizer.state[m._opacity]
    assert float(st["exp_avg"].abs().sum()) == 0.0 and float(st["exp_avg_sq"].abs().sum()) == 0.0
    with torch.no_grad():
        m._opacity[:] = 2.0
    seen = torch.zeros(10, dtype=torch.bool)
    seen[:4] = True
    m.reset_opacity_nonvisible([seen])
    assert torch.allclose(m.get_opacity[4:], torch.full((6, 1), 0.4), atol=1e-6)
    assert torch.allclose(m._opacity[:4].detach(), torch.sigmoid(torch.tensor(2.0)).expand(4, 1))  # upstream's behaviour
    try:
        gm.UPSTREAM_NONVISIBLE_RESET = False
        with torch.no_grad():
            m._opacity[:] = 2.0
        m.reset_opacity_nonvisible([seen])
        assert torch.allclose(m._opacity[:4].detach(), torch.full((4, 1), 2.0))
    finally:
        gm.UPSTREAM_NONVISIBLE_RESET = True


def test_extend_appends_with_zero_moments_and_kf_ids():
    m = _model(n=8)
    _one_adam_step(m)
    avg = m.optimizer.state[m._xyz]["exp_avg"].clone()
    feats = torch.zeros(5, 3, 1)
    feats[:, :, 0] = 0.25
    m.extend_from_pcd(torch.ones(5, 3), feats, torch.zeros(5, 3), torch.tensor([[1.0, 0, 0, 0]] * 5), torch.zeros(5, 1), kf_id=17)
    assert m.get_xyz.shape == (13, 3) and m.get_features.shape == (13, 1, 3)
    st = m.optimizer.state[m._xyz]
    assert torch.equal(st["exp_avg"][:8], avg) and float(st["exp_avg"][8:].abs().sum()) == 0.0
    assert m.unique_kfIDs.tolist()[8:] == [17] * 5 and m.n_obs.tolist()[8:] == [0] * 5
    assert torch.equal(m.get_features[8:, 0], torch.full((5, 3), 0.25))


def test_ply_round_trip(tmp_path):
    m = _model(n=12)
    path = str(tmp_path / "point_cloud" / "point_cloud.ply")
    m.save_ply(path)
    raw = open(path, "rb").read()
    assert raw.startswith(b"ply\nformat binary_little_endian 1.0\nelement vertex 12\n")
    assert m.construct_list_of_attributes() == ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2", "opacity",
                                                "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
    assert len(raw) - raw.index(b"end_header\n") - len(b"end_header\n") == 12 * 17 * 4
    r = GaussianModel(0, device="cpu")
    r.load_ply(path)
    for a, b in zip(m.parameters(), r.parameters()):
        assert torch.equal(a.detach(), b.detach())


def test_empty_model_accepts_a_first_keyframe_after_training_setup():
    """The reference builds the optimizer before the first keyframe exists (empty tensors), then extends."""
    m = GaussianModel(0, device="cpu")
    m.init_lr(6.0)
    m.training_setup(OPT)
    feats = torch.zeros(4, 3, 1)
    m.extend_from_pcd(torch.zeros(4, 3), feats, torch.zeros(4, 3), torch.tensor([[1.0, 0, 0, 0]] * 4), torch.zeros(4, 1), kf_id=0)
    assert m.get_xyz.shape == (4, 3) and m._features_rest.shape == (4, 0, 3)
    _one_adam_step(m)
    assert m.optimizer.state[m._xyz]["exp_avg"].shape == (4, 3)
    assert np.isfinite(m.get_xyz.detach().numpy()).all()
