"""Trajectory metrics: Umeyama alignment recovers a known similarity, the ATE statistics and the fallbacks of
utils/eval_utils_0806.py:33-98."""
import numpy as np
import pytest
import torch

from lvdgs import eval_utils as ev
from lvdgs.pose_utils import SE3_exp


def _trajectory(n=25, seed=0):
    rng = np.random.default_rng(seed)
    poses, T = [], np.eye(4)
    for _ in range(n):
        step = SE3_exp(torch.tensor(np.concatenate([rng.normal(0, 0.3, 3), rng.normal(0, 0.05, 3)]), dtype=torch.float32)).double().numpy()
        T = T @ step
        poses.append(T.copy())
    return poses


def _similarity(poses, s, tau):
    S = SE3_exp(torch.tensor(tau, dtype=torch.float32)).double().numpy()
    out = []
    for p in poses:
        q = p.copy()
        q[:3, 3] *= s
        out.append(S @ q)
    return out


def test_umeyama_recovers_rotation_translation_and_scale():
    rng = np.random.default_rng(1)
    x = rng.normal(size=(3, 40))
    S = SE3_exp(torch.tensor([0.4, -1.0, 2.0, 0.3, -0.2, 0.9])).double().numpy()
    y = 2.5 * S[:3, :3] @ x + S[:3, 3:4]
    r, t, c = ev.umeyama_alignment(x, y, with_scale=True)
    assert np.allclose(r, S[:3, :3], atol=1e-6) and np.allclose(t, S[:3, 3], atol=1e-6) and abs(c - 2.5) < 1e-6  # S is a float32 rotation
    r, t, c = ev.umeyama_alignment(x, S[:3, :3] @ x + S[:3, 3:4], with_scale=False)
    assert c == 1.0 and np.allclose(r, S[:3, :3], atol=1e-6) and abs(np.linalg.det(r) - 1.0) < 1e-12
    # a reflection in the data must not produce an improper rotation
    r, _, _ = ev.umeyama_alignment(x, np.diag([1.0, 1.0, -1.0]) @ x)
    assert abs(np.linalg.det(r) - 1.0) < 1e-12
    with pytest.raises(ev.AlignmentError):
        ev.umeyama_alignment(np.zeros((3, 5)), np.zeros((3, 5)))


def test_ate_is_zero_under_a_similarity_for_monocular_and_not_otherwise():
    gt = _trajectory()
    est = _similarity(gt, 1.0 / 3.0, [1.0, 2.0, -0.5, 0.2, 0.1, -0.3])  # gt = 3 * S^-1-ish of est
    rmse, stats, aligned = ev.evaluate_ate(gt, est, monocular=True)
    assert rmse < 1e-6 and stats["max"] < 1e-6
    assert np.allclose(aligned[7][:3, 3], gt[7][:3, 3], atol=1e-6)
    rmse_rigid, _, _ = ev.evaluate_ate(gt, est, monocular=False)
    assert rmse_rigid > 0.1
    rigid = _similarity(gt, 1.0, [1.0, 2.0, -0.5, 0.2, 0.1, -0.3])
    assert ev.evaluate_ate(gt, rigid, monocular=False)[0] < 1e-6


def test_ate_statistics_with_known_noise():
    gt = _trajectory(n=200, seed=3)
    rng = np.random.default_rng(4)
    est = [p.copy() for p in gt]
    noise = rng.normal(0, 0.02, (200, 3))
    for p, e in zip(est, noise):
        p[:3, 3] += e
    rmse, stats, _ = ev.evaluate_ate(gt, est, monocular=False)
    assert 0.9 * np.sqrt(3) * 0.02 < rmse < 1.1 * np.sqrt(3) * 0.02  # alignment can only remove error
    assert stats["min"] <= stats["median"] <= stats["max"] and abs(stats["sse"] - 200 * rmse ** 2) < 1e-9
    assert abs(stats["rmse"] ** 2 - (stats["mean"] ** 2 + stats["std"] ** 2)) < 1e-12


def test_low_diversity_falls_back_to_origin_alignment():
    still = [np.eye(4) for _ in range(5)]
    for i, p in enumerate(still):
        p[:3, 3] = [0.001 * i, 0.0, 0.0]
    assert not ev.trajectory_has_diversity(still) and not ev.trajectory_has_diversity(_trajectory(n=2))
    moved = _similarity(still, 1.0, [5.0, 0.0, 0.0, 0.0, 0.0, 0.3])
    rmse, _, aligned = ev.evaluate_ate(still, moved, monocular=True)
    assert rmse < 1e-6 and np.allclose(aligned[0], still[0], atol=1e-6)


def test_eval_ate_reads_keyframe_poses():
    from types import SimpleNamespace
    gt = _trajectory(n=12, seed=5)
    est = _similarity(gt, 0.5, [0.3, 0.1, 0.2, 0.0, 0.2, 0.1])
    frames = {}
    for i, (g, e) in enumerate(zip(gt, est)):
        wg, we = np.linalg.inv(g), np.linalg.inv(e)
        frames[i] = SimpleNamespace(R=torch.tensor(we[:3, :3]), T=torch.tensor(we[:3, 3]), R_gt=torch.tensor(wg[:3, :3]),
                                    T_gt=torch.tensor(wg[:3, 3]), uid=i)
    assert ev.eval_ate(frames, list(range(12)), monocular=True) < 1e-6
    assert ev.eval_ate(frames, [0, 1], monocular=True) is None
