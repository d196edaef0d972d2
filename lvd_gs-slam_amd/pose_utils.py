"""SE(3) retraction used by the per-frame pose optimiser.

Host-side mirror of the reference's ``utils/pose_utils.py`` (same names, argument
meaning and return values) so ``slam_frontend.py:1521`` / ``slam_backend.py:389``
call sites work unchanged.  Pinned by ``tests/golden/se3_exp.npz`` and
``tests/golden/update_pose.npz``.

Conventions (reference ``utils/pose_utils.py:59-87``): tau = [rho; theta] with the
translation part first, left-multiplicative update ``T_w2c <- Exp(tau) @ T_w2c``,
small-angle series below ``1e-5`` rad, convergence when ``||tau|| < 1e-4``.
"""
import numpy as np
import torch

_SMALL_ANGLE = 1e-5  # reference utils/pose_utils.py:30,46


def rt2mat(R, T):
    """4x4 numpy matrix from a rotation and a translation (utils/pose_utils.py:4-8)."""
    out = np.eye(4)
    out[:3, :3] = R
    out[:3, 3] = T
    return out


def skew_sym_mat(x):
    """[x]_x such that [x]_x @ y == cross(x, y) (utils/pose_utils.py:10-21)."""
    zero = torch.zeros((), device=x.device, dtype=x.dtype)
    return torch.stack(
        [
            torch.stack([zero, -x[2], x[1]]),
            torch.stack([x[2], zero, -x[0]]),
            torch.stack([-x[1], x[0], zero]),
        ]
    )


def _rodrigues_coeffs(angle):
    """(A, B, C) = (sin a / a, (1-cos a)/a^2, (a - sin a)/a^3) with the reference's
    truncated series below the small-angle threshold (A=1, B=1/2, C=1/6)."""
    if angle < _SMALL_ANGLE:
        one = torch.ones((), device=angle.device, dtype=angle.dtype)
        return one, 0.5 * one, one / 6.0
    a2 = angle * angle
    return torch.sin(angle) / angle, (1 - torch.cos(angle)) / a2, (angle - torch.sin(angle)) / (a2 * angle)


def SO3_exp(theta):
    """Rotation matrix of the axis-angle vector ``theta`` (utils/pose_utils.py:23-38)."""
    K = skew_sym_mat(theta)
    A, B, _ = _rodrigues_coeffs(torch.norm(theta))
    eye = torch.eye(3, device=theta.device, dtype=theta.dtype)
    return eye + A * K + B * (K @ K)


def V(theta):
    """Left Jacobian of SO(3) (utils/pose_utils.py:40-55)."""
    K = skew_sym_mat(theta)
    _, B, C = _rodrigues_coeffs(torch.norm(theta))
    eye = torch.eye(3, device=theta.device, dtype=theta.dtype)
    return eye + B * K + C * (K @ K)


def SE3_exp(tau):
    """4x4 rigid transform of the twist ``tau = [rho, theta]`` (utils/pose_utils.py:57-68)."""
    rho, theta = tau[:3], tau[3:]
    out = torch.eye(4, device=tau.device, dtype=tau.dtype)
    out[:3, :3] = SO3_exp(theta)
    out[:3, 3] = V(theta) @ rho
    return out


def update_pose(camera, converged_threshold=1e-4):
    """Apply the camera's accumulated deltas to its pose and zero them.

    ``T_w2c <- SE3_exp([cam_trans_delta, cam_rot_delta]) @ T_w2c``; returns a 0-dim
    bool tensor ``||tau|| < converged_threshold`` (utils/pose_utils.py:70-87).
    """
    tau = torch.cat([camera.cam_trans_delta, camera.cam_rot_delta], dim=0)
    w2c = torch.eye(4, device=tau.device)
    w2c[:3, :3] = camera.R
    w2c[:3, 3] = camera.T
    new_w2c = SE3_exp(tau) @ w2c
    converged = tau.norm() < converged_threshold
    camera.update_RT(new_w2c[:3, :3], new_w2c[:3, 3])
    camera.cam_rot_delta.data.fill_(0)
    camera.cam_trans_delta.data.fill_(0)
    return converged
