"""Projection / view matrices that feed the rasterizer.

The reference imports ``getProjectionMatrix2`` / ``getWorld2View2`` / ``focal2fov`` from
``gaussian_splatting.utils.graphics_utils`` (``utils/camera_utils.py:4``,
``utils/slam_backend.py:12``, ``utils/slam_frontend.py:1743-1749``); that package is absent
from the reference checkout, so these are restated from the call
sites and the published 3DGS / MonoGS conventions:

* column-vector world->camera matrix ``[R t; 0 1]``; callers transpose it
  (``utils/camera_utils.py:106-108``) into the row-vector layout the rasterizer reads;
* OpenGL-style clip matrix from pinhole intrinsics with ``z_sign = +1`` (camera looks down
  +z), principal point honoured, ``w_clip = z_view``.
"""
import math

import torch


def getWorld2View2(R, t, translate=None, scale=1.0):
    """4x4 world->camera (column-vector convention) from torch ``R`` (3,3), ``t`` (3,)."""
    if translate is None:
        translate = torch.zeros(3, device=R.device, dtype=R.dtype)
    Rt = torch.zeros((4, 4), device=R.device, dtype=R.dtype)
    Rt[:3, :3] = R
    Rt[:3, 3] = t
    Rt[3, 3] = 1.0
    C2W = torch.linalg.inv(Rt)
    center = (C2W[:3, 3] + translate.to(R.device)) * scale
    C2W = C2W.clone()
    C2W[:3, 3] = center
    return torch.linalg.inv(C2W)


def getProjectionMatrix2(znear, zfar, cx, cy, fx, fy, W, H):
    """Clip matrix (column-vector convention; callers transpose, slam_frontend.py:1743-1749)."""
    left = ((2 * cx - W) / W - 1.0) * W / 2.0
    right = ((2 * cx - W) / W + 1.0) * W / 2.0
    top = ((2 * cy - H) / H + 1.0) * H / 2.0
    bottom = ((2 * cy - H) / H - 1.0) * H / 2.0
    left, right = znear / fx * left, znear / fx * right
    top, bottom = znear / fy * top, znear / fy * bottom
    P = torch.zeros(4, 4)
    z_sign = 1.0
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = z_sign
    P[2, 2] = z_sign * zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def getProjectionMatrix(znear, zfar, fovX, fovY):
    """Symmetric-frustum clip matrix from fields of view."""
    tx, ty = math.tan(fovX / 2), math.tan(fovY / 2)
    P = torch.zeros(4, 4)
    P[0, 0] = 1.0 / tx
    P[1, 1] = 1.0 / ty
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def fov2focal(fov, pixels):
    return pixels / (2 * math.tan(fov / 2))


def focal2fov(focal, pixels):
    return 2 * math.atan(pixels / (2 * focal))
