"""Minimal Gaussian parameter container with the accessors ``render()`` reads.

The reference's ``gaussian_splatting.scene.GaussianModel`` is absent from the checkout
(SURVEY.md section 2 row 10); map bookkeeping (densify / prune / Adam groups) is outside the hot
path and scheduled as a "next" row.  This class holds the raw parameters with the published
activations -- exp for scales, sigmoid for opacity, L2-normalised quaternions -- so that the
benchmark and the tests exercise the same ``render(viewpoint, gaussians, pipe, bg)`` call chain
as ``utils/slam_frontend.py:1493`` / ``utils/slam_backend.py:184``.
"""
import torch
from torch import nn

from .sh_utils import RGB2SH


class GaussianModel:
    standard_activations = True  # exp / normalize / sigmoid, as published: render() may fuse them

    def __init__(self, sh_degree=0, device="cuda"):
        self.max_sh_degree = sh_degree
        self.active_sh_degree = sh_degree
        self.device = device
        z = lambda *s: nn.Parameter(torch.empty(*s, device=device))
        self._xyz, self._features_dc, self._features_rest = z(0, 3), z(0, 1, 3), z(0, 0, 3)
        self._scaling, self._rotation, self._opacity = z(0, 3), z(0, 4), z(0, 1)

    @classmethod
    def from_activated(cls, means3D, scales, rotations, opacities, shs=None, colors=None, sh_degree=0, device="cuda"):
        """Build from the activated values the rasterizer consumes (inverse activations applied)."""
        m = cls(sh_degree, device)
        p = lambda t: nn.Parameter(t.detach().to(device=device, dtype=torch.float32).contiguous())
        if shs is None:
            shs = RGB2SH(colors)[:, None, :]
        m._xyz = p(means3D)
        m._features_dc = p(shs[:, :1])
        m._features_rest = p(shs[:, 1:])
        m._scaling = p(torch.log(scales))
        m._rotation = p(rotations)
        o = opacities.clamp(1e-6, 1 - 1e-6)
        m._opacity = p(torch.log(o / (1 - o)))
        return m

    def parameters(self):
        return [self._xyz, self._features_dc, self._features_rest, self._scaling, self._rotation, self._opacity]

    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_features(self):
        if self._features_rest.shape[1] == 0:  # SH degree 0: nothing to concatenate
            return self._features_dc
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    @property
    def get_scaling(self):
        return torch.exp(self._scaling)

    @property
    def get_rotation(self):
        return torch.nn.functional.normalize(self._rotation)

    @property
    def get_opacity(self):
        return torch.sigmoid(self._opacity)

    def get_covariance(self, scaling_modifier=1.0):
        q = self.get_rotation
        r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
        R = torch.stack([torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)], -1),
                         torch.stack([2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)], -1),
                         torch.stack([2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1)], 1)
        M = R * (scaling_modifier * self.get_scaling)[:, None, :]
        S = M @ M.transpose(1, 2)
        return torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], 1)
