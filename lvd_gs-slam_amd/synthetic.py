"""Seeded synthetic scenes for parity tests and the benchmark.

The reference defines no synthetic input; this is the build's definition, generated on the CPU
with ``torch.Generator().manual_seed(seed)`` in float32 so that the CPU oracle and the GPU path
see identical bits.

* camera: pinhole fx = fy = W, cx = W/2, cy = H/2, znear 0.01, zfar 100
  (``utils/slam_frontend.py:1743-1748``); by default world == camera (R = I, T = 0),
  optionally a seeded rigid pose for the multi-keyframe cases;
* Gaussians: depth log-uniform [1, 50]; pixel uniform over [-0.1W, 1.1W] x [-0.1H, 1.1H];
  xyz by back-projection; pixel radius log-uniform [0.5, 8]; per-axis scale
  z*r/fx*U[0.7, 1.3]; unit quaternion from N(0,1)^4; opacity sigmoid(N(0, 1.5));
  colour rgb ~ U[0,1] (stored as the SH DC term (rgb - 0.5)/C0); SH degree 0.

Values are the *activated* quantities the rasterizer consumes (scale, not log-scale; opacity,
not logit), as ``render()`` passes them.
"""
import math
from types import SimpleNamespace

import torch

from .graphics_utils import focal2fov, getProjectionMatrix2
from .pose_utils import SE3_exp

SH_C0 = 0.28209479177387814


def make_camera(W, H, pose_seed=None, fx=None, fy=None, cx=None, cy=None, pose_scale=0.05):
    """Camera namespace with the attributes ``render()`` reads (float32 CPU tensors)."""
    fx = float(W) if fx is None else fx
    fy = float(W) if fy is None else fy
    cx = W / 2.0 if cx is None else cx
    cy = H / 2.0 if cy is None else cy
    w2c = torch.eye(4)
    if pose_seed is not None:
        g = torch.Generator().manual_seed(1000 + pose_seed)
        w2c = SE3_exp(torch.randn(6, generator=g) * pose_scale)
    proj = getProjectionMatrix2(znear=0.01, zfar=100.0, fx=fx, fy=fy, cx=cx, cy=cy, W=W, H=H).transpose(0, 1)
    view = w2c.transpose(0, 1).contiguous()
    cam = SimpleNamespace(
        image_width=W, image_height=H, fx=fx, fy=fy, cx=cx, cy=cy,
        FoVx=focal2fov(fx, W), FoVy=focal2fov(fy, H),
        R=w2c[:3, :3].contiguous(), T=w2c[:3, 3].contiguous(),
        world_view_transform=view, projection_matrix=proj.contiguous(),
        full_proj_transform=(view @ proj).contiguous(),
        camera_center=torch.linalg.inv(view)[3, :3].contiguous(),
        cam_rot_delta=torch.zeros(3), cam_trans_delta=torch.zeros(3),
    )
    cam.tanfovx = math.tan(cam.FoVx * 0.5)
    cam.tanfovy = math.tan(cam.FoVy * 0.5)
    return cam


def make_gaussians(N, W, H, seed=0, sh_degree=0, r_min=0.5, r_max=8.0, z_min=1.0, z_max=50.0):
    """dict of float32 CPU tensors: means3D (N,3), scales (N,3), rotations (N,4), opacities (N,1),
    shs (N,(deg+1)^2,3), colors (N,3)."""
    g = torch.Generator().manual_seed(seed)
    u = lambda *s: torch.rand(*s, generator=g)
    fx = float(W)
    z = torch.exp(u(N) * (math.log(z_max) - math.log(z_min)) + math.log(z_min))
    px = (u(N) * 1.2 - 0.1) * W
    py = (u(N) * 1.2 - 0.1) * H
    x = (px - W / 2.0) * z / fx
    y = (py - H / 2.0) * z / fx
    r = torch.exp(u(N) * (math.log(r_max) - math.log(r_min)) + math.log(r_min))
    scales = (z * r / fx)[:, None] * (0.7 + 0.6 * u(N, 3))
    q = torch.randn(N, 4, generator=g)
    q = q / q.norm(dim=1, keepdim=True)
    opac = torch.sigmoid(torch.randn(N, 1, generator=g) * 1.5)
    rgb = u(N, 3)
    K = (sh_degree + 1) ** 2
    shs = torch.zeros(N, K, 3)
    shs[:, 0] = (rgb - 0.5) / SH_C0
    if K > 1:
        shs[:, 1:] = torch.randn(N, K - 1, 3, generator=g) * 0.1
    return dict(means3D=torch.stack([x, y, z], 1).contiguous(), scales=scales.contiguous(), rotations=q.contiguous(),
                opacities=opac.contiguous(), shs=shs.contiguous(), colors=rgb.contiguous())


def make_surface_gaussians(N, W, H, seed=0, r_min=4.0, r_max=64.0):
    """The regime real SLAM maps live in, which ``make_gaussians``' small random blobs never reach: opaque surfaces made
    of large, flat Gaussians.  Three slanted planes (depths around 3, 6 and 12, half of the Gaussians on the front one),
    footprints of r_min..r_max pixels (log-uniform, the in-plane sigma; a tenth of it along the normal), opacity
    sigmoid(N(2, 1)) ~ 0.88: tile lists of 500-2000 entries at ~100 Gaussians per 16x16 tile's worth of area, nearly
    every pixel saturating (transmittance below 1e-4) long before its list ends, rectangles of up to several hundred
    tiles.  Same dict as ``make_gaussians``."""
    g = torch.Generator().manual_seed(seed)
    u = lambda *s: torch.rand(*s, generator=g)
    fx = float(W)
    plane = torch.multinomial(torch.tensor([0.5, 0.3, 0.2]), N, replacement=True, generator=g)
    z0 = torch.tensor([3.0, 6.0, 12.0])[plane]
    slant = torch.tensor([[0.25, -0.1], [-0.2, 0.15], [0.1, 0.2]])[plane]
    px = (u(N) * 1.2 - 0.1) * W
    py = (u(N) * 1.2 - 0.1) * H
    z = z0 * (1.0 + slant[:, 0] * (px / W - 0.5) + slant[:, 1] * (py / H - 0.5)) * (1.0 + 0.01 * torch.randn(N, generator=g))
    x = (px - W / 2.0) * z / fx
    y = (py - H / 2.0) * z / fx
    r = torch.exp(u(N) * (math.log(r_max) - math.log(r_min)) + math.log(r_min))
    s_in = (z * r / fx)[:, None] * (0.7 + 0.6 * u(N, 2))
    scales = torch.cat([s_in, 0.1 * s_in.mean(1, keepdim=True)], 1)      # thin along the (local) z axis
    q = torch.cat([torch.ones(N, 1), 0.15 * torch.randn(N, 3, generator=g)], 1)   # near the identity: the thin axis faces the camera
    q = q / q.norm(dim=1, keepdim=True)
    opac = torch.sigmoid(2.0 + torch.randn(N, 1, generator=g))
    rgb = u(N, 3)
    shs = ((rgb - 0.5) / SH_C0)[:, None, :].contiguous()
    return dict(means3D=torch.stack([x, y, z], 1).contiguous(), scales=scales.contiguous(), rotations=q.contiguous(),
                opacities=opac.contiguous(), shs=shs, colors=rgb.contiguous())


def make_workload_gaussians(name, seed=0):
    """The Gaussians of ``CONFIGS[name]`` (``kind``: "blobs", the default, or "surface")."""
    cfg = CONFIGS[name]
    make = make_surface_gaussians if cfg.get("kind") == "surface" else make_gaussians
    return make(cfg["N"], cfg["W"], cfg["H"], seed=seed)


def make_image_grads(W, H, seed=0):
    """Fixed seeded N(0,1) upstream gradients for (render, depth, opacity)."""
    g = torch.Generator().manual_seed(7000 + seed)
    return (torch.randn(3, H, W, generator=g), torch.randn(1, H, W, generator=g), torch.randn(1, H, W, generator=g))


CONFIGS = {
    # BASELINE.json configs[0..2]
    "cfg1_10k_640x480": dict(N=10_000, W=640, H=480),
    "cfg2_100k_640x480": dict(N=100_000, W=640, H=480),
    "cfg3_500k_1920x1080": dict(N=500_000, W=1920, H=1080),
    # KITTI-07 geometry (configs/mono/KITTI/07.yaml:8-18)
    "kitti07_geom": dict(N=200_000, W=1226, H=370, fx=707.0912, fy=707.0912, cx=601.8873, cy=183.1104),
    # the same pair density per tile as kitti07_geom on four times the area (diagnostic: is a 1848-tile frame slow per
    # pair because it leaves compute units idle?)
    "kitti07_x4": dict(N=800_000, W=2452, H=740, fx=1414.1824, fy=1414.1824, cx=1203.7746, cy=366.2208),
    # BASELINE.json configs[4] shape: 2 M Gaussians, waymo-sized frames (configs/mono/waymo/405841.yaml:15-16)
    "cfg5_2m_1920x1280": dict(N=2_000_000, W=1920, H=1280),
    # opaque surfaces of large flat Gaussians (make_surface_gaussians): lists of ~1000 entries per tile, saturating pixels
    "surface_100k_1920x1080": dict(N=100_000, W=1920, H=1080, kind="surface"),
    "surface_12k_640x480": dict(N=12_000, W=640, H=480, kind="surface"),
}
