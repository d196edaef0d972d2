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


def make_surface_gaussians(N, W, H, seed=0, r_min=4.0, r_max=64.0, margin=0.1):
    """The regime real SLAM maps live in, which ``make_gaussians``' small random blobs never reach: opaque surfaces made
    of large, flat Gaussians.  Three slanted planes (depths around 3, 6 and 12, half of the Gaussians on the front one),
    footprints of r_min..r_max pixels (log-uniform, the in-plane sigma; a tenth of it along the normal), opacity
    sigmoid(N(2, 1)) ~ 0.88: tile lists of 500-2000 entries at ~100 Gaussians per 16x16 tile's worth of area, nearly
    every pixel saturating (transmittance below 1e-4) long before its list ends, rectangles of up to several hundred
    tiles.  Same dict as ``make_gaussians``.  ``margin``: how far past the frame of the camera at the origin the surfaces reach,
    in frame widths / heights (0.1, the workloads' value; the sequences of ``make_sequence`` move the camera and ask for more)."""
    g = torch.Generator().manual_seed(seed)
    u = lambda *s: torch.rand(*s, generator=g)
    fx = float(W)
    plane = torch.multinomial(torch.tensor([0.5, 0.3, 0.2]), N, replacement=True, generator=g)
    z0 = torch.tensor([3.0, 6.0, 12.0])[plane]
    slant = torch.tensor([[0.25, -0.1], [-0.2, 0.15], [0.1, 0.2]])[plane]
    span = 1.2 if margin == 0.1 else 1.0 + 2.0 * margin
    px = (u(N) * span - margin) * W
    py = (u(N) * span - margin) * H
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


# ---------------------------------------------------------------------------------------------------------------------------
# Sequences: what FrontEnd.run reads from its dataset, frame by frame (reference utils/slam_frontend.py:1794-1797 ->
# utils/camera_utils.py Camera.init_from_dataset: ``dataset[idx] -> (image, depth, pose, mono_depth)`` and the intrinsics)
# ---------------------------------------------------------------------------------------------------------------------------
def dynamic_object_mask(H, W, seed):
    """A frame's ``static_mask`` (bool (H,W), True = static) as LVD-GS's front end makes it from GroundingDINO + SAM detections
    (utils/slam_frontend.py:1309-1329; the models are out of scope): two to four "vehicles" -- rectangles of 8-25 % of the image's
    width and 10-35 % of its height in the lower two thirds of the frame -- marked dynamic, seeded per frame."""
    g = torch.Generator().manual_seed(7000 + seed)
    m = torch.ones(H, W, dtype=torch.bool)
    for _ in range(2 + int(torch.randint(0, 3, (1,), generator=g))):
        w = int(W * (0.08 + 0.17 * float(torch.rand(1, generator=g)))); h = int(H * (0.10 + 0.25 * float(torch.rand(1, generator=g))))
        x0 = int(torch.randint(0, max(W - w, 1), (1,), generator=g)); y0 = H // 3 + int(torch.randint(0, max(H - H // 3 - h, 1), (1,), generator=g))
        m[y0:y0 + h, x0:x0 + w] = False
    return m


def vehicle_trajectory(n_frames, step=0.02, sway=0.15, yaw=0.03, period=40.0):
    """World-to-camera poses of a forward-moving camera: ``step`` along +z per frame with a sinusoidal lateral sway (amplitude
    ``sway``), a smaller vertical one and a yaw oscillation of ``yaw`` radians, period ``period`` frames; frame 0 at the identity
    (the front end adopts the first ground-truth pose, utils/slam_frontend.py:1718)."""
    poses = []
    for i in range(n_frames):
        ph = 2.0 * math.pi * i / period
        c2w = torch.eye(4)
        a = yaw * math.sin(ph)
        c2w[0, 0], c2w[0, 2], c2w[2, 0], c2w[2, 2] = math.cos(a), math.sin(a), -math.sin(a), math.cos(a)
        c2w[:3, 3] = torch.tensor([sway * (1.0 - math.cos(ph)) * 0.5, 0.2 * sway * math.sin(ph), step * i])
        poses.append(torch.linalg.inv(c2w))
    return poses


class SequenceDataset:
    """Frames held in memory with the interface ``Camera.init_from_dataset`` reads (reference utils/camera_utils.py:56-75):
    ``dataset[idx] -> (image (3,H,W) on the device, depth (H,W) numpy or None, ground-truth world-to-camera pose (4,4), mono
    depth (H,W) numpy float32)``, the intrinsics and ``device``.  ``static_mask(idx)``: the frame's static mask where the
    sequence carries dynamic objects (what the reference's ``dynamic_masker`` would return), else None."""

    def __init__(self, images, mono_depths, poses, W, H, fx, fy, cx, cy, device, depths=None, static_masks=None):
        self.images, self.mono_depths, self.poses, self.depths, self.static_masks = images, mono_depths, poses, depths, static_masks
        self.width, self.height, self.fx, self.fy, self.cx, self.cy, self.device = W, H, fx, fy, cx, cy, device
        self.fovx, self.fovy = focal2fov(fx, W), focal2fov(fy, H)
        self.dist_coeffs = None

    def __len__(self):
        return len(self.images)

    def __getitem__(self, idx):
        return (self.images[idx], None if self.depths is None else self.depths[idx], self.poses[idx].to(self.device), self.mono_depths[idx])

    def static_mask(self, idx):
        return None if self.static_masks is None else self.static_masks[idx]

    def to(self, device):
        """The same frames held on another device."""
        return SequenceDataset([t.to(device) for t in self.images], self.mono_depths, self.poses, self.width, self.height, self.fx, self.fy,
                               self.cx, self.cy, device, depths=self.depths,
                               static_masks=None if self.static_masks is None else [m.to(device) for m in self.static_masks])


def make_sequence(truth, render_fn, pipe, W, H, n_frames, device, fx=None, fy=None, cx=None, cy=None, seed=0, depth_noise=0.02,
                  image_noise=0.0, dynamic_objects=False, step=0.02, sway=0.15, yaw=0.03, period=40.0, camera_cls=None):
    """``n_frames`` frames of ``vehicle_trajectory`` through the map ``truth`` (a GaussianModel), rendered ONCE by ``render_fn``
    (the product's ``render`` on the GPU, the dense float64 renderer in the CPU tests): image = clamped render (+ seeded pixel
    noise), mono depth = expected depth of the opaque pixels x (1 + ``depth_noise`` N(0,1)) -- a metric depth predictor's output,
    what ``get_depth`` (MASt3R, out of scope) hands the reference's front end.  ``dynamic_objects``: every frame also carries two
    to four flat-coloured rectangles ("vehicles", ``dynamic_object_mask``, redrawn per frame: they move) painted over the image
    and marked dynamic in ``static_mask(idx)``."""
    if camera_cls is None:
        from .camera_utils import Camera as camera_cls
    fx = float(W) if fx is None else fx
    fy = float(W) if fy is None else fy
    cx = W / 2.0 if cx is None else cx
    cy = H / 2.0 if cy is None else cy
    proj = getProjectionMatrix2(znear=0.01, zfar=100.0, fx=fx, fy=fy, cx=cx, cy=cy, W=W, H=H).transpose(0, 1)
    bg = torch.zeros(3, device=device)
    gen = torch.Generator().manual_seed(9000 + seed)
    images, monos, masks = [], [], []
    poses = vehicle_trajectory(n_frames, step, sway, yaw, period)
    for i, w2c in enumerate(poses):
        cam = camera_cls(i, torch.zeros(3, H, W, device=device), None, None, torch.eye(4), proj.to(device), fx, fy, cx, cy,
                         focal2fov(fx, W), focal2fov(fy, H), H, W, device=device)
        cam.update_RT(w2c[:3, :3].to(device), w2c[:3, 3].to(device))
        with torch.no_grad():
            pkg = render_fn(cam, truth, pipe, bg)
        img = pkg["render"].detach().float()
        if image_noise:
            img = img + image_noise * torch.randn(3, H, W, generator=gen).to(img.device)
        img = img.clamp(0.0, 1.0)
        opac = pkg["opacity"][0].detach().float()
        depth = torch.where(opac > 0.5, pkg["depth"][0].detach().float() / opac.clamp(min=1e-3), torch.zeros_like(opac))
        depth = depth * (1.0 + depth_noise * torch.randn(H, W, generator=gen).to(depth.device))
        if dynamic_objects:
            m = dynamic_object_mask(H, W, 100 * seed + i)
            colour = torch.rand(3, generator=gen)
            img = torch.where(m.to(img.device)[None], img, colour.to(img.device)[:, None, None].expand_as(img))
            masks.append(m.to(device))
        images.append(img.contiguous().to(device))
        monos.append(depth.cpu().numpy().astype("float32"))
    return SequenceDataset(images, monos, poses, W, H, fx, fy, cx, cy, device, static_masks=masks if dynamic_objects else None)
