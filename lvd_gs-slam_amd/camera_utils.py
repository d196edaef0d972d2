"""``Camera``: the per-frame state the rasterizer and the pose optimiser read.

Mirror of the reference's ``utils/camera_utils.py`` (attributes, properties and static
constructors keep their names; ``utils/camera_utils.py:8-166``).  The rasterizer reads
``world_view_transform``, ``full_proj_transform``, ``projection_matrix``, ``camera_center``,
``FoVx/FoVy``, ``image_height/width`` and the two pose deltas; tracking / mapping step
``cam_rot_delta``, ``cam_trans_delta``, ``exposure_a``, ``exposure_b`` with Adam and fold
the deltas into ``R, T`` with ``pose_utils.update_pose``.
"""
import torch
from torch import nn

from .graphics_utils import getProjectionMatrix2, getWorld2View2
from .slam_utils import image_gradient, image_gradient_mask


class Camera(nn.Module):
    def __init__(self, uid, color, depth, mono_depth, gt_T, projection_matrix, fx, fy, cx, cy,
                 fovx, fovy, image_height, image_width, device="cuda:0"):
        super().__init__()
        self.uid = uid
        self.device = device
        eye = torch.eye(4, device=device)
        self.R, self.T = eye[:3, :3], eye[:3, 3]
        self.R_gt, self.T_gt = gt_T[:3, :3], gt_T[:3, 3]
        self.original_image = color
        self.depth = depth
        self.mono_depth = mono_depth
        self.grad_mask = None
        self.fx, self.fy, self.cx, self.cy = fx, fy, cx, cy
        self.FoVx, self.FoVy = fovx, fovy
        self.image_height, self.image_width = image_height, image_width
        self.cam_rot_delta = nn.Parameter(torch.zeros(3, device=device))
        self.cam_trans_delta = nn.Parameter(torch.zeros(3, device=device))
        self.exposure_a = nn.Parameter(torch.zeros(1, device=device))
        self.exposure_b = nn.Parameter(torch.zeros(1, device=device))
        self.projection_matrix = projection_matrix.to(device=device)

    @staticmethod
    def init_from_dataset(dataset, idx, projection_matrix):
        color, depth, pose, mono_depth = dataset[idx]
        return Camera(idx, color, depth, mono_depth, pose, projection_matrix, dataset.fx, dataset.fy,
                      dataset.cx, dataset.cy, dataset.fovx, dataset.fovy, dataset.height,
                      dataset.width, device=dataset.device)

    @staticmethod
    def init_from_gui(uid, T, FoVx, FoVy, fx, fy, cx, cy, H, W, device="cuda:0"):
        proj = getProjectionMatrix2(znear=0.01, zfar=100.0, fx=fx, fy=fy, cx=cx, cy=cy, W=W, H=H)
        return Camera(uid, None, None, None, T, proj.transpose(0, 1), fx, fy, cx, cy, FoVx, FoVy,
                      H, W, device=device)

    @property
    def world_view_transform(self):
        return getWorld2View2(self.R, self.T).transpose(0, 1)

    @property
    def full_proj_transform(self):
        return self.world_view_transform.unsqueeze(0).bmm(self.projection_matrix.unsqueeze(0)).squeeze(0)

    @property
    def camera_center(self):
        return self.world_view_transform.inverse()[3, :3]

    def update_RT(self, R, t):
        self.R = R.to(device=self.device)
        self.T = t.to(device=self.device)

    def compute_grad_mask(self, config):
        """Edge mask used by the tracking loss (utils/camera_utils.py:126-155)."""
        thr = config["Training"]["edge_threshold"]
        gray = self.original_image.mean(dim=0, keepdim=True)
        gv, gh = image_gradient(gray)
        mv, mh = image_gradient_mask(gray)
        mag = torch.sqrt((gv * mv) ** 2 + (gh * mh) ** 2)
        if config["Dataset"]["type"] == "replica":
            rows = cols = 32
            _, h, w = self.original_image.shape
            bh, bw = int(h / rows), int(w / cols)
            for r in range(rows):
                for c in range(cols):
                    blk = mag[:, r * bh:(r + 1) * bh, c * bw:(c + 1) * bw]
                    cut = blk.median() * thr
                    blk[blk > cut] = 1  # sequential, like the reference: the second
                    blk[blk <= cut] = 0  # test sees the ones written by the first
            self.grad_mask = mag
        else:
            self.grad_mask = mag > mag.median() * thr

    def clean(self):
        self.original_image = None
        self.depth = None
        self.grad_mask = None
        self.cam_rot_delta = None
        self.cam_trans_delta = None
        self.exposure_a = None
        self.exposure_b = None
