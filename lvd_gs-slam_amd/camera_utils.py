"""``Camera``: the per-frame state the rasterizer and the pose optimiser read.

Same constructor, attributes, properties and static constructors as the class in the reference's
``utils/camera_utils.py:8-166`` (the SLAM loops and ``pose_utils.update_pose`` use them by name), with one
difference in how the derived matrices are produced: the reference recomputes ``world_view_transform``,
``full_proj_transform`` and ``camera_center`` from ``R, T`` with a handful of small PyTorch kernels at every
attribute access -- three accesses per ``render()``.  Here they are computed once per pose and cached until
``R`` or ``T`` is replaced or written in place (the cache entry keeps references to both tensors and their
versions and compares by identity), which takes about ten launches off every tracking / mapping iteration.
"""
import torch
from torch import nn

from .graphics_utils import getProjectionMatrix2, getWorld2View2
from .slam_utils import image_gradient, image_gradient_mask

_POSE_PARAMS = ("cam_rot_delta", "cam_trans_delta")
_EXPOSURE_PARAMS = ("exposure_a", "exposure_b")


class Camera(nn.Module):
    def __init__(self, uid, color, depth, mono_depth, gt_T, projection_matrix, fx, fy, cx, cy,
                 fovx, fovy, image_height, image_width, device="cuda:0"):
        super().__init__()
        self.uid, self.device = uid, device
        # estimated pose starts at the identity, the ground truth is kept for evaluation
        self.R = torch.eye(3, device=device)
        self.T = torch.zeros(3, device=device)
        self.R_gt, self.T_gt = gt_T[:3, :3], gt_T[:3, 3]
        self.original_image, self.depth, self.mono_depth = color, depth, mono_depth
        self.grad_mask = None
        self.fx, self.fy, self.cx, self.cy = fx, fy, cx, cy
        self.FoVx, self.FoVy = fovx, fovy
        self.image_height, self.image_width = image_height, image_width
        for name in _POSE_PARAMS:
            setattr(self, name, nn.Parameter(torch.zeros(3, device=device)))
        for name in _EXPOSURE_PARAMS:
            setattr(self, name, nn.Parameter(torch.zeros(1, device=device)))
        self.projection_matrix = projection_matrix.to(device=device).contiguous()   # (callers pass a transposed view: copy it once here, not at every render)
        self._derived_key = None
        self._derived = None

    # ---- constructors the front end uses --------------------------------------------------------
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

    # ---- matrices derived from the pose, cached per (R, T, projection) ----------------------------
    def _matrices(self):
        # The cache entry holds the tensors themselves (compared by identity) next to their versions: an `id()` can be
        # handed to a new tensor as soon as the old one is freed (two update_RT() calls in a row do exactly that,
        # utils/slam_frontend.py sync_backend), a held reference cannot.
        R, T, P = self.R, self.T, self.projection_matrix
        k = self._derived_key
        if (k is None or k[0] is not R or k[1] != R._version or k[2] is not T or k[3] != T._version
                or k[4] is not P or k[5] != P._version):
            view = getWorld2View2(R, T).transpose(0, 1).contiguous()   # row-vector convention (contiguous: the renderer takes its pointer)
            full = view.unsqueeze(0).bmm(P.unsqueeze(0)).squeeze(0)
            centre = view.inverse()[3, :3].contiguous()   # (the inverse comes back column-major: the row is a strided view, and every render would copy it)
            self._derived, self._derived_key = (view, full, centre), (R, R._version, T, T._version, P, P._version)
        return self._derived

    @property
    def world_view_transform(self):
        return self._matrices()[0]

    @property
    def full_proj_transform(self):
        return self._matrices()[1]

    @property
    def camera_center(self):
        return self._matrices()[2]

    def update_RT(self, R, t):
        self.R = R.to(device=self.device)
        self.T = t.to(device=self.device)

    # ---- edge mask of the tracking loss (utils/camera_utils.py:126-155) ---------------------------
    def compute_grad_mask(self, config):
        thr = config["Training"]["edge_threshold"]
        gray = self.original_image.mean(dim=0, keepdim=True)
        gv, gh = image_gradient(gray)
        mv, mh = image_gradient_mask(gray)
        mag = torch.sqrt((gv * mv) ** 2 + (gh * mh) ** 2)
        if config["Dataset"]["type"] != "replica":
            self.grad_mask = mag > mag.median() * thr
            return
        # replica: 32 x 32 blocks, each thresholded against its own median.  The reference assigns 1 above the cut
        # and THEN 0 at or below it, so with a cut of 1 or more the ones just written are cleared again.
        rows = cols = 32
        _, h, w = self.original_image.shape
        bh, bw = int(h / rows), int(w / cols)
        body = mag[:, : rows * bh, : cols * bw]
        blocks = body.reshape(1, rows, bh, cols, bw).permute(0, 1, 3, 2, 4).reshape(1, rows, cols, bh * bw)
        cut = blocks.median(dim=-1, keepdim=True).values * thr
        marked = torch.where(blocks > cut, torch.ones_like(blocks), blocks)
        marked = torch.where(marked <= cut, torch.zeros_like(marked), marked)
        mag[:, : rows * bh, : cols * bw] = marked.reshape(1, rows, cols, bh, bw).permute(0, 1, 3, 2, 4).reshape(1, rows * bh, cols * bw)
        self.grad_mask = mag

    def clean(self):
        for name in ("original_image", "depth", "grad_mask") + _POSE_PARAMS + _EXPOSURE_PARAMS:
            setattr(self, name, None)
