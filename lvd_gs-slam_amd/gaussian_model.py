"""Gaussian map: parameters, Adam groups, seeding from keyframes, densification and pruning.

Counterpart of ``gaussian_splatting.scene.gaussian_model.GaussianModel`` (absent from the reference
checkout).  What is pinned are the members the reference's files use:

* ``render()`` reads ``get_xyz, get_features, get_opacity, get_scaling, get_rotation, active_sh_degree``
  (and ``get_covariance`` under ``pipe.compute_cov3D_python``);
* ``utils/slam_backend.py`` calls ``extend_from_pcd_seq(viewpoint, kf_id=, init=, scale=, depthmap=)`` (:76),
  ``prune_points(mask)`` (:89, :339), ``max_radii2D`` (:124, :351, :460), ``add_densification_stats`` (:128, :355),
  ``densify_and_prune(max_grad, min_opacity, extent, max_screen_size)`` (:132, :364), ``reset_opacity()`` (:142),
  ``reset_opacity_nonvisible(filters)`` (:375), ``optimizer`` (:144, :378), ``update_learning_rate(it)`` (:380),
  ``get_scaling`` (:303), ``n_obs`` / ``unique_kfIDs`` as CPU tensors (:322-336);
* ``utils/eval_utils_0806.py:449`` calls ``save_ply(path)``.

The bodies follow the published 3DGS / MonoGS behaviour [recalled, not verifiable against this checkout]:
exp / sigmoid / normalize activations, per-group Adam with eps 1e-15, exponential position-lr decay,
clone small / split large Gaussians whose mean screen-space gradient exceeds the threshold, prune
transparent / oversized ones, opacity resets.  Keyframe seeding back-projects a depth map through the
keyframe's intrinsics and pose, keeps a random ``1 / pcd_downsample`` of the valid pixels, and sets the
isotropic scale from the mean squared distance to the 3 nearest neighbours (``distCUDA2``, HIP).

Everything is torch on the model's device; no open3d / plyfile dependency.
"""
import math
import os

import numpy as np
import torch
from torch import nn

from .sh_utils import RGB2SH


def inverse_sigmoid(x):
    return torch.log(x / (1 - x))


def build_rotation(q):
    """(N,4) quaternions (r,x,y,z), normalised here -> (N,3,3)."""
    q = q / q.norm(dim=1, keepdim=True)
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    return torch.stack([torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)], -1),
                        torch.stack([2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)], -1),
                        torch.stack([2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1)], 1)


def get_expon_lr_func(lr_init, lr_final, lr_delay_steps=0, lr_delay_mult=1.0, max_steps=1000000):
    """Log-linear interpolation from lr_init to lr_final over max_steps, optionally eased in (Plenoxels / 3DGS)."""

    def helper(step):
        if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
            return 0.0
        if lr_delay_steps > 0:
            delay_rate = lr_delay_mult + (1 - lr_delay_mult) * math.sin(0.5 * math.pi * min(max(step / lr_delay_steps, 0.0), 1.0))
        else:
            delay_rate = 1.0
        t = min(max(step / max_steps, 0.0), 1.0)
        return delay_rate * math.exp(math.log(lr_init) * (1 - t) + math.log(lr_final) * t)

    return helper


UPSTREAM_NONVISIBLE_RESET = True  # see GaussianModel.reset_opacity_nonvisible


class FusedAdam(torch.optim.Adam):
    """``torch.optim.Adam`` whose ``step()`` updates every parameter of every group in ONE HIP launch (lvdgs_adam_step):
    one pass over grad / exp_avg / exp_avg_sq / parameter instead of PyTorch's ~60 multi-tensor launches per step.
    Same state layout (``exp_avg``, ``exp_avg_sq``, ``step``), same ``param_groups``, so the densification code that
    edits the state (and anything written against torch.optim.Adam) keeps working; float32 parameters on the GPU, no
    weight decay / amsgrad / maximize -- anything else goes to the parent class."""

    def _fusable(self):
        for group in self.param_groups:
            if group.get("weight_decay", 0) or group.get("amsgrad", False) or group.get("maximize", False):
                return False
            for p in group["params"]:
                if p.grad is not None and (not p.is_cuda or p.dtype is not torch.float32 or not p.is_contiguous()
                                           or p.grad.is_sparse or not p.grad.is_contiguous() or p.grad.dtype is not torch.float32):
                    return False
        return True

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None or not self._fusable():
            # the parent's step, without its hook wrapper (this method already ran the optimizer's step hooks once)
            parent = torch.optim.Adam.step
            return getattr(parent, "__wrapped__", parent)(self, closure)
        import ctypes as C
        from . import _lib
        items = []
        for group in self.param_groups:
            beta1, beta2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] = st["step"] + 1 if torch.is_tensor(st["step"]) else st["step"] + 1
                items.append((p, st, float(group["lr"]), float(beta1), float(beta2), float(group["eps"])))
        dev = None
        i = 0
        while i < len(items):
            # one launch per run of up to 8 tensors that share betas / eps / device (the map has one such run)
            b1, b2, eps, dev = items[i][3], items[i][4], items[i][5], items[i][0].device
            arr = (_lib.AdamTensor * 8)()
            n = 0
            while i < len(items) and n < 8 and items[i][3:6] == (b1, b2, eps) and items[i][0].device == dev:
                p, st, lr = items[i][0], items[i][1], items[i][2]
                t = arr[n]
                t.param, t.grad = C.c_void_p(p.data_ptr()), C.c_void_p(p.grad.data_ptr())
                t.exp_avg, t.exp_avg_sq = C.c_void_p(st["exp_avg"].data_ptr()), C.c_void_p(st["exp_avg_sq"].data_ptr())
                t.numel, t.step, t.lr = p.numel(), int(st["step"]), lr
                n += 1
                i += 1
            with _lib.on_device(dev):
                _lib.check(_lib.lib().lvdgs_adam_step(arr, n, b1, b2, eps, _lib.raw_stream(dev)), "lvdgs_adam_step")
        return None


class GaussianModel:
    standard_activations = True  # exp / normalize / sigmoid, as published: render() may fuse them into the kernels

    def __init__(self, sh_degree=0, config=None, device="cuda"):
        self.active_sh_degree = 0
        self.max_sh_degree = sh_degree
        self.device = device
        self.config = config
        self.isotropic = False
        self.ply_input = None
        z = lambda *s: nn.Parameter(torch.empty(*s, device=device))
        self._xyz, self._features_dc, self._features_rest = z(0, 3), z(0, 1, 3), z(0, (sh_degree + 1) ** 2 - 1, 3)
        self._scaling, self._rotation, self._opacity = z(0, 3), z(0, 4), z(0, 1)
        self.max_radii2D = torch.empty(0, device=device)
        self.xyz_gradient_accum = torch.empty(0, 1, device=device)
        self.denom = torch.empty(0, 1, device=device)
        self.unique_kfIDs = torch.empty(0, dtype=torch.int32)  # CPU, like upstream (slam_backend.py:332 compares, :339 .cuda()s)
        self.n_obs = torch.empty(0, dtype=torch.int32)         # CPU (slam_backend.py:324 adds visibility.cpu())
        self.optimizer = None
        self.percent_dense = 0.0
        self.spatial_lr_scale = 0.0
        self.scaling_activation, self.scaling_inverse_activation = torch.exp, torch.log
        self.opacity_activation, self.inverse_opacity_activation = torch.sigmoid, inverse_sigmoid
        self.rotation_activation = torch.nn.functional.normalize
        self.rng = np.random.default_rng(0)  # pixel subsampling of new keyframes
        # Samples of densify_and_split.  A CPU generator with a fixed seed: the draw is made on the host and copied,
        # so every replica of the map (one per GPU in the sharded mapping loop) and a CPU run of the same loop split
        # their Gaussians identically.  Set to None for torch's global generator on the model's device.
        self.generator = torch.Generator().manual_seed(0)

    # ------------------------------------------------------------------ construction helpers
    @classmethod
    def from_activated(cls, means3D, scales, rotations, opacities, shs=None, colors=None, sh_degree=0, device="cuda"):
        """Build from the activated values the rasterizer consumes (inverse activations applied)."""
        m = cls(sh_degree, device=device)
        m.active_sh_degree = sh_degree
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
        m._reset_bookkeeping(kf_id=0)
        return m

    def _reset_bookkeeping(self, kf_id=0):
        n = self._xyz.shape[0]
        self.max_radii2D = torch.zeros(n, device=self.device)
        self.xyz_gradient_accum = torch.zeros(n, 1, device=self.device)
        self.denom = torch.zeros(n, 1, device=self.device)
        self.unique_kfIDs = torch.full((n,), kf_id, dtype=torch.int32)
        self.n_obs = torch.zeros(n, dtype=torch.int32)

    def parameters(self):
        return [self._xyz, self._features_dc, self._features_rest, self._scaling, self._rotation, self._opacity]

    # ------------------------------------------------------------------ accessors render() reads
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
        return self.scaling_activation(self._scaling)

    @property
    def get_rotation(self):
        return self.rotation_activation(self._rotation)

    @property
    def get_opacity(self):
        return self.opacity_activation(self._opacity)

    def get_covariance(self, scaling_modifier=1.0):
        M = build_rotation(self._rotation) * (scaling_modifier * self.get_scaling)[:, None, :]
        S = M @ M.transpose(1, 2)
        return torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], 1)

    def oneupSHdegree(self):
        if self.active_sh_degree < self.max_sh_degree:
            self.active_sh_degree += 1

    # ------------------------------------------------------------------ optimizer
    def init_lr(self, spatial_lr_scale):
        self.spatial_lr_scale = spatial_lr_scale

    def training_setup(self, training_args):
        """``training_args``: the reference's ``opt_params`` block (configs/mono/KITTI/base_config.yaml:58-76)."""
        g = (lambda k: training_args[k]) if isinstance(training_args, dict) else (lambda k: getattr(training_args, k))
        self.percent_dense = g("percent_dense")
        n = self._xyz.shape[0]
        self.xyz_gradient_accum = torch.zeros(n, 1, device=self.device)
        self.denom = torch.zeros(n, 1, device=self.device)
        groups = [
            {"params": [self._xyz], "lr": g("position_lr_init") * self.spatial_lr_scale, "name": "xyz"},
            {"params": [self._features_dc], "lr": g("feature_lr"), "name": "f_dc"},
            {"params": [self._features_rest], "lr": g("feature_lr") / 20.0, "name": "f_rest"},
            {"params": [self._opacity], "lr": g("opacity_lr"), "name": "opacity"},
            {"params": [self._scaling], "lr": g("scaling_lr") * self.spatial_lr_scale, "name": "scaling"},
            {"params": [self._rotation], "lr": g("rotation_lr"), "name": "rotation"},
        ]
        # FusedAdam is torch.optim.Adam with a one-launch step() on the GPU (and the parent's step() anywhere else)
        self.optimizer = FusedAdam(groups, lr=0.0, eps=1e-15)
        self.lr_init = g("position_lr_init") * self.spatial_lr_scale
        self.lr_final = g("position_lr_final") * self.spatial_lr_scale
        self.lr_delay_mult = g("position_lr_delay_mult")
        self.max_steps = g("position_lr_max_steps")
        self.xyz_scheduler_args = get_expon_lr_func(lr_init=self.lr_init, lr_final=self.lr_final,
                                                    lr_delay_mult=self.lr_delay_mult, max_steps=self.max_steps)

    def update_learning_rate(self, iteration):
        for group in self.optimizer.param_groups:
            if group["name"] == "xyz":
                lr = self.xyz_scheduler_args(iteration)
                group["lr"] = lr
                return lr

    def _set_params(self, tensors):
        self._xyz, self._features_dc, self._features_rest = tensors["xyz"], tensors["f_dc"], tensors["f_rest"]
        self._opacity, self._scaling, self._rotation = tensors["opacity"], tensors["scaling"], tensors["rotation"]

    def _params_by_name(self):
        return {"xyz": self._xyz, "f_dc": self._features_dc, "f_rest": self._features_rest, "opacity": self._opacity,
                "scaling": self._scaling, "rotation": self._rotation}

    def replace_tensor_to_optimizer(self, tensor, name):
        """Swap one group's parameter for ``tensor`` and zero its Adam moments."""
        out = {}
        for group in self.optimizer.param_groups:
            if group["name"] != name:
                continue
            old = group["params"][0]
            state = self.optimizer.state.pop(old, None)
            group["params"][0] = nn.Parameter(tensor.requires_grad_(True))
            if state is not None:
                state["exp_avg"] = torch.zeros_like(tensor)
                state["exp_avg_sq"] = torch.zeros_like(tensor)
                self.optimizer.state[group["params"][0]] = state
            out[name] = group["params"][0]
        return out

    def _rebuild_groups(self, transform, moment_transform):
        """Apply ``transform`` to every parameter and ``moment_transform`` to its Adam moments."""
        if self.optimizer is None:
            new = {k: nn.Parameter(transform(k, v.data).requires_grad_(True)) for k, v in self._params_by_name().items()}
            self._set_params(new)
            return
        new = {}
        for group in self.optimizer.param_groups:
            old = group["params"][0]
            state = self.optimizer.state.pop(old, None)
            param = nn.Parameter(transform(group["name"], old.data).requires_grad_(True))
            if state is not None:
                state["exp_avg"] = moment_transform(group["name"], state["exp_avg"])
                state["exp_avg_sq"] = moment_transform(group["name"], state["exp_avg_sq"])
                self.optimizer.state[param] = state
            group["params"][0] = param
            new[group["name"]] = param
        self._set_params(new)

    # ------------------------------------------------------------------ prune / append
    def prune_points(self, mask):
        """Remove the Gaussians where ``mask`` is True (slam_backend.py:89, :339)."""
        keep = ~mask.to(device=self._xyz.device, dtype=torch.bool)
        # The surviving rows as INDICES, found once: every `t[keep]` with a boolean mask is a count on the device the host waits for before
        # it can size the result -- eighteen of them here (six parameters, twelve Adam moments) plus the three statistics, each a
        # round trip of its own behind whatever is queued (tools/stall_probe.py: 50 ms per call of this function in the mapping loop of
        # the synthetic drive).  index_select with the same rows in the same order: the same tensors.
        idx = torch.nonzero(keep).squeeze(1)
        pick = lambda n, t: t.index_select(0, idx)
        self._rebuild_groups(pick, pick)
        self.xyz_gradient_accum = self.xyz_gradient_accum.index_select(0, idx)
        self.denom = self.denom.index_select(0, idx)
        self.max_radii2D = self.max_radii2D.index_select(0, idx)
        idx_cpu = idx.cpu()
        self.unique_kfIDs = self.unique_kfIDs.index_select(0, idx_cpu)
        self.n_obs = self.n_obs.index_select(0, idx_cpu)

    def densification_postfix(self, new_xyz, new_features_dc, new_features_rest, new_opacities, new_scaling, new_rotation,
                              new_kf_ids=None, new_n_obs=None):
        ext = {"xyz": new_xyz, "f_dc": new_features_dc, "f_rest": new_features_rest, "opacity": new_opacities,
               "scaling": new_scaling, "rotation": new_rotation}
        self._rebuild_groups(lambda n, t: torch.cat((t, ext[n].to(t)), dim=0),
                             lambda n, t: torch.cat((t, torch.zeros_like(ext[n], dtype=t.dtype, device=t.device)), dim=0))
        n = self._xyz.shape[0]
        self.xyz_gradient_accum = torch.zeros(n, 1, device=self.device)
        self.denom = torch.zeros(n, 1, device=self.device)
        self.max_radii2D = torch.zeros(n, device=self.device)
        if new_kf_ids is not None:
            self.unique_kfIDs = torch.cat((self.unique_kfIDs, new_kf_ids.to(torch.int32).cpu()))
        if new_n_obs is not None:
            self.n_obs = torch.cat((self.n_obs, new_n_obs.to(torch.int32).cpu()))

    # ------------------------------------------------------------------ seeding from a keyframe
    def _cfg(self, section, key, default=None):
        try:
            return self.config[section][key]
        except (TypeError, KeyError):
            return default

    def create_pcd_from_image(self, cam, init=False, scale=2.0, depthmap=None):
        """Back-project one keyframe: -> (xyz, features, log-scales, quaternions, opacity logits)."""
        dev = self.device
        image_ab = (torch.exp(cam.exposure_a.detach()) * cam.original_image.to(dev) + cam.exposure_b.detach()).clamp(0.0, 1.0)
        rgb = (image_ab * 255).to(torch.uint8).to(torch.float32) / 255.0  # colours go through 8 bits, as with an RGBD image
        H, W = int(cam.image_height), int(cam.image_width)
        if depthmap is not None:
            depth = torch.as_tensor(np.asarray(depthmap) if not torch.is_tensor(depthmap) else depthmap, dtype=torch.float32, device=dev)
        else:
            depth = None if getattr(cam, "depth", None) is None else torch.as_tensor(cam.depth, dtype=torch.float32, device=dev)
            if self._cfg("Dataset", "sensor_type", "monocular") == "monocular" or depth is None:
                noise = torch.from_numpy(self.rng.standard_normal((H, W)).astype(np.float32)).to(dev)
                depth = (1.0 + (noise - 0.5) * 0.05) * scale  # a fronto-parallel slab at `scale` until depth is learnt
        return self.create_pcd_from_image_and_depth(cam, rgb, depth.reshape(H, W), init)

    def create_pcd_from_image_and_depth(self, cam, rgb, depth, init=False):
        from .simple_knn import distCUDA2

        dev = self.device
        downsample = self._cfg("Dataset", "pcd_downsample_init" if init else "pcd_downsample", 32 if init else 64)
        point_size = self._cfg("Dataset", "point_size", 0.01)
        valid = (depth > 0) & (depth <= 100.0)  # depth_trunc 100
        if self._cfg("Dataset", "adaptive_pointsize", False):
            point_size = min(0.05, point_size * float(depth.median()))
        v, u = torch.nonzero(valid, as_tuple=True)
        n_keep = int(v.numel() * (1.0 / downsample))
        pick = torch.from_numpy(np.sort(self.rng.choice(v.numel(), size=n_keep, replace=False))).to(dev) if n_keep else \
            torch.empty(0, dtype=torch.long, device=dev)
        v, u = v[pick], u[pick]
        z = depth[v, u]
        cam_pts = torch.stack(((u.float() - cam.cx) * z / cam.fx, (v.float() - cam.cy) * z / cam.fy, z), dim=1)
        R, T = cam.R.to(dev).float(), cam.T.to(dev).float()
        xyz = (cam_pts - T[None, :]) @ R  # p_cam = R p_world + T
        colors = rgb[:, v, u].t().contiguous()
        self.ply_input = (xyz, colors)

        n = xyz.shape[0]
        features = torch.zeros(n, 3, (self.max_sh_degree + 1) ** 2, device=dev)
        features[:, :3, 0] = RGB2SH(colors)
        dist2 = torch.clamp_min(distCUDA2(xyz), 0.0000001) * point_size if n else torch.empty(0, device=dev)
        scales = torch.log(torch.sqrt(dist2))[:, None]
        if not self.isotropic:
            scales = scales.repeat(1, 3)
        rots = torch.zeros(n, 4, device=dev)
        rots[:, 0] = 1
        opacities = inverse_sigmoid(0.5 * torch.ones(n, 1, device=dev))
        return xyz, features, scales, rots, opacities

    def extend_from_pcd(self, xyz, features, scales, rots, opacities, kf_id):
        n = xyz.shape[0]
        self.densification_postfix(
            xyz, features[:, :, 0:1].transpose(1, 2).contiguous(), features[:, :, 1:].transpose(1, 2).contiguous(),
            opacities, scales, rots, new_kf_ids=torch.full((n,), kf_id, dtype=torch.int32), new_n_obs=torch.zeros(n, dtype=torch.int32))

    def extend_from_pcd_seq(self, cam_info, kf_id=-1, init=False, scale=2.0, depthmap=None):
        """Seed Gaussians from one keyframe (utils/slam_backend.py:75-78)."""
        self.extend_from_pcd(*self.create_pcd_from_image(cam_info, init, scale=scale, depthmap=depthmap), kf_id)

    # ------------------------------------------------------------------ densification
    def add_densification_stats(self, viewspace_point_tensor, update_filter):
        """``xyz_gradient_accum[f] += |grad[f, :2]|``; ``denom[f] += 1`` -- as element-wise statements over all Gaussians (x + 0 = x: the same
        bits) instead of boolean gathers and scatters, each of which makes the host wait for the device to learn how many elements were
        selected: four such waits per iteration of a 0.3 ms loop (``initialize_map``)."""
        grad = viewspace_point_tensor.grad
        if update_filter is None or update_filter.dtype is not torch.bool or update_filter.shape[0] != self.denom.shape[0]:
            self.xyz_gradient_accum[update_filter] += torch.norm(grad[update_filter, :2], dim=-1, keepdim=True)
            self.denom[update_filter] += 1
            return
        f = update_filter[:, None]
        norm = torch.norm(grad[:, :2], dim=-1, keepdim=True)
        self.xyz_gradient_accum += torch.where(f, norm, torch.zeros_like(norm)).to(self.xyz_gradient_accum.dtype)
        self.denom += f.to(self.denom.dtype)

    def densify_and_clone(self, grads, grad_threshold, scene_extent):
        sel = (torch.norm(grads, dim=-1) >= grad_threshold) & (self.get_scaling.max(dim=1).values <= self.percent_dense * scene_extent)
        idx = torch.nonzero(sel).squeeze(1)     # (one count-and-wait instead of one per tensor: see prune_points)
        idx_cpu = idx.cpu()
        take = lambda t: t.index_select(0, idx)
        self.densification_postfix(take(self._xyz), take(self._features_dc), take(self._features_rest), take(self._opacity),
                                   take(self._scaling), take(self._rotation),
                                   new_kf_ids=self.unique_kfIDs.index_select(0, idx_cpu), new_n_obs=self.n_obs.index_select(0, idx_cpu))

    def densify_and_split(self, grads, grad_threshold, scene_extent, N=2):
        n_init = self._xyz.shape[0]
        padded = torch.zeros(n_init, device=self.device)
        padded[: grads.shape[0]] = grads.squeeze()
        sel = (padded >= grad_threshold) & (self.get_scaling.max(dim=1).values > self.percent_dense * scene_extent)
        idx = torch.nonzero(sel).squeeze(1)     # (one count-and-wait instead of one per tensor: see prune_points)
        idx_cpu = idx.cpu()
        take = lambda t: t.index_select(0, idx)
        scaling_sel = take(self.get_scaling)
        stds = scaling_sel.repeat(N, 1)
        if self.generator is not None and self.generator.device.type == "cpu":
            samples = torch.randn(stds.shape, generator=self.generator).to(stds.device) * stds
        else:
            samples = torch.randn(stds.shape, device=stds.device, generator=self.generator) * stds
        rotation_sel = take(self._rotation)
        rots = build_rotation(rotation_sel).repeat(N, 1, 1)
        new_xyz = torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + take(self._xyz).repeat(N, 1)
        new_scaling = self.scaling_inverse_activation(scaling_sel.repeat(N, 1) / (0.8 * N))
        self.densification_postfix(new_xyz, take(self._features_dc).repeat(N, 1, 1), take(self._features_rest).repeat(N, 1, 1),
                                   take(self._opacity).repeat(N, 1), new_scaling, rotation_sel.repeat(N, 1),
                                   new_kf_ids=self.unique_kfIDs.index_select(0, idx_cpu).repeat(N), new_n_obs=self.n_obs.index_select(0, idx_cpu).repeat(N))
        self.prune_points(torch.cat((sel, torch.zeros(N * int(idx.numel()), device=self.device, dtype=torch.bool))))

    def densify_and_prune(self, max_grad, min_opacity, extent, max_screen_size):
        """slam_backend.py:132-137, :364-369."""
        grads = self.xyz_gradient_accum / self.denom
        grads = torch.where(grads.isnan(), torch.zeros_like(grads), grads)   # (`grads[grads.isnan()] = 0.0` without the host waiting for the count of NaNs)
        self.densify_and_clone(grads, max_grad, extent)
        self.densify_and_split(grads, max_grad, extent)
        prune = (self.get_opacity < min_opacity).squeeze(-1)
        if max_screen_size:
            big_vs = self.max_radii2D > max_screen_size
            big_ws = self.get_scaling.max(dim=1).values > 0.1 * extent
            prune = prune | big_vs | big_ws
        self.prune_points(prune)

    def reset_opacity(self):
        new = inverse_sigmoid(torch.ones_like(self.get_opacity) * 0.01)
        self._opacity = self.replace_tensor_to_optimizer(new.detach(), "opacity")["opacity"]

    def reset_opacity_nonvisible(self, visibility_filters):
        """Gaussians no view of the window saw go back to opacity 0.4 (slam_backend.py:372-375).

        As recalled from upstream, the visible ones get their ACTIVATED opacity written into the logit tensor
        (so 0.9 becomes sigmoid(0.9) = 0.71).  ``UPSTREAM_NONVISIBLE_RESET = False`` keeps their logits instead."""
        new = inverse_sigmoid(torch.ones_like(self.get_opacity) * 0.4)
        src = self.get_opacity.detach() if UPSTREAM_NONVISIBLE_RESET else self._opacity.detach()
        for f in visibility_filters:
            new[f] = src[f]
        self._opacity = self.replace_tensor_to_optimizer(new.detach(), "opacity")["opacity"]

    # ------------------------------------------------------------------ PLY (binary little endian, 3DGS attribute order)
    def construct_list_of_attributes(self):
        names = ["x", "y", "z", "nx", "ny", "nz"]
        names += [f"f_dc_{i}" for i in range(self._features_dc.shape[1] * self._features_dc.shape[2])]
        names += [f"f_rest_{i}" for i in range(self._features_rest.shape[1] * self._features_rest.shape[2])]
        names += ["opacity"]
        names += [f"scale_{i}" for i in range(self._scaling.shape[1])]
        names += [f"rot_{i}" for i in range(self._rotation.shape[1])]
        return names

    def save_ply(self, path):
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        n = self._xyz.shape[0]
        c = lambda t: t.detach().cpu().numpy().astype(np.float32)
        cols = np.concatenate([c(self._xyz), np.zeros((n, 3), np.float32),
                               c(self._features_dc.transpose(1, 2).flatten(start_dim=1)),
                               c(self._features_rest.transpose(1, 2).flatten(start_dim=1)),
                               c(self._opacity), c(self._scaling), c(self._rotation)], axis=1)
        names = self.construct_list_of_attributes()
        assert cols.shape[1] == len(names)
        header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % n
        header += "".join(f"property float {a}\n" for a in names) + "end_header\n"
        with open(path, "wb") as f:
            f.write(header.encode("ascii"))
            f.write(np.ascontiguousarray(cols, dtype="<f4").tobytes())

    def load_ply(self, path):
        with open(path, "rb") as f:
            names, n = [], 0
            while True:
                line = f.readline().decode("ascii").strip()
                if line.startswith("element vertex"):
                    n = int(line.split()[-1])
                elif line.startswith("property float"):
                    names.append(line.split()[-1])
                elif line == "end_header":
                    break
            data = np.frombuffer(f.read(n * len(names) * 4), dtype="<f4").reshape(n, len(names))
        col = {a: i for i, a in enumerate(names)}
        pick = lambda prefix: data[:, [col[a] for a in sorted((a for a in names if a.startswith(prefix)), key=lambda s: int(s.split("_")[-1]))]]
        p = lambda a: nn.Parameter(torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=self.device).requires_grad_(True))
        self._xyz = p(data[:, [col["x"], col["y"], col["z"]]])
        self._features_dc = p(pick("f_dc_").reshape(n, 3, -1).transpose(0, 2, 1))
        rest = pick("f_rest_")
        self._features_rest = p(rest.reshape(n, 3, -1).transpose(0, 2, 1) if rest.shape[1] else np.zeros((n, 0, 3), np.float32))
        self._opacity = p(data[:, [col["opacity"]]])
        self._scaling = p(pick("scale_"))
        self._rotation = p(pick("rot_"))
        self.active_sh_degree = self.max_sh_degree
        self._reset_bookkeeping(kf_id=0)
