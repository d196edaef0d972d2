"""``render()`` facade: builds the raster settings from a ``Camera`` and returns the dict the
SLAM loops consume.

Counterpart of ``gaussian_splatting.gaussian_renderer.render`` (absent from the reference
checkout).  The contract is pinned by the reference's call sites:

* four positional arguments ``render(viewpoint, gaussians, pipeline_params, background)``
  (utils/slam_frontend.py:1493, utils/slam_backend.py:98,184,277,407, utils/eval_utils_0806.py:215);
* returned keys ``render, viewspace_points, visibility_filter, radii, depth, opacity, n_touched``
  (utils/slam_backend.py:110-116);
* ``render_with_custom_resolution(..., target_width=, target_height=)["depth"]`` (utils/init_pose.py:145-146);
* read from the camera: ``world_view_transform, full_proj_transform, projection_matrix, camera_center,
  FoVx, FoVy, image_height, image_width, cam_rot_delta, cam_trans_delta`` (utils/camera_utils.py);
* read from the Gaussian model: ``get_xyz, get_opacity, get_scaling, get_rotation, get_features,
  active_sh_degree`` (and ``get_covariance(scaling_modifier)`` when ``pipe.compute_cov3D_python``);
* ``pipeline_params.convert_SHs_python`` / ``compute_cov3D_python`` (configs/mono/KITTI/base_config.yaml:94-96).
Returns ``None`` for an empty model, like upstream.
"""
import math

import torch

from . import rasterizer as _rz
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer

SH_C0 = 0.28209479177387814

# When the model exposes its raw parameters under the published 3DGS names (_scaling, _rotation, _opacity) and
# uses the published activations (exp, normalize, sigmoid), render() hands the raw tensors to the rasterizer,
# which applies the activations inside its projection kernel and the chain rule inside its backward kernel:
# same values as pc.get_scaling / get_rotation / get_opacity, about a dozen elementwise kernels fewer per
# iteration.  Set to False to always go through the model's accessors.
FUSE_ACTIVATIONS = True


def _raw_parameters(pc):
    """(log-scales, raw quaternions, opacity logits) if the fused path is safe for this model, else None."""
    if not FUSE_ACTIVATIONS:
        return None
    raw = tuple(getattr(pc, n, None) for n in ("_scaling", "_rotation", "_opacity"))
    if any(not torch.is_tensor(t) for t in raw):
        return None
    if getattr(pc, "standard_activations", False):
        return raw
    acts = (getattr(pc, "scaling_activation", None), getattr(pc, "rotation_activation", None),
            getattr(pc, "opacity_activation", None))
    if acts[0] is torch.exp and acts[1] is torch.nn.functional.normalize and acts[2] is torch.sigmoid:
        return raw
    return None


def _eval_sh_python(pc, viewpoint_camera):
    """SH -> RGB in PyTorch (only for pipe.convert_SHs_python; degree 0..1 exactness is not the hot path)."""
    from .sh_utils import eval_sh

    shs_view = pc.get_features.transpose(1, 2).view(-1, 3, (pc.max_sh_degree + 1) ** 2)
    dirs = pc.get_xyz - viewpoint_camera.camera_center.repeat(pc.get_features.shape[0], 1)
    dirs = dirs / dirs.norm(dim=1, keepdim=True)
    return torch.clamp_min(eval_sh(pc.active_sh_degree, shs_view, dirs) + 0.5, 0.0)


def _render(viewpoint_camera, pc, pipe, bg_color, image_height, image_width, scaling_modifier=1.0,
            override_color=None, mask=None):
    xyz = pc.get_xyz
    if xyz.shape[0] == 0:
        return None
    # leaf that receives d(loss)/d(NDC xy) in .grad after backward (upstream builds it as zeros + 0 with
    # retain_grad(); a plain leaf gives callers the same .grad with one kernel and one autograd node less)
    screenspace_points = torch.zeros_like(xyz, dtype=xyz.dtype, requires_grad=True, device=xyz.device)

    raster_settings = GaussianRasterizationSettings(
        image_height=int(image_height), image_width=int(image_width),
        tanfovx=math.tan(viewpoint_camera.FoVx * 0.5), tanfovy=math.tan(viewpoint_camera.FoVy * 0.5),
        bg=bg_color, scale_modifier=scaling_modifier,
        viewmatrix=viewpoint_camera.world_view_transform, projmatrix=viewpoint_camera.full_proj_transform,
        projmatrix_raw=viewpoint_camera.projection_matrix, sh_degree=pc.active_sh_degree,
        campos=viewpoint_camera.camera_center, prefiltered=False, debug=False)

    means3D, means2D = xyz, screenspace_points
    scales = rotations = cov3D_precomp = None
    activations = 0
    raw = None if getattr(pipe, "compute_cov3D_python", False) else _raw_parameters(pc)
    if raw is not None:
        scales, rotations, opacity = raw
        activations = _rz.ACT_EXP_SCALES | _rz.ACT_NORMALIZE_ROTATIONS | _rz.ACT_SIGMOID_OPACITIES
    else:
        opacity = pc.get_opacity
        if getattr(pipe, "compute_cov3D_python", False):
            cov3D_precomp = pc.get_covariance(scaling_modifier)
        else:
            scales, rotations = pc.get_scaling, pc.get_rotation

    shs = colors_precomp = None
    if override_color is not None:
        colors_precomp = override_color
    elif getattr(pipe, "convert_SHs_python", False):
        colors_precomp = _eval_sh_python(pc, viewpoint_camera)
    else:
        shs = pc.get_features

    sel = (lambda t: t) if mask is None else (lambda t: None if t is None else t[mask])
    # GaussianRasterizer(raster_settings)(...) without building an nn.Module per frame (same autograd function)
    rendered_image, radii, depth, opacity_img, n_touched = _rz.rasterize_gaussians(
        sel(means3D), sel(means2D), sel(shs), sel(colors_precomp), sel(opacity), sel(scales), sel(rotations),
        sel(cov3D_precomp), viewpoint_camera.cam_rot_delta, viewpoint_camera.cam_trans_delta, raster_settings, activations)
    return {"render": rendered_image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0,
            "radii": radii, "depth": depth, "opacity": opacity_img, "n_touched": n_touched}


def render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None, mask=None):
    return _render(viewpoint_camera, pc, pipe, bg_color, viewpoint_camera.image_height, viewpoint_camera.image_width,
                   scaling_modifier, override_color, mask)


def render_with_custom_resolution(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None,
                                  mask=None, target_width=None, target_height=None):
    """Same view rendered at another raster size (the fields of view are kept, utils/init_pose.py:141-146)."""
    H = viewpoint_camera.image_height if target_height is None else target_height
    W = viewpoint_camera.image_width if target_width is None else target_width
    return _render(viewpoint_camera, pc, pipe, bg_color, H, W, scaling_modifier, override_color, mask)
