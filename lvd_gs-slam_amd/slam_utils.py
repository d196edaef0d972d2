"""Tracking / mapping photometric losses that consume the rasterizer's outputs.

Host-side mirror of the reference's ``utils/slam_utils.py`` (same function names,
arguments and config keys) so ``slam_frontend.py:1506`` and ``slam_backend.py:108,264,299``
work unchanged.  Pinned by ``tests/golden/loss_tracking.npz``, ``loss_mapping.npz`` and
``median_depth.npz`` (generated from the reference by ``tests/golden/make_golden.py``).

Differences from the reference, none of which change results: tensors are moved with
``.to(image.device)`` instead of a hard-coded ``.cuda()`` (utils/slam_utils.py:54,96,111),
and the Scharr kernels are created on the input's device (utils/slam_utils.py:9,12,28-29),
which is what lets the CPU test-suite run them.
"""
import torch
import torch.nn.functional as F

_SCHARR = ((3.0, 10.0, 3.0), (0.0, 0.0, 0.0), (-3.0, -10.0, -3.0))


def _depthwise3x3(x, k3x3, taps=None):
    """Valid 3 x 3 cross-correlation of every channel of ``x`` (1, C, H + 2, W + 2) with ``k3x3`` (``taps``: the same nine weights as
    nested Python tuples).  On the GPU as shifted multiply-adds (zero taps skipped): ``F.conv2d`` goes through MIOpen, whose set-up
    costs 10-15 ms per CALL for a KITTI-size frame (tools/soak_densify.py under cProfile) -- every new frame of a sequence pays it
    three times in ``Camera.compute_grad_mask``.  The CPU keeps ``conv2d`` (the goldens of the host mirrors were taken with it)."""
    if not x.is_cuda or taps is None:
        c = x.shape[1]
        return F.conv2d(x, k3x3.view(1, 1, 3, 3).repeat(c, 1, 1, 1), groups=c)
    h, w = x.shape[-2] - 2, x.shape[-1] - 2
    out = None
    for i in range(3):
        for j in range(3):
            v = float(taps[i][j])
            if v != 0.0:
                term = x[..., i:i + h, j:j + w] * v
                out = term if out is None else out + term
    return out if out is not None else torch.zeros_like(x[..., :h, :w])


def image_gradient(image):
    """Normalised Scharr gradients (vertical, horizontal) of a (C,H,W) image with
    reflect padding (utils/slam_utils.py:5-22)."""
    kv = torch.tensor(_SCHARR, dtype=torch.float32, device=image.device)
    kh = kv.t().contiguous()
    scale = 1.0 / kv.abs().sum()
    padded = F.pad(image, (1, 1, 1, 1), mode="reflect")[None]
    kh_taps = tuple(zip(*_SCHARR))
    return (scale * _depthwise3x3(padded, kv, _SCHARR))[0], (scale * _depthwise3x3(padded, kh, kh_taps))[0]


def image_gradient_mask(image, eps=0.01):
    """True where the whole 3x3 neighbourhood has |I| > eps (utils/slam_utils.py:25-39).
    Returned twice (v, h) like the reference."""
    ones = torch.ones((3, 3), dtype=torch.float32, device=image.device)
    padded = F.pad(image, (1, 1, 1, 1), mode="reflect")[None]
    valid = (padded.abs() > eps).float()
    full = _depthwise3x3(valid, ones, ((1.0,) * 3,) * 3)[0] == 9.0
    return full, full.clone()


def _fusable(image):
    """The fused HIP loss handles float32 GPU renders; anything else takes the PyTorch formulas below."""
    return image.is_cuda and image.dtype == torch.float32 and USE_FUSED_LOSS


USE_FUSED_LOSS = True


def _gt_image(viewpoint, like):
    return viewpoint.original_image.to(like.device)


def _mono_depth(viewpoint, like):
    md = viewpoint.mono_depth
    cache = getattr(viewpoint, "_lvdgs_mono_depth", None)
    if cache is not None and cache[0] is md and cache[1].device == like.device:
        return cache[1]
    t = md if torch.is_tensor(md) else torch.from_numpy(md)
    t = t.to(dtype=torch.float32, device=like.device)[None]
    try:  # the reference re-uploads the numpy depth every call; keep the device copy with the viewpoint
        viewpoint._lvdgs_mono_depth = (md, t)
    except Exception:
        pass
    return t


def _static_mask_bytes(viewpoint, like):
    """The viewpoint's ``static_mask`` (bool, True = static; (H,W) or with a unit dimension) as H*W bytes on ``like``'s device,
    16-byte aligned, for the masked-loss kernels.  The reference uploads and converts the mask in every iteration of every
    window view (utils/slam_backend.py:199); here the device copy stays with the viewpoint for as long as the mask tensor is the
    same object and has not been written to (``Tensor._version``)."""
    m = viewpoint.static_mask
    version = m._version if torch.is_tensor(m) else None   # (a numpy mask: keyed on identity alone)
    cache = getattr(viewpoint, "_lvdgs_static_mask", None)
    if cache is not None and cache[0] is m and cache[1] == version and cache[2].device == like.device:
        return cache[2]
    t = (m if torch.is_tensor(m) else torch.as_tensor(m)).to(like.device).reshape(-1)
    t = (t.view(torch.uint8) if t.dtype == torch.bool else t.ne(0).view(torch.uint8)).contiguous()
    if t is m or (torch.is_tensor(m) and m.is_cuda and t.data_ptr() == m.data_ptr()) or t.data_ptr() % 16:
        t = t.clone()   # (never an alias of the caller's tensor: a later in-place edit must not change the cached bytes unnoticed)
    try:
        viewpoint._lvdgs_static_mask = (m, version, t)
    except Exception:
        pass
    return t


def _exposure(image, viewpoint):
    return torch.exp(viewpoint.exposure_a) * image + viewpoint.exposure_b


def get_loss_tracking(config, image, depth, opacity, viewpoint, initialization=False):
    """Tracking loss on the exposure-corrected render (utils/slam_utils.py:42-50).
    With ``monocular`` set the RGB-only branch is taken whether or not
    ``Dataset.depth_loss`` is set, exactly as in the reference."""
    if _fusable(image):
        from .fused_loss import photometric_loss
        thr = config["Training"]["rgb_boundary_threshold"]
        common = dict(opacity=opacity, exposure_a=viewpoint.exposure_a, exposure_b=viewpoint.exposure_b,
                      grad_mask=viewpoint.grad_mask, rgb_boundary_threshold=thr, weight_by_opacity=True)
        if config["Training"]["monocular"]:
            return photometric_loss(image, _gt_image(viewpoint, image), **common)
        alpha = config["Training"].get("alpha", 0.95)
        return photometric_loss(image, _gt_image(viewpoint, image), depth=depth, gt_depth=_mono_depth(viewpoint, image),
                                weight_rgb=alpha, weight_depth=1 - alpha, depth_needs_opaque=True, **common)
    image_ab = _exposure(image, viewpoint)
    if config["Training"]["monocular"]:
        return get_loss_tracking_rgb(config, image_ab, depth, opacity, viewpoint)
    return get_loss_tracking_rgbd(config, image_ab, depth, opacity, viewpoint)


def get_loss_tracking_rgb(config, image, depth, opacity, viewpoint):
    """mean(opacity * |image - gt| on (sum_c gt > thr) & grad_mask) (utils/slam_utils.py:53-62)."""
    gt = _gt_image(viewpoint, image)
    _, h, w = gt.shape
    keep = (gt.sum(dim=0) > config["Training"]["rgb_boundary_threshold"]).view(1, h, w)
    keep = keep * viewpoint.grad_mask
    return (opacity * torch.abs(image * keep - gt * keep)).mean()


def get_loss_tracking_rgbd(config, image, depth, opacity, viewpoint, initialization=False):
    """alpha * rgb + (1-alpha) * mean|depth - gt_depth| on valid & opaque pixels
    (utils/slam_utils.py:65-79)."""
    alpha = config["Training"].get("alpha", 0.95)
    gt_depth = _mono_depth(viewpoint, image)
    keep = (gt_depth > 0.01).view(*depth.shape) * (opacity > 0.95).view(*depth.shape)
    l1_rgb = get_loss_tracking_rgb(config, image, depth, opacity, viewpoint)
    l1_depth = torch.abs(depth * keep - gt_depth * keep)
    return alpha * l1_rgb + (1 - alpha) * l1_depth.mean()


def get_loss_mapping(config, image, viewpoint, depth=None, initialization=False, monodepth=True):
    """Mapping loss dispatcher (utils/slam_utils.py:82-92)."""
    rgb_only = config["Training"]["monocular"] and not monodepth
    if _fusable(image):
        from .fused_loss import photometric_loss
        kw = dict(rgb_boundary_threshold=config["Training"]["rgb_boundary_threshold"])
        if not initialization:
            kw.update(exposure_a=viewpoint.exposure_a, exposure_b=viewpoint.exposure_b)
        if rgb_only:
            return photometric_loss(image, _gt_image(viewpoint, image), **kw)
        alpha = config["Training"].get("alpha", 0.95)
        return photometric_loss(image, _gt_image(viewpoint, image), depth=depth, gt_depth=_mono_depth(viewpoint, image),
                                weight_rgb=alpha, weight_depth=1 - alpha, **kw)
    image_ab = image if initialization else _exposure(image, viewpoint)
    if rgb_only:
        return get_loss_mapping_rgb(config, image_ab, viewpoint)
    return get_loss_mapping_rgbd(config, image_ab, depth, viewpoint)


def get_loss_mapping_rgb(config, image, viewpoint):
    """mean|image - gt| where sum_c gt > thr (utils/slam_utils.py:95-104)."""
    gt = _gt_image(viewpoint, image)
    _, h, w = gt.shape
    keep = (gt.sum(dim=0) > config["Training"]["rgb_boundary_threshold"]).view(1, h, w)
    return torch.abs(image * keep - gt * keep).mean()


def get_loss_mapping_rgbd(config, image, depth, viewpoint, initialization=False):
    """alpha * mean|d rgb| + (1-alpha) * mean|d depth| (utils/slam_utils.py:107-121)."""
    alpha = config["Training"].get("alpha", 0.95)
    gt = _gt_image(viewpoint, image)
    gt_depth = _mono_depth(viewpoint, image)
    keep_rgb = (gt.sum(dim=0) > config["Training"]["rgb_boundary_threshold"]).view(*depth.shape)
    keep_d = (gt_depth > 0.01).view(*depth.shape)
    l1_rgb = torch.abs(image * keep_rgb - gt * keep_rgb)
    l1_depth = torch.abs(depth * keep_d - gt_depth * keep_d)
    return alpha * l1_rgb.mean() + (1 - alpha) * l1_depth.mean()


def get_median_depth(depth, opacity=None, mask=None, return_std=False):
    """Median of rendered depth over depth>0, opacity>0.95 (and mask)
    (utils/slam_utils.py:124-134)."""
    depth = depth.detach().clone()
    valid = depth > 0
    if opacity is not None:
        valid = torch.logical_and(valid, opacity.detach() > 0.95)
    if mask is not None:
        valid = torch.logical_and(valid, mask)
    picked = depth[valid]
    if return_std:
        return picked.median(), picked.std(), valid
    return picked.median()
