"""Keyframe management on the renderer's outputs: ``is_keyframe`` and ``add_to_window``.

The two decisions the front end takes right after tracking a frame (reference utils/slam_frontend.py:1579-1674), restated
as functions of what they consume -- the tracked frame's ``n_touched > 0`` from ``render()`` and the window keyframes'
``occ_aware_visibility`` rows from the mapping iteration (both produced by this package's kernels) plus poses and the
median depth.  They are host logic on a handful of counts and 4x4 matrices; the point of having them here is the
fixture: ``tests/golden/loops.npz`` holds what the reference's own methods decided on the toy scene, and the replay
tests feed these functions the product's outputs (CPU renderer: exact; HIP renderer: same decisions).
"""
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from .graphics_utils import getWorld2View2


def _relative_translation(cam_a, cam_b) -> torch.Tensor:
    """|| translation of T_a @ inv(T_b) ||, T = world-to-camera of the keyframes."""
    T_a = getWorld2View2(cam_a.R, cam_a.T)
    T_b = getWorld2View2(cam_b.R, cam_b.T)
    return torch.norm((T_a @ torch.linalg.inv(T_b))[0:3, 3])


def covisibility(cur_visibility: torch.Tensor, kf_visibility: torch.Tensor) -> Tuple[int, int, int, int]:
    """(intersection, union, count of the frame, count of the keyframe) of two per-Gaussian visibility vectors."""
    a, b = cur_visibility.bool(), kf_visibility.bool()
    return (int(torch.logical_and(a, b).count_nonzero()), int(torch.logical_or(a, b).count_nonzero()),
            int(a.count_nonzero()), int(b.count_nonzero()))


def is_keyframe(config, cameras: Dict, cur_frame_idx, last_keyframe_idx, cur_frame_visibility_filter, occ_aware_visibility,
                median_depth) -> bool:
    """Does the tracked frame become a keyframe?  (reference :1579-1619)

    Yes when it has moved more than ``kf_translation`` median depths from the last keyframe, or more than
    ``kf_min_translation`` median depths while sharing less than ``kf_overlap`` (intersection over union) of the visible
    Gaussians with it.  A frame whose expanded static mask covers under 30 % of the image lowers the overlap bar to 70 %."""
    T = config["Training"]
    cur, last = cameras[cur_frame_idx], cameras[last_keyframe_idx]
    dist = float(_relative_translation(cur, last))
    far = dist > T["kf_translation"] * float(median_depth)
    moved = dist > T["kf_min_translation"] * float(median_depth)
    inter, union, _, _ = covisibility(cur_frame_visibility_filter, occ_aware_visibility[last_keyframe_idx])
    overlap_bar = T["kf_overlap"]
    mask = getattr(cur, "expanded_static_mask", None)
    if mask is not None and float(mask.float().mean()) < 0.3:
        overlap_bar = overlap_bar * 0.7
    ratio = inter / union if union else float("nan")
    return bool((ratio < overlap_bar and moved) or far)


def add_to_window(config, cameras: Dict, cur_frame_idx, cur_frame_visibility_filter, occ_aware_visibility, window: Sequence,
                  initialized: bool = True) -> Tuple[List, Optional[int]]:
    """The window with the new keyframe in front, and the keyframe that left it, if any (reference :1621-1674).

    The two newest keyframes always stay.  Of the others, the LAST one (oldest) whose covisibility with the new keyframe
    -- intersection over the smaller of the two visible sets -- is at most ``kf_cutoff`` leaves when the window is over
    ``window_size``; if the window is still too large, the keyframe with the largest
    sqrt(distance to the new keyframe) x sum of inverse distances to the other old keyframes leaves."""
    T = config["Training"]
    keep_newest = 2
    window = [cur_frame_idx] + list(window)
    removed = None
    cut_off = T.get("kf_cutoff", 0.4) if initialized else 0.4
    candidates = []
    for kf_idx in window[keep_newest:]:
        inter, _, n_cur, n_kf = covisibility(cur_frame_visibility_filter, occ_aware_visibility[kf_idx])
        denom = min(n_cur, n_kf)
        ratio = inter / denom if denom else float("nan")
        if ratio <= cut_off and len(window) > T["window_size"]:
            candidates.append(kf_idx)
    if candidates:
        removed = candidates[-1]
        window.remove(removed)
    if len(window) > T["window_size"]:
        cur = cameras[cur_frame_idx]
        old = window[keep_newest:]
        scores = []
        for i in old:
            others = sum(1.0 / (float(_relative_translation(cameras[i], cameras[j])) + 1e-6) for j in old if j != i)
            scores.append(float(torch.sqrt(_relative_translation(cameras[i], cur))) * others)
        removed = old[max(range(len(scores)), key=lambda k: (scores[k], -k))]   # the first of equal maxima, like argmax
        window.remove(removed)
    return window, removed
