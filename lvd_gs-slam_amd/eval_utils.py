"""Trajectory and rendering metrics of the evaluation harness, without its plotting / logging dependencies.

Counterpart of the numbers ``utils/eval_utils_0806.py`` produces (it needs ``evo``, ``wandb``, ``lpips``,
``cv2`` and ``matplotlib``, none of which is part of the hot path):

* ``evaluate_ate`` -- ``evaluate_evo`` (:33-98): Umeyama alignment of the estimated keyframe trajectory onto
  the ground truth (with scale for monocular runs), falling back to aligning the first poses when either
  trajectory spans less than 0.1 m (:42-56); APE on the translation part; ``rmse`` and evo's other statistics.
  ``evo`` is not installed, so the alignment follows the published algorithm (Umeyama, PAMI 1991) [unpinned].
* ``eval_ate`` -- (:101-169): camera-to-world poses from the keyframes' ``R, T`` / ``R_gt, T_gt``.
* ``frame_metrics`` -- the per-frame block of ``eval_rendering`` (:231-306): PSNR on non-black pixels, SSIM on
  the full frame, and the static-region variants (mask = non-black & static; SSIM with the dynamic pixels of
  both images overwritten by the background colour).  SSIM runs on the fused HIP kernel.
"""
import numpy as np
import torch


class AlignmentError(ValueError):
    pass


def umeyama_alignment(x, y, with_scale=False):
    """Least-squares similarity (r, t, c) with ``y ~ c * r @ x + t`` for point sets ``x, y`` of shape (m, n)."""
    x, y = np.asarray(x, np.float64), np.asarray(y, np.float64)
    if x.shape != y.shape:
        raise AlignmentError("point sets must have the same shape")
    m, n = x.shape
    mean_x, mean_y = x.mean(axis=1), y.mean(axis=1)
    sigma_x = ((x - mean_x[:, None]) ** 2).sum() / n
    cov = (y - mean_y[:, None]) @ (x - mean_x[:, None]).T / n
    u, d, vt = np.linalg.svd(cov)
    if np.count_nonzero(d > np.finfo(d.dtype).eps) < m - 1:
        raise AlignmentError("degenerate covariance rank, Umeyama alignment is not possible")
    s = np.eye(m)
    if np.linalg.det(u) * np.linalg.det(vt) < 0.0:
        s[m - 1, m - 1] = -1.0
    r = u @ s @ vt
    c = float(np.trace(np.diag(d) @ s) / sigma_x) if with_scale else 1.0
    t = mean_y - c * (r @ mean_x)
    return r, t, c


def trajectory_has_diversity(poses, min_translation=0.1):
    """utils/eval_utils_0806.py:42-48."""
    if len(poses) < 3:
        return False
    positions = np.array([p[:3, 3] for p in poses])
    return float(np.sqrt(np.sum(np.ptp(positions, axis=0) ** 2))) > min_translation


def align_trajectory(poses_est, poses_ref, correct_scale=False):
    """Similarity-align estimated camera-to-world poses onto the reference ones (positions drive the fit)."""
    est = [np.asarray(p, np.float64).copy() for p in poses_est]
    ref = [np.asarray(p, np.float64) for p in poses_ref]
    r, t, c = umeyama_alignment(np.array([p[:3, 3] for p in est]).T, np.array([p[:3, 3] for p in ref]).T, correct_scale)
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = r, t
    out = []
    for p in est:
        p[:3, 3] *= c
        out.append(T @ p)
    return out


def align_trajectory_origin(poses_est, poses_ref):
    """Rigidly move the estimate so that its first pose coincides with the reference's first pose."""
    to_ref = np.asarray(poses_ref[0], np.float64) @ np.linalg.inv(np.asarray(poses_est[0], np.float64))
    return [to_ref @ np.asarray(p, np.float64) for p in poses_est]


def evaluate_ate(poses_gt, poses_est, monocular=False):
    """-> (rmse, statistics dict, aligned estimate); mirrors evaluate_evo's choice of alignment."""
    try:
        if not trajectory_has_diversity(poses_gt) or not trajectory_has_diversity(poses_est):
            aligned = align_trajectory_origin(poses_est, poses_gt)
        else:
            aligned = align_trajectory(poses_est, poses_gt, correct_scale=monocular)
    except AlignmentError:
        aligned = [np.asarray(p, np.float64) for p in poses_est]
    err = np.array([np.linalg.norm(a[:3, 3] - np.asarray(g, np.float64)[:3, 3]) for a, g in zip(aligned, poses_gt)])
    stats = {"rmse": float(np.sqrt(np.mean(err ** 2))), "mean": float(err.mean()), "median": float(np.median(err)),
             "std": float(err.std()), "min": float(err.min()), "max": float(err.max()), "sse": float(np.sum(err ** 2))}
    return stats["rmse"], stats, aligned


def eval_ate(frames, kf_ids, monocular=False):
    """ATE RMSE over the keyframes (utils/eval_utils_0806.py:101-169); None with fewer than 3."""
    if len(frames) < 3 or len(kf_ids) < 3:
        return None

    def c2w(R, T):
        pose = np.eye(4)
        pose[:3, :3] = R.detach().cpu().numpy()
        pose[:3, 3] = T.detach().cpu().numpy()
        return np.linalg.inv(pose)

    est = [c2w(frames[k].R, frames[k].T) for k in kf_ids]
    gt = [c2w(frames[k].R_gt, frames[k].T_gt) for k in kf_ids]
    return evaluate_ate(gt, est, monocular=monocular)[0]


def frame_metrics(rendering, gt_image, static_mask=None, background=None):
    """PSNR / SSIM of one rendered frame and their static-region variants (utils/eval_utils_0806.py:231-306)."""
    from .image_utils import psnr
    from .loss_utils import ssim

    image = torch.clamp(rendering.detach(), 0.0, 1.0)
    basic = gt_image > 0
    out = {"psnr": float(psnr(image[basic].unsqueeze(0), gt_image[basic].unsqueeze(0))),
           "ssim": float(ssim(image.unsqueeze(0), gt_image.unsqueeze(0)))}
    out["psnr_static"], out["ssim_static"], out["static_ratio"] = out["psnr"], out["ssim"], None
    if static_mask is not None:
        m = static_mask
        if m.dim() == 3:
            m = m.squeeze(-1) if m.shape[-1] == 1 else m.squeeze(0)
        if m.dim() == 2:
            m = m.unsqueeze(0).expand(3, -1, -1)
        keep = basic & m.to(device=image.device, dtype=torch.bool)
        if bool(keep.any()):
            bg = torch.zeros(3, device=image.device) if background is None else background.to(image.device)
            fill = bg.view(3, 1, 1).expand_as(image)
            out["psnr_static"] = float(psnr(image[keep].unsqueeze(0), gt_image[keep].unsqueeze(0)))
            out["ssim_static"] = float(ssim(torch.where(keep, image, fill).unsqueeze(0), torch.where(keep, gt_image, fill).unsqueeze(0)))
            out["static_ratio"] = float(keep.float().mean())
    return out
