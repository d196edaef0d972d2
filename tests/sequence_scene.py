"""Scaffolding of the sequence tests (tests/test_sequence.py on the CPU, tests/test_gpu_sequence.py on the GPU): the merged KITTI-07
config with the schedule scaled down to a few dozen frames, a seeded surface scene, the frames of a camera driving through it.

The schedule keeps the reference's STRUCTURE at a smaller period (configs/mono/KITTI/base_config.yaml:22-52): map initialisation
with its own densification / reset cadence, a per-keyframe burst of mapping iterations, the free-running iterations with a pruning
pass every ten, densify / prune every ``gaussian_update_every`` at offset ``gaussian_update_offset``, the opacity reset of the
non-visible every ``gaussian_reset``, a window that fills (the monocular "initial BA" burst when it does) and then slides."""
import copy
import json
import os
from types import SimpleNamespace

import torch

HERE = os.path.dirname(os.path.abspath(__file__))


import sys as _sys
_sys.path.insert(0, os.path.join(HERE, "..", "tools"))
from sequence import PIPE, empty_map, sequence_config, truth_model  # noqa: E402,F401  (tools/sequence.py: shared with bench.py)


TOY = dict(W=64, H=48, n_true=420, r_min=2.0, r_max=9.0, margin=0.5, n_frames=18, step=0.025, sway=0.08, yaw=0.02, period=14.0)
TOY_TRAINING = dict(init_itr_num=48, init_gaussian_update=15, init_gaussian_reset=12, init_gaussian_th=0.005, init_gaussian_extent=30,
                    tracking_itr_num=16, mapping_itr_num=5, mapping_itr_nosingle=4, initial_ba_itr_num=6, gaussian_update_every=6,
                    gaussian_update_offset=2, gaussian_reset=17, gaussian_th=0.4, size_threshold=30, window_size=3, pose_window=2,
                    kf_interval=2, kf_overlap=0.95, kf_translation=0.03, kf_min_translation=0.02, prune_num=1, depth_lambda=0.1,
                    lr=dict(cam_trans_delta=0.005))   # (sixteen tracking iterations per frame instead of a hundred: the step a frame may need stays in reach)


def toy_dataset_overrides(cfg):
    cfg["Dataset"].update(pcd_downsample=24, pcd_downsample_init=10, point_size=0.3, adaptive_pointsize=False)
    cfg["opt_params"]["opacity_lr"] = 0.2   # (an opacity reset is followed by tens of iterations here, not hundreds: logits must be able to come back)
    cfg["opt_params"]["densify_grad_threshold"] = 0.002   # (64-pixel frames: a pixel is 1 / 32 of NDC, gradients per unit of NDC are small multiples of it)
    return cfg


def cpu_hooks():
    """The dense float64 renderer, the loss oracle and the brute-force neighbour search where the product has HIP kernels only."""
    import sys
    sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
    import aux_oracle
    import test_loop_golden as tl
    from dense_render import dense_render

    def knn(points):
        import numpy as np
        return torch.from_numpy(aux_oracle.dist2_knn3(points.detach().cpu().numpy().astype(np.float64))).to(points)

    def psnr_only(rendering, gt_image, static_mask=None, background=None):
        image = torch.clamp(rendering.detach(), 0.0, 1.0)
        keep = gt_image > 0
        mse = ((image[keep] - gt_image[keep]) ** 2).mean()
        return {"psnr": float(20 * torch.log10(1.0 / torch.sqrt(mse)))}

    return dict(render_fn=dense_render, view_loss_fn=tl._cpu_view_loss, refine_loss_fn=tl._cpu_refine_loss), knn, psnr_only


def toy_sequence_on_cpu(dynamic_objects=False, n_frames=None):
    """(config, dataset on the CPU rendered by the dense renderer, hooks)."""
    from lvdgs import synthetic
    hooks, knn, psnr_only = cpu_hooks()
    t = TOY
    cfg = toy_dataset_overrides(sequence_config(t["W"], t["H"], **TOY_TRAINING))
    truth = truth_model(t["W"], t["H"], t["n_true"], t["r_min"], t["r_max"], t["margin"], "cpu")
    ds = synthetic.make_sequence(truth, hooks["render_fn"], PIPE, t["W"], t["H"], n_frames or t["n_frames"], "cpu", seed=3, depth_noise=0.01,
                                 dynamic_objects=dynamic_objects, step=t["step"], sway=t["sway"], yaw=t["yaw"], period=t["period"])
    return cfg, ds, hooks, knn, psnr_only
