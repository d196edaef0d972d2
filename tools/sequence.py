#!/usr/bin/env python3
"""The loops as a system: a synthetic drive through an opaque-surface scene at KITTI-07's geometry, from an EMPTY map --
initialize_map on frame 0, then per frame track_frame -> keyframe test -> seeding -> masked map_window bursts with densification /
pruning / opacity resets at the reference's cadence -> pruning pass, the back end's free-running iterations between frames, and at
the end ATE (Umeyama), PSNR before / after a colour refinement (lvdgs.slam_sequence.SlamSequence; reference utils/slam_frontend.py:1740-1899,
utils/slam_backend.py:485-609, utils/eval_utils_0806.py:33-306).

    python tools/sequence.py [--frames 60] [--scale 1.0] [--cadence reference|short] [--no-fused] [--idle 10] [--refine 500] [--no-masks]

Prints one JSON object (bench.py's config.side.sequence_kitti07_geom is the same record).  `--cadence reference`: the iteration
counts and densification schedule of configs/mono/KITTI/base_config.yaml as they are; `short`: every burst a fifth of it (the GPU
test's schedule).  The first three tracked frames of a reference-cadence run are the map's slowest (the map is one keyframe old)."""
import argparse
import copy
import json
import os
import random
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import lvdgs  # noqa: E402,F401
from lvdgs import synthetic  # noqa: E402
from lvdgs.gaussian_renderer import render  # noqa: E402
from lvdgs.slam_sequence import SlamSequence  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
PIPE = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False)


def sequence_config(W, H, **training):
    """The merged KITTI-07 config (tests/golden/config_07.json: ``load_config("configs/mono/KITTI/07.yaml")`` of the reference, captured by
    tests/golden/make_golden.py) with ``training`` laid over its Training block and the frame size set."""
    training = dict(training)
    cfg = copy.deepcopy(json.load(open(os.path.join(GOLDEN, "config_07.json"))))
    cfg["Training"]["monocular"] = cfg["Dataset"]["sensor_type"] == "monocular"   # (set by the absent slam.py entry point)
    lr = training.pop("lr", None)
    cfg["Training"].update(training)
    if lr:
        cfg["Training"]["lr"].update(lr)
    cfg["Dataset"]["Calibration"].update(width=W, height=H)
    cfg["Results"].update(save_results=False, use_gui=False)
    return cfg


def truth_model(W, H, n_true, r_min, r_max, margin, device, seed=11):
    from lvdgs.gaussian_model import GaussianModel
    g = synthetic.make_surface_gaussians(n_true, W, H, seed=seed, r_min=r_min, r_max=r_max, margin=margin)
    return GaussianModel.from_activated(g["means3D"], g["scales"], g["rotations"], g["opacities"], shs=g["shs"], device=device)


def empty_map(cfg, device):
    from lvdgs.gaussian_model import GaussianModel
    m = GaussianModel(cfg["model_params"]["sh_degree"], config=cfg, device=device)
    m.init_lr(cfg["opt_params"]["init_lr"])
    m.training_setup(cfg["opt_params"])
    return m


SHORT = dict(init_itr_num=210, init_gaussian_update=20, init_gaussian_reset=100, tracking_itr_num=40, mapping_itr_num=30,
             mapping_itr_nosingle=10, initial_ba_itr_num=60, gaussian_update_every=30, gaussian_update_offset=10, gaussian_reset=401)


# frame geometries: KITTI-07 (configs/mono/KITTI/07.yaml:8-18) and the waymo segment of BASELINE configs[4]'s size (configs/mono/waymo/405841.yaml:5-16,
# whose Training block also runs 30 mapping iterations per keyframe instead of 10)
GEOMETRY = {"kitti07": dict(W=1226, H=370, fx=707.0912, cx=601.8873, cy=183.1104, training={}),
            "waymo": dict(W=1920, H=1280, fx=2066.697564417299, cx=950.5512774150723, cy=641.1870541472169, training={"mapping_itr_nosingle": 30})}


def kitti_sequence(dev, frames=60, scale=1.0, cadence="reference", masks=True, n_true=None, seed=0, training=None, window_size=None, geometry="kitti07",
                   pcd_downsample=None):
    """(config, dataset, true map): a frame geometry of ``GEOMETRY`` (x ``scale``), the merged KITTI-07 config with the chosen cadence."""
    geo = GEOMETRY[geometry]
    W, H = int(round(geo["W"] * scale)), int(round(geo["H"] * scale))
    fx = fy = geo["fx"] * scale
    cx, cy = geo["cx"] * scale, geo["cy"] * scale
    tr = dict(SHORT) if cadence == "short" else {}
    tr.update(geo["training"])
    if window_size is not None:
        tr.update(window_size=window_size, pose_window=min(3, window_size - 1))
    tr.update(training or {})
    cfg = sequence_config(W, H, **tr)
    cfg["Dataset"]["Calibration"].update(fx=fx, fy=fy, cx=cx, cy=cy)
    if pcd_downsample is not None:   # (init, later keyframes): seeds per keyframe = valid pixels / this (configs/mono/KITTI/base_config.yaml:13-14: 32, 64)
        cfg["Dataset"].update(pcd_downsample_init=int(pcd_downsample[0]), pcd_downsample=int(pcd_downsample[1]))
    n_true = int(600_000 * scale * scale * (W * H) / (1226.0 * scale * 370.0 * scale) * (707.0912 / geo["fx"]) ** 2) if n_true is None and geometry != "kitti07" else n_true
    n_true = int(600_000 * scale * scale) if n_true is None else n_true
    # opaque surfaces of 1.5..16-pixel footprints (x scale) -- texture at the scale of a few pixels, which is what keeps a SLAM map's
    # Gaussians small and many (the 4..64-pixel footprints of the surface WORKLOADS render to a blur that a few hundred large Gaussians
    # reproduce: the map then prunes itself down to them) -- reaching 0.9 frame widths past the first frame's edges: the camera's field
    # of view is wider than the generator's (fx < W) and the camera moves
    truth = truth_model(W, H, n_true, 1.5 * scale, 16.0 * scale, 0.9, dev, seed=11 + seed)
    ds = synthetic.make_sequence(truth, render, PIPE, W, H, frames, dev, fx=fx, fy=fy, cx=cx, cy=cy, seed=seed, depth_noise=0.02,
                                 image_noise=0.01, dynamic_objects=masks, step=0.02, sway=0.15, yaw=0.03, period=40.0)
    return cfg, ds, truth


def run_sequence(dev, frames=60, scale=1.0, cadence="reference", fused="auto", idle=10, refine=500, masks=True, seed=0, training=None,
                 window_size=None, on_event=None, geometry="kitti07", pcd_downsample=None, **sequence_kwargs):
    torch.manual_seed(seed)
    random.seed(seed)
    cfg, ds, truth = kitti_sequence(dev, frames, scale, cadence, masks, seed=seed, training=training, window_size=window_size, geometry=geometry,
                                    pcd_downsample=pcd_downsample)
    del truth
    m = empty_map(cfg, dev)
    seq = SlamSequence(cfg, ds, m, PIPE, torch.zeros(3, device=dev), fused=fused, idle_map_iters=idle, on_event=on_event, **sequence_kwargs)
    seq.run()
    out = seq.summary()
    out["ate_rmse"] = seq.eval_ate()
    before = seq.eval_rendering()
    out["psnr_before_refinement"], out["ssim_before_refinement"] = before.get("psnr"), before.get("ssim")
    out["psnr_static_before_refinement"] = before.get("psnr_static")
    if refine:
        seq.refine(refine)
        after = seq.eval_rendering()
        out["psnr"], out["ssim"], out["psnr_static"] = after.get("psnr"), after.get("ssim"), after.get("psnr_static")
        s = seq.seconds["refinement"]
        out["refinement_ms_per_iteration"] = round(1e3 * s / refine, 4)
    err = seq.pose_errors()
    out["pose_error_unaligned_mean"], out["pose_error_unaligned_max"] = sum(err.values()) / len(err), max(err.values())
    out["trajectory_length"] = float(sum(float(torch.linalg.inv(ds.poses[i + 1])[:3, 3].sub(torch.linalg.inv(ds.poses[i])[:3, 3]).norm())
                                         for i in range(len(ds) - 1)))
    out["window_log"] = seq.window_log
    out["batched_window_runs"] = int(getattr(getattr(seq.backend, "_lvdgs_window_batch", None), "runs", 0))
    out.update(geometry=geometry, width=ds.width, height=ds.height, cadence=cadence, fused=fused, idle_map_iters=idle, keyframes_carry_static_mask=masks)
    return out, seq


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=60)
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--cadence", choices=["reference", "short"], default="reference")
    ap.add_argument("--no-fused", action="store_true")
    ap.add_argument("--idle", type=int, default=10)
    ap.add_argument("--refine", type=int, default=500)
    ap.add_argument("--no-masks", action="store_true")
    ap.add_argument("--window-size", type=int, default=None)
    ap.add_argument("--geometry", choices=sorted(GEOMETRY), default="kitti07")
    ap.add_argument("--pcd-downsample", type=int, nargs=2, default=None, metavar=("INIT", "KEYFRAME"),
                    help="seed one Gaussian per INIT valid pixels of frame 0 and per KEYFRAME of every later keyframe (the config's 32 / 64): smaller = a larger map")
    ap.add_argument("--seed", type=int, default=0, help="scene, trajectory noise, dynamic objects and the loops' random draws")
    ap.add_argument("--verbose", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    t0 = time.perf_counter()
    ev = (lambda e, s: print(f"  frame {s.counts['frames']:3d} {e:16s} N = {s._n()}", file=sys.stderr)) if a.verbose else None
    out, _ = run_sequence(dev, a.frames, a.scale, a.cadence, False if a.no_fused else "auto", a.idle, a.refine, not a.no_masks, seed=a.seed,
                          window_size=a.window_size, on_event=ev, geometry=a.geometry, pcd_downsample=a.pcd_downsample)
    out["seed"] = a.seed
    out["tool_seconds"] = round(time.perf_counter() - t0, 2)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
