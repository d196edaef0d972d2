#!/usr/bin/env python3
"""The two single-view loops of the back end at KITTI-07's geometry: milliseconds and device launches per iteration.

  * ``initialize_map`` (reference utils/slam_backend.py:95-149; 1050 iterations, configs/mono/KITTI/base_config.yaml:22): frame 0 of the
    synthetic drive (tools/sequence.py), seeds from its depth map, the reference's cadence of densification (every 100) and the
    opacity reset (iteration 500) -- the map grows from ~14 k to ~23 k Gaussians under the loop;
  * ``color_refinement`` (:393-468; 26000 iterations in a run): the 200 k-Gaussian KITTI-geometry map of bench.py with its twelve
    keyframes, with and without static masks, iterations timed in blocks after a warm-up.

    python tools/single_view_loops.py [init|refine|both] [--iters 400]

Prints one JSON object (bench.py's config.side.initialize_map_kitti07_geom / color_refinement_kitti07_geom[_masked] are these records).
"launches_per_iteration" counts the library's own launches (its HIP-event profile, the C ABI's lvdgs_profile_*): PyTorch's kernels
in the loop (optimizer bookkeeping, index updates) are NOT in it -- the rocprofv3 trace of this command shows those."""
import argparse
import gc
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

import bench  # noqa: E402
import lvdgs  # noqa: E402,F401
from lvdgs import _lib  # noqa: E402
from lvdgs.slam_loops import color_refinement, initialize_map  # noqa: E402


def launches_per_iteration(fn, iterations):
    _lib.profile_reset(); _lib.profile_enable(True)
    fn()
    torch.cuda.synchronize()
    _lib.profile_enable(False)
    times = _lib.profile_read()
    return round(sum(n for n, _ in times.values()) / iterations, 2), {k: round(1e3 * ms / iterations, 2) for k, (n, ms) in sorted(times.items(), key=lambda kv: -kv[1][1])}


def time_initialize_map(dev, iters=None, fused="auto"):
    import sequence as tool
    ss = tool
    from lvdgs.slam_sequence import SlamSequence
    torch.manual_seed(0); random.seed(0)
    cfg, ds, truth = tool.kitti_sequence(dev, frames=1, masks=False)
    del truth
    if iters is not None:
        cfg["Training"]["init_itr_num"] = iters
    seq = SlamSequence(cfg, ds, ss.empty_map(cfg, dev), ss.PIPE, torch.zeros(3, device=dev), fused=fused)
    vp = seq.new_viewpoint(0)
    seq.cameras[0] = vp
    vp.update_RT(vp.R_gt, vp.T_gt)
    depth_map = seq.add_new_keyframe(0, init=True)
    seq.gaussians.extend_from_pcd_seq(vp, kf_id=0, init=True, scale=2.0, depthmap=depth_map)
    n0 = seq._n()
    be = seq.backend
    be.viewpoints[0] = vp
    torch.cuda.synchronize()
    gc.collect(); gc.freeze()
    t = time.perf_counter()
    initialize_map(be, 0, vp, fused=fused)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    return {"iterations": be.init_itr_num, "ms_per_iteration": round(1e3 * dt / be.init_itr_num, 4), "seconds": round(dt, 3),
            "gaussians_first": n0, "gaussians_last": seq._n(), "width": ds.width, "height": ds.height,
            "densify_every": be.init_gaussian_update, "opacity_reset_at": be.init_gaussian_reset, "fused": fused}


def time_color_refinement(dev, masked, iters=400, workload="kitti07_geom", fused="auto"):
    torch.manual_seed(0); random.seed(0)
    model, _, _, (N, W, H) = bench.build_scene(workload, 0, dev)
    be, window = bench.build_window(workload, 12, dev, model, n_window=8, masked=masked)
    color_refinement(be, iteration_total=60, fused=fused)          # warm-up (buffers, code objects, clocks)
    torch.cuda.synchronize()
    gc.collect(); gc.freeze()
    blocks = []
    for _ in range(3):
        t = time.perf_counter()
        color_refinement(be, iteration_total=iters, fused=fused)
        torch.cuda.synchronize()
        blocks.append((time.perf_counter() - t) / iters)
    per = sorted(blocks)[1]
    n, kern = launches_per_iteration(lambda: color_refinement(be, iteration_total=100, fused=fused), 100)
    return {"iterations_per_block": iters, "ms_per_iteration": round(1e3 * per, 4), "ms_per_iteration_of_the_three_blocks": [round(1e3 * b, 4) for b in blocks],
            "library_launches_per_iteration": n, "library_kernels_us_per_iteration": kern, "gaussians": N, "width": W, "height": H,
            "keyframes": len(be.viewpoints), "keyframes_carry_static_mask": masked, "fused": fused}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="?", default="both", choices=["init", "refine", "both"])
    ap.add_argument("--iters", type=int, default=400)
    ap.add_argument("--no-fused", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    fused = False if a.no_fused else "auto"
    out = {}
    if a.what in ("init", "both"):
        out["initialize_map_kitti07_geom"] = time_initialize_map(dev, fused=fused)
    if a.what in ("refine", "both"):
        out["color_refinement_kitti07_geom"] = time_color_refinement(dev, False, a.iters, fused=fused)
        out["color_refinement_kitti07_geom_masked"] = time_color_refinement(dev, True, a.iters, fused=fused)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
