#!/usr/bin/env python3
"""Benchmark of the hot path: render + loss + backward iterations per second.

    python bench.py [--gpus N] [--steps K] [--warmup W]

N = 1 (default): BASELINE.json configs[2] -- 500k Gaussians, 1920x1080, one tracking iteration per
step: render(viewpoint, gaussians, pipe, bg) -> get_loss_tracking -> backward, with gradients to
every Gaussian parameter and to the 6-DoF camera pose (cam_rot_delta / cam_trans_delta) and the
exposure parameters (reference utils/slam_frontend.py:1492-1521).
N > 1 (launched by torch.distributed.run, one process per GPU): the mapping-window case -- every
rank renders its own keyframe of the same Gaussians with the mapping loss and the packed
Gaussian gradient (N x 14 floats) is sum-all-reduced with RCCL each step (weak scaling: one
keyframe per GPU, reference utils/slam_backend.py:180-306 sums the window's losses before one backward).

Rank 0 prints ONE JSON line.  `value` is whole-job iterations/s with all inputs resident in HBM.
`roofline` prices the dominant kernel (by HIP-event time measured here) against the 8 TB/s HBM
peak with the algorithmic byte count of SURVEY.md section 8(d); `cpu_baseline` times the CPU oracle
(a scalar C port, 1 core) on one iteration of the same scene.
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import lvdgs  # noqa: E402,F401
from lvdgs import _lib, rasterizer, slam_utils, synthetic, window_shard  # noqa: E402
from lvdgs.gaussian_model import GaussianModel  # noqa: E402
from lvdgs.gaussian_renderer import render  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
CONFIG = {"Training": {"monocular": True, "rgb_boundary_threshold": 0.01, "alpha": 0.98},
          "Dataset": {"depth_loss": True}}


def algorithmic_bytes(kernel, N, V, D, P, T):
    """SURVEY.md section 8(d) per-unit figures, split by kernel (documented in DESIGN.md)."""
    k = (32 + max(1, (T - 1).bit_length()) + 7) // 8
    table = {
        "preprocess_fwd": 56 * N + 48 * V,
        "radix": 12 * D + 24 * D * k,
        "tile_ranges": 8 * D + 8 * T,
        "blend_fwd": 44 * D + 28 * P,
        "blend_bwd": 44 * D + 28 * P + 44 * V,
        "preprocess_bwd": 56 * N + 92 * V + 56 * N + 12 * N + 24,
    }
    return table.get(kernel)


def build_scene(workload, rank, dev):
    cfg = synthetic.CONFIGS[workload]
    N, W, H = cfg["N"], cfg["W"], cfg["H"]
    g = synthetic.make_gaussians(N, W, H, seed=0)
    cam = synthetic.make_camera(W, H, pose_seed=None if rank == 0 else rank,
                                **{k: cfg[k] for k in ("fx", "fy", "cx", "cy") if k in cfg})
    for k in ("world_view_transform", "projection_matrix", "full_proj_transform", "camera_center"):
        setattr(cam, k, getattr(cam, k).to(dev))
    cam.cam_rot_delta = torch.nn.Parameter(torch.zeros(3, device=dev))
    cam.cam_trans_delta = torch.nn.Parameter(torch.zeros(3, device=dev))
    cam.exposure_a = torch.nn.Parameter(torch.zeros(1, device=dev))
    cam.exposure_b = torch.nn.Parameter(torch.zeros(1, device=dev))
    gen = torch.Generator().manual_seed(4242 + rank)
    cam.original_image = torch.rand(3, H, W, generator=gen).to(dev)
    cam.grad_mask = (torch.rand(1, H, W, generator=gen) > 0.5).to(dev)
    cam.mono_depth = (torch.rand(H, W, generator=gen) * 40 + 1).to(dev)
    model = GaussianModel.from_activated(g["means3D"], g["scales"], g["rotations"], g["opacities"], shs=g["shs"],
                                         sh_degree=0, device=dev)
    return model, cam, g, (N, W, H)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default=os.environ.get("LVDGS_BENCH_WORKLOAD", "cfg3_500k_1920x1080"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--step", choices=["auto", "tracking", "mapping"], default="auto",
                    help="auto: tracking iteration on 1 GPU, mapping-window iteration on N > 1")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch N > 1 as `python -m torch.distributed.run --nnodes=1 "
                         f"--nproc-per-node {args.gpus} --master-addr 127.0.0.1 --master-port P bench.py --gpus {args.gpus} ...`")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU path for the product)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    model, cam, g_cpu, (N, W, H) = build_scene(args.workload, rank, dev)
    pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False)
    bg = torch.zeros(3, device=dev)
    params = model.parameters()
    pose_params = [cam.cam_rot_delta, cam.cam_trans_delta, cam.exposure_a, cam.exposure_b]
    bucket = window_shard.GradientBucket(params) if world > 1 else None
    tracking = (world == 1) if args.step == "auto" else args.step == "tracking"
    stats = {}

    def step():
        for p in params + pose_params:
            p.grad = None
        pkg = render(cam, model, pipe, bg)
        if tracking:
            loss = slam_utils.get_loss_tracking(CONFIG, pkg["render"], pkg["depth"], pkg["opacity"], cam)
        else:
            loss = slam_utils.get_loss_mapping(CONFIG, pkg["render"], cam, depth=pkg["depth"])
        loss.backward()
        if world > 1:
            # one flat bucket: a single RCCL all-reduce per step (28 MB at 500k Gaussians)
            bucket.all_reduce()
        return pkg

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    rasterizer.KEEP_DEBUG_STATE = True
    for _ in range(args.warmup):
        pkg = step()
    sync()
    stats["D"] = int(rasterizer._DEBUG_LAST.get("num_rendered", 0))
    stats["V"] = int((pkg["radii"] > 0).sum().item())
    rasterizer.KEEP_DEBUG_STATE = False
    rasterizer._DEBUG_LAST.clear()

    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- per-kernel durations: the same steps again with HIP events around every launch ----
    roofline = None
    kernels = {}
    if rank == 0:
        _lib.profile_reset()
        _lib.profile_enable(True)
    for _ in range(args.steps):
        step()
    sync()
    if rank == 0:
        _lib.profile_enable(False)
        times = _lib.profile_read()
        P, T = W * H, ((W + 15) // 16) * ((H + 15) // 16)
        for name, (n, ms) in sorted(times.items(), key=lambda kv: -kv[1][1]):
            kernels[name] = {"launches_per_step": n / args.steps, "avg_us": 1e3 * ms / max(n, 1),
                             "us_per_step": 1e3 * ms / args.steps}
        dom = max(times, key=lambda k: times[k][1])
        n, ms = times[dom]
        avg_s = ms / 1e3 / max(n, 1)
        nbytes = algorithmic_bytes(dom, N, stats["V"], stats["D"], P, T)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            ent = tj.get(args.workload, {}).get(dom)
            if ent:
                traffic = ent.get("hbm_bytes_per_launch")
        if nbytes:
            achieved = nbytes / avg_s / 1e9
            roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                        "algorithmic_bytes_per_launch": int(nbytes), "avg_launch_us": round(avg_s * 1e6, 2)}

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_baseline = run_cpu_baseline(g_cpu, args.workload, N, W, H)

    if rank == 0:
        value = world * args.steps / elapsed
        out = {
            "metric": "render+backward iters/sec @500k Gaussians 1080p; 1/2/4/8-GPU scaling",
            "value": round(value, 3), "unit": "iters/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "gaussians": N, "width": W, "height": H, "visible": stats["V"],
                       "pairs": stats["D"], "sh_degree": 0,
                       "step": ("tracking iteration: render + get_loss_tracking + backward (pose + all Gaussian grads)"
                                if tracking else
                                "mapping window: one keyframe per GPU, render + get_loss_mapping + backward + RCCL all-reduce of the N x 14 gradient"),
                       "parallelism": f"keyframe-per-gpu x{world}" if world > 1 else "single"},
            "roofline": roofline, "cpu_baseline": cpu_baseline, "kernels_us_per_step": {k: round(v["us_per_step"], 2) for k, v in kernels.items()},
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


CPU_BASELINE_ITERS = 2  # about 16 s of CPU work at config 3: inside the 10-30 s window, never extrapolated


def run_cpu_baseline(g, workload, N, W, H):
    """Forward + backward of the CPU oracle (scalar C port, 1 core) on the same scene and camera, CPU_BASELINE_ITERS
    times.  The backward is fed fixed image gradients (colour, depth, opacity), not the tracking loss's: the oracle
    restates the rasterizer, the loss is outside it."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import oracle as orc

    cfg = synthetic.CONFIGS[workload]
    cam = synthetic.make_camera(W, H, **{k: cfg[k] for k in ("fx", "fy", "cx", "cy") if k in cfg})
    gc, gd, go = synthetic.make_image_grads(W, H, 0)
    o = orc.Oracle("f32")
    t0 = time.perf_counter()
    for _ in range(CPU_BASELINE_ITERS):
        o.forward(means3D=g["means3D"].numpy(), opacities=g["opacities"].numpy(), W=W, H=H, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
                  viewmatrix=cam.world_view_transform.numpy(), projmatrix=cam.full_proj_transform.numpy(),
                  projmatrix_raw=cam.projection_matrix.numpy(), campos=cam.camera_center.numpy(), bg=np.zeros(3),
                  scales=g["scales"].numpy(), rotations=g["rotations"].numpy(), colors_precomp=g["colors"].numpy())
        o.backward(gc.numpy(), gd.numpy(), go.numpy())
    dt = time.perf_counter() - t0
    o.free()
    return {"value": round(CPU_BASELINE_ITERS / dt, 5), "unit": "iters/s", "cores": 1, "kind": "port",
            "sample": f"{CPU_BASELINE_ITERS} iterations (rasterizer forward + backward with fixed image gradients; the loss is "
                      f"not part of the oracle) of the same {N}-Gaussian {W}x{H} scene, scalar C oracle on 1 core, {dt:.1f} s",
            "host_cores_available": os.cpu_count()}


if __name__ == "__main__":
    main()
