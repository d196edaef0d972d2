#!/usr/bin/env python3
"""Benchmark of the hot path: render + loss + backward iterations per second.

    python bench.py [--gpus N] [--steps K] [--warmup W]

N = 1 (default): BASELINE.json configs[2] -- 500k Gaussians, 1920x1080, one tracking iteration per
step: render(viewpoint, gaussians, pipe, bg) -> get_loss_tracking -> backward, with gradients to
every Gaussian parameter and to the 6-DoF camera pose (cam_rot_delta / cam_trans_delta) and the
exposure parameters (reference utils/slam_frontend.py:1492-1521).
N > 1 (launched by torch.distributed.run, one process per GPU): one WHOLE mapping iteration of the back end
(lvdgs.backend_map.map_window = reference utils/slam_backend.py:167-390) on the reference's window -- 8 keyframes + 2
random older ones per iteration (configs/mono/KITTI/base_config.yaml:37, utils/slam_backend.py:275) -- whose ten views
are dealt to the GPUs whole and in bands of tile rows (backend_map.plan_pieces: 1.25 views of work per GPU at N = 8),
STRONG scaling: render + mapping loss + backward of the rank's pieces, the RCCL collectives (float SUM bucket: Gaussian
gradients, keyframe pose / exposure gradients, densification statistics; int32 MAX: radii; uint8 MAX: visibility
flags), the bookkeeping, the Adam step over all Gaussians, the keyframe Adam step and the pose retraction.  `value`
counts view renders+backwards per second over the job (10 per iteration); config.same_step_on_one_gpu_iters_per_s is
the SAME window on one GPU measured inside the job, config.phases_us_per_step the iteration by phase.
`--window weak` is the round-2 line instead: a window of N keyframes, one whole keyframe per GPU, no random views.

Rank 0 prints ONE JSON line.  `value` is whole-job iterations/s with all inputs resident in HBM.
`roofline` prices the dominant kernel (by HIP-event time measured here) against the 8 TB/s HBM
peak with the algorithmic byte count of SURVEY.md section 8(d); `cpu_baseline` times the CPU oracle
(a scalar C port, 1 core) on one iteration of the same scene.
"""
import argparse
import gc
import json
import os
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# (the pool's host driver supports dmabuf IPC only: without this RCCL's buffer sharing between the ranks of a node fails with
# `hipIpcGetMemHandle: invalid argument`; exported by the environment already, kept here for a launcher that drops it)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

import lvdgs  # noqa: E402,F401
from lvdgs import _lib, backend_map, rasterizer, slam_utils, synthetic  # noqa: E402
from lvdgs.gaussian_model import GaussianModel  # noqa: E402
from lvdgs.gaussian_renderer import render  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_G = 1228.8   # 256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction
VALU_MEASURED_G = 853.0  # independent v_fma_f32 at 8 waves/SIMD, tools/valu_clock.hip (the chip clocks down to ~2.05 GHz)
# the values of configs/mono/KITTI/base_config.yaml the step reads (densification and opacity resets are pushed out of the
# benchmark's horizon so that the Gaussian count stays the workload's)
CONFIG = {"Training": {"monocular": True, "rgb_boundary_threshold": 0.01, "alpha": 0.98, "pose_window": 3, "window_size": 8,
                       "prune_mode": "slam", "prune_num": 1, "lr": {"cam_rot_delta": 0.003, "cam_trans_delta": 0.001}},
          "Dataset": {"depth_loss": True}}
OPT = dict(position_lr_init=0.0016, position_lr_final=0.00016, position_lr_delay_mult=0.01, position_lr_max_steps=30000,
           feature_lr=0.0025, opacity_lr=0.05, scaling_lr=0.001, rotation_lr=0.001, percent_dense=0.01,
           densify_grad_threshold=0.0002, lambda_dssim=0.2)


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
    """The workload's Gaussians as a GaussianModel on `dev` and one Camera (seeded pose, target image, edge mask, mono depth)."""
    from lvdgs.camera_utils import Camera
    from lvdgs.graphics_utils import focal2fov, getProjectionMatrix2
    from lvdgs.pose_utils import SE3_exp
    cfg = synthetic.CONFIGS[workload]
    N, W, H = cfg["N"], cfg["W"], cfg["H"]
    g = synthetic.make_workload_gaussians(workload, seed=0)
    fx, fy = cfg.get("fx", float(W)), cfg.get("fy", float(W))
    cx, cy = cfg.get("cx", W / 2.0), cfg.get("cy", H / 2.0)
    proj = getProjectionMatrix2(znear=0.01, zfar=100.0, fx=fx, fy=fy, cx=cx, cy=cy, W=W, H=H).transpose(0, 1).contiguous().to(dev)
    gen = torch.Generator().manual_seed(4242 + rank)
    image = torch.rand(3, H, W, generator=gen).to(dev)
    cam = Camera(rank + 1, image, None, (torch.rand(H, W, generator=gen) * 40 + 1).numpy(), torch.eye(4), proj, fx, fy, cx, cy,
                 focal2fov(fx, W), focal2fov(fy, H), H, W, device=dev)
    cam.grad_mask = (torch.rand(1, H, W, generator=gen) > 0.5).to(dev)
    if rank:
        pose = SE3_exp(torch.randn(6, generator=torch.Generator().manual_seed(1000 + rank)) * 0.05)
        cam.update_RT(pose[:3, :3], pose[:3, 3])
    model = GaussianModel.from_activated(g["means3D"], g["scales"], g["rotations"], g["opacities"], shs=g["shs"],
                                         sh_degree=0, device=dev)
    return model, cam, g, (N, W, H)


dynamic_object_mask = synthetic.dynamic_object_mask   # (a keyframe's static_mask: seeded "vehicle" rectangles marked dynamic)


def build_window(workload, world, dev, model, n_window=None, masked=False):
    """A BackEnd-shaped object (the attributes reference utils/slam_backend.py:21-72 sets) holding `world` keyframes
    of the workload's scene, every one from its own seeded pose with its own seeded target image, built identically
    on every rank.  The window is the newest `n_window` of them (default: all); the others are the older keyframes the
    iteration draws its two random views from.  `masked`: every keyframe carries a ``static_mask`` -- the reference's default
    (``dynamic_filtering.enabled``: utils/slam_frontend.py:1218,1429-1433), which sends the window's keyframes down the L1 + SSIM +
    masked-depth branch of the mapping loss (utils/slam_backend.py:196-261)."""
    from lvdgs.camera_utils import Camera
    from lvdgs.graphics_utils import focal2fov, getProjectionMatrix2
    from lvdgs.pose_utils import SE3_exp
    cfg = synthetic.CONFIGS[workload]
    W, H = cfg["W"], cfg["H"]
    fx, fy = cfg.get("fx", float(W)), cfg.get("fy", float(W))
    cx, cy = cfg.get("cx", W / 2.0), cfg.get("cy", H / 2.0)
    proj = getProjectionMatrix2(znear=0.01, zfar=100.0, fx=fx, fy=fy, cx=cx, cy=cy, W=W, H=H).transpose(0, 1).to(dev)
    viewpoints = {}
    pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False)
    for k in range(world):
        gen = torch.Generator().manual_seed(4242 + k)
        cam = Camera(k + 1, torch.zeros(3, H, W, device=dev), None, None, torch.eye(4), proj, fx, fy, cx, cy, focal2fov(fx, W), focal2fov(fy, H), H, W, device=dev)
        pose = SE3_exp(torch.randn(6, generator=torch.Generator().manual_seed(1000 + k)) * 0.05)
        cam.update_RT(pose[:3, :3], pose[:3, 3])
        # The keyframe's target image and mono depth are what the map itself shows from the keyframe's pose, plus fixed pixel
        # noise (sigma 0.05 / 2 %): the state of a converged SLAM map, whose iterations leave the map where it is.  (Targets
        # unrelated to the map -- random images -- make Adam tear the map apart within a hundred iterations: lists of thousands
        # of entries per tile, a different workload at every step.)
        with torch.no_grad():
            pkg = render(cam, model, pipe, torch.zeros(3, device=dev))
            cam.original_image = (pkg["render"] + 0.05 * torch.randn(3, H, W, generator=gen).to(dev)).clamp_(0.02, 1.0).contiguous()
            cam.mono_depth = (pkg["depth"][0] * (1.0 + 0.02 * torch.randn(H, W, generator=gen).to(dev))).clamp_min_(0.05).cpu().numpy()
        if masked:
            cam.static_mask = dynamic_object_mask(H, W, k).to(dev)
        viewpoints[k + 1] = cam
    n_window = world if n_window is None else n_window
    window = list(range(world, world - n_window, -1))  # newest first
    model.init_lr(6.0)
    model.training_setup(OPT)
    kf_groups = []
    for idx, kf in enumerate(window):
        vp = viewpoints[kf]
        if idx < CONFIG["Training"]["pose_window"]:
            kf_groups += [{"params": [vp.cam_rot_delta], "lr": 0.0015, "name": f"rot_{kf}"},
                          {"params": [vp.cam_trans_delta], "lr": 0.0005, "name": f"trans_{kf}"}]
        kf_groups += [{"params": [vp.exposure_a], "lr": 0.01, "name": f"exposure_a_{kf}"},
                      {"params": [vp.exposure_b], "lr": 0.01, "name": f"exposure_b_{kf}"}]
    far = 1 << 60
    be = SimpleNamespace(
        config=CONFIG, gaussians=model, pipeline_params=SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False),
        background=torch.zeros(3, device=dev), opt_params=SimpleNamespace(**OPT), monocular=True, iteration_count=0, last_sent=0,
        occ_aware_visibility={}, viewpoints=viewpoints, current_window=window, initialized=True,
        keyframe_optimizers=torch.optim.Adam(kf_groups), gaussian_update_every=far, gaussian_update_offset=far - 1,
        gaussian_th=0.7, gaussian_extent=6.0, gaussian_reset=far, size_threshold=20)
    return be, window


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default=os.environ.get("LVDGS_BENCH_WORKLOAD", "cfg3_500k_1920x1080"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side", action="store_true", help="skip config.side (the other workloads, bounded to ~10 s) and the steady-state block")
    ap.add_argument("--pose-only", action="store_true", help="tracking step: the backward the product's tracking loop runs (pose + exposure gradients only)")
    ap.add_argument("--window", choices=["real", "weak"], default="real",
                    help="mapping step: 'real' = the reference's window, 8 keyframes + 2 random older ones per iteration, its ten "
                         "views sharded over the GPUs (strong scaling); 'weak' = one keyframe per GPU, no random views")
    ap.add_argument("--no-masks", action="store_true",
                    help="mapping step: keyframes WITHOUT a static_mask (every window view scored by get_loss_mapping).  Default: every keyframe "
                         "carries one, as LVD-GS's front end attaches it (dynamic_filtering.enabled = True, utils/slam_frontend.py:1218,1429-1433): "
                         "the eight window views take the L1 + SSIM + masked-depth branch (utils/slam_backend.py:196-261), the two random views "
                         "get_loss_mapping (:275-300)")
    ap.add_argument("--step", choices=["auto", "tracking", "tracking-autograd", "mapping"], default="auto",
                    help="auto: tracking iteration on 1 GPU, mapping-window iteration on N > 1.  tracking: one iteration of "
                         "the product's tracking loop (fast_tracking.TrackingSession: render, tracking loss, backward, pose "
                         "optimiser step -- three C-ABI calls, no autograd); tracking-autograd: render() -> "
                         "get_loss_tracking -> backward through the public autograd API, without the optimiser step")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch N > 1 as `python -m torch.distributed.run --nnodes=1 "
                         f"--nproc-per-node {args.gpus} --master-addr 127.0.0.1 --master-port P bench.py --gpus {args.gpus} ...`")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU path for the product)")
    # LVDGS_BENCH_BACKEND=gloo lets the N > 1 code path be exercised on a box with fewer GPUs than ranks (several ranks
    # share a device, collectives go through the host); numbers from such a run are not benchmark results.
    backend_name = os.environ.get("LVDGS_BENCH_BACKEND", "nccl")
    local_dev = local_rank if backend_name == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    dist = None
    rccl_log = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend_name == "nccl":
            # RCCL's own version line, kept out of stdout (rank 0 prints ONE JSON line there) and quoted in the JSON: config.comm.rccl_version_line
            if "NCCL_DEBUG" not in os.environ:
                os.environ["NCCL_DEBUG"] = "VERSION"
                os.environ.setdefault("NCCL_DEBUG_FILE", "/tmp/lvdgs_rccl_%p.log")
            rccl_log = os.environ.get("NCCL_DEBUG_FILE", "").replace("%p", str(os.getpid())).replace("%h", os.uname().nodename) or None
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend_name, rank=rank, world_size=world)

    notes = []   # what did not go as planned in the parts around the timed region (printed in the line: config.notes)
    torch.manual_seed(0)   # (the single-GPU mapping window draws its two random views from the global generator)
    model, cam, g_cpu, (N, W, H) = build_scene(args.workload, rank, dev)
    pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False)
    bg = torch.zeros(3, device=dev)
    params = model.parameters()
    pose_params = [cam.cam_rot_delta, cam.cam_trans_delta, cam.exposure_a, cam.exposure_b]
    tracking = (world == 1) if args.step == "auto" else args.step.startswith("tracking")
    use_session = tracking and args.step != "tracking-autograd"
    stats = {}
    backend = window = session = None
    real_window = not tracking and args.window == "real"
    masked = not args.no_masks
    make_window = ((lambda m: build_window(args.workload, 12, dev, m, n_window=8, masked=masked)) if real_window else
                   (lambda m: build_window(args.workload, world, dev, m, masked=masked)))
    comm = None
    if world > 1:
        # Preflight, before anything is built or timed: one tiny collective of every (operation, dtype, communicator) the iteration
        # issues, each in a try of its own with the verdict agreed over the ranks, and a fallback for everything but the two the
        # iteration cannot do without (backend_map.collective_preflight).  The first multi-GPU run must not lose its line to a
        # uint8 reduction or a second communicator.
        aux_group = None
        if not tracking and os.environ.get("LVDGS_BENCH_AUX_GROUP", "1") != "0":
            # the two small MAX collectives on a communicator of their own: in flight under the gradient all-reduce
            # (LVDGS_BENCH_AUX_GROUP=0: all three collectives on the one communicator, as in round 3).  new_group is collective: it
            # fails on every rank or on none, so every rank takes the same branch
            try:
                if "aux" in os.environ.get("LVDGS_PREFLIGHT_FAIL", "").split(","):
                    raise RuntimeError("failed on purpose (LVDGS_PREFLIGHT_FAIL)")
                aux_group = dist.new_group(backend=backend_name if backend_name != "nccl" else None)
            except Exception as e:   # noqa: BLE001
                notes.append(f"no second communicator ({type(e).__name__}: {e}): MAX collectives on the first")
        want_sharded = os.environ.get("LVDGS_BENCH_SHARDED_ADAM", "0") == "1"
        comm = backend_map.collective_preflight(dev, None, aux_group, sharded_adam=want_sharded and not tracking)
        notes.extend(comm["notes"])
        if comm["fatal"]:
            if rank == 0:
                print(json.dumps({"metric": "render+backward iters/sec @500k Gaussians 1080p; 1/2/4/8-GPU scaling", "value": None, "n_gpus": world,
                                  "error": "collective preflight: " + comm["fatal"], "comm": comm}))
            dist.destroy_process_group()
            raise SystemExit(3)
        ranks_seen = [None] * world
        dist.all_gather_object(ranks_seen, (rank, local_dev, os.uname().nodename))
        comm["ranks_seen"] = ranks_seen
        try:
            comm["rccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version()) if backend_name == "nccl" else None
        except Exception:   # noqa: BLE001
            comm["rccl_version"] = None
        comm["rccl_version_line"] = None
        if rccl_log and os.path.exists(rccl_log):
            for ln in open(rccl_log, errors="replace"):
                if "version" in ln.lower():
                    comm["rccl_version_line"] = ln.strip()
                    break
        comm["backend"] = backend_name
    if not tracking:
        backend, window = make_window(model)
        if comm is not None and comm["use_aux_group"]:
            backend.shard_aux_group = aux_group
        # LVDGS_BENCH_SHARDED_ADAM=1: the Gaussian Adam as reduce-scatter -> every rank steps its share -> all-gather
        # (backend_map.ShardedAdam; off by default: modelled, never measured on more than one GPU)
        backend.shard_optimizer = bool(comm is not None and comm["use_sharded_adam"])
    elif use_session:
        from lvdgs.fast_tracking import TrackingSession
        # BASELINE configs[2] names the full pose + map backward: the headline computes every Gaussian gradient.  (The product's
        # tracking loop does not -- the reference's tracking optimiser holds pose and exposure alone, LVDGS_FLAG_POSE_ONLY -- and
        # its rate is printed beside the headline: config.pose_only_iters_per_s.)
        session = TrackingSession(cam, model, CONFIG, pipe, bg, gaussian_gradients=not args.pose_only)

    def autograd_step():
        for p in params + pose_params:
            p.grad = None
        pkg = render(cam, model, pipe, bg)
        slam_utils.get_loss_tracking(CONFIG, pkg["render"], pkg["depth"], pkg["opacity"], cam).backward()
        return pkg["radii"]

    def step():
        if session is not None:
            session.step()
            return session.radii
        if tracking:
            return autograd_step()
        backend_map.map_window(backend, window, iters=1)
        return None

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Everything built so far (torch, the scene) leaves the garbage collector's generations BEFORE the warm-up: a full
    # collection over that heap is one stall of about 40 ms, which otherwise lands somewhere in the first hundred iterations
    # of whichever loop allocates Python objects (tools/autograd_variance.py) -- and taken between the warm-up and the timed
    # region it let the GPU fall idle (and its clocks drop) right before the measurement.  The collector stays on.
    gc.collect()
    gc.freeze()
    rasterizer.KEEP_DEBUG_STATE = True
    for _ in range(max(args.warmup, 1)):
        radii = step()
    debug_pairs = int(rasterizer._DEBUG_LAST.get("num_rendered", 0))   # (a Python int the autograd API left: no GPU work)
    rasterizer.KEEP_DEBUG_STATE = False
    rasterizer._DEBUG_LAST.clear()
    # Nothing sits between the warm-up and the timed region but the barrier: a pause of a millisecond here (reading the
    # scene's statistics back used to be done at this point) lets the GPU drop its clocks, and the next ~17 ms -- 30
    # iterations -- then run 12 % slower (tools/short_region.py: per-step times after a synchronisation).
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        radii = step()
    sync()
    elapsed = time.perf_counter() - t0
    if session is not None:
        stats["D"] = int(session.num_rendered)
    elif backend is not None and getattr(backend, "_lvdgs_view_pass", None) is not None:
        stats["D"] = int(backend._lvdgs_view_pass.a.num_rendered)   # the mapping views bypass the autograd rasterizer
    else:
        stats["D"] = debug_pairs
    stats["V"] = int((radii > 0).sum().item()) if radii is not None else int((backend.gaussians.max_radii2D > 0).sum().item())
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- steady state, same process: >= 50 ms of load (the clocks are up, tools/short_region.py), then 200 steps untimed by
    # events, then -- without a pause -- the same steps with HIP events around every launch (per-kernel durations).  The
    # roofline below is priced from THIS block's kernel time; the headline `value` stays the short region above.
    roofline = None
    kernels = {}
    steady = None
    ss_steps = max(args.steps, 200) if not args.no_side else args.steps
    n_warm = max(args.warmup, 1)
    if not args.no_side:
        n_warm = max(n_warm, int(0.050 / max(elapsed / args.steps, 1e-6)) + 1)
    for _ in range(n_warm):   # (the read-backs above let the GPU idle: back to the clocks first)
        step()
    sync()
    t1 = time.perf_counter()
    for _ in range(ss_steps):
        step()
    sync()
    ss_elapsed = time.perf_counter() - t1
    if rank == 0:
        _lib.profile_reset()
        _lib.profile_enable(True)
    t1 = time.perf_counter()
    for _ in range(ss_steps):
        step()
    sync()
    prof_elapsed = time.perf_counter() - t1
    if world > 1:
        t = torch.tensor([ss_elapsed, prof_elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ss_elapsed, prof_elapsed = (float(x) for x in t.tolist())
    if rank == 0:
        _lib.profile_enable(False)
        times = _lib.profile_read()
        P, T = W * H, ((W + 15) // 16) * ((H + 15) // 16)
        for name, (n, ms) in sorted(times.items(), key=lambda kv: -kv[1][1]):
            kernels[name] = {"launches_per_step": n / ss_steps, "avg_us": 1e3 * ms / max(n, 1),
                             "us_per_step": 1e3 * ms / ss_steps}
        steady = {"warmup_steps": n_warm, "steps": ss_steps, "ms_per_step": round(1e3 * ss_elapsed / ss_steps, 4),
                  "value": round((1 if tracking else (10 if real_window else world)) * ss_steps / ss_elapsed, 3),
                  "ms_per_step_with_events": round(1e3 * prof_elapsed / ss_steps, 4),
                  "kernels_us_per_step": {k: round(v["us_per_step"], 2) for k, v in kernels.items()},
                  "kernels_sum_us": round(sum(v["us_per_step"] for v in kernels.values()), 2),
                  "note": "kernels_us_per_step: HIP events recorded around every launch on its stream, in the loop whose wall time is "
                          "ms_per_step_with_events (an event pair costs ~2 us of stream time per launch: the sum is below that figure and "
                          "may exceed the event-free ms_per_step)"}
        dom = max(times, key=lambda k: times[k][1])
        n, ms = times[dom]
        avg_s = ms / 1e3 / max(n, 1)
        nbytes = algorithmic_bytes(dom, N, stats["V"], stats["D"], P, T)
        traffic = valu_insts = source = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        # (the PMC passes were taken on the tracking iteration with the full backward; the mapping iteration's scene moves under
        # Adam, its pair count differs: no traffic / instruction figures are attached to it)
        if os.path.exists(tpath) and session is not None and not args.pose_only:
            tj = json.load(open(tpath))
            ent = tj.get(args.workload, {}).get(dom)
            if ent:
                traffic = ent.get("hbm_bytes_per_launch")
                valu_insts = ent.get("valu_wave_instructions_per_launch")
                source = tj.get(args.workload, {}).get("_source")
        if nbytes:
            achieved = nbytes / avg_s / 1e9
            roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                        "traffic_source": ("separate rocprofv3 --pmc passes of this workload, NOT measured in this run: profiles/traffic.json"
                                           + (f" <- {source}" if source else "")) if traffic else None,
                        "algorithmic_bytes_per_launch": int(nbytes), "avg_launch_us": round(avg_s * 1e6, 2),
                        "measured_in": "steady_state block of this run (HIP events around the launch on its stream)"}
            if valu_insts:
                # The resource that actually binds the blend kernels: vector-instruction issue.
                # peak: one wave64 instruction per 2 cycles per SIMD, 1024 SIMDs, 2.4 GHz (MI355X_MICROARCH.md);
                # ceiling: what a stream of independent v_fma_f32 reaches on this chip at the clock it then runs at
                # (tools/valu_clock.hip: 2.88 cycles at the nominal clock, 8 waves/SIMD).
                g_per_s = valu_insts / avg_s / 1e9
                roofline["valu"] = {"wave_instructions_per_launch": int(valu_insts), "achieved": round(g_per_s, 1), "unit": "G wave-instructions/s",
                                    "peak": VALU_PEAK_G, "frac": round(g_per_s / VALU_PEAK_G, 4), "measured_issue_ceiling": VALU_MEASURED_G,
                                    "frac_of_measured_ceiling": round(g_per_s / VALU_MEASURED_G, 4),
                                    "source": "SQ_INSTS_VALU of a separate rocprofv3 --pmc pass (profiles/traffic.json), not measured in this run"}

    autograd_rate = None
    if rank == 0 and session is not None:
        # the same iteration through the public autograd API (render() / get_loss_tracking / backward), for comparison
        for _ in range(10):
            autograd_step()
        gc.collect()
        gc.freeze()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            autograd_step()
        torch.cuda.synchronize()
        autograd_rate = round(args.steps / (time.perf_counter() - t1), 3)

    pose_only_rate = None
    side = None
    if rank == 0 and session is not None and world == 1 and not args.pose_only and not args.no_side:
        from lvdgs.fast_tracking import TrackingSession
        po = TrackingSession(cam, model, CONFIG, pipe, bg)   # the product's default: pose + exposure gradients only
        pose_only_rate = time_session(po, 60, 200)
        del po
        side = run_side(dev, pipe)

    # The two measurements below come AFTER the timed region and are extras: what goes wrong in them is recorded in config.notes and
    # does not cost the line its headline.  Every collective of this part is issued by every rank whatever happened to it: the
    # rank-local work (building a second model and window, the timed loops -- an out-of-memory or a HIP error there hits ONE rank)
    # sits in try blocks of its own, and a MIN all-reduce of an "ok" flag after each decides for all ranks alike how to go on --
    # a rank that skipped ahead on an exception of its own would leave the others waiting in their next collective for ever.
    same_step_single = None
    comm_us = phases = None

    def all_ok(ok):
        if world == 1:
            return bool(ok)
        t = torch.tensor([1.0 if ok else 0.0], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item() > 0.5)

    if world > 1 and not tracking:
        # The same step on ONE GPU, measured inside this job (every rank alone in a group of its own: no collective),
        # so that the N-GPU number can be read against the right single-GPU number: bench.py's N = 1 default is the
        # tracking iteration of BASELINE configs[2], a different (lighter) step than the mapping iteration timed here.
        solo_groups, ok = None, True
        try:
            solo_groups = [dist.new_group(ranks=[r]) for r in range(world)]   # (collective: raises on every rank or on none)
        except Exception as e:   # noqa: BLE001
            ok = False
            notes.append(f"no single-rank groups ({type(e).__name__}: {e}): same_step_on_one_gpu not measured")
        solo_s = 0.0
        if ok:
            try:
                solo_model = GaussianModel.from_activated(g_cpu["means3D"], g_cpu["scales"], g_cpu["rotations"], g_cpu["opacities"], shs=g_cpu["shs"],
                                                          sh_degree=0, device=dev)
                solo_backend, solo_window = make_window(solo_model) if real_window else build_window(args.workload, 1, dev, solo_model, masked=masked)
                for _ in range(3):
                    backend_map.map_window(solo_backend, solo_window, iters=1, group=solo_groups[rank])
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    backend_map.map_window(solo_backend, solo_window, iters=1, group=solo_groups[rank])
                torch.cuda.synchronize()
                solo_s = time.perf_counter() - t1
                del solo_backend, solo_model
            except Exception as e:   # noqa: BLE001
                ok = False
                notes.append(f"rank {rank}: the one-GPU run of the same step failed ({type(e).__name__}: {e})")
        if solo_groups is not None:   # (every rank is here: the two collectives below are unconditional)
            ok = all_ok(ok)
            solo = torch.tensor([solo_s], device=dev, dtype=torch.float64)
            dist.all_reduce(solo, op=dist.ReduceOp.MAX)
            if ok:
                same_step_single = round(args.steps / float(solo.item()), 3)
            elif rank == 0 and not notes:
                notes.append("the one-GPU run of the same step failed on another rank: not reported")

    other_masks_value = None
    if world > 1 and not tracking and real_window:
        # The same window with the keyframes' masks the OTHER way round (default run: without masks -- every view scored by
        # get_loss_mapping, the step of rounds 2-4 under this metric name; --no-masks run: with them), so that lines of different
        # rounds can be compared like for like.  Built on every rank alike; its collectives are map_window's own.
        ok = True
        try:
            other_model = GaussianModel.from_activated(g_cpu["means3D"], g_cpu["scales"], g_cpu["rotations"], g_cpu["opacities"], shs=g_cpu["shs"],
                                                       sh_degree=0, device=dev)
            torch.manual_seed(0)
            other_backend, other_window = build_window(args.workload, 12, dev, other_model, n_window=8, masked=not masked)
            other_backend.shard_aux_group = getattr(backend, "shard_aux_group", None)
            other_backend.shard_optimizer = backend.shard_optimizer
        except Exception as e:   # noqa: BLE001
            ok = False
            notes.append(f"rank {rank}: the window with the masks the other way round could not be built ({type(e).__name__}: {e})")
        if all_ok(ok):
            for _ in range(max(args.warmup, 1)):
                backend_map.map_window(other_backend, other_window, iters=1)
            sync()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                backend_map.map_window(other_backend, other_window, iters=1)
            sync()
            t = torch.tensor([time.perf_counter() - t1], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            other_masks_value = round(10 * args.steps / float(t.item()), 3)
            del other_backend, other_model

    if not tracking:
        # the iteration by phase (events on the stream at the phase boundaries, one wait at the end of each iteration).  map_window's
        # own collectives are the benchmark's: a rank-local failure inside them is fatal for the job as it was in the timed region.
        st = {}
        for _ in range(5):
            backend_map.map_window(backend, window, iters=1, stats=st)
        per = [r["phases"].seconds() for r in st["iterations"]]
        names = list(per[0])
        t = torch.tensor([sum(p[n] for p in per) / len(per) for n in names], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        phases = {n: round(float(v) * 1e6, 1) for n, v in zip(names, t.tolist())}
        comm_us = phases.get("collectives")

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_baseline = run_cpu_baseline(g_cpu, args.workload, N, W, H)

    if rank == 0:
        views_per_step = 1 if tracking else (10 if real_window else world)
        value = views_per_step * args.steps / elapsed
        out = {
            "metric": "render+backward iters/sec @500k Gaussians 1080p; 1/2/4/8-GPU scaling",
            "value": round(value, 3), "unit": "iters/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True,
            # N > 1 shards the reference's ten-view mapping window over the GPUs (total work fixed: strong), `--window weak` one
            # keyframe per GPU; the N = 1 line is the single-GPU tracking iteration BASELINE configs[2] names, and carries the one-GPU
            # rate of the N > 1 step as config.mapping_window_views_per_s (the same-step anchor of a scaling curve)
            "scaling": "weak" if (not tracking and not real_window) else "strong",
            "value_definition": ("tracking iterations per second (render + tracking loss + full pose/map backward + pose step), one GPU" if tracking else
                                 ("views (render + loss + backward) per second of whole mapping-window iterations: 10 x iterations/s, strong scaling "
                                  "(NOT comparable with the weak-scaling values of BENCH_r01/r02)" if real_window else
                                  "keyframes per second of mapping iterations with one keyframe per GPU (weak scaling)")),
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            # what an N-GPU `value` is to be read against: the SAME step (the ten-view mapping window) on ONE GPU -- measured inside this
            # job when N > 1, the side run of the same window when N = 1 -- never the N = 1 `value`, which is the tracking iteration
            "scaling_anchor_value": (None if same_step_single is None else round(same_step_single * views_per_step / (1 if real_window else world), 3)) if world > 1
                                    else (None if not side else side.get("mapping_window_" + args.workload + "_masked", {}).get("views_per_s")),
            "scaling_anchor_definition": "views per second of the ten-view mapping window (keyframes with static masks) on one GPU: the 1-GPU point of the N-GPU curve",
            "config": {"workload": args.workload, "gaussians": N, "width": W, "height": H, "visible": stats["V"],
                       "pairs": stats["D"], "sh_degree": 0,
                       "step": (("tracking iteration of slam_loops.track_frame on a TrackingSession: render + get_loss_tracking + backward "
                                 "(pose + all Gaussian grads) + pose optimiser step (Adam, SE(3) retraction, camera matrices) -- two C-ABI calls: lvdgs_forward_backward_fused_loss (= lvdgs_forward then lvdgs_backward_fused_loss; one blend launch for both passes on frames of up to 4096 tiles, not at this size), lvdgs_tracking_tail"
                                 if session is not None else
                                 "tracking iteration through the autograd API: render() + get_loss_tracking + backward (pose + all Gaussian grads)")
                                if tracking else
                                ("mapping iteration (backend_map.map_window) on the reference's window: 8 keyframes + 2 random older ones = 10 views "
                                 "per iteration, dealt to the GPUs whole and in bands of tile rows -- render + mapping loss (keyframes with a static mask: "
                                 "L1 + SSIM + masked depth, whole views; the others: get_loss_mapping) + backward of every piece, RCCL collectives (gradients + statistics SUM, radii MAX, flags MAX), bookkeeping, Adam over all Gaussians, "
                                 "keyframe Adam, pose retraction; value = views (render + backward) per second = 10 x iterations/s"
                                 if real_window else
                                 "mapping iteration (backend_map.map_window), WEAK scaling: a window of N keyframes, one whole keyframe per GPU, no "
                                 "random views -- render + get_loss_mapping + backward, the collectives, bookkeeping, Adam over all Gaussians, "
                                 "keyframe Adam, pose retraction; value = keyframes per second")),
                       "parallelism": ((f"10 views in pieces over {world} GPUs" if real_window else f"keyframe-per-gpu x{world}") if world > 1 else "single"),
                       "views_per_step": views_per_step, "window_keyframes_carry_static_mask": None if tracking else masked, "comm_us_per_step": comm_us, "phases_us_per_step": phases,
                       "autograd_api_iters_per_s": autograd_rate, "pose_only_iters_per_s": pose_only_rate,
                       "mapping_window_views_per_s": None if not side else side.get("mapping_window_" + args.workload, {}).get("views_per_s"),
                       "side": side, "comm": comm,
                       ("value_with_static_masks" if not masked else "value_without_static_masks"): other_masks_value,
                       "notes": notes or None, "same_step_on_one_gpu_iters_per_s": same_step_single,
                       "same_step_on_one_gpu_value": None if same_step_single is None else round(same_step_single * views_per_step / (1 if real_window else world), 3)},
            **({"collective_backend": backend_name + " (functional check only, not a benchmark result)"} if world > 1 and backend_name != "nccl" else {}),
            **({"timing_note": "warm-up + timed region last under ~40 ms: after start-up idle this GPU needs ~17 ms of load to reach its clocks, "
                               "and every kernel runs 5-12 % slower until then (tools/short_region.py; 20 steps after 60 of warm-up read the "
                               "same per-step time as 200 after 20)"}
               if (max(args.warmup, 1) + args.steps) * elapsed / args.steps < 0.040 else {}),
            "roofline": roofline, "cpu_baseline": cpu_baseline, "steady_state": steady,
            "kernels_us_per_step": {k: round(v["us_per_step"], 2) for k, v in kernels.items()},
        }
        # the loops as a system, once more at the END of the line in short (a record that keeps only the line's tail still shows it):
        # config.side.sequence_kitti07_geom has the full entry
        sq = (side or {}).get("sequence_kitti07_geom")
        if sq:
            out["sequence_kitti07_geom_summary"] = {k: sq.get(k) for k in (
                "frames", "keyframes", "tracking_iterations", "mapping_iterations", "gaussians_first", "gaussians_max", "gaussians_last",
                "tracking_plus_mapping_iterations_per_s", "frames_per_s", "ate_rmse", "trajectory_length", "psnr_static", "psnr")}
            for k in ("initialize_map_kitti07_geom", "color_refinement_kitti07_geom", "color_refinement_kitti07_geom_masked"):
                if k in side:
                    out["sequence_kitti07_geom_summary"][k + "_ms_per_iteration"] = side[k].get("ms_per_iteration")
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def time_session(sess, warm, steps):
    """Iterations per second of a TrackingSession after `warm` untimed iterations."""
    for _ in range(warm):
        sess.step()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(steps):
        sess.step()
    torch.cuda.synchronize()
    return round(steps / (time.perf_counter() - t), 3)


def run_side(dev, pipe):
    """The workloads next to the headline one, in the same process and bounded to about ten seconds: the tracking iteration (full
    backward, and the product's pose-only one) at KITTI-07 geometry, on the opaque-surface scene and at BASELINE's largest size,
    and the reference's 8 + 2 mapping window on ONE GPU at KITTI-07 and config-3 geometry (views per second: the N = 1 anchor of
    the N > 1 lines).  Builder-side detail of each: tools/side_benchmarks.sh."""
    from lvdgs.fast_tracking import TrackingSession
    out = {}
    bg = torch.zeros(3, device=dev)
    for w in ("kitti07_geom", "surface_100k_1920x1080", "cfg5_2m_1920x1280"):
        t0 = time.perf_counter()
        model, cam, _, (N, W, H) = build_scene(w, 0, dev)
        ent = {"gaussians": N, "width": W, "height": H}
        for key, full in (("tracking_iters_per_s", True), ("pose_only_iters_per_s", False)):
            sess = TrackingSession(cam, model, CONFIG, pipe, bg, gaussian_gradients=full)
            per = 1.0 / time_session(sess, 10, 20)
            ent[key] = time_session(sess, max(10, int(0.05 / per)), max(50, min(400, int(0.4 / per))))
            ent["pairs"] = int(sess.num_rendered)
            del sess
        ent["seconds_spent"] = round(time.perf_counter() - t0, 2)
        out[w] = ent
        del model, cam
        torch.cuda.empty_cache()
    # the 8 + 2 mapping window on one GPU.  `_masked`: every keyframe carries a static_mask -- the reference's default configuration
    # (utils/slam_frontend.py:1218,1429-1433) -- so the eight window views take the L1 + SSIM + masked-depth branch of the mapping loss
    # (utils/slam_backend.py:196-261); without: every view get_loss_mapping
    for w, masked in (("kitti07_geom", False), ("kitti07_geom", True), ("cfg3_500k_1920x1080", False), ("cfg3_500k_1920x1080", True),
                      ("cfg5_2m_1920x1280", False), ("cfg5_2m_1920x1280", True), ("surface_100k_1920x1080", True)):
        t0 = time.perf_counter()
        torch.manual_seed(0)
        model, _, _, (N, W, H) = build_scene(w, 0, dev)
        backend, window = build_window(w, 12, dev, model, n_window=8, masked=masked)
        for _ in range(8):
            backend_map.map_window(backend, window, iters=1)
        # (as in main(): what the set-up built leaves the collector's generations before the timed loop -- a full collection over a
        # heap with torch in it is one stall of 40-110 ms, and the mapping loop's per-iteration Python objects trigger one every few
        # dozen iterations: tools/_diag_side.py saw single iterations of 45 and 110 ms in a 1.8 ms loop)
        gc.collect()
        gc.freeze()
        torch.cuda.synchronize()
        iters = 40 if w == "kitti07_geom" else (25 if w == "cfg3_500k_1920x1080" else (16 if w.startswith("surface") else 12))
        # three timed blocks, the MEDIAN reported (all three printed): a bounded side run of a few dozen iterations is otherwise at the
        # mercy of one host-side stall -- a first-use code-object load, an allocator growth, a garbage collection: single iterations of
        # 20-80 ms were seen in this 1.7 ms loop (tools/side_stall_diag.py) and put 3.4 ms into a line whose kernels ran as always
        blocks = []
        for _ in range(3):
            t = time.perf_counter()
            for _ in range(iters):
                backend_map.map_window(backend, window, iters=1)
            torch.cuda.synchronize()
            blocks.append((time.perf_counter() - t) / iters)
        per = sorted(blocks)[1]
        out["mapping_window_" + w + ("_masked" if masked else "")] = {
            "views_per_s": round(10 / per, 2), "ms_per_iteration": round(1e3 * per, 3), "views_per_iteration": 10,
            "ms_per_iteration_of_the_three_blocks": [round(1e3 * b, 3) for b in blocks], "iterations_per_block": iters,
            "window_keyframes_carry_static_mask": masked, "gaussians": N, "width": W, "height": H, "seconds_spent": round(time.perf_counter() - t0, 2)}
        del backend, model
        torch.cuda.empty_cache()
    # The loops as a system (tools/sequence.py -> lvdgs.slam_sequence.SlamSequence): a 60-frame synthetic drive at KITTI-07's geometry from an
    # EMPTY map, the reference's cadence -- initialize_map (1050 iterations), per frame track_frame (<= 100) -> keyframe test -> seeding ->
    # map_window bursts on keyframes with static masks, densify / prune every 150, the back end's free-running iterations with a pruning pass
    # every ten -> ATE (Umeyama), PSNR before / after 500 colour-refinement iterations.  And the two single-view loops by themselves
    # (tools/single_view_loops.py).  A failure here costs the side entry, not the line.
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import sequence as seq_tool
        import single_view_loops as svl
        t0 = time.perf_counter()
        rec, _ = seq_tool.run_sequence(dev, frames=60, refine=500)
        keep = ("frames", "keyframes", "tracking_iterations", "mapping_iterations", "init_iterations", "prune_passes", "refinement_iterations",
                "gaussians_first", "gaussians_last", "gaussians_max", "size_changes_by_densification", "size_changes_by_pruning", "seconds",
                "frames_per_s", "tracking_plus_mapping_iterations_per_s", "tracking_iterations_per_s", "mapping_iterations_per_s", "ate_rmse",
                "pose_error_unaligned_mean", "pose_error_unaligned_max", "trajectory_length", "psnr_before_refinement", "psnr", "psnr_static_before_refinement",
                "psnr_static", "ssim", "refinement_ms_per_iteration", "batched_window_runs", "width", "height", "cadence", "idle_map_iters",
                "keyframes_carry_static_mask")
        out["sequence_kitti07_geom"] = {**{k: rec.get(k) for k in keep}, "seconds_spent": round(time.perf_counter() - t0, 2),
                                        "what": "synthetic drive from an empty map: initialize_map -> per frame track_frame / keyframe test / seeding / masked map_window "
                                                "bursts with densification and pruning at configs/mono/KITTI/base_config.yaml's cadence / free-running mapping -> ATE, PSNR, "
                                                "colour refinement (lvdgs.slam_sequence; reference utils/slam_frontend.py:1740-1899, utils/slam_backend.py:485-609)"}
        torch.cuda.empty_cache()
        t0 = time.perf_counter()
        out["initialize_map_kitti07_geom"] = {**svl.time_initialize_map(dev), "seconds_spent": round(time.perf_counter() - t0, 2)}
        for masked in (False, True):
            t0 = time.perf_counter()
            out["color_refinement_kitti07_geom" + ("_masked" if masked else "")] = {**svl.time_color_refinement(dev, masked, iters=300),
                                                                                     "seconds_spent": round(time.perf_counter() - t0, 2)}
        torch.cuda.empty_cache()
    except Exception as e:   # noqa: BLE001
        out["sequence_and_single_view_loops_error"] = f"{type(e).__name__}: {e}"
    return out


CPU_BASELINE_SECONDS = 12.0  # the scalar build is given 2 iterations (about 16 s at config 3), the OpenMP build as many as fit


def run_cpu_baseline(g, workload, N, W, H):
    """Forward + backward of the CPU oracle on the same scene and camera: the OpenMP build over all host cores (the
    baseline the JSON line carries), and the scalar build the parity tests use on one core beside it.  The backward is
    fed fixed image gradients (colour, depth, opacity), not the tracking loss's: the oracle restates the rasterizer, the
    loss is outside it."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import oracle as orc

    cfg = synthetic.CONFIGS[workload]
    cam = synthetic.make_camera(W, H, **{k: cfg[k] for k in ("fx", "fy", "cx", "cy") if k in cfg})
    gc, gd, go = synthetic.make_image_grads(W, H, 0)

    def time_it(prec, min_iters, budget_s):
        o = orc.Oracle(prec)
        n, t0 = 0, time.perf_counter()
        while n < min_iters or (time.perf_counter() - t0 < budget_s and n < 200):
            o.forward(means3D=g["means3D"].numpy(), opacities=g["opacities"].numpy(), W=W, H=H, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
                      viewmatrix=cam.world_view_transform.numpy(), projmatrix=cam.full_proj_transform.numpy(),
                      projmatrix_raw=cam.projection_matrix.numpy(), campos=cam.camera_center.numpy(), bg=np.zeros(3),
                      scales=g["scales"].numpy(), rotations=g["rotations"].numpy(), colors_precomp=g["colors"].numpy())
            o.backward(gc.numpy(), gd.numpy(), go.numpy())
            n += 1
        dt = time.perf_counter() - t0
        threads = o.threads
        o.free()
        return n, dt, threads

    n1, dt1, _ = time_it("f32", 2, 0.0)
    nt, dtt, threads = time_it("f32_omp", 3, CPU_BASELINE_SECONDS)
    return {"value": round(nt / dtt, 5), "unit": "iters/s", "cores": threads, "kind": "port",
            "sample": f"{nt} iterations (rasterizer forward + backward with fixed image gradients; the loss is not part of the "
                      f"oracle) of the same {N}-Gaussian {W}x{H} scene, C oracle with OpenMP on {threads} threads, {dtt:.1f} s",
            "one_core": {"value": round(n1 / dt1, 5), "unit": "iters/s", "sample": f"{n1} iterations of the scalar build (the parity tests' checker), {dt1:.1f} s"},
            "host_cores_available": os.cpu_count()}


if __name__ == "__main__":
    main()
