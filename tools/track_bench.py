#!/usr/bin/env python3
"""Tracking iterations per second through the two paths, on the benchmark workloads:
  autograd : render() -> get_loss_tracking -> backward   (the public API; bench.py --step tracking-autograd)
  session  : fast_tracking.TrackingSession.step()         (the same + Adam / retraction / camera matrices on the device)
  loop     : slam_loops.track_frame(fused=False): the PyTorch loop incl. torch.optim.Adam and update_pose (what the
             reference's FrontEnd.tracking does per iteration)
usage: python tools/track_bench.py [workload ...]"""
import os
import sys
import gc
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import lvdgs  # noqa: E402,F401
from lvdgs import slam_utils, synthetic  # noqa: E402
from lvdgs.camera_utils import Camera  # noqa: E402
from lvdgs.fast_tracking import TrackingSession  # noqa: E402
from lvdgs.gaussian_renderer import render  # noqa: E402
from lvdgs.graphics_utils import focal2fov, getProjectionMatrix2  # noqa: E402
from lvdgs.slam_loops import track_frame  # noqa: E402

dev = torch.device("cuda", 0)
CFG = dict(bench.CONFIG)


def camera_for(workload):
    cfg = synthetic.CONFIGS[workload]
    W, H = cfg["W"], cfg["H"]
    fx, fy, cx, cy = cfg.get("fx", float(W)), cfg.get("fy", float(W)), cfg.get("cx", W / 2.0), cfg.get("cy", H / 2.0)
    proj = getProjectionMatrix2(znear=0.01, zfar=100.0, fx=fx, fy=fy, cx=cx, cy=cy, W=W, H=H).transpose(0, 1).to(dev)
    gen = torch.Generator().manual_seed(4242)
    cam = Camera(1, torch.rand(3, H, W, generator=gen).to(dev), None, None, torch.eye(4), proj, fx, fy, cx, cy, focal2fov(fx, W),
                 focal2fov(fy, H), H, W, device=dev)
    cam.grad_mask = (torch.rand(1, H, W, generator=gen) > 0.5).to(dev)
    return cam


def timed(fn, n, warm=5):
    for _ in range(warm):
        fn()
    gc.collect(); gc.freeze()   # no full collection over torch's heap (one ~50 ms stall) inside the timed loop: tools/autograd_variance.py
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return n / (time.perf_counter() - t0)


for workload in (sys.argv[1:] or ["kitti07_geom", "cfg2_100k_640x480", "cfg3_500k_1920x1080"]):
    model, _, _, (N, W, H) = bench.build_scene(workload, 0, dev)
    pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False)
    bg = torch.zeros(3, device=dev)
    cam = camera_for(workload)
    params = model.parameters()
    pose = [cam.cam_rot_delta, cam.cam_trans_delta, cam.exposure_a, cam.exposure_b]

    def autograd_step():
        for p in params + pose:
            p.grad = None
        pkg = render(cam, model, pipe, bg)
        slam_utils.get_loss_tracking(CFG, pkg["render"], pkg["depth"], pkg["opacity"], cam).backward()

    r_auto = timed(autograd_step, 100)
    sess = TrackingSession(cam, model, CFG, pipe, bg)
    r_sess = timed(sess.step, 200)
    sess.finish()
    def frame(fused):   # one whole track_frame call (session set-up, 50 iterations, convergence polling, read-back); a warm one
        cams = [camera_for(workload), camera_for(workload)]   # (building a camera uploads its image: not part of the loop)
        track_frame(cams[0], model, CFG, pipe, bg, tracking_itr_num=50, fused=fused)
        gc.collect(); gc.freeze()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, _, n_it = track_frame(cams[1], model, CFG, pipe, bg, tracking_itr_num=50, fused=fused)
        torch.cuda.synchronize()
        return n_it / (time.perf_counter() - t0)

    r_loop, r_fused_loop = frame(False), frame(True)
    print(f"{workload:22s} N={N:7d} {W}x{H}: autograd step {r_auto:7.0f} it/s | session step {r_sess:7.0f} it/s | "
          f"PyTorch tracking loop (Adam + update_pose) {r_loop:6.0f} it/s | fused tracking loop {r_fused_loop:6.0f} it/s")
