import sys, time, cProfile, pstats, io
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from types import SimpleNamespace
import bench, lvdgs
from lvdgs.fast_tracking import TrackingSession, track_frame_fused
dev = torch.device('cuda', 0)
for workload in ('kitti07_geom', 'cfg3_500k_1920x1080'):
    model, cam, g, (N, W, H) = bench.build_scene(workload, 0, dev)
    pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False)
    bg = torch.zeros(3, device=dev)
    CFG = dict(bench.CONFIG)
    for rep in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        s = TrackingSession(cam, model, CFG, pipe, bg)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for _ in range(50): s.step(); s.converged_lagging(2)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        s.finish(); pkg = s.render_package()
        torch.cuda.synchronize(); t3 = time.perf_counter()
        print(workload, rep, f"init {1e3*(t1-t0):.2f} ms  50 steps+poll {1e3*(t2-t1):.2f} ms ({50/(t2-t1):.0f} it/s)  finish {1e3*(t3-t2):.2f} ms")
    pr = cProfile.Profile(); pr.enable()
    s = TrackingSession(cam, model, CFG, pipe, bg); torch.cuda.synchronize()
    pr.disable(); st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats('cumulative').print_stats(12); print(st.getvalue()[:2500])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): s.step()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print("50 steps without polling", f"{50/(t1-t0):.0f} it/s")
