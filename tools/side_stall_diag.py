"""Per-iteration wall time of the KITTI mapping windows run the way bench.py's config.side runs them (tracking sessions of three\nworkloads first): found single iterations of 45-110 ms -- full garbage collections -- in a 1.8 ms loop.  usage: python tools/side_stall_diag.py"""
import sys, time, torch
from types import SimpleNamespace
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from lvdgs import backend_map
from lvdgs.fast_tracking import TrackingSession
import gc
gc_log = []
_t0 = [0.0]
def _cb(phase, info):
    if phase == "start": _t0[0] = time.perf_counter()
    else: gc_log.append((info["generation"], 1e3 * (time.perf_counter() - _t0[0])))
gc.callbacks.append(_cb)
dev = torch.device("cuda", 0)
pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False)
bg = torch.zeros(3, device=dev)
for w in ("kitti07_geom", "surface_100k_1920x1080", "cfg5_2m_1920x1280"):
    model, cam, _, _ = bench.build_scene(w, 0, dev)
    for full in (True, False):
        sess = TrackingSession(cam, model, bench.CONFIG, pipe, bg, gaussian_gradients=full)
        bench.time_session(sess, 10, 100)
        del sess
    del model, cam
    torch.cuda.empty_cache()
for w, masked in (("kitti07_geom", False), ("kitti07_geom", True), ("kitti07_geom", True)):
    torch.manual_seed(0)
    model, _, _, (N, W, H) = bench.build_scene(w, 0, dev)
    backend, window = bench.build_window(w, 12, dev, model, n_window=8, masked=masked)
    ts = []
    del gc_log[:]
    for k in range(96):
        if k == 8 and os.environ.get("SIDE_DIAG_FREEZE", "1") != "0":   # (as bench.py's run_side does behind its eight warm-up iterations)
            gc.collect(); gc.freeze()
        torch.cuda.synchronize(); t = time.perf_counter()
        backend_map.map_window(backend, window, iters=1)
        torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t))
    b = backend._lvdgs_window_batch
    slow = [(k, round(x, 1)) for k, x in enumerate(ts) if x > 4.0]
    print(w, masked, "median %.2f ms; iterations over 4 ms:" % sorted(ts)[len(ts) // 2], slow, "| collections (generation, ms) over 1 ms:", [(g, round(ms, 1)) for g, ms in gc_log if ms > 1.0], "| caps", [p.cap for p in b.passes][:3], "D", [int(p.a.num_rendered) for p in b.passes][:10], flush=True)
    del backend, model
    torch.cuda.empty_cache()
