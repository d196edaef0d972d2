import sys, time, torch, os, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from lvdgs import backend_map
dev = torch.device("cuda", 0)
out = bench.run_side.__wrapped__ if hasattr(bench.run_side, "__wrapped__") else None
# the tracking part of run_side first (as bench does), then the two KITTI windows with per-iteration host times
from types import SimpleNamespace
from lvdgs.fast_tracking import TrackingSession
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
for w, masked in (("kitti07_geom", False), ("kitti07_geom", True)):
    torch.manual_seed(0)
    model, _, _, (N, W, H) = bench.build_scene(w, 0, dev)
    backend, window = bench.build_window(w, 12, dev, model, n_window=8, masked=masked)
    for _ in range(8):
        backend_map.map_window(backend, window, iters=1)
    gc.collect(); gc.freeze(); torch.cuda.synchronize()
    for b in range(3):
        ts = []
        t0 = time.perf_counter()
        for k in range(40):
            t = time.perf_counter()
            backend_map.map_window(backend, window, iters=1)
            ts.append(1e3 * (time.perf_counter() - t))
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(w, masked, "block", b, "ms/it %.3f" % (1e3 * (t2 - t0) / 40), "host per it median %.2f" % sorted(ts)[20], "host its > 3 ms:", [(k, round(x, 1)) for k, x in enumerate(ts) if x > 3.0], "final sync wait %.1f ms" % (1e3 * (t2 - t1)), flush=True)
    del backend, model
    torch.cuda.empty_cache()
