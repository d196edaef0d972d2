#!/usr/bin/env python3
"""Soak of the mapping window UNDER the reference's densification cadence, on targets that are not the map's own renders: the synthetic
drive of tools/sequence.py is run to its last keyframe, then the back end keeps mapping its last window (8 keyframes with static masks +
2 random older ones per iteration) for `iters` iterations with gaussian_update_every 150 / offset 50 / gaussian_reset 2001
(configs/mono/KITTI/base_config.yaml:30-34) -- densify-and-prune every 150 iterations, an opacity reset of the non-visible at 2001, the
pruning pass of the free-running back end every 10 (utils/slam_backend.py:487-499).  Every iteration is bracketed by events on the stream:
ordinary iterations and the ones that change the map's size (re-plan of the batched window's per-view buffers, the cached memory verdict,
Adam state surgery) are reported separately; device memory and the map's size block by block.

    python tools/soak_densify.py [iters=3300] [frames=60]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gc  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402

import lvdgs  # noqa: E402,F401
import sequence as tool  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3300
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 60
dev = torch.device("cuda", 0)
rec, seq = tool.run_sequence(dev, frames=frames, refine=0)
be, window = seq.backend, list(seq.current_window)
print(f"sequence: {rec['frames']} frames, {rec['keyframes']} keyframes, window {window}, map {seq._n()} Gaussians, iteration count {be.iteration_count}, "
      f"ATE {rec['ate_rmse']:.5f}, PSNR (static) {rec['psnr_static_before_refinement']:.2f}")
print(f"cadence: densify every {be.gaussian_update_every} at offset {be.gaussian_update_offset}, opacity reset every {be.gaussian_reset}, pruning pass every 10 iterations")
gc.collect(); gc.freeze()
torch.cuda.synchronize()
mem0 = torch.cuda.memory_allocated(dev)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
kinds, sizes, phase_rec, host_ms = [], [], [], []
ev[0].record()
t0 = time.perf_counter()
block = max(iters // 6, 1)
for k in range(iters):
    n0 = seq._n()
    th = time.perf_counter()
    st = {}
    seq._map(window, stats=st)
    phase_rec.append(st["iterations"][-1]["phases"] if st.get("iterations") else None)
    kind = "ordinary"
    if be.last_sent >= 10:
        seq._map(window, prune=True, iters=10)
        seq._sync_backend()
        kind = "with pruning pass"
    if seq._n() != n0:
        kind = "map size changed (" + kind + ")" if kind != "ordinary" else "map size changed"
    elif k and sizes[-1] != (sizes[-2] if k > 1 else sizes[-1]) and kind == "ordinary":
        kind = "ordinary, first after a size change"
    if be.iteration_count % be.gaussian_reset == 0 or (be.iteration_count - 1) % be.gaussian_reset == 0:
        kind = "opacity reset"
    ev[k + 1].record()
    host_ms.append(1e3 * (time.perf_counter() - th))
    kinds.append(kind); sizes.append(seq._n())
    if (k + 1) % block == 0:
        torch.cuda.synchronize()
        print(f"  iterations ..{k + 1:5d}: wall {1e3 * (time.perf_counter() - t0) / (k + 1):.3f} ms each so far, map {seq._n():6d} Gaussians, "
              f"device memory {torch.cuda.memory_allocated(dev) / 2**20:.1f} MiB (reserved {torch.cuda.memory_reserved(dev) / 2**20:.1f}), "
              f"parameters finite: {all(bool(torch.isfinite(p).all()) for p in be.gaussians.parameters())}")
torch.cuda.synchronize()
ms = np.array([ev[k].elapsed_time(ev[k + 1]) for k in range(iters)])
print(f"map size over the soak: min {min(sizes)}, max {max(sizes)}, last {sizes[-1]}; size changed in {sum(1 for a, b in zip([sizes[0]] + sizes, sizes) if a != b)} iterations")
for kind in sorted(set(kinds)):
    sel = ms[[i for i, q in enumerate(kinds) if q == kind]]
    print(f"  {kind:45s} {len(sel):5d} iterations: median {np.median(sel):.3f} ms, mean {sel.mean():.3f}, p95 {np.percentile(sel, 95):.3f}, max {sel.max():.3f}")
hm = np.array(host_ms)
print(f"host time per iteration (enqueueing + its own waits, no synchronisation added): median {np.median(hm):.3f} ms, mean {hm.mean():.3f}, "
      f"over 20 ms: {(hm > 20).sum()} of {len(hm)} (sum {hm[hm > 20].sum():.0f} ms); event time: median {np.median(ms):.3f}, mean {ms.mean():.3f}, over 20 ms: {(ms > 20).sum()} (sum {ms[ms > 20].sum():.0f} ms)")
print("phases of the mapping iteration itself (events on the stream at the phase boundaries; medians / means in ms):")
for kind in sorted(set(kinds)):
    rows = [phase_rec[i].seconds() for i, q in enumerate(kinds) if q == kind and phase_rec[i] is not None]
    if rows:
        print(f"  {kind:45s} " + ", ".join(f"{name} {1e3 * np.median([r[name] for r in rows]):.3f} / {1e3 * np.mean([r[name] for r in rows]):.3f}" for name in rows[0]))
mem1 = torch.cuda.memory_allocated(dev)
print(f"device memory in use {mem0 / 2**20:.1f} MiB -> {mem1 / 2**20:.1f} MiB; batched window runs {getattr(getattr(be, '_lvdgs_window_batch', None), 'runs', 0)}")
ate = seq.eval_ate()
print(f"after the soak: ATE {ate:.5f}, PSNR {seq.eval_rendering()}")
assert all(bool(torch.isfinite(p).all()) for p in be.gaussians.parameters())
print("soak ok")
