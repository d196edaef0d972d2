#!/usr/bin/env python3
"""Which call is the host in when a mapping iteration stalls?  The drive of tools/sequence.py to its last keyframe, then mapping
iterations one by one under cProfile; an iteration whose HOST time exceeds 20 ms has its profile printed (top entries by own time).
usage: python tools/stall_probe.py [iterations=600]"""
import cProfile
import gc
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

import lvdgs  # noqa: E402,F401
import sequence as tool  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 600
dev = torch.device("cuda", 0)
rec, seq = tool.run_sequence(dev, frames=40, refine=0)
be, window = seq.backend, list(seq.current_window)
gc.collect(); gc.freeze()
shown = 0
host = []
for k in range(iters):
    pr = cProfile.Profile()
    t = time.perf_counter()
    pr.enable()
    seq._map(window)
    if be.last_sent >= 10:
        seq._map(window, prune=True, iters=10)
        seq._sync_backend()
    pr.disable()
    dt = time.perf_counter() - t
    host.append(dt)
    if dt > 0.020 and shown < 8:
        shown += 1
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(6)
        print(f"--- iteration {k}: host {1e3 * dt:.1f} ms, N = {seq._n()}, gc counts {gc.get_count()}, allocator: {torch.cuda.memory_stats(dev).get('num_alloc_retries', 0)} retries, "
              f"{torch.cuda.memory_stats(dev).get('num_device_alloc', 0)} device allocs, {torch.cuda.memory_stats(dev).get('num_device_free', 0)} device frees")
        print("\n".join(s.getvalue().splitlines()[4:16]))
torch.cuda.synchronize()
import numpy as np
h = np.array(host) * 1e3
print(f"host ms per iteration: median {np.median(h):.3f}, mean {h.mean():.3f}, > 20 ms: {(h > 20).sum()} of {len(h)}, their sum {h[h > 20].sum():.0f} ms")
