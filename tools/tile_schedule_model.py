"""How much of a blend kernel's time is the shape of its grid?  One workgroup per tile, workgroup time ~ a + b * (length of
the tile's list); blockIdx b goes to XCD b % 8, and an XCD's 32 CUs take its workgroups in order, `slots` at a time per
CU.  Greedy list scheduling of the real tile lists of a workload (read back from the rasterizer's state on the GPU)
gives the makespan of that order against the perfectly balanced time, and the same for longest-first orders.

    python tools/tile_schedule_model.py [workload] [slots_per_cu]
"""
import ctypes as C
import heapq
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from lvdgs import _lib, rasterizer  # noqa: E402
from lvdgs.gaussian_renderer import render  # noqa: E402


def tile_of_workgroup(b, n):
    xcd, k, q, r = b & 7, b >> 3, n >> 3, n & 7
    return (xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q) + k


def makespan(durations, slots):
    free = [0.0] * slots
    heapq.heapify(free)
    end = 0.0
    for d in durations:
        t = heapq.heappop(free) + d
        end = max(end, t)
        heapq.heappush(free, t)
    return end


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "cfg3_500k_1920x1080"
    slots_per_cu = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    dev = torch.device("cuda:0")
    model, cam, g, (N, W, H) = bench.build_scene(workload, 0, dev)
    rasterizer.KEEP_DEBUG_STATE = True
    bg = torch.zeros(3, device=dev)
    pipe = type("P", (), dict(convert_SHs_python=False, compute_cov3D_python=False))()
    render(cam, model, pipe, bg)
    torch.cuda.synchronize()
    st = dict(rasterizer._DEBUG_LAST)
    lay = _lib.StateLayout()
    _lib.check(_lib.lib().lvdgs_state_layout_query(N, st["binning_pairs"], W, H, C.byref(lay)), "layout")
    NT = ((W + 15) // 16) * ((H + 15) // 16)
    raw = st["image"][lay.img_ranges:lay.img_ranges + NT * 8].cpu().numpy().view(np.uint32).reshape(NT, 2).astype(np.int64)
    length = np.maximum(raw[:, 1] - raw[:, 0], 0)
    print(f"{workload}: {NT} tiles, list length mean {length.mean():.1f}  max {length.max()}  p99 {np.percentile(length, 99):.0f}")
    for fixed in (0.0, 64.0):  # per-workgroup cost that does not depend on the list, in entries
        dur = fixed + length.astype(np.float64)
        slots = 32 * slots_per_cu
        ideal = dur.sum() / (8 * slots)
        per_xcd = [[dur[tile_of_workgroup(b, NT)] for b in range(x, NT, 8)] for x in range(8)]
        as_is = max(makespan(d, slots) for d in per_xcd)
        longest_first = max(makespan(sorted(d, reverse=True), slots) for d in per_xcd)
        # every XCD's share of the total differs too (contiguous runs of tiles)
        share = max(sum(d) for d in per_xcd) / (dur.sum() / 8)
        print(f"  fixed cost {fixed:4.0f} entries: launch order {as_is / ideal:.3f} x balanced | longest first inside an XCD "
              f"{longest_first / ideal:.3f} x | heaviest XCD holds {share:.3f} x its share")
        # runs of `chunk` row-major tiles dealt round-robin to the XCDs instead of one contiguous run per XCD
        for chunk in (8, 16, 32, 64, 128):
            order = [[] for _ in range(8)]
            for c in range(0, NT, chunk):
                order[(c // chunk) % 8].extend(dur[c:c + chunk])
            ms = max(makespan(d, slots) for d in order)
            sh = max(sum(d) for d in order) / (dur.sum() / 8)
            print(f"      runs of {chunk:3d} tiles round-robin: {ms / ideal:.3f} x balanced, heaviest XCD {sh:.3f} x its share")
            # the same runs, heaviest run first (neighbouring tiles stay together; the grid's last workgroups are the lightest)
            runs = sorted((dur[c:c + chunk] for c in range(0, NT, chunk)), key=lambda r: -r.sum())
            order = [[] for _ in range(8)]
            for k, r in enumerate(runs):
                order[k % 8].extend(r)
            ms = max(makespan(d, slots) for d in order)
            print(f"      runs of {chunk:3d} tiles, heaviest run first:  {ms / ideal:.3f} x balanced")


if __name__ == "__main__":
    main()
