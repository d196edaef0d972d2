#!/usr/bin/env python3
"""CPU model (numpy, on the oracle's lists) of TWO-LEVEL BINNING for the opaque-surface scene: (Gaussian, super-tile) pairs -- a
super-tile = S x S tiles -- emitted, scattered and depth-sorted instead of (Gaussian, tile) pairs; every tile's workgroup stages its
super-tile's sorted list and keeps the entries that pass the exact tile-reach test it already has (common.hpp: reaches_rect).

What it prices, per S in {2, 4}:
  pairs_super / pairs_tile      what the scatter writes and the tile sort orders (today: pairs_tile);
  staged / listed               entries a tile's workgroup reads and tests (its super-tile's whole list) per entry it blends today;
  sort work                     sum n log2 n over the lists (per-tile lists today, per-super-tile lists then);
  what does NOT change          the projection kernel still evaluates the per-tile reach mask (tiles_touched -> slot_base: the backward's
                                per-(Gaussian, tile) gradient records keep their rect_rank slots), preprocess_bwd still reads one record per
                                (Gaussian, tile) pair.
usage: python tools/supertile_model.py [workload]   (default surface_100k_1920x1080; minutes on one core)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402

import lvdgs  # noqa: E402,F401
from lvdgs import synthetic  # noqa: E402
import oracle as orc  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "surface_100k_1920x1080"
cfg = synthetic.CONFIGS[wl]
N, W, H = cfg["N"], cfg["W"], cfg["H"]
g = synthetic.make_workload_gaussians(wl, seed=0)
cam = synthetic.make_camera(W, H)
t0 = time.time()
o = orc.Oracle("f32_omp" if os.cpu_count() > 2 else "f32")
f = o.forward(means3D=g["means3D"].numpy(), opacities=g["opacities"].numpy(), W=W, H=H, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
              viewmatrix=cam.world_view_transform.numpy(), projmatrix=cam.full_proj_transform.numpy(), projmatrix_raw=cam.projection_matrix.numpy(),
              campos=cam.camera_center.numpy(), bg=np.zeros(3), scales=g["scales"].numpy(), rotations=g["rotations"].numpy(), colors_precomp=g["colors"].numpy())
print(f"oracle forward {time.time() - t0:.1f} s", flush=True)
ids = f["ids_sorted"].astype(np.int64)
tiles = (f["keys_sorted"] >> np.uint64(32)).astype(np.int64)
gx, gy = (W + 15) // 16, (H + 15) // 16
m2 = f["means2D"].astype(np.float64); co = f["conic_opacity"].astype(np.float64)


def reaches(idx, x0, y0, x1, y1):
    m, c4 = m2[idx], co[idx]
    a, b, c, op = c4[:, 0], c4[:, 1], c4[:, 2], c4[:, 3]
    dx_lo, dx_hi, dy_lo, dy_hi = m[:, 0] - x1, m[:, 0] - x0, m[:, 1] - y1, m[:, 1] - y0
    inside = (dx_lo <= 0) & (dx_hi >= 0) & (dy_lo <= 0) & (dy_hi >= 0)

    def along_y(dx):
        dy = np.clip(-b * dx / c, dy_lo, dy_hi); return 0.5 * (a * dx * dx + c * dy * dy) + b * dx * dy

    def along_x(dy):
        dx = np.clip(-b * dy / a, dx_lo, dx_hi); return 0.5 * (a * dx * dx + c * dy * dy) + b * dx * dy
    qmin = np.minimum(np.minimum(along_y(dx_lo), along_y(dx_hi)), np.minimum(along_x(dy_lo), along_x(dy_hi)))
    return (op >= 1 / 255) & (inside | (qmin <= np.log(np.maximum(op, 1e-30) * 255) + 0.02))


tx, ty = tiles % gx, tiles // gx
keep = reaches(ids, tx * 16.0, ty * 16.0, tx * 16.0 + 15, ty * 16.0 + 15)
D_ref, D_tile = len(ids), int(keep.sum())
ids_t, tiles_t = ids[keep], tiles[keep]
per_tile = np.bincount(tiles_t, minlength=gx * gy)
sort_tile = float((per_tile * np.log2(np.maximum(per_tile, 2))).sum())
print(f"{wl}: N {N}, tiles {gx * gy}; (Gaussian, tile) pairs: reference list {D_ref}, listed today (tile-reach culled) {D_tile}, "
      f"per tile mean {per_tile.mean():.0f} max {per_tile.max()}; sort work sum n log2 n = {sort_tile / 1e6:.1f} M")
for S in (2, 4):
    sx, sy = (gx + S - 1) // S, (gy + S - 1) // S
    # (Gaussian, super-tile) pairs: unique over the reference list, then the reach test on the super-tile's pixel rectangle
    st = (ty // S) * sx + (tx // S)
    key = np.unique(ids * (sx * sy) + st)
    gi, si = key // (sx * sy), key % (sx * sy)
    px, py = (si % sx) * 16.0 * S, (si // sx) * 16.0 * S
    k2 = reaches(gi, px, py, np.minimum(px + 16 * S - 1, W - 1), np.minimum(py + 16 * S - 1, H - 1))
    D_super = int(k2.sum())
    per_super = np.bincount(si[k2], minlength=sx * sy)
    sort_super = float((per_super * np.log2(np.maximum(per_super, 2))).sum())
    # staged entries: every tile reads its super-tile's whole list
    tile_super = (np.arange(gx * gy) // gx // S) * sx + (np.arange(gx * gy) % gx) // S
    staged = per_super[tile_super]
    print(f"  S = {S} ({16 * S} x {16 * S} pixels, {sx * sy} super-tiles): pairs {D_super} = {D_super / D_tile:.3f} x today's ({D_tile / D_super:.2f} x fewer); "
          f"list per super-tile mean {per_super.mean():.0f} max {per_super.max()}; sort work {sort_super / 1e6:.1f} M = {sort_super / sort_tile:.3f} x; "
          f"staged-and-tested entries per tile mean {staged.mean():.0f} = {staged.sum() / D_tile:.2f} x the entries it blends (max {staged.max()} vs {per_tile.max()})")
