"""Test helper: run the HIP rasterizer (through GaussianRasterizer -> C ABI) on CPU tensors and
return numpy results, including intermediates read out of the state buffers."""
import ctypes as C
import os

import numpy as np
import torch

from lvdgs import _lib, rasterizer
from lvdgs.rasterizer import GaussianRasterizationSettings, GaussianRasterizer


def settings_from_cam(cam, W, H, bg, sh_degree=0, scale_modifier=1.0, dev="cuda"):
    return GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=bg.to(dev),
        scale_modifier=scale_modifier, viewmatrix=cam.world_view_transform.to(dev),
        projmatrix=cam.full_proj_transform.to(dev), projmatrix_raw=cam.projection_matrix.to(dev),
        sh_degree=sh_degree, campos=cam.camera_center.to(dev), prefiltered=False, debug=bool(os.environ.get("LVDGS_TEST_DEBUG")))


def _view(buf, off, count, dtype):
    nbytes = count * np.dtype(dtype).itemsize
    return buf[off:off + nbytes].cpu().numpy().view(dtype).copy()


def run_hip(g, cam, W, H, bg, use_sh=False, sh_degree=0, cov_precomp=None, grads=None, pose=True, dev="cuda",
            propagate_opacity=True, tile_cull=True):
    """g: dict of float32 CPU tensors (synthetic.make_gaussians).  grads: optional (dL_dcolor, dL_ddepth,
    dL_dopacity) CPU tensors.  Returns (forward dict, backward dict or None).

    tile_cull=False (rasterizer.LIST_ALL_TILES for the call: LVDGS_FLAG_LIST_ALL_TILES in lvdgs_args.flags) lists every tile of
    a Gaussian's rectangle, which makes the pair list, the ranges and n_contrib the reference's bit for bit; the default
    drops the tiles the Gaussian cannot reach."""
    before = rasterizer.LIST_ALL_TILES
    rasterizer.LIST_ALL_TILES = not tile_cull
    try:
        return _run_hip(g, cam, W, H, bg, use_sh, sh_degree, cov_precomp, grads, pose, dev, propagate_opacity, tile_cull)
    finally:
        rasterizer.LIST_ALL_TILES = before


def _run_hip(g, cam, W, H, bg, use_sh, sh_degree, cov_precomp, grads, pose, dev, propagate_opacity, tile_cull):
    rasterizer.KEEP_DEBUG_STATE = True
    # the parity tests feed a gradient of the opacity image too: switch that (non-default) path on for the run
    propagate_before = rasterizer.PROPAGATE_OPACITY_GRAD
    rasterizer.PROPAGATE_OPACITY_GRAD = bool(propagate_opacity)
    rs = settings_from_cam(cam, W, H, bg, sh_degree=sh_degree, dev=dev)
    leaf = lambda t: t.to(dev).clone().requires_grad_(True)
    means3D, opac = leaf(g["means3D"]), leaf(g["opacities"])
    N = means3D.shape[0]
    means2D = torch.zeros(N, 3, device=dev, requires_grad=True)
    kw = {}
    if cov_precomp is not None:
        cov = leaf(cov_precomp); kw["cov3D_precomp"] = cov
    else:
        sc, rot = leaf(g["scales"]), leaf(g["rotations"]); kw.update(scales=sc, rotations=rot)
    if use_sh:
        shs = leaf(g["shs"]); kw["shs"] = shs
    else:
        col = leaf(g["colors"]); kw["colors_precomp"] = col
    theta = torch.zeros(3, device=dev, requires_grad=pose)
    rho = torch.zeros(3, device=dev, requires_grad=pose)
    color, radii, depth, opacity, n_touched = GaussianRasterizer(rs)(
        means3D=means3D, means2D=means2D, opacities=opac, theta=theta, rho=rho, **kw)
    torch.cuda.synchronize()
    st = dict(rasterizer._DEBUG_LAST)
    D = st["num_rendered"]
    lay = _lib.StateLayout()
    _lib.check(_lib.lib().lvdgs_state_layout_query(N, st["binning_pairs"], W, H, C.byref(lay)), "layout")
    NT = ((W + 15) // 16) * ((H + 15) // 16)
    RF = int(lay.geom_rec_floats)
    rec = _view(st["geom"], lay.geom_rec, N * RF, np.float32).reshape(N, RF) if N else np.zeros((0, RF), np.float32)
    fwd = dict(
        color=color.detach().cpu().numpy(), depth=depth.detach().cpu().numpy(), opacity=opacity.detach().cpu().numpy(),
        radii=radii.cpu().numpy(), n_touched=n_touched.cpu().numpy(), num_rendered=D, rec=rec, overflowed=st["overflowed"],
        tiles_touched=_view(st["geom"], lay.geom_tiles_touched, N, np.uint32) if N else np.zeros(0, np.uint32),
        slot_base=_view(st["geom"], lay.geom_slot_base, N, np.uint32) if N else np.zeros(0, np.uint32),
        point_list=_view(st["binning"], lay.bin_point_list, D, np.uint32) if D else np.zeros(0, np.uint32),
        ranges=_view(st["image"], lay.img_ranges, NT * 2, np.uint32).reshape(NT, 2),
        final_T=_view(st["image"], lay.img_final_T, W * H, np.float32).reshape(H, W),
        n_contrib=_view(st["image"], lay.img_n_contrib, W * H, np.uint32).reshape(H, W), tile_cull=bool(tile_cull),
    )
    # tile id of every entry of point_list, from the ranges (the counting path never materialises tile keys)
    r = fwd["ranges"].astype(np.int64)
    fwd["tile_keys"] = np.repeat(np.arange(NT, dtype=np.uint32), np.maximum(r[:, 1] - r[:, 0], 0))
    assert fwd["overflowed"] or len(fwd["tile_keys"]) == D, (len(fwd["tile_keys"]), D)
    bwd = None
    if grads is not None:
        gc, gd, go = (None if t is None else t.to(dev) for t in grads)
        loss = (color * gc).sum()
        if gd is not None:
            loss = loss + (depth * gd).sum()
        if go is not None:
            loss = loss + (opacity * go).sum()
        loss.backward()
        torch.cuda.synchronize()
        z = lambda t: None if t.grad is None else t.grad.detach().cpu().numpy()
        bwd = dict(means3D=z(means3D), means2D=z(means2D), opacities=z(opac).reshape(-1))
        if cov_precomp is not None:
            bwd["cov3D"] = z(cov)
        else:
            bwd["scales"], bwd["rotations"] = z(sc), z(rot)
        if use_sh:
            bwd["shs"] = z(shs)
        else:
            bwd["colors"] = z(col)
        if pose:
            bwd["tau"] = np.concatenate([z(rho), z(theta)])
    rasterizer.KEEP_DEBUG_STATE = False
    rasterizer.PROPAGATE_OPACITY_GRAD = propagate_before
    return fwd, bwd


def run_oracle(orc, g, cam, W, H, bg, use_sh=False, sh_degree=0, cov_precomp=None, grads=None, prec="f32"):
    o = orc.Oracle(prec)
    kw = dict(scales=g["scales"].numpy(), rotations=g["rotations"].numpy())
    if cov_precomp is not None:
        kw = dict(cov3D_precomp=cov_precomp.numpy())
    fwd = o.forward(means3D=g["means3D"].numpy(), opacities=g["opacities"].numpy(), W=W, H=H, tanfovx=cam.tanfovx,
                    tanfovy=cam.tanfovy, viewmatrix=cam.world_view_transform.numpy(),
                    projmatrix=cam.full_proj_transform.numpy(), projmatrix_raw=cam.projection_matrix.numpy(),
                    campos=cam.camera_center.numpy(), bg=bg.numpy(), shs=g["shs"].numpy() if use_sh else None,
                    colors_precomp=None if use_sh else g["colors"].numpy(), sh_degree=sh_degree, **kw)
    bwd = None
    if grads is not None:
        gc, gd, go = grads
        bwd = o.backward(gc.numpy(), None if gd is None else gd.numpy(), None if go is None else go.numpy())
    o.free()
    return fwd, bwd


def check_pair_lists(f_hip, f_ora, W, H, max_dropped_checked=200_000):
    """The per-tile lists of the HIP path against the oracle's (= the reference's: every tile of the 3-sigma rectangle).

    Without tile culling they are identical.  With it the HIP list must be the oracle's list with some pairs REMOVED
    (same order otherwise), and every removed (Gaussian, tile) pair must be one that cannot contribute: alpha < 1/255
    on all 256 pixels of the tile, evaluated here in float64 from the oracle's projected means / conics / opacities.
    Returns the fraction of pairs removed."""
    N = f_ora["radii"].shape[0]
    D = f_ora["num_rendered"]
    ora_ids = f_ora["ids_sorted"].astype(np.int64)
    ora_tiles = (f_ora["keys_sorted"] >> np.uint64(32)).astype(np.int64)
    if not f_hip["tile_cull"]:
        np.testing.assert_array_equal(f_hip["tiles_touched"], f_ora["tiles_touched"])
        assert f_hip["num_rendered"] == D
        np.testing.assert_array_equal(f_hip["point_list"], f_ora["ids_sorted"])
        np.testing.assert_array_equal(f_hip["tile_keys"], ora_tiles.astype(np.uint32))
        np.testing.assert_array_equal(f_hip["ranges"], f_ora["ranges"])
        f_hip["kept_in_oracle_list"] = np.ones(D, bool)
        return 0.0
    assert (f_hip["tiles_touched"] <= f_ora["tiles_touched"]).all()
    assert f_hip["num_rendered"] == int(f_hip["tiles_touched"].sum()) <= D
    hip_key = f_hip["tile_keys"].astype(np.int64) * max(N, 1) + f_hip["point_list"].astype(np.int64)
    ora_key = ora_tiles * max(N, 1) + ora_ids
    kept = np.isin(ora_key, hip_key)
    np.testing.assert_array_equal(ora_key[kept], hip_key)     # a sub-list, order (tile, depth, id) preserved
    f_hip["kept_in_oracle_list"] = kept
    r = f_hip["ranges"].astype(np.int64)
    assert (np.diff(r[r[:, 1] > r[:, 0]].reshape(-1)) >= 0).all()
    np.testing.assert_array_equal(np.bincount(f_hip["point_list"], minlength=N).astype(np.uint32), f_hip["tiles_touched"])
    dropped = np.nonzero(~kept)[0]
    if len(dropped) > max_dropped_checked:
        dropped = dropped[np.random.default_rng(0).choice(len(dropped), max_dropped_checked, replace=False)]
    if len(dropped):
        gx = (W + 15) // 16
        gid, tile = ora_ids[dropped], ora_tiles[dropped]
        m = f_ora["means2D"].astype(np.float64)[gid]
        co = f_ora["conic_opacity"].astype(np.float64)[gid]
        worst = np.zeros(len(dropped))
        for lo in range(0, len(dropped), 50_000):
            sl = slice(lo, lo + 50_000)
            px = (tile[sl] % gx * 16)[:, None] + (np.arange(256) % 16)[None, :]
            py = (tile[sl] // gx * 16)[:, None] + (np.arange(256) // 16)[None, :]
            dx, dy = m[sl, 0:1] - px, m[sl, 1:2] - py
            power = -0.5 * (co[sl, 0:1] * dx * dx + co[sl, 2:3] * dy * dy) - co[sl, 1:2] * dx * dy
            alpha = np.where(power > 0, 0.0, co[sl, 3:4] * np.exp(np.minimum(power, 0)))
            worst[sl] = alpha.max(1)
        assert worst.max() < 1.0 / 255.0, f"a dropped pair reaches alpha {worst.max():.6f} >= 1/255"
    return 1.0 - len(hip_key) / max(D, 1)


def expected_n_contrib(f_hip, f_ora, field="n_contrib"):
    """The oracle's per-pixel count of list entries up to the last contributor, restated for the HIP path's (possibly
    shorter) lists: the number of KEPT entries among the tile's first n_contrib ones.  check_pair_lists() first.
    ``field``: "n_contrib", or its bounds "n_contrib_lo" / "n_contrib_hi" (the last contributor with every near-threshold
    comparison of the pixel gone the other way: what a correctly rounding implementation may report on a fragile pixel)."""
    kept = f_hip["kept_in_oracle_list"]
    if kept.all():
        return f_ora[field]
    H, W = f_ora["n_contrib"].shape
    gx = (W + 15) // 16
    prefix = np.concatenate([[0], np.cumsum(kept)]).astype(np.int64)
    ys, xs = np.mgrid[0:H, 0:W]
    start = f_ora["ranges"].astype(np.int64)[(ys // 16) * gx + xs // 16, 0]
    return (prefix[start + f_ora[field].astype(np.int64)] - prefix[start]).astype(np.uint32)
