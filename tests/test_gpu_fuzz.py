"""Randomised parity sweep: scene sizes, image shapes (not multiples of the tile), footprint and opacity ranges and
camera poses drawn from a seeded generator; every case is checked like the fixed parity cases (integers exact, floats
1e-4).  Sizes are kept small enough for the CPU oracle to finish each case in about a second."""
import os

import numpy as np
import pytest
import torch

import test_gpu_parity as tp

pytestmark = pytest.mark.gpu


def _case(rng):
    W = int(rng.integers(17, 420))
    H = int(rng.integers(17, 300))
    N = int(rng.choice([1, 2, 7, 63, 64, 65, 500, 3000, 12000]))
    r_min = float(rng.choice([0.3, 1.0, 4.0, 20.0]))
    r_max = r_min * float(rng.choice([1.5, 4.0, 12.0]))
    z_min = float(rng.choice([0.25, 1.0, 5.0]))
    c = dict(N=N, W=W, H=H, r_min=r_min, r_max=r_max, z_min=z_min, z_max=z_min * float(rng.choice([1.2, 10.0, 60.0])),
             pose=None if rng.random() < 0.3 else int(rng.integers(0, 50)), opacity_scale=float(rng.choice([1.0, 0.3, 0.05])),
             seed=int(rng.integers(0, 10_000)))
    # one case in five: opaque surfaces of large flat Gaussians (synthetic.make_surface_gaussians) -- rectangles of hundreds
    # of tiles (block culling, wave-walked grouping, wave-summed gradients), lists beyond one wave's sort, saturating pixels
    if rng.random() < 0.2:
        c.update(kind="surface", N=int(rng.choice([40, 700, 2500, 6000])), r_min=float(rng.choice([2.0, 4.0, 8.0])), r_max=float(rng.choice([16.0, 64.0])))
    return c


_FIRST = int(os.environ.get("LVDGS_FUZZ_FIRST", "0"))   # (a sweep over other scenes than the first LVDGS_FUZZ_CASES)


@pytest.mark.parametrize("case_seed", list(range(_FIRST, _FIRST + int(os.environ.get("LVDGS_FUZZ_CASES", "80")))))  # more: set the variable
def test_random_scene_matches_oracle(case_seed):
    _check_case(_case(np.random.default_rng(1000 + case_seed)))


@pytest.mark.parametrize("opacity_scale", [0.02, 1.0])
def test_wide_faint_gaussians_keep_every_quadrant_busy(opacity_scale):
    """Footprints of 25-70 px on a 112 x 90 image: every tile's list is long and nearly every entry reaches every 8 x 8
    quadrant (all 64 entries of a backward round survive the quadrant test, every batch of the splat pass is full) --
    faint ones (nothing saturates, the lists are walked to the end) and opaque ones (early termination everywhere)."""
    _check_case(dict(N=700, W=112, H=90, r_min=25.0, r_max=70.0, z_min=1.0, z_max=10.0, pose=3, opacity_scale=opacity_scale, seed=77))


def _check_case(c):
    orc, hr, syn = tp._mods()
    if c.get("kind") == "surface":
        g = syn.make_surface_gaussians(c["N"], c["W"], c["H"], seed=c["seed"], r_min=c["r_min"], r_max=c["r_max"])
    else:
        g = syn.make_gaussians(c["N"], c["W"], c["H"], seed=c["seed"], r_min=c["r_min"], r_max=c["r_max"], z_min=c["z_min"], z_max=c["z_max"])
    with torch.no_grad():
        g["opacities"].mul_(c["opacity_scale"])
    cam = syn.make_camera(c["W"], c["H"], pose_seed=c["pose"])
    bg = torch.tensor([0.3, 0.1, 0.6])
    grads = syn.make_image_grads(c["W"], c["H"], c["seed"])
    f_hip, b_hip = hr.run_hip(g, cam, c["W"], c["H"], bg, grads=grads)
    f_ora, b_ora = hr.run_oracle(orc, g, cam, c["W"], c["H"], bg, grads=grads)
    # (same checks as the fixed cases, without the 98 % solid-pixel expectation: tiny images can be mostly fragile)
    np.testing.assert_array_equal(f_hip["radii"], f_ora["radii"], err_msg=str(c))
    try:   # the per-tile lists: the oracle's, less pairs that contribute to no pixel (hip_runner.check_pair_lists)
        hr.check_pair_lists(f_hip, f_ora, c["W"], c["H"])
    except AssertionError as e:
        raise AssertionError(f"{c}: {e}") from e
    solid = f_ora["fragile"] == 0
    for k in ("color", "depth", "opacity"):
        m = np.broadcast_to(solid, f_ora[k].shape)
        tp._close(np.where(m, f_hip[k], 0), np.where(m, f_ora[k], 0), what=f"{k} {c}")
    np.testing.assert_array_equal(f_hip["n_contrib"][solid], hr.expected_n_contrib(f_hip, f_ora)[solid], err_msg=str(c))
    names = ["means3D", "means2D", "opacities", "scales", "rotations", "colors"]
    clean = _gaussians_without_fragile_pixels(c, f_ora, solid)
    same = clean & (f_ora["radii"] > 0)   # integer bookkeeping: exact wherever no fragile pixel can reach
    np.testing.assert_array_equal(f_hip["n_touched"][same], f_ora["n_touched"][same], err_msg=str(c))
    try:
        _check_backward_of_case(c, b_hip, b_ora, solid, clean, names)
    except AssertionError as strict:
        _second_look(c, g, cam, bg, grads, b_hip, b_ora, solid, names, strict)


def _second_look(c, g, cam, bg, grads, b_hip, b_ora, solid, names, strict):
    """About one random scene in a thousand fails the strict bounds for a reason that is not a defect; two are known, each
    with a check of its own that still fails on anything else (15 000-scene sweep: cases 413, 524, 2172, 2845, 3324, 6294,
    8171 of `tools/fuzz_case.py`):

    * fragile pixels under tiny or faint Gaussians -- a contribution that is in on one side and out on the other can be
      most of a Gaussian's gradient.  The image gradients are zeroed on the fragile pixels and BOTH sides run again: what
      the remaining pixels contribute must agree strictly, for every Gaussian and the pose;
    * a gradient float32 cannot hold to the bounds -- the opacity gradient of a faint, screen-filling Gaussian is a sum of
      tens of thousands of sign-alternating terms; stacks of opaque Gaussians recover T by division.  The float64 oracle
      says how far the float32 ORACLE is from the truth and how ill-conditioned the sums are (their change when every
      pixel's image gradient moves by a relative 1e-6 with a random sign): the kernels pass if they are no farther from the float64 result than the
      float32 restatement (x 1.5) plus the strict tolerance plus sixteen float32 roundings times the condition number."""
    orc, hr, syn = tp._mods()
    if not solid.all():
        masked = tuple(torch.where(torch.from_numpy(np.broadcast_to(solid, t.shape[1:]).copy())[None], t, torch.zeros_like(t)) for t in grads)
        _, bh = hr.run_hip(g, cam, c["W"], c["H"], bg, grads=masked)
        _, bo = hr.run_oracle(orc, g, cam, c["W"], c["H"], bg, grads=masked)
        try:
            tp._check_backward(bh, bo, names + ["tau"])
            return
        except AssertionError as again:
            strict = AssertionError(f"{strict} | with the fragile pixels' gradients zeroed: {again}")
            b_hip, b_ora, grads = bh, bo, masked   # ... and perhaps ill-conditioned on top: the float64 look, on the masked problem
    _, b64 = hr.run_oracle(orc, g, cam, c["W"], c["H"], bg, grads=grads, prec="f64")
    # condition number of every gradient as a SUM over pixels: its change when every pixel's image gradient moves by a
    # relative 1e-6 with a random sign (what independent roundings of the summands do)
    gen = torch.Generator().manual_seed(12345)
    jitter = tuple(t.double() * (1.0 + 1e-6 * (torch.randint(0, 2, t.shape, generator=gen).double() * 2 - 1)) for t in grads)
    _, b64p = hr.run_oracle(orc, g, cam, c["W"], c["H"], bg, grads=jitter, prec="f64")
    for n in names + ["tau"]:
        ref64 = np.asarray(b64[n], np.float64).reshape(b_hip[n].shape)
        ref32 = np.asarray(b_ora[n], np.float64).reshape(b_hip[n].shape)
        pert = np.asarray(b64p[n], np.float64).reshape(b_hip[n].shape)
        hip = np.asarray(b_hip[n], np.float64)
        norm = max(np.linalg.norm(ref64), 1e-300)
        e_hip, e_ora = np.linalg.norm(hip - ref64) / norm, np.linalg.norm(ref32 - ref64) / norm
        cond = np.linalg.norm(pert - ref64) / norm / 1e-6
        assert e_hip <= 2e-5 + 1.5 * e_ora + 16 * 6e-8 * cond, (
            f"{n}: kernels {e_hip:.3e} from float64, float32 oracle {e_ora:.3e}, condition number {cond:.3g} | strict check said: {strict}")
        m_hip, m_ora = np.abs(hip - ref64).max(), np.abs(ref32 - ref64).max()
        scale = max(np.abs(ref64).max(), 1e-300)
        assert m_hip <= 1e-5 * scale + 1.5 * m_ora + 16 * 6e-8 * np.abs(pert - ref64).max() / 1e-6, (
            f"{n}: max error {m_hip:.3e} vs {m_ora:.3e} (scale {scale:.3e}) | {strict}")


def _gaussians_without_fragile_pixels(c, f_ora, solid):
    """Some pixel may sit within 1e-5 of a threshold: one faint Gaussian can be in on one side and out on the other, which
    perturbs that pixel's contribution to every Gaussian composited in it.  All of those overlap the pixel's tile, so
    every Gaussian whose tile rectangle holds no fragile pixel must still agree to the full tolerance."""
    if solid.all():
        return np.ones(f_ora["radii"].shape[0], bool)
    gx, gy = (c["W"] + 15) // 16, (c["H"] + 15) // 16
    pad = np.zeros((gy * 16, gx * 16), bool)
    pad[:c["H"], :c["W"]] = ~solid
    frag_tile = pad.reshape(gy, 16, gx, 16).any(axis=(1, 3))
    cum = np.zeros((gy + 1, gx + 1), np.int64)
    cum[1:, 1:] = frag_tile.cumsum(0).cumsum(1)
    r = f_ora["rect"].astype(np.int64)
    x0, y0, x1, y1 = r[:, 0], r[:, 1], r[:, 2], r[:, 3]
    n_frag = cum[y1, x1] - cum[y0, x1] - cum[y1, x0] + cum[y0, x0]
    return n_frag == 0


def _check_backward_of_case(c, b_hip, b_ora, solid, clean, names):
    if solid.all():
        tp._check_backward(b_hip, b_ora, names + ["tau"])
        return
    # the Gaussians a fragile pixel can reach (and the pose gradient, a sum over all of them): to the size of the
    # perturbation, alpha ~ 4e-3
    for n in names:
        ref = b_ora[n].reshape(b_hip[n].shape)
        if clean.any():
            # the other Gaussians take the reference's values: the error of the clean ones is then measured against the
            # norm and scale of the WHOLE gradient, not against that of a subset that may consist of a few nearly hidden
            # Gaussians (case 413 of a 4000-case sweep: 5 fragile pixels under wide opaque Gaussians left such a subset,
            # 4.9e-5 of ITS norm while the whole tensor agreed to 4e-7)
            sel = clean.reshape((-1,) + (1,) * (ref.ndim - 1))
            tp._close(np.where(sel, b_hip[n], ref), ref, what=f"grad {n} (Gaussians away from fragile pixels) {c}")
        scale = max(np.abs(ref).max(), 1e-30)
        assert np.abs(b_hip[n] - ref).max() <= 2e-2 * scale, (n, c)
    assert np.abs(b_hip["tau"] - b_ora["tau"]).max() <= 2e-2 * max(np.abs(b_ora["tau"]).max(), 1e-30), c


@pytest.mark.parametrize("case_seed", list(range(16)))
def test_random_ssim_l1_cases(case_seed):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
    import loss_oracle as lo
    from lvdgs.loss_utils import l1_dssim_loss
    rng = np.random.default_rng(2000 + case_seed)
    C = int(rng.choice([1, 3]))
    H, W = int(rng.integers(1, 150)), int(rng.integers(1, 200))
    lam = float(rng.choice([0.0, 0.2, 1.0]))
    g = torch.Generator().manual_seed(case_seed)
    a = torch.rand(C, H, W, generator=g)
    b = (a + float(rng.choice([0.02, 0.3])) * torch.randn(C, H, W, generator=g)).clamp(0, 1)
    mask = (torch.rand(H, W, generator=g) > float(rng.choice([0.1, 0.6]))) if rng.random() < 0.6 else None
    bg = torch.rand(C, generator=g)
    ad = a.double().requires_grad_(True)
    want = lo.l1_dssim_loss(ad, b, lam, mask, bg)
    want.backward()
    ag = a.cuda().requires_grad_(True)
    got = l1_dssim_loss(ag, b.cuda(), lam, None if mask is None else mask.cuda(), bg.cuda())
    got.backward()
    assert abs(float(got.detach()) - float(want.detach())) < 3e-6, (C, H, W, lam)
    ref = ad.grad.numpy()
    err = np.abs(ag.grad.cpu().numpy() - ref).max() / max(np.abs(ref).max(), 1e-30)
    assert err < 1e-4, (C, H, W, lam, err)


@pytest.mark.parametrize("case_seed", list(range(12)))
def test_random_point_clouds_knn(case_seed):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
    import aux_oracle
    from lvdgs.simple_knn import distCUDA2
    rng = np.random.default_rng(3000 + case_seed)
    n = int(rng.choice([1, 2, 3, 5, 63, 64, 65, 129, 1000, 4097, 9000]))
    g = torch.Generator().manual_seed(case_seed)
    kind = rng.choice(["ball", "sheet", "line", "clusters", "grid"])
    pts = torch.randn(n, 3, generator=g)
    if kind == "sheet":
        pts[:, 1] *= 1e-3
    elif kind == "line":
        pts[:, 1:] *= 1e-4
    elif kind == "clusters":
        pts = pts * 0.01 + torch.randint(0, 5, (n, 1), generator=g).float() * 10.0
    elif kind == "grid":
        pts = torch.round(pts * 2.0)  # many exact duplicates and ties
    out = distCUDA2(pts.cuda()).cpu().numpy()
    np.testing.assert_allclose(out, aux_oracle.dist2_knn3(pts.numpy()), rtol=3e-5, atol=1e-9, err_msg=f"{n} {kind}")
