"""Dense, differentiable float64 PyTorch statement of the splatting maths (test infrastructure).

Every pixel looks at every Gaussian (no tiles lists, no loops), the 3DGS constants are applied
as masks, and gradients come from autograd.  It is independent of both the C oracle's analytic
backward and the HIP kernels, and is only usable at toy sizes (P*N floats).  Used to pin the C
oracle's forward and analytic backward, including dL/dtau via ``SE3_exp(tau) @ T_w2c``.
"""
import math

import torch

from lvdgs.pose_utils import SE3_exp

TILE = 16
SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
         -0.4570457994644658, 1.445305721320277, -0.5900435899266435)


def sh_to_rgb(deg, shs, means, campos):
    d = means - campos[None]
    d = d / d.norm(dim=1, keepdim=True)
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    res = SH_C0 * shs[:, 0]
    if deg > 0:
        res = res - SH_C1 * y * shs[:, 1] + SH_C1 * z * shs[:, 2] - SH_C1 * x * shs[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        res = (res + SH_C2[0] * xy * shs[:, 4] + SH_C2[1] * yz * shs[:, 5]
               + SH_C2[2] * (2 * zz - xx - yy) * shs[:, 6] + SH_C2[3] * xz * shs[:, 7]
               + SH_C2[4] * (xx - yy) * shs[:, 8])
    if deg > 2:
        res = (res + SH_C3[0] * y * (3 * xx - yy) * shs[:, 9] + SH_C3[1] * xy * z * shs[:, 10]
               + SH_C3[2] * y * (4 * zz - xx - yy) * shs[:, 11]
               + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * shs[:, 12]
               + SH_C3[4] * x * (4 * zz - xx - yy) * shs[:, 13] + SH_C3[5] * z * (xx - yy) * shs[:, 14]
               + SH_C3[6] * x * (xx - 3 * yy) * shs[:, 15])
    return torch.clamp_min(res + 0.5, 0.0)


def quat_to_rot(q):
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    return torch.stack([
        torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)], -1),
        torch.stack([2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)], -1),
        torch.stack([2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1),
    ], 1)


def render_dense(means3D, opacities, H, W, tanfovx, tanfovy, bg, viewmatrix, projmatrix, campos,
                 scales=None, rotations=None, cov3D_precomp=None, colors_precomp=None, shs=None,
                 sh_degree=0, scale_modifier=1.0, ndc_offset=None):
    """Returns dict(color (3,H,W), depth (1,H,W), opacity (1,H,W), radii, n_touched, n_contrib,
    means2D_pix (N,2) retain-grad leaf-like tensor for the viewspace gradient)."""
    N = means3D.shape[0]
    dt = means3D.dtype
    ones = torch.ones(N, 1, dtype=dt)
    ph = torch.cat([means3D, ones], 1)
    p_view = ph @ viewmatrix
    p_hom = ph @ projmatrix
    p_w = 1.0 / (p_hom[:, 3] + 1e-7)
    p_proj = p_hom[:, :3] * p_w[:, None]
    in_front = p_view[:, 2] > 0.2

    if cov3D_precomp is None:
        R = quat_to_rot(rotations)
        S = torch.diag_embed(scales * scale_modifier)
        M = R @ S
        Sigma = M @ M.transpose(1, 2)
    else:
        c = cov3D_precomp
        Sigma = torch.stack([torch.stack([c[:, 0], c[:, 1], c[:, 2]], -1),
                             torch.stack([c[:, 1], c[:, 3], c[:, 4]], -1),
                             torch.stack([c[:, 2], c[:, 4], c[:, 5]], -1)], 1)

    fx = W / (2.0 * tanfovx)
    fy = H / (2.0 * tanfovy)
    tz = p_view[:, 2]
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    txtz, tytz = p_view[:, 0] / tz, p_view[:, 1] / tz
    # clamped coordinates carry no gradient (the clamp, and its dependence on tz, are dropped)
    tx = torch.where((txtz < -limx) | (txtz > limx), (txtz.clamp(-limx, limx) * tz).detach(), p_view[:, 0])
    ty = torch.where((tytz < -limy) | (tytz > limy), (tytz.clamp(-limy, limy) * tz).detach(), p_view[:, 1])
    zero = torch.zeros_like(tz)
    J = torch.stack([torch.stack([fx / tz, zero, -fx * tx / (tz * tz)], -1),
                     torch.stack([zero, fy / tz, -fy * ty / (tz * tz)], -1)], 1)  # (N,2,3)
    Wrot = viewmatrix[:3, :3].t()  # world->camera rotation (column-vector form)
    T = J @ Wrot[None]
    cov2 = T @ Sigma @ T.transpose(1, 2)
    a = cov2[:, 0, 0] + 0.3
    b = cov2[:, 0, 1]
    c = cov2[:, 1, 1] + 0.3
    det = a * c - b * b
    ok = in_front & (det != 0)
    det_safe = torch.where(det != 0, det, torch.ones_like(det))
    ca, cb, cc = c / det_safe, -b / det_safe, a / det_safe
    mid = 0.5 * (a + c)
    lam = mid + torch.sqrt(torch.clamp_min(mid * mid - det, 0.1))
    radius = torch.ceil(3.0 * torch.sqrt(lam.detach())).to(torch.int64)
    px = ((p_proj[:, 0] + 1.0) * W - 1.0) * 0.5
    py = ((p_proj[:, 1] + 1.0) * H - 1.0) * 0.5
    if ndc_offset is not None:
        # render()'s "viewspace points": an (N,3) zero leaf added to the NDC position, whose .grad the densification
        # statistics read (d pixel = W/2 d ndc_x, H/2 d ndc_y)
        px = px + 0.5 * W * ndc_offset[:, 0].to(dt)
        py = py + 0.5 * H * ndc_offset[:, 1].to(dt)
    pix = torch.stack([px, py], -1)
    if pix.requires_grad:
        pix.retain_grad()
    px, py = pix[:, 0], pix[:, 1]

    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    pxd, pyd, rf = px.detach(), py.detach(), radius.to(dt)
    trunc = lambda v: torch.trunc(v).to(torch.int64)
    rx0 = trunc((pxd - rf) / TILE).clamp(0, gx)
    ry0 = trunc((pyd - rf) / TILE).clamp(0, gy)
    rx1 = trunc((pxd + rf + TILE - 1) / TILE).clamp(0, gx)
    ry1 = trunc((pyd + rf + TILE - 1) / TILE).clamp(0, gy)
    ok = ok & ((rx1 - rx0) * (ry1 - ry0) > 0)
    radii = torch.where(ok, radius, torch.zeros_like(radius))

    if colors_precomp is not None:
        rgb = colors_precomp
    else:
        rgb = sh_to_rgb(sh_degree, shs, means3D, campos)

    # depth order, ties by index (stable)
    depth = tz
    key = torch.where(ok, depth.detach(), torch.full_like(depth, float("inf")))
    order = torch.sort(key, stable=True).indices
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    pxl = xs.reshape(-1).to(dt)
    pyl = ys.reshape(-1).to(dt)
    tile_x = (xs.reshape(-1) // TILE)
    tile_y = (ys.reshape(-1) // TILE)

    o = order
    in_rect = (ok[o][None] & (tile_x[:, None] >= rx0[o][None]) & (tile_x[:, None] < rx1[o][None])
               & (tile_y[:, None] >= ry0[o][None]) & (tile_y[:, None] < ry1[o][None]))  # (P,N)
    dx = px[o][None] - pxl[:, None]
    dy = py[o][None] - pyl[:, None]
    power = -0.5 * (ca[o][None] * dx * dx + cc[o][None] * dy * dy) - cb[o][None] * dx * dy
    raw = opacities.reshape(-1)[o][None] * torch.exp(power)
    # value min(0.99, raw); gradient of raw (the published backward ignores the clamp)
    alpha = raw - torch.clamp_min(raw - 0.99, 0.0).detach()
    valid = in_rect & (power <= 0) & (alpha >= 1.0 / 255.0)
    a_eff = torch.where(valid, alpha, torch.zeros_like(alpha))
    one_minus = 1.0 - a_eff
    T_incl = torch.cumprod(one_minus, dim=1)
    T_before = torch.cat([torch.ones_like(T_incl[:, :1]), T_incl[:, :-1]], 1)
    stop = valid & (T_incl.detach() < 1e-4)
    done = torch.cummax(stop.to(torch.int8), dim=1).values.bool()
    live = valid & ~done
    w = torch.where(live, a_eff * T_before, torch.zeros_like(a_eff))
    T_final = torch.where(live, one_minus, torch.ones_like(one_minus)).prod(dim=1)
    color = w @ rgb[o] + T_final[:, None] * bg[None]
    dep = w @ depth[o]
    opac = 1.0 - T_final
    n_touched = torch.zeros(N, dtype=torch.int64)
    n_touched[o] = (live & (T_incl.detach() > 0.5)).sum(0)
    pos = torch.cumsum(in_rect.to(torch.int64), dim=1)  # 1-based position in the tile's list
    n_contrib = torch.where(live, pos, torch.zeros_like(pos)).max(dim=1).values
    return dict(color=color.t().reshape(3, H, W), depth=dep.reshape(1, H, W), opacity=opac.reshape(1, H, W),
                radii=radii, n_touched=n_touched, n_contrib=n_contrib.reshape(H, W), means2D_pix=pix,
                tiles_touched=torch.where(ok, (rx1 - rx0) * (ry1 - ry0), torch.zeros_like(rx0)),
                order=order, ok=ok)


def camera_matrices(R, T, tau, proj_raw_rowvec):
    """viewmatrix / projmatrix / campos (row-vector layout) for w2c = SE3_exp(tau) @ [R T]."""
    w2c = torch.eye(4, dtype=R.dtype)
    w2c[:3, :3] = R
    w2c[:3, 3] = T
    w2c = SE3_exp(tau) @ w2c
    view = w2c.t()
    proj = view @ proj_raw_rowvec
    campos = torch.linalg.inv(view)[3, :3]
    return view, proj, campos


def fov_from_focal(f, pixels):
    return 2 * math.atan(pixels / (2 * f))
