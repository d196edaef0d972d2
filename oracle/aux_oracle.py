"""CPU restatements (numpy) of the two auxiliary native ops.  TEST INFRASTRUCTURE ONLY.

* dist2_knn3: exact brute-force 3-nearest-neighbour mean squared distance (what simple_knn's
  distCUDA2 computes; reference README.md:42, source absent -> parity unpinned).
* rope2d: croco's RoPE2D in its published PyTorch form (rotate_half formulation) applied to
  (B, N, H, D) tokens (reference README.md:49-50, source absent -> parity unpinned).
"""
import numpy as np


def dist2_knn3(points):
    p = np.asarray(points, np.float64)
    n = p.shape[0]
    out = np.zeros(n)
    k = min(3, n - 1)
    if k <= 0:
        return out
    for s in range(0, n, 1024):
        d = ((p[s:s + 1024, None, :] - p[None, :, :]) ** 2).sum(-1)
        d[np.arange(d.shape[0]), np.arange(s, s + d.shape[0])] = np.inf
        out[s:s + d.shape[0]] = np.sort(d, axis=1)[:, :k].mean(1)
    return out


def rope2d(tokens, positions, base=100.0, fwd=1.0):
    """tokens (B,N,H,D), positions (B,N,2) -> rotated copy (float64 maths)."""
    t = np.asarray(tokens, np.float64)
    B, N, H, D = t.shape
    Dh = D // 2
    inv_freq = fwd / (base ** (np.arange(0, Dh, 2) / Dh))
    out = t.copy()
    for half in range(2):
        ang = np.asarray(positions)[..., half].astype(np.float64)[..., None] * inv_freq  # (B,N,Dh/2)
        ang = np.concatenate([ang, ang], -1)[:, :, None, :]                              # (B,N,1,Dh)
        x = t[..., half * Dh:(half + 1) * Dh]
        rot = np.concatenate([-x[..., Dh // 2:], x[..., :Dh // 2]], -1)
        out[..., half * Dh:(half + 1) * Dh] = x * np.cos(ang) + rot * np.sin(ang)
    return out
