"""CPU checks of the auxiliary-op restatements (oracle/aux_oracle.py) against independent formulations."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import aux_oracle  # noqa: E402


def test_knn_oracle_against_kdtree():
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(0)
    p = rng.normal(size=(3000, 3))
    d, _ = cKDTree(p).query(p, k=4)
    np.testing.assert_allclose(aux_oracle.dist2_knn3(p), (d[:, 1:] ** 2).mean(1), rtol=1e-10)
    assert aux_oracle.dist2_knn3(np.zeros((1, 3)))[0] == 0.0


def test_rope_oracle_against_rotate_half_formulation():
    """croco's published RoPE2D: tokens*cos + rotate_half(tokens)*sin on each half, (B,H,N,D) layout."""
    torch.manual_seed(0)
    B, H, N, D = 2, 3, 11, 16
    tokens = torch.randn(B, H, N, D, dtype=torch.float64)
    pos = torch.randint(0, 20, (B, N, 2))
    Dh = D // 2
    inv_freq = 1.0 / (100.0 ** (torch.arange(0, Dh, 2).double() / Dh))

    def rope1d(t, p):
        fr = torch.einsum("bn,f->bnf", p.double(), inv_freq)
        fr = torch.cat((fr, fr), -1)[:, None]
        rot = torch.cat((-t[..., Dh // 2:], t[..., :Dh // 2]), -1)
        return t * fr.cos() + rot * fr.sin()

    y, x = tokens.chunk(2, dim=-1)
    ref = torch.cat((rope1d(y, pos[..., 0]), rope1d(x, pos[..., 1])), -1)
    got = aux_oracle.rope2d(tokens.transpose(1, 2).numpy(), pos.numpy())
    np.testing.assert_allclose(got, ref.transpose(1, 2).numpy(), rtol=1e-12, atol=1e-12)
    inv = aux_oracle.rope2d(got, pos.numpy(), fwd=-1.0)
    np.testing.assert_allclose(inv, tokens.transpose(1, 2).numpy(), atol=1e-12)
