"""HIP simple-knn (distCUDA2) and RoPE-2D against their CPU restatements."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,kind", [(1, "u"), (3, "u"), (4, "u"), (257, "u"), (5000, "u"), (20000, "clustered"), (3000, "plane")])
def test_dist2_knn3_matches_brute_force(n, kind):
    import aux_oracle
    from lvdgs.simple_knn import distCUDA2
    g = torch.Generator().manual_seed(n)
    pts = torch.rand(n, 3, generator=g) * 10 - 5
    if kind == "clustered":
        pts = torch.randn(n, 3, generator=g) * torch.tensor([0.1, 3.0, 0.5]) + (torch.randint(0, 4, (n, 1), generator=g) * 7.0)
    if kind == "plane":
        pts[:, 2] = 1.5  # degenerate axis (zero extent)
    out = distCUDA2(pts.cuda()).cpu().numpy()
    ref = aux_oracle.dist2_knn3(pts.numpy())
    np.testing.assert_allclose(out, ref, rtol=2e-5, atol=1e-9)


def test_dist2_knn3_large_input_sampled_against_brute_force():
    """Above 200 000 points the search switches to 256-point boxes: check 400 random points of 260 000 exactly."""
    from lvdgs.simple_knn import distCUDA2
    g = torch.Generator().manual_seed(7)
    n = 260_000
    pts = torch.randn(n, 3, generator=g) * torch.tensor([10.0, 2.0, 30.0])
    out = distCUDA2(pts.cuda()).cpu().numpy()
    pick = torch.randperm(n, generator=g)[:400]
    d2 = ((pts[pick].double()[:, None, :] - pts.double()[None, :, :]) ** 2).sum(-1)  # 400 x n
    d2[torch.arange(400), pick] = float("inf")
    ref = d2.topk(3, dim=1, largest=False).values.mean(1).numpy()
    np.testing.assert_allclose(out[pick.numpy()], ref, rtol=2e-5, atol=1e-9)


def test_dist2_knn3_duplicates():
    import aux_oracle
    from lvdgs.simple_knn import distCUDA2
    pts = torch.rand(500, 3)
    pts[100:110] = pts[0]
    out = distCUDA2(pts.cuda()).cpu().numpy()
    np.testing.assert_allclose(out, aux_oracle.dist2_knn3(pts.numpy()), rtol=2e-5, atol=1e-9)
    assert out[0] == 0.0


@pytest.mark.parametrize("B,N,H,D", [(1, 7, 1, 4), (2, 196, 12, 64), (1, 1024, 16, 64), (3, 33, 5, 24)])
def test_rope2d_forward_and_inverse(B, N, H, D):
    import aux_oracle
    from lvdgs.curope import cuRoPE2D, rope_2d
    g = torch.Generator().manual_seed(B * 1000 + N)
    tokens = torch.randn(B, H, N, D, generator=g)
    pos = torch.stack([torch.randint(0, 37, (B, N), generator=g), torch.randint(0, 53, (B, N), generator=g)], -1)
    ref = aux_oracle.rope2d(tokens.transpose(1, 2).numpy(), pos.numpy(), base=100.0, fwd=1.0)
    t = tokens.cuda().clone()
    out = cuRoPE2D(freq=100.0)(t, pos.cuda())
    assert out.data_ptr() == t.data_ptr()  # in place
    np.testing.assert_allclose(out.transpose(1, 2).cpu().numpy(), ref, rtol=1e-5, atol=2e-5)
    # the backward pass is the inverse rotation
    back = out.transpose(1, 2).contiguous()
    rope_2d(back, pos.cuda(), 100.0, -1.0)
    np.testing.assert_allclose(back.cpu().numpy(), tokens.transpose(1, 2).numpy(), rtol=1e-5, atol=2e-5)


def test_rope2d_autograd():
    import aux_oracle
    from lvdgs.curope import cuRoPE2D_func
    B, N, H, D = 1, 16, 2, 8
    x = torch.randn(B, N, H, D, device="cuda", requires_grad=True)
    pos = torch.randint(0, 9, (B, N, 2), device="cuda")
    w = torch.randn(B, N, H, D, device="cuda")
    y = cuRoPE2D_func.apply(x.clone(), pos, 100.0, 1.0)
    (y * w).sum().backward()
    # d/dx of a rotation applied to x, contracted with w, is the inverse rotation of w
    ref = aux_oracle.rope2d(w.cpu().numpy(), pos.cpu().numpy(), 100.0, -1.0)
    np.testing.assert_allclose(x.grad.cpu().numpy(), ref, rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 2e-3), (torch.bfloat16, 1.6e-2), (torch.float32, 2e-5)])
@pytest.mark.parametrize("B,N,H,D", [(2, 196, 12, 64), (1, 577, 16, 64), (2, 33, 5, 24), (1, 9, 3, 4)])
def test_rope2d_half_precision_in_the_attention_layout(dtype, tol, B, N, H, D):
    """croco applies curope to q / k as (B, H, N, D) tensors, in half precision under autocast (reference README.md:49-50,
    MASt3R inference utils/slam_frontend.py:1448,1455).  The kernel rotates the strided (B, N, H, D) view where it lies:
    same storage, no transposed copy, float32 arithmetic, one rounding at the store (so the error is half an ulp of the
    result: 2^-11 relative for float16, 2^-8 for bfloat16)."""
    import aux_oracle
    from lvdgs.curope import cuRoPE2D
    g = torch.Generator().manual_seed(B * 977 + N + D)
    tokens = torch.randn(B, H, N, D, generator=g).to(dtype)
    pos = torch.stack([torch.randint(0, 37, (B, N), generator=g), torch.randint(0, 53, (B, N), generator=g)], -1)
    ref = aux_oracle.rope2d(tokens.float().transpose(1, 2).numpy(), pos.numpy(), base=100.0, fwd=1.0)
    t = tokens.cuda().clone()
    before = t.data_ptr()
    out = cuRoPE2D(freq=100.0)(t, pos.cuda())
    assert out is t and out.data_ptr() == before and out.dtype == dtype and out.shape == (B, H, N, D)
    got = out.float().transpose(1, 2).cpu().numpy()
    np.testing.assert_allclose(got, ref, rtol=tol / 2, atol=tol * np.abs(ref).max() / 8)
    if dtype is not torch.float32:
        # exactly the float32 rotation of the same (already rounded) inputs, rounded once
        f = tokens.float().cuda().clone()
        cuRoPE2D(freq=100.0)(f, pos.cuda())
        assert torch.equal(f.to(dtype), out)


def test_rope2d_strided_views_and_autograd_in_half():
    """A slice of a larger buffer (token stride larger than H*D, odd base offset -> scalar path) and the autograd
    function on a float16 (B, H, N, D) tensor: the gradient is the inverse rotation, in place on the incoming gradient."""
    import aux_oracle
    from lvdgs.curope import cuRoPE2D, cuRoPE2D_func, rope_2d
    g = torch.Generator().manual_seed(5)
    B, N, H, D = 2, 50, 4, 32
    big = torch.randn(B, N, H, D + 3, generator=g).cuda()
    view = big[..., 1:1 + D]            # last axis contiguous, every other stride off the 4-element grid
    keep = big.clone()
    pos = torch.randint(0, 20, (B, N, 2), generator=g)
    ref = aux_oracle.rope2d(view.cpu().numpy(), pos.numpy(), 100.0, 1.0)
    rope_2d(view, pos.cuda(), 100.0, 1.0)
    np.testing.assert_allclose(view.cpu().numpy(), ref, rtol=1e-5, atol=2e-5)
    assert torch.equal(big[..., 0], keep[..., 0]) and torch.equal(big[..., 1 + D:], keep[..., 1 + D:])  # neighbours untouched
    with pytest.raises(ValueError):
        rope_2d(big.transpose(2, 3)[:, :, :H, :], pos.cuda(), 100.0, 1.0)  # feature axis not contiguous

    x = torch.randn(B, H, N, D, generator=g).half().cuda().requires_grad_(True)
    w = torch.randn(B, H, N, D, generator=g).half().cuda()
    y = cuRoPE2D(freq=100.0)(x.clone(), pos.cuda())
    (y.float() * w.float()).sum().backward()
    ref_g = aux_oracle.rope2d(w.float().transpose(1, 2).cpu().numpy(), pos.numpy(), 100.0, -1.0)
    np.testing.assert_allclose(x.grad.float().transpose(1, 2).cpu().numpy(), ref_g, rtol=1e-3, atol=2e-3)
    assert x.grad.dtype == torch.float16
