"""Timings of the two auxiliary HIP ops (simple-knn's distCUDA2, croco's RoPE-2D) at representative sizes."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import lvdgs  # noqa: E402,F401
from lvdgs.curope import rope_2d  # noqa: E402
from lvdgs.simple_knn import distCUDA2  # noqa: E402


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6


out = {}
for P in (14_000, 100_000, 500_000):  # a KITTI keyframe at pcd_downsample 32 seeds ~14k points
    pts = torch.randn(P, 3, device="cuda") * torch.tensor([10.0, 2.0, 30.0], device="cuda")
    out[f"distCUDA2 {P} points, us"] = round(timed(lambda: distCUDA2(pts)), 1)
for (B, N, H, D) in ((2, 576, 16, 64), (2, 2304, 16, 64)):  # MASt3R ViT-L tokens at 512x288 and 1024x576, two views
    tok = torch.randn(B, N, H, D, device="cuda")
    side = int(N ** 0.5)
    pos = torch.stack(torch.meshgrid(torch.arange(N // side), torch.arange(side), indexing="ij"), -1).reshape(1, N, 2).expand(B, N, 2).contiguous().cuda()
    out[f"rope2d {B}x{N}x{H}x{D}, us"] = round(timed(lambda: rope_2d(tok, pos, 100.0, 1.0)), 1)
print(out)
