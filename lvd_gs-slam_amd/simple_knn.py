"""``distCUDA2``: mean squared distance of every point to its three nearest neighbours.

Counterpart of ``simple_knn._C.distCUDA2`` (reference README.md:42; absent from the checkout and
not called by any file that is present -- upstream it seeds the scale of newly created Gaussians).
Runs ``lvdgs_dist2_knn3`` (HIP): Morton sort, 256-point boxes, box-pruned exact search.
"""
import ctypes as C

import torch

from . import _lib


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    if points.device.type != "cuda":
        raise _lib.LvdgsError("distCUDA2 needs a GPU tensor (there is no CPU path)")
    pts = points.detach().to(torch.float32).contiguous()
    P = int(pts.shape[0])
    out = torch.empty(P, dtype=torch.float32, device=pts.device)
    L = _lib.lib()
    scratch = torch.empty(max(int(L.lvdgs_knn_scratch_bytes(P)), 256), dtype=torch.uint8, device=pts.device)
    with _lib.on_device(pts.device):
        st = L.lvdgs_dist2_knn3(P, C.c_void_p(pts.data_ptr()), C.c_void_p(out.data_ptr()), C.c_void_p(scratch.data_ptr()),
                                scratch.numel(), _lib.raw_stream(pts.device))
    _lib.check(st, "lvdgs_dist2_knn3")
    return out
