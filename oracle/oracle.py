"""ctypes binding of the CPU oracle (oracle/lvdgs_oracle.c).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the
product package.  Parity unpinned -- see the header of lvdgs_oracle.c.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


def build(force=False):
    out = os.path.join(HERE, "_build")
    need = force or not all(os.path.exists(os.path.join(out, f"liblvdgs_oracle_{p}.so")) for p in ("f32", "f64", "f32_omp"))
    if not need:
        src = os.path.getmtime(os.path.join(HERE, "lvdgs_oracle.c"))
        need = any(os.path.getmtime(os.path.join(out, f"liblvdgs_oracle_{p}.so")) < src for p in ("f32", "f64", "f32_omp"))
    if need:
        subprocess.check_call(["make", "-C", HERE, "-s"])


def _fields(real):
    rp, ip, up, bp = C.POINTER(real), C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)
    return [
        ("N", C.c_int32), ("W", C.c_int32), ("H", C.c_int32), ("sh_degree", C.c_int32), ("M", C.c_int32),
        ("prefiltered", C.c_int32), ("tanfovx", real), ("tanfovy", real), ("scale_modifier", real),
        ("means3D", rp), ("scales", rp), ("rotations", rp), ("opacities", rp), ("shs", rp),
        ("colors_precomp", rp), ("cov3D_precomp", rp), ("viewmatrix", rp), ("projmatrix", rp),
        ("projmatrix_raw", rp), ("campos", rp), ("bg", rp),
        ("num_rendered", C.c_int64), ("out_color", rp), ("out_depth", rp), ("out_opacity", rp),
        ("radii", ip), ("n_touched", ip), ("means2D", rp), ("depths", rp), ("conic_opacity", rp), ("rgb", rp),
        ("cov3D", rp), ("clamped", bp), ("tiles_touched", up), ("rect", ip),
        ("keys_sorted", C.POINTER(C.c_uint64)), ("ids_sorted", up), ("ranges", up), ("final_T", rp),
        ("n_contrib", up), ("fragile", bp), ("n_contrib_lo", up), ("n_contrib_hi", up),
        ("dL_dcolor", rp), ("dL_ddepth", rp), ("dL_dopacity_img", rp),
        ("dL_dmeans3D", rp), ("dL_dmeans2D", rp), ("dL_dscales", rp), ("dL_drotations", rp), ("dL_dopacity", rp),
        ("dL_dcolors", rp), ("dL_dshs", rp), ("dL_dcov3D", rp), ("dL_dtau", rp),
    ]


class _CtxF32(C.Structure):
    _fields_ = _fields(C.c_float)


class _CtxF64(C.Structure):
    _fields_ = _fields(C.c_double)


def _lib(prec):
    if prec not in _LIBS:
        build()
        lib = C.CDLL(os.path.join(HERE, "_build", f"liblvdgs_oracle_{prec}.so"))
        assert lib.oracle_real_bytes() == (8 if prec == "f64" else 4)
        _LIBS[prec] = lib
    return _LIBS[prec]


def _np(ptr, shape, dtype):
    n = int(np.prod(shape))
    if n == 0:
        return np.zeros(shape, dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).view(dtype).reshape(shape).copy()


class Oracle:
    """One forward (+ optional backward) of the CPU oracle on numpy inputs.

    ``precision`` is "f32" (parity target of the HIP path), "f64" (gradient pinning) or "f32_omp" (the f32 restatement
    spread over the host's cores with OpenMP -- bench.py's CPU baseline; its per-Gaussian sums are added in another order,
    so it is not what the parity tests compare with).
    Inputs: means3D (N,3), opacities (N,), scales (N,3), rotations (N,4), shs (N,M,3) or
    colors_precomp (N,3), optional cov3D_precomp (N,6); viewmatrix/projmatrix/projmatrix_raw
    (4,4 row-vector layout as the reference's Camera produces them), campos (3,), bg (3,).
    """

    def __init__(self, precision="f32"):
        self.prec = precision
        self.lib = _lib(precision)
        self.real = np.float64 if precision == "f64" else np.float32
        self.creal = C.c_double if precision == "f64" else C.c_float
        self.ctx = (_CtxF64 if precision == "f64" else _CtxF32)()
        self.threads = int(self.lib.oracle_threads())
        self._keep = []
        self._live = False

    def _arr(self, a, shape=None):
        if a is None:
            return None
        a = np.ascontiguousarray(np.asarray(a, dtype=self.real))
        if shape is not None:
            assert a.shape == tuple(shape), (a.shape, shape)
        self._keep.append(a)
        return a.ctypes.data_as(C.POINTER(self.creal))

    def forward(self, *, means3D, opacities, W, H, tanfovx, tanfovy, viewmatrix, projmatrix, projmatrix_raw=None,
                campos=None, bg=None, scales=None, rotations=None, cov3D_precomp=None, shs=None,
                colors_precomp=None, sh_degree=0, scale_modifier=1.0, prefiltered=False):
        if self._live:
            self.free()
        c = self.ctx
        N = int(np.asarray(means3D).shape[0])
        c.N, c.W, c.H, c.sh_degree = N, int(W), int(H), int(sh_degree)
        c.M = int(np.asarray(shs).shape[1]) if shs is not None else 0
        c.prefiltered = int(prefiltered)
        c.tanfovx, c.tanfovy, c.scale_modifier = float(tanfovx), float(tanfovy), float(scale_modifier)
        c.means3D = self._arr(means3D, (N, 3))
        c.scales = self._arr(scales)
        c.rotations = self._arr(rotations)
        c.opacities = self._arr(np.asarray(opacities).reshape(-1), (N,))
        c.shs = self._arr(shs)
        c.colors_precomp = self._arr(colors_precomp)
        c.cov3D_precomp = self._arr(cov3D_precomp)
        c.viewmatrix = self._arr(viewmatrix, (4, 4))
        c.projmatrix = self._arr(projmatrix, (4, 4))
        c.projmatrix_raw = self._arr(projmatrix_raw if projmatrix_raw is not None else np.eye(4), (4, 4))
        c.campos = self._arr(campos if campos is not None else np.zeros(3), (3,))
        c.bg = self._arr(bg if bg is not None else np.zeros(3), (3,))
        assert (colors_precomp is None) != (shs is None)
        assert (cov3D_precomp is None) != (scales is None)
        rc = self.lib.oracle_forward(C.byref(c))
        assert rc == 0
        self._live = True
        P, D = c.W * c.H, int(c.num_rendered)
        NT = ((c.W + 15) // 16) * ((c.H + 15) // 16)
        r = self.real
        out = dict(
            num_rendered=D,
            color=_np(c.out_color, (3, c.H, c.W), r), depth=_np(c.out_depth, (1, c.H, c.W), r),
            opacity=_np(c.out_opacity, (1, c.H, c.W), r), radii=_np(c.radii, (N,), np.int32),
            n_touched=_np(c.n_touched, (N,), np.int32), means2D=_np(c.means2D, (N, 2), r),
            depths=_np(c.depths, (N,), r), conic_opacity=_np(c.conic_opacity, (N, 4), r), rgb=_np(c.rgb, (N, 3), r),
            cov3D=_np(c.cov3D, (N, 6), r), clamped=_np(c.clamped, (N, 3), np.uint8),
            tiles_touched=_np(c.tiles_touched, (N,), np.uint32), rect=_np(c.rect, (N, 4), np.int32),
            keys_sorted=_np(c.keys_sorted, (D,), np.uint64), ids_sorted=_np(c.ids_sorted, (D,), np.uint32),
            ranges=_np(c.ranges, (NT, 2), np.uint32), final_T=_np(c.final_T, (c.H, c.W), r),
            n_contrib=_np(c.n_contrib, (c.H, c.W), np.uint32), fragile=_np(c.fragile, (c.H, c.W), np.uint8),
            n_contrib_lo=_np(c.n_contrib_lo, (c.H, c.W), np.uint32), n_contrib_hi=_np(c.n_contrib_hi, (c.H, c.W), np.uint32),
        )
        return out

    def backward(self, dL_dcolor, dL_ddepth=None, dL_dopacity=None):
        assert self._live, "forward first"
        c = self.ctx
        N, M = c.N, max(c.M, 1)
        c.dL_dcolor = self._arr(dL_dcolor, (3, c.H, c.W))
        c.dL_ddepth = self._arr(None if dL_ddepth is None else np.asarray(dL_ddepth).reshape(c.H, c.W))
        c.dL_dopacity_img = self._arr(None if dL_dopacity is None else np.asarray(dL_dopacity).reshape(c.H, c.W))
        rc = self.lib.oracle_backward(C.byref(c))
        assert rc == 0
        r = self.real
        return dict(
            means3D=_np(c.dL_dmeans3D, (N, 3), r), means2D=_np(c.dL_dmeans2D, (N, 3), r),
            scales=_np(c.dL_dscales, (N, 3), r), rotations=_np(c.dL_drotations, (N, 4), r),
            opacities=_np(c.dL_dopacity, (N,), r), colors=_np(c.dL_dcolors, (N, 3), r),
            shs=_np(c.dL_dshs, (N, M, 3), r), cov3D=_np(c.dL_dcov3D, (N, 6), r), tau=_np(c.dL_dtau, (6,), r),
        )

    def free(self):
        if self._live:
            self.lib.oracle_free(C.byref(self.ctx))
            self._live = False
        self._keep.clear()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def mark_visible(means3D, viewmatrix, precision="f32"):
    lib = _lib(precision)
    real = np.float32 if precision == "f32" else np.float64
    creal = C.c_float if precision == "f32" else C.c_double
    m = np.ascontiguousarray(means3D, dtype=real)
    v = np.ascontiguousarray(viewmatrix, dtype=real)
    out = np.zeros(m.shape[0], np.uint8)
    lib.oracle_mark_visible(C.c_int(m.shape[0]), m.ctypes.data_as(C.POINTER(creal)), v.ctypes.data_as(C.POINTER(creal)),
                            out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out.astype(bool)
