"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/lvdgs.h declares,
the ctypes mirror of struct lvdgs_args matches the C layout, and argument validation works
without touching a GPU."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

from lvdgs import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "lvdgs.h")


def test_library_loads_and_reports_version():
    L = _lib.lib()
    assert b"gfx950" in L.lvdgs_version()


def test_every_declared_symbol_is_exported():
    text = open(HEADER).read()
    declared = set(re.findall(r"\b(lvdgs_[a-z0-9_]+)\s*\(", text))
    declared -= {"lvdgs_status"}
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    L = _lib.lib()
    for name in declared:
        assert hasattr(L, name), name


def test_flag_and_status_values_match_the_header():
    text = open(HEADER).read()
    enums = {k: int(v) for k, v in re.findall(r"\b(LVDGS_[A-Z_]+)\s*=\s*(\d+)", text)}
    assert enums["LVDGS_FLAG_LIST_ALL_TILES"] == _lib.FLAG_LIST_ALL_TILES
    assert enums["LVDGS_FLAG_ACCUMULATE_PARAM_GRADS"] == _lib.FLAG_ACCUMULATE_PARAM_GRADS
    assert enums["LVDGS_FLAG_POSE_ONLY"] == _lib.FLAG_POSE_ONLY
    assert enums["LVDGS_FLAG_NO_BLEND"] == _lib.FLAG_NO_BLEND
    flags = [v for k, v in enums.items() if k.startswith("LVDGS_FLAG_")]
    assert len(set(flags)) == len(flags) and all(v & (v - 1) == 0 for v in flags)   # distinct single bits
    assert (enums["LVDGS_OK"], enums["LVDGS_E_INVALID"], enums["LVDGS_E_HIP"], enums["LVDGS_E_RANGE"], enums["LVDGS_E_CAPACITY"]) == \
        (_lib.OK, _lib.E_INVALID, _lib.E_HIP, _lib.E_RANGE, _lib.E_CAPACITY)


def test_ctypes_struct_matches_c_layout(tmp_path):
    src = tmp_path / "probe.c"
    fields = [f for f, _ in _lib.Args._fields_]
    lines = "\n".join(f'    printf("{f} %zu\\n", offsetof(lvdgs_args, {f}));' for f in fields)
    src.write_text(f'#include <stdio.h>\n#include <stddef.h>\n#include "{HEADER}"\nint main(void) {{\n'
                   f'    printf("sizeof %zu\\n", sizeof(lvdgs_args));\n{lines}\n'
                   f'    printf("layout %zu\\n", sizeof(lvdgs_state_layout));\n'
                   f'    printf("ktime %zu\\n", sizeof(lvdgs_kernel_time));\n    return 0;\n}}\n')
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", "-std=c11", "-o", str(exe), str(src)])
    out = dict(line.split() for line in subprocess.check_output([str(exe)]).decode().splitlines())
    assert int(out["sizeof"]) == C.sizeof(_lib.Args)
    for f in fields:
        assert int(out[f]) == getattr(_lib.Args, f).offset, f
    assert int(out["layout"]) == C.sizeof(_lib.StateLayout)
    assert int(out["ktime"]) == C.sizeof(_lib.KernelTime)


@pytest.mark.parametrize("ctype,cname", [("LossArgs", "lvdgs_loss_args"), ("MaskedDepthArgs", "lvdgs_masked_depth_args"),
                                         ("SsimArgs", "lvdgs_ssim_args"), ("PoseStepArgs", "lvdgs_pose_step_args"),
                                         ("MaskedLossArgs", "lvdgs_masked_loss_args"),
                                         ("AdamTensor", "lvdgs_adam_tensor"), ("ViewStatsArgs", "lvdgs_view_stats_args")])
def test_every_other_ctypes_struct_matches_its_c_layout(tmp_path, ctype, cname):
    """Field offsets and sizes of the ctypes mirrors against a C probe compiled from include/lvdgs.h."""
    cls = getattr(_lib, ctype)
    fields = [f for f, _ in cls._fields_]
    lines = "\n".join(f'    printf("{f} %zu\\n", offsetof({cname}, {f}));' for f in fields)
    src = tmp_path / "probe.c"
    src.write_text(f'#include <stdio.h>\n#include <stddef.h>\n#include "{HEADER}"\nint main(void) {{\n'
                   f'    printf("sizeof %zu\\n", sizeof({cname}));\n{lines}\n    return 0;\n}}\n')
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", "-std=c11", "-o", str(exe), str(src)])
    out = dict(line.split() for line in subprocess.check_output([str(exe)]).decode().splitlines())
    assert int(out["sizeof"]) == C.sizeof(cls), ctype
    for f in fields:
        assert int(out[f]) == getattr(cls, f).offset, (ctype, f)


def test_pose_and_adam_argument_validation_without_gpu():
    L = _lib.lib()
    pa = _lib.PoseStepArgs()
    assert L.lvdgs_pose_step(C.byref(pa), None) == _lib.E_INVALID and b"NULL" in L.lvdgs_last_error()
    arr = (_lib.AdamTensor * 8)()
    assert L.lvdgs_adam_step(arr, 9, 0.9, 0.999, 1e-15, None) == _lib.E_INVALID
    assert L.lvdgs_adam_step(arr, 1, 1.0, 0.999, 1e-15, None) == _lib.E_INVALID and b"betas" in L.lvdgs_last_error()
    arr[0].numel, arr[0].step = 4, 0
    assert L.lvdgs_adam_step(arr, 1, 0.9, 0.999, 1e-15, None) == _lib.E_INVALID      # step counts start at 1
    arr[0].step = 1
    assert L.lvdgs_adam_step(arr, 1, 0.9, 0.999, 1e-15, None) == _lib.E_INVALID and b"NULL" in L.lvdgs_last_error()
    assert L.lvdgs_adam_step(arr, 0, 0.9, 0.999, 1e-15, None) == _lib.OK             # nothing to do
    md = _lib.MaskedDepthArgs()
    assert L.lvdgs_masked_depth_l1_forward(C.byref(md), None) == _lib.E_INVALID
    ml = _lib.MaskedLossArgs()
    views = (C.POINTER(_lib.MaskedLossArgs) * 1)(C.pointer(ml))
    assert L.lvdgs_masked_loss_batch(views, 1, None) == _lib.E_INVALID and b"image size" in L.lvdgs_last_error()
    ml.width, ml.height = 64, 48
    assert L.lvdgs_masked_loss_batch(views, 1, None) == _lib.E_INVALID and b"NULL" in L.lvdgs_last_error()
    assert L.lvdgs_masked_loss_batch(views, 0, None) == _lib.OK
    assert L.lvdgs_masked_loss_scratch_bytes(1226, 370) % 256 == 0 and L.lvdgs_masked_loss_scratch_bytes(1226, 370) >= 4 * 39 * 12 * 8
    a = _lib.Args()
    assert L.lvdgs_backward_masked_loss(C.byref(a), C.byref(ml), None) == _lib.E_INVALID
    assert L.lvdgs_blend_backward_window_batch(None, None, None, 1, 0, None) == _lib.E_INVALID


def test_sizes_are_monotone_and_aligned():
    L = _lib.lib()
    assert L.lvdgs_geom_bytes(0) > 0
    prev = 0
    for n in (1, 1000, 100_000, 2_000_000):
        b = L.lvdgs_geom_bytes(n)
        assert b % 256 == 0 and b > prev and b >= n * 60
        prev = b
    assert L.lvdgs_binning_bytes(3_000_000) >= 3_000_000 * 8
    assert L.lvdgs_image_bytes(1920, 1080) >= 1920 * 1080 * 8
    assert L.lvdgs_backward_scratch_bytes(500_000, 3_000_000) >= 3_000_000 * 40   # ten floats per (Gaussian, tile) pair


def test_argument_validation_without_gpu():
    L = _lib.lib()
    a = _lib.Args()
    n = C.c_int64(-1)
    assert L.lvdgs_forward_prepare(C.byref(a), C.byref(n), None) == _lib.E_INVALID
    assert b"image size" in L.lvdgs_last_error()
    a.image_width, a.image_height, a.tanfovx, a.tanfovy = 64, 64, 1.0, 1.0
    assert L.lvdgs_forward_prepare(C.byref(a), C.byref(n), None) == _lib.E_INVALID  # bg / matrices NULL
    a.sh_degree = 7
    assert L.lvdgs_forward_render(C.byref(a), None) == _lib.E_INVALID
    assert b"sh_degree" in L.lvdgs_last_error()
    with pytest.raises(_lib.LvdgsError):
        _lib.check(_lib.E_INVALID, "probe")


def test_every_export_is_in_the_integration_guide():
    """INTEGRATION.md's binding table names every entry point include/lvdgs.h declares (what each replaces on the reference's side)."""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing = [n for n in _lib.EXPORTS if n not in doc]
    assert not missing, missing


def test_gaussian_backward_batch_argument_validation_without_gpu():
    """lvdgs_gaussian_backward_batch refuses view lists it has no single launch for -- before anything is enqueued."""
    L = _lib.lib()
    assert L.lvdgs_gaussian_backward_batch(None, 0, None) == _lib.OK   # nothing to do
    assert L.lvdgs_gaussian_backward_batch(None, 2, None) == _lib.E_INVALID and b"view list" in L.lvdgs_last_error()
    a, b = _lib.Args(), _lib.Args()
    views = (C.POINTER(_lib.Args) * 2)(C.pointer(a), C.pointer(b))
    a.flags = b.flags = _lib.FLAG_POSE_ONLY
    assert L.lvdgs_gaussian_backward_batch(views, 2, None) == _lib.E_INVALID and b"POSE_ONLY" in L.lvdgs_last_error()
    a.flags = b.flags = 0
    assert L.lvdgs_gaussian_backward_batch(views, 2, None) == _lib.E_INVALID and b"one coefficient" in L.lvdgs_last_error()   # no shs
    a.shs = b.shs = 256   # (never dereferenced: every call here is rejected before a launch)
    a.sh_coeffs = b.sh_coeffs = 1
    b.num_gaussians = 5
    assert L.lvdgs_gaussian_backward_batch(views, 2, None) == _lib.E_INVALID and b"differ in map" in L.lvdgs_last_error()
    b.num_gaussians = 0
    assert L.lvdgs_gaussian_backward_batch(views, 2, None) == _lib.E_INVALID and b"ACCUMULATE_PARAM_GRADS" in L.lvdgs_last_error()
    views[1] = None
    assert L.lvdgs_gaussian_backward_batch(views, 2, None) == _lib.E_INVALID and b"view 1 is NULL" in L.lvdgs_last_error()


def test_rope2d_strided_argument_validation_without_gpu():
    L = _lib.lib()
    tok, pos = C.c_void_p(256), C.c_void_p(512)  # never dereferenced: every call below is rejected before a launch
    call = lambda dtype, B, N, H, D, sb, sn, sh: L.lvdgs_rope2d_strided(tok, dtype, pos, B, N, H, D, sb, sn, sh, 100.0, 1.0, None)
    assert call(0, 1, 4, 2, 6, 48, 12, 6) == _lib.E_INVALID and b"multiple of 4" in L.lvdgs_last_error()
    assert call(3, 1, 4, 2, 8, 64, 16, 8) == _lib.E_INVALID and b"dtype" in L.lvdgs_last_error()
    assert call(1, 1, 4, 2, 8, 64, 16, 4) == _lib.E_INVALID and b"strides" in L.lvdgs_last_error()   # heads overlap
    assert call(2, 1, 4, 2, 8, 64, -16, 8) == _lib.E_INVALID
    assert call(0, 0, 4, 2, 8, 64, 16, 8) == _lib.OK                                                    # no tokens: nothing to do
    assert L.lvdgs_rope2d_strided(None, 0, pos, 1, 4, 2, 8, 64, 16, 8, 100.0, 1.0, None) == _lib.E_INVALID


def test_product_package_never_imports_the_oracle():
    """The product path may not import, link or execute anything under oracle/."""
    pkg = os.path.join(ROOT, "lvd_gs-slam_amd")
    bad = re.compile(r"import\s+oracle|from\s+oracle|oracle/|oracle\.py|liblvdgs_oracle|lvdgs_oracle")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not bad.search(text), (dirpath, f)


def test_rasterizer_refuses_cpu_tensors():
    import torch
    from lvdgs.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    rs = GaussianRasterizationSettings(16, 16, 1.0, 1.0, torch.zeros(3), 1.0, torch.eye(4), torch.eye(4), torch.eye(4),
                                       0, torch.zeros(3))
    with pytest.raises(_lib.LvdgsError):
        GaussianRasterizer(rs)(means3D=torch.zeros(2, 3), means2D=torch.zeros(2, 3), opacities=torch.ones(2, 1),
                               colors_precomp=torch.ones(2, 3), scales=torch.ones(2, 3), rotations=torch.ones(2, 4))
