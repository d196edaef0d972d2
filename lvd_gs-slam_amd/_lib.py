"""ctypes binding of ``lib/liblvdgs.so`` (C ABI: ``include/lvdgs.h``).

The library is the only compute path: if it is missing or fails to load, importing a symbol
from here raises -- there is no CPU or PyTorch fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# LVDGS_LIB: another build of the library (same-box A/B of compile-time variants: `make OUT=../lib_b EXTRA=-D...`)
LIB_PATH = os.environ.get("LVDGS_LIB") or os.path.join(_HERE, "lib", "liblvdgs.so")

OK, E_INVALID, E_HIP, E_RANGE, E_CAPACITY = 0, 1, 2, 3, 4
FLAG_LIST_ALL_TILES = 1
FLAG_ACCUMULATE_PARAM_GRADS = 2   # lvdgs_args.flags
FLAG_POSE_ONLY = 4
FLAG_NO_BLEND = 8   # the call leaves its blend pass to lvdgs_blend_*_batch
FLAG_SUPER_TILES = 16   # hint: group and depth-sort the pairs per 64 x 64-pixel super-tile (same outputs; rasterizer.super_tiles_flag)

_fp = C.c_void_p


class Args(C.Structure):
    """struct lvdgs_args (include/lvdgs.h) -- field order must match the header."""
    _fields_ = [
        ("image_height", C.c_int32), ("image_width", C.c_int32),
        ("tanfovx", C.c_float), ("tanfovy", C.c_float), ("scale_modifier", C.c_float),
        ("sh_degree", C.c_int32), ("prefiltered", C.c_int32), ("debug", C.c_int32),
        ("bg", _fp), ("viewmatrix", _fp), ("projmatrix", _fp), ("projmatrix_raw", _fp), ("campos", _fp),
        ("num_gaussians", C.c_int32), ("sh_coeffs", C.c_int32),
        ("means3D", _fp), ("opacities", _fp), ("scales", _fp), ("rotations", _fp), ("cov3D_precomp", _fp),
        ("shs", _fp), ("colors_precomp", _fp),
        ("geom_state", _fp), ("geom_bytes", C.c_size_t),
        ("binning_state", _fp), ("binning_bytes", C.c_size_t),
        ("image_state", _fp), ("image_bytes", C.c_size_t),
        ("scratch", _fp), ("scratch_bytes", C.c_size_t),
        ("num_rendered", C.c_int64),
        ("radii", _fp), ("out_color", _fp), ("out_depth", _fp), ("out_opacity", _fp), ("n_touched", _fp),
        ("dL_dout_color", _fp), ("dL_dout_depth", _fp), ("dL_dout_opacity", _fp),
        ("dL_dmeans3D", _fp), ("dL_dmeans2D", _fp), ("dL_dopacities", _fp), ("dL_dscales", _fp),
        ("dL_drotations", _fp), ("dL_dcov3D", _fp), ("dL_dshs", _fp), ("dL_dcolors", _fp), ("dL_dtau", _fp),
        ("pair_capacity", C.c_int64), ("activations", C.c_int32),
        ("flags", C.c_int32), ("tile_row_begin", C.c_int32), ("tile_row_end", C.c_int32),
    ]


class LossArgs(C.Structure):
    """struct lvdgs_loss_args (include/lvdgs.h)."""
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32),
        ("image", _fp), ("depth", _fp), ("opacity", _fp), ("gt_image", _fp), ("gt_depth", _fp), ("grad_mask", _fp),
        ("exposure_a", _fp), ("exposure_b", _fp),
        ("rgb_boundary_threshold", C.c_float), ("weight_rgb", C.c_float), ("weight_depth", C.c_float),
        ("weight_by_opacity", C.c_int32), ("depth_needs_opaque", C.c_int32),
        ("scratch", _fp), ("scratch_bytes", C.c_size_t),
        ("loss", _fp), ("grad_loss", _fp), ("d_image", _fp), ("d_depth", _fp), ("d_opacity", _fp),
        ("d_exposure_a", _fp), ("d_exposure_b", _fp),
    ]


class MaskedDepthArgs(C.Structure):
    """struct lvdgs_masked_depth_args (include/lvdgs.h)."""
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32),
        ("depth", _fp), ("gt_depth", _fp), ("static_mask", _fp),
        ("scratch", _fp), ("scratch_bytes", C.c_size_t),
        ("out", _fp), ("grad_loss", _fp), ("d_depth", _fp),
    ]


class PoseStepArgs(C.Structure):
    """struct lvdgs_pose_step_args (include/lvdgs.h)."""
    _fields_ = [
        ("R", _fp), ("T", _fp), ("cam_rot_delta", _fp), ("cam_trans_delta", _fp), ("exposure_a", _fp), ("exposure_b", _fp),
        ("grad_tau", _fp), ("grad_exposure_a", _fp), ("grad_exposure_b", _fp), ("state", _fp),
        ("lr_rot", C.c_double), ("lr_trans", C.c_double), ("lr_exposure", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double),
        ("eps", C.c_double), ("converged_threshold", C.c_float),
        ("projmatrix_raw", _fp), ("viewmatrix", _fp), ("projmatrix", _fp), ("campos", _fp),
        ("grad_rot", _fp), ("grad_trans", _fp), ("host_flags", _fp),
    ]


class ViewStatsArgs(C.Structure):
    """struct lvdgs_view_stats_args (include/lvdgs.h)."""
    _fields_ = [("radii_max", _fp), ("norm_sum", _fp), ("vis_count", _fp), ("touched_row", _fp), ("split_xy", _fp)]


class AdamTensor(C.Structure):
    """struct lvdgs_adam_tensor (include/lvdgs.h)."""
    _fields_ = [("param", _fp), ("grad", _fp), ("exp_avg", _fp), ("exp_avg_sq", _fp), ("numel", C.c_int64), ("step", C.c_int64),
                ("lr", C.c_double)]


class SsimArgs(C.Structure):
    """struct lvdgs_ssim_args (include/lvdgs.h)."""
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32), ("planes", C.c_int32), ("channels", C.c_int32),
        ("img1", _fp), ("img2", _fp), ("keep_mask", _fp), ("bg", _fp),
        ("weight_l1", C.c_float), ("weight_ssim", C.c_float),
        ("scratch", _fp), ("scratch_bytes", C.c_size_t),
        ("out", _fp), ("d_img1", _fp),
    ]


class MaskedLossArgs(C.Structure):
    """struct lvdgs_masked_loss_args (include/lvdgs.h)."""
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32),
        ("image", _fp), ("gt_image", _fp), ("static_mask", _fp), ("bg", _fp), ("depth", _fp), ("gt_depth", _fp),
        ("lambda_dssim", C.c_float), ("depth_lambda", C.c_float),
        ("scratch", _fp), ("scratch_bytes", C.c_size_t),
        ("d_image", _fp), ("out", _fp),
    ]


class StateLayout(C.Structure):
    _fields_ = [(n, C.c_size_t) for n in (
        "geom_rec", "geom_tiles_touched", "geom_slot_base", "bin_point_list", "bin_tile_keys",
        "img_ranges", "img_final_T", "img_n_contrib", "geom_rec_floats")]


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_int64), ("total_ms", C.c_double)]


# every symbol include/lvdgs.h declares (checked by tests/test_abi.py)
EXPORTS = (
    "lvdgs_geom_bytes", "lvdgs_prepare_scratch_bytes", "lvdgs_binning_bytes", "lvdgs_image_bytes",
    "lvdgs_render_scratch_bytes", "lvdgs_backward_scratch_bytes", "lvdgs_forward_prepare", "lvdgs_forward_render",
    "lvdgs_forward",
    "lvdgs_backward", "lvdgs_mark_visible", "lvdgs_state_layout_query", "lvdgs_knn_scratch_bytes",
    "lvdgs_dist2_knn3", "lvdgs_rope2d", "lvdgs_rope2d_strided", "lvdgs_loss_scratch_bytes", "lvdgs_photometric_loss_forward",
    "lvdgs_photometric_loss_backward", "lvdgs_photometric_loss_value_and_grad", "lvdgs_photometric_loss_partials", "lvdgs_tracking_tail", "lvdgs_backward_fused_loss", "lvdgs_blend_forward_batch", "lvdgs_blend_backward_fused_loss_batch", "lvdgs_masked_depth_scratch_bytes", "lvdgs_masked_depth_l1_forward",
    "lvdgs_masked_depth_l1_backward", "lvdgs_pose_step", "lvdgs_host_device_pointer", "lvdgs_pose_step_batch", "lvdgs_adam_step", "lvdgs_isotropic_scratch_bytes", "lvdgs_isotropic_reg", "lvdgs_view_stats", "lvdgs_map_stats_apply", "lvdgs_map_view_tail", "lvdgs_ssim_scratch_bytes", "lvdgs_ssim_l1",
    "lvdgs_masked_loss_scratch_bytes", "lvdgs_masked_loss_batch", "lvdgs_backward_masked_loss", "lvdgs_blend_backward_window_batch", "lvdgs_forward_batch", "lvdgs_forward_backward_fused_loss", "lvdgs_map_view_tail_batch", "lvdgs_gaussian_backward_batch", "lvdgs_last_error", "lvdgs_version", "lvdgs_profile_enable",
    "lvdgs_profile_reset", "lvdgs_profile_read",
)

_lib = None


class LvdgsError(RuntimeError):
    pass


def lib():
    """The loaded library (loads on first use; raises if the HIP build is missing)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LvdgsError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C lvd_gs-slam_amd/csrc`). There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        L.lvdgs_last_error.restype = C.c_char_p
        L.lvdgs_version.restype = C.c_char_p
        for name in ("lvdgs_geom_bytes", "lvdgs_prepare_scratch_bytes"):
            getattr(L, name).restype = C.c_size_t
            getattr(L, name).argtypes = [C.c_int32]
        L.lvdgs_binning_bytes.restype = C.c_size_t
        L.lvdgs_binning_bytes.argtypes = [C.c_int64]
        L.lvdgs_image_bytes.restype = C.c_size_t
        L.lvdgs_image_bytes.argtypes = [C.c_int32, C.c_int32]
        L.lvdgs_render_scratch_bytes.restype = C.c_size_t
        L.lvdgs_render_scratch_bytes.argtypes = [C.c_int32, C.c_int64, C.c_int32, C.c_int32]
        L.lvdgs_backward_scratch_bytes.restype = C.c_size_t
        L.lvdgs_backward_scratch_bytes.argtypes = [C.c_int32, C.c_int64]
        L.lvdgs_forward_prepare.argtypes = [C.POINTER(Args), C.POINTER(C.c_int64), C.c_void_p]
        L.lvdgs_forward_render.argtypes = [C.POINTER(Args), C.c_void_p]
        L.lvdgs_forward.argtypes = [C.POINTER(Args), C.POINTER(C.c_int64), C.c_void_p]
        L.lvdgs_backward.argtypes = [C.POINTER(Args), C.c_void_p]
        L.lvdgs_blend_forward_batch.argtypes = [C.POINTER(C.POINTER(Args)), C.c_int32, C.c_void_p]
        L.lvdgs_blend_backward_fused_loss_batch.argtypes = [C.POINTER(C.POINTER(Args)), C.POINTER(C.POINTER(LossArgs)), C.c_int32, C.c_int32, C.c_void_p]
        L.lvdgs_mark_visible.argtypes = [C.c_int32, _fp, _fp, _fp, _fp, C.c_void_p]
        L.lvdgs_host_device_pointer.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.lvdgs_state_layout_query.argtypes = [C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.POINTER(StateLayout)]
        L.lvdgs_knn_scratch_bytes.restype = C.c_size_t
        L.lvdgs_knn_scratch_bytes.argtypes = [C.c_int32]
        L.lvdgs_dist2_knn3.argtypes = [C.c_int32, _fp, _fp, _fp, C.c_size_t, C.c_void_p]
        L.lvdgs_rope2d.argtypes = [_fp, _fp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_void_p]
        L.lvdgs_rope2d_strided.argtypes = [_fp, C.c_int32, _fp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int64,
                                           C.c_int64, C.c_float, C.c_float, C.c_void_p]
        L.lvdgs_loss_scratch_bytes.restype = C.c_size_t
        L.lvdgs_loss_scratch_bytes.argtypes = [C.c_int32, C.c_int32]
        L.lvdgs_photometric_loss_forward.argtypes = [C.POINTER(LossArgs), C.c_void_p]
        L.lvdgs_photometric_loss_backward.argtypes = [C.POINTER(LossArgs), C.c_void_p]
        L.lvdgs_photometric_loss_value_and_grad.argtypes = [C.POINTER(LossArgs), C.c_void_p]
        L.lvdgs_masked_depth_scratch_bytes.restype = C.c_size_t
        L.lvdgs_masked_depth_scratch_bytes.argtypes = [C.c_int32, C.c_int32]
        L.lvdgs_masked_depth_l1_forward.argtypes = [C.POINTER(MaskedDepthArgs), C.c_void_p]
        L.lvdgs_masked_depth_l1_backward.argtypes = [C.POINTER(MaskedDepthArgs), C.c_void_p]
        L.lvdgs_pose_step.argtypes = [C.POINTER(PoseStepArgs), C.c_void_p]
        L.lvdgs_pose_step_batch.argtypes = [C.POINTER(PoseStepArgs), C.c_int32, C.c_void_p]
        L.lvdgs_photometric_loss_partials.argtypes = [C.POINTER(LossArgs), C.c_void_p]
        L.lvdgs_tracking_tail.argtypes = [C.POINTER(LossArgs), C.POINTER(Args), C.POINTER(PoseStepArgs), C.c_void_p, C.c_int32, C.c_void_p]
        L.lvdgs_backward_fused_loss.argtypes = [C.POINTER(Args), C.POINTER(LossArgs), C.c_int32, C.c_void_p]
        L.lvdgs_adam_step.argtypes = [C.POINTER(AdamTensor), C.c_int32, C.c_double, C.c_double, C.c_double, C.c_void_p]
        L.lvdgs_isotropic_scratch_bytes.restype = C.c_size_t
        L.lvdgs_isotropic_scratch_bytes.argtypes = [C.c_int32]
        L.lvdgs_isotropic_reg.argtypes = [C.c_int32, _fp, _fp, C.c_float, _fp, C.c_size_t, _fp, C.c_void_p]
        L.lvdgs_view_stats.argtypes = [C.c_int32, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, C.c_void_p]
        L.lvdgs_map_view_tail.argtypes = [C.POINTER(LossArgs), C.POINTER(Args), _fp, C.POINTER(ViewStatsArgs), C.c_void_p]
        L.lvdgs_map_view_tail_batch.argtypes = [C.POINTER(C.POINTER(LossArgs)), C.POINTER(C.POINTER(Args)), C.POINTER(C.c_void_p),
                                                C.POINTER(C.POINTER(ViewStatsArgs)), C.c_int32, C.c_void_p]
        L.lvdgs_map_stats_apply.argtypes = [C.c_int32, _fp, _fp, _fp, _fp, C.c_int32, _fp, _fp, _fp, C.c_void_p]
        L.lvdgs_ssim_scratch_bytes.restype = C.c_size_t
        L.lvdgs_ssim_scratch_bytes.argtypes = [C.c_int32, C.c_int32, C.c_int32]
        L.lvdgs_ssim_l1.argtypes = [C.POINTER(SsimArgs), C.c_void_p]
        L.lvdgs_forward_backward_fused_loss.argtypes = [C.POINTER(Args), C.POINTER(LossArgs), C.c_int32, C.POINTER(C.c_int64), C.c_void_p]
        L.lvdgs_forward_batch.argtypes = [C.POINTER(C.POINTER(Args)), C.c_int32, C.POINTER(C.c_int64), C.c_void_p]
        L.lvdgs_masked_loss_scratch_bytes.restype = C.c_size_t
        L.lvdgs_masked_loss_scratch_bytes.argtypes = [C.c_int32, C.c_int32]
        L.lvdgs_masked_loss_batch.argtypes = [C.POINTER(C.POINTER(MaskedLossArgs)), C.c_int32, C.c_void_p]
        L.lvdgs_backward_masked_loss.argtypes = [C.POINTER(Args), C.POINTER(MaskedLossArgs), C.c_void_p]
        L.lvdgs_gaussian_backward_batch.argtypes = [C.POINTER(C.POINTER(Args)), C.c_int32, C.c_void_p]
        L.lvdgs_blend_backward_window_batch.argtypes = [C.POINTER(C.POINTER(Args)), C.POINTER(C.POINTER(LossArgs)), C.POINTER(C.POINTER(MaskedLossArgs)),
                                                        C.c_int32, C.c_int32, C.c_void_p]
        L.lvdgs_profile_enable.argtypes = [C.c_int]
        L.lvdgs_profile_read.argtypes = [C.POINTER(KernelTime), C.c_int]
        _lib = L
    return _lib


def raw_stream(device):
    """hipStream_t of PyTorch's current stream ON `device` (the tensor's device, which need not be the current one),
    as a c_void_p.  A direct C call: torch.cuda.current_stream() costs tens of microseconds of Python per call."""
    import torch
    idx = device.index if (device is not None and device.index is not None) else torch.cuda.current_device()
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(idx))


class on_device:
    """Make `device` the current HIP device for the duration of a library call when it is not already (kernels are
    launched on the current device; a stream of another device would be rejected)."""
    __slots__ = ("idx", "prev")

    def __init__(self, device):
        self.idx = device.index

    def __enter__(self):
        import torch
        self.prev = None
        if self.idx is not None:
            cur = torch.cuda.current_device()
            if cur != self.idx:
                self.prev = cur
                torch.cuda.set_device(self.idx)

    def __exit__(self, *exc):
        if self.prev is not None:
            import torch
            torch.cuda.set_device(self.prev)
        return False


_gc_policy = None      # "scoped" | "process" | "off"; None: not decided yet (LVDGS_GC_FREEZE / set_gc_policy / first loop entry)
_gc_depth = 0


def set_gc_policy(policy):
    """How the product's loops keep CPython's full-heap garbage collections out of their iterations.

    The loops run at 0.2-2 ms per iteration and allocate a few hundred small Python objects in each, which makes CPython start a
    FULL collection every few dozen to few hundred iterations; with the whole heap of an application that has PyTorch imported
    (about a million objects) in the oldest generation that is one stall of 40-110 ms (measured: ``tools/side_stall_diag.py``) --
    20-60 iterations' worth of time.

    * ``"scoped"`` (the default): ``gc.freeze()`` when the outermost product loop is entered (``track_frame``, ``map_window``,
      ``initialize_map``, ``color_refinement``, ``slam_sequence``) -- everything alive moves to the permanent generation, two list
      splices, no collection -- and ``gc.unfreeze()`` when it returns: inside the loop the collector walks only what the loop itself
      allocated, and afterwards the host application's collector sees its heap exactly as before (``gc.get_freeze_count() == 0``).
      An application that calls ``gc.freeze()`` ITSELF should pick one of the other two: the scoped exit would thaw its objects too
      (a freeze found in place at the first loop entry of the process switches the policy to "off" by itself).
    * ``"process"`` (``LVDGS_GC_FREEZE=1``, what every round up to 5 did unasked): one ``gc.collect(); gc.freeze()`` at the first loop
      entry, for the life of the process.  Cyclic garbage among the objects alive at that moment is never reclaimed.
    * ``"off"`` (``LVDGS_GC_FREEZE=0``): the collector is left alone.
    """
    global _gc_policy
    if policy not in ("scoped", "process", "off"):
        raise ValueError("gc policy: 'scoped', 'process' or 'off'")
    if policy == "process" and _gc_policy != "process":
        import gc
        gc.collect()
        gc.freeze()
    _gc_policy = policy


class quiet_gc:
    """Context manager around a product loop: see ``set_gc_policy``.  Re-entrant (only the outermost scope acts)."""
    __slots__ = ("acted",)

    def __enter__(self):
        global _gc_policy, _gc_depth
        import gc
        self.acted = False
        if _gc_policy is None:
            env = os.environ.get("LVDGS_GC_FREEZE", "scoped")
            _gc_policy = {"0": "off", "1": "process"}.get(env, "scoped")
            if _gc_policy == "scoped" and gc.get_freeze_count() > 0:   # (walks the permanent generation: asked once per process)
                _gc_policy = "off"   # the host application manages a freeze of its own: hands off
            if _gc_policy == "process":
                gc.collect()
                gc.freeze()
        _gc_depth += 1
        if _gc_policy == "scoped" and _gc_depth == 1:
            gc.freeze()
            self.acted = True
        return self

    def __exit__(self, *exc):
        global _gc_depth
        _gc_depth -= 1
        if self.acted:
            import gc
            gc.unfreeze()
        return False


def check(status, what):
    if status != OK:
        raise LvdgsError(f"{what} failed ({status}): {lib().lvdgs_last_error().decode()}")


def profile_enable(on=True):
    lib().lvdgs_profile_enable(1 if on else 0)


def profile_reset():
    lib().lvdgs_profile_reset()


def profile_read():
    """{kernel name: (launches, total_ms)} measured with HIP events on the launch stream."""
    buf = (KernelTime * 64)()
    n = lib().lvdgs_profile_read(buf, 64)
    return {buf[i].name.decode(): (int(buf[i].launches), float(buf[i].total_ms)) for i in range(n)}
