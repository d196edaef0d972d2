"""The C ABI driven directly with ctypes, the way INTEGRATION.md's binding does: the two-call forward
(lvdgs_forward_prepare + lvdgs_forward_render) and lvdgs_backward give bit-identical results to the autograd
path (single-call lvdgs_forward)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def test_two_call_forward_and_backward_through_ctypes():
    import hip_runner
    from lvdgs import _lib, synthetic
    L = _lib.lib()
    N, W, H = 3000, 200, 120
    g = synthetic.make_gaussians(N, W, H, seed=3)
    cam = synthetic.make_camera(W, H, pose_seed=4)
    grads = synthetic.make_image_grads(W, H, 5)
    bg = torch.tensor([0.3, 0.2, 0.1])
    f_ref, b_ref = hip_runner.run_hip(g, cam, W, H, bg, grads=grads)

    dev = torch.device("cuda")
    t = {k: v.to(dev).contiguous() for k, v in g.items()}
    mats = {k: getattr(cam, k).to(dev).contiguous() for k in ("world_view_transform", "full_proj_transform", "projection_matrix",
                                                              "camera_center")}
    bgd = bg.to(dev)
    buf = lambda n: torch.empty(max(int(n), 256), dtype=torch.uint8, device=dev)
    a = _lib.Args()
    a.image_height, a.image_width, a.tanfovx, a.tanfovy = H, W, cam.tanfovx, cam.tanfovy
    a.scale_modifier, a.sh_degree = 1.0, 0
    a.bg, a.viewmatrix, a.projmatrix = _p(bgd), _p(mats["world_view_transform"]), _p(mats["full_proj_transform"])
    a.projmatrix_raw, a.campos = _p(mats["projection_matrix"]), _p(mats["camera_center"])
    a.num_gaussians, a.sh_coeffs = N, 0
    a.means3D, a.opacities, a.scales, a.rotations = _p(t["means3D"]), _p(t["opacities"]), _p(t["scales"]), _p(t["rotations"])
    a.colors_precomp = _p(t["colors"])
    radii = torch.empty(N, dtype=torch.int32, device=dev)
    n_touched = torch.empty(N, dtype=torch.int32, device=dev)
    color, depth, opacity = (torch.empty(c, H, W, device=dev) for c in (3, 1, 1))
    geom, image = buf(L.lvdgs_geom_bytes(N)), buf(L.lvdgs_image_bytes(W, H))
    scratch = buf(L.lvdgs_prepare_scratch_bytes(N))
    a.radii, a.n_touched, a.out_color, a.out_depth, a.out_opacity = _p(radii), _p(n_touched), _p(color), _p(depth), _p(opacity)
    a.geom_state, a.geom_bytes, a.image_state, a.image_bytes = _p(geom), geom.numel(), _p(image), image.numel()
    a.scratch, a.scratch_bytes = _p(scratch), scratch.numel()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    D = C.c_int64()
    _lib.check(L.lvdgs_forward_prepare(C.byref(a), C.byref(D), stream), "prepare")
    assert D.value == f_ref["num_rendered"]
    binning, scratch2 = buf(L.lvdgs_binning_bytes(D.value)), buf(L.lvdgs_render_scratch_bytes(N, D.value, W, H))
    a.num_rendered, a.binning_state, a.binning_bytes = D.value, _p(binning), binning.numel()
    a.scratch, a.scratch_bytes = _p(scratch2), scratch2.numel()
    _lib.check(L.lvdgs_forward_render(C.byref(a), stream), "render")
    torch.cuda.synchronize()
    np.testing.assert_array_equal(color.cpu().numpy(), f_ref["color"])
    np.testing.assert_array_equal(depth.cpu().numpy(), f_ref["depth"])
    np.testing.assert_array_equal(opacity.cpu().numpy(), f_ref["opacity"])
    np.testing.assert_array_equal(radii.cpu().numpy(), f_ref["radii"])
    np.testing.assert_array_equal(n_touched.cpu().numpy(), f_ref["n_touched"])

    gc, gd, go = (x.to(dev).contiguous() for x in grads)
    scratch3 = buf(L.lvdgs_backward_scratch_bytes(N, D.value))
    a.scratch, a.scratch_bytes = _p(scratch3), scratch3.numel()
    a.dL_dout_color, a.dL_dout_depth, a.dL_dout_opacity = _p(gc), _p(gd), _p(go)
    e = lambda *s: torch.empty(*s, device=dev)
    out = dict(means3D=e(N, 3), means2D=e(N, 3), opacities=e(N, 1), scales=e(N, 3), rotations=e(N, 4), colors=e(N, 3), tau=e(6))
    a.dL_dmeans3D, a.dL_dmeans2D, a.dL_dopacities = _p(out["means3D"]), _p(out["means2D"]), _p(out["opacities"])
    a.dL_dscales, a.dL_drotations, a.dL_dcolors, a.dL_dtau = _p(out["scales"]), _p(out["rotations"]), _p(out["colors"]), _p(out["tau"])
    _lib.check(L.lvdgs_backward(C.byref(a), stream), "backward")
    torch.cuda.synchronize()
    for k in ("means3D", "means2D", "opacities", "scales", "rotations", "colors"):
        np.testing.assert_array_equal(out[k].cpu().numpy().reshape(b_ref[k].shape), b_ref[k], err_msg=k)
    np.testing.assert_array_equal(out["tau"].cpu().numpy(), b_ref["tau"].reshape(-1))
    assert b"lvdgs" in L.lvdgs_version()

    # LVDGS_FLAG_ACCUMULATE_PARAM_GRADS: the same backward ADDS the parameter gradients to what the buffers hold (a later
    # view of a mapping iteration) -- one addition per element, so exactly x + g -- and writes dL_dmeans2D / dL_dtau as always
    gen = torch.Generator().manual_seed(9)
    start = {k: torch.randn(out[k].shape, generator=gen).to(dev) for k in out}
    acc = {k: start[k].clone() for k in out}
    a.dL_dmeans3D, a.dL_dmeans2D, a.dL_dopacities = _p(acc["means3D"]), _p(acc["means2D"]), _p(acc["opacities"])
    a.dL_dscales, a.dL_drotations, a.dL_dcolors, a.dL_dtau = _p(acc["scales"]), _p(acc["rotations"]), _p(acc["colors"]), _p(acc["tau"])
    a.flags = _lib.FLAG_ACCUMULATE_PARAM_GRADS
    _lib.check(L.lvdgs_backward(C.byref(a), stream), "backward (accumulating)")
    torch.cuda.synchronize()
    for k in ("means3D", "opacities", "scales", "rotations", "colors"):
        assert torch.equal(acc[k], start[k] + out[k]), k
    assert torch.equal(acc["means2D"], out["means2D"]) and torch.equal(acc["tau"], out["tau"])

    # LVDGS_FLAG_POSE_ONLY: the pose gradient alone -- bit for bit the full backward's -- with every other gradient pointer NULL,
    # and nothing written through the ones that are given
    tau_only = torch.full((6,), float("nan"), device=dev)
    a.flags = _lib.FLAG_POSE_ONLY
    a.dL_dmeans3D = a.dL_dmeans2D = a.dL_dopacities = a.dL_dscales = a.dL_drotations = a.dL_dcolors = None
    a.dL_dtau = _p(tau_only)
    _lib.check(L.lvdgs_backward(C.byref(a), stream), "backward (pose only)")
    torch.cuda.synchronize()
    assert torch.equal(tau_only, out["tau"])
    sentinel = {k: torch.full(out[k].shape, 7.0, device=dev) for k in out if k != "tau"}
    a.dL_dmeans3D, a.dL_dmeans2D, a.dL_dopacities = _p(sentinel["means3D"]), _p(sentinel["means2D"]), _p(sentinel["opacities"])
    a.dL_dscales, a.dL_drotations, a.dL_dcolors = _p(sentinel["scales"]), _p(sentinel["rotations"]), _p(sentinel["colors"])
    a.dL_dout_depth = a.dL_dout_opacity = None     # (the form without a depth gradient: another kernel instantiation)
    _lib.check(L.lvdgs_backward(C.byref(a), stream), "backward (pose only, colour gradient alone)")
    a.flags = 0
    a.dL_dtau = _p(acc["tau"])
    _lib.check(L.lvdgs_backward(C.byref(a), stream), "backward (colour gradient alone)")
    torch.cuda.synchronize()
    assert torch.equal(tau_only, acc["tau"]) and bool(tau_only.any())
    # (the full call has just written the sentinels' buffers: pose-only before it must not have)
    a.flags = _lib.FLAG_POSE_ONLY
    for v in sentinel.values():
        v.fill_(7.0)
    _lib.check(L.lvdgs_backward(C.byref(a), stream), "backward (pose only)")
    torch.cuda.synchronize()
    assert all(bool((v == 7.0).all()) for v in sentinel.values())
    # a view-dependent colour feeds the pose gradient through the colour gradient: refused, with a message
    shs = torch.zeros(N, 4, 3, device=dev)
    a.colors_precomp, a.shs, a.sh_coeffs, a.sh_degree = None, _p(shs), 4, 1
    assert L.lvdgs_backward(C.byref(a), stream) == _lib.E_INVALID and b"POSE_ONLY" in L.lvdgs_last_error()
    # the plain backward has no batched blend pass to leave its own to
    a.flags = _lib.FLAG_NO_BLEND
    assert L.lvdgs_backward(C.byref(a), stream) == _lib.E_INVALID and b"NO_BLEND" in L.lvdgs_last_error()


def test_errors_are_reported_not_thrown():
    from lvdgs import _lib
    L = _lib.lib()
    a = _lib.Args()
    a.image_height, a.image_width, a.num_gaussians = 64, 64, 10
    D = C.c_int64()
    assert L.lvdgs_forward(C.byref(a), C.byref(D), None) == _lib.E_INVALID
    assert len(L.lvdgs_last_error()) > 0
    assert L.lvdgs_backward(C.byref(a), None) == _lib.E_INVALID


def test_forward_batch_equals_forward_view_by_view_state_and_images():
    """lvdgs_forward_batch (every stage of several views of one map in one launch) against lvdgs_forward per view: pair counts, the
    whole geometry / binning / image state and every output image, bit for bit -- with and without LVDGS_FLAG_NO_BLEND, with a view
    whose pair capacity is too small (LVDGS_E_CAPACITY for that view alone, the others complete), and the argument checks."""
    from lvdgs import _lib, synthetic
    L = _lib.lib()
    dev = torch.device("cuda")
    N, W, H, V = 20000, 320, 200, 3
    g = synthetic.make_gaussians(N, W, H, seed=11)
    t = {k: v.to(dev).contiguous() for k, v in g.items()}
    bgd = torch.tensor([0.1, 0.2, 0.3], device=dev)
    buf = lambda n: torch.zeros(max(int(n), 256), dtype=torch.uint8, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    keep = []

    def view(k, cap):
        cam = synthetic.make_camera(W, H, pose_seed=20 + k)
        mats = {n: getattr(cam, n).to(dev).contiguous() for n in ("world_view_transform", "full_proj_transform", "projection_matrix", "camera_center")}
        a = _lib.Args()
        a.image_height, a.image_width, a.tanfovx, a.tanfovy = H, W, cam.tanfovx, cam.tanfovy
        a.scale_modifier, a.sh_degree = 1.0, 0
        a.bg, a.viewmatrix, a.projmatrix = _p(bgd), _p(mats["world_view_transform"]), _p(mats["full_proj_transform"])
        a.projmatrix_raw, a.campos = _p(mats["projection_matrix"]), _p(mats["camera_center"])
        a.num_gaussians, a.sh_coeffs = N, 0
        a.means3D, a.opacities, a.scales, a.rotations, a.colors_precomp = _p(t["means3D"]), _p(t["opacities"]), _p(t["scales"]), _p(t["rotations"]), _p(t["colors"])
        st = dict(radii=torch.zeros(N, dtype=torch.int32, device=dev), n_touched=torch.zeros(N, dtype=torch.int32, device=dev),
                  color=torch.zeros(3, H, W, device=dev), depth=torch.zeros(1, H, W, device=dev), opacity=torch.zeros(1, H, W, device=dev),
                  geom=buf(L.lvdgs_geom_bytes(N)), image=buf(L.lvdgs_image_bytes(W, H)), binning=buf(L.lvdgs_binning_bytes(cap)),
                  scratch=buf(max(L.lvdgs_prepare_scratch_bytes(N), L.lvdgs_render_scratch_bytes(N, cap, W, H))))
        a.radii, a.n_touched, a.out_color, a.out_depth, a.out_opacity = _p(st["radii"]), _p(st["n_touched"]), _p(st["color"]), _p(st["depth"]), _p(st["opacity"])
        a.geom_state, a.geom_bytes, a.image_state, a.image_bytes = _p(st["geom"]), st["geom"].numel(), _p(st["image"]), st["image"].numel()
        a.binning_state, a.binning_bytes, a.scratch, a.scratch_bytes = _p(st["binning"]), st["binning"].numel(), _p(st["scratch"]), st["scratch"].numel()
        a.pair_capacity = cap
        keep.append((mats, st))
        return a, st

    cap = 400_000
    single = [view(k, cap) for k in range(V)]
    counts = []
    for a, st in single:
        n = C.c_int64()
        _lib.check(L.lvdgs_forward(C.byref(a), C.byref(n), stream), "lvdgs_forward")
        counts.append(n.value)
    torch.cuda.synchronize()
    assert min(counts) > 10_000 and len(set(counts)) == V   # three different views
    lay = _lib.StateLayout()

    def compare(st, ref, D, images=True):
        L.lvdgs_state_layout_query(N, cap, W, H, C.byref(lay))
        assert torch.equal(st["radii"], ref["radii"]) and torch.equal(st["geom"], ref["geom"])
        assert torch.equal(st["binning"][lay.bin_point_list:lay.bin_point_list + 4 * D], ref["binning"][lay.bin_point_list:lay.bin_point_list + 4 * D])
        T = ((W + 15) // 16) * ((H + 15) // 16)
        assert torch.equal(st["image"][lay.img_ranges:lay.img_ranges + 8 * T], ref["image"][lay.img_ranges:lay.img_ranges + 8 * T])
        if images:
            assert torch.equal(st["image"], ref["image"])
            for n_ in ("color", "depth", "opacity", "n_touched"):
                assert torch.equal(st[n_], ref[n_]), n_

    for no_blend in (False, True):
        batch = [view(k, cap) for k in range(V)]
        for a, _ in batch:
            a.flags = _lib.FLAG_NO_BLEND if no_blend else 0
        arr = (C.POINTER(_lib.Args) * V)(*[C.pointer(a) for a, _ in batch])
        nums = (C.c_int64 * V)()
        _lib.check(L.lvdgs_forward_batch(arr, V, nums, stream), "lvdgs_forward_batch")
        assert list(nums) == counts
        if no_blend:
            for k, (a, _) in enumerate(batch):
                a.num_rendered = nums[k]
            _lib.check(L.lvdgs_blend_forward_batch(arr, V, stream), "lvdgs_blend_forward_batch")
        torch.cuda.synchronize()
        for k in range(V):
            compare(batch[k][1], single[k][1], counts[k])

    # more views than one launch carries (ten): the call goes through them in groups, every group with its own sequence number
    many = [view(k % V, cap) for k in range(12)]
    for k, (a, _) in enumerate(many):   # (views k and k + 3 share a camera: the same counts and lists)
        a.viewmatrix, a.projmatrix, a.campos = single[k % V][0].viewmatrix, single[k % V][0].projmatrix, single[k % V][0].campos
    arr12 = (C.POINTER(_lib.Args) * 12)(*[C.pointer(a) for a, _ in many])
    nums12 = (C.c_int64 * 12)()
    _lib.check(L.lvdgs_forward_batch(arr12, 12, nums12, stream), "lvdgs_forward_batch (12 views)")
    torch.cuda.synchronize()
    assert list(nums12) == [counts[k % V] for k in range(12)]
    for k in (0, 9, 10, 11):
        compare(many[k][1], single[k % V][1], counts[k % V])

    # a view with too small a capacity: E_CAPACITY, its count reported, the other views complete
    small = counts[1] // 2
    batch = [view(k, small if k == 1 else cap) for k in range(V)]
    arr = (C.POINTER(_lib.Args) * V)(*[C.pointer(a) for a, _ in batch])
    nums = (C.c_int64 * V)()
    assert L.lvdgs_forward_batch(arr, V, nums, stream) == _lib.E_CAPACITY and b"view 1" in L.lvdgs_last_error()
    assert list(nums) == counts
    torch.cuda.synchronize()
    compare(batch[0][1], single[0][1], counts[0]); compare(batch[2][1], single[2][1], counts[2])
    a1, st1 = batch[1]
    st1["binning"], st1["scratch"] = buf(L.lvdgs_binning_bytes(counts[1])), buf(L.lvdgs_render_scratch_bytes(N, counts[1], W, H))
    a1.binning_state, a1.binning_bytes, a1.scratch, a1.scratch_bytes = _p(st1["binning"]), st1["binning"].numel(), _p(st1["scratch"]), st1["scratch"].numel()
    a1.num_rendered = counts[1]
    _lib.check(L.lvdgs_forward_render(C.byref(a1), stream), "lvdgs_forward_render")
    torch.cuda.synchronize()
    for n_ in ("color", "depth", "opacity", "n_touched", "radii"):
        assert torch.equal(st1[n_], single[1][1][n_]), n_

    # argument checks: views of different image sizes, a NULL view, no count array
    bad, _ = view(0, cap)
    bad.image_width = W - 16
    arr2 = (C.POINTER(_lib.Args) * 2)(C.pointer(batch[0][0]), C.pointer(bad))
    assert L.lvdgs_forward_batch(arr2, 2, nums, stream) == _lib.E_INVALID and b"differ" in L.lvdgs_last_error()
    assert L.lvdgs_forward_batch(arr, V, None, stream) == _lib.E_INVALID
    assert L.lvdgs_forward_batch(arr, 0, nums, stream) == _lib.OK
