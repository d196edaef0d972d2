"""The import names the reference's files use resolve to the lvdgs implementation, and the reference's
own utils/camera_utils.py (which needs gaussian_splatting.utils.graphics_utils) becomes importable."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DROPIN = os.path.join(ROOT, "lvd_gs-slam_amd", "dropin")


def test_shim_names_resolve(monkeypatch):
    monkeypatch.syspath_prepend(DROPIN)
    import lvdgs.rasterizer as R
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from gaussian_splatting.gaussian_renderer import render, render_with_custom_resolution
    from gaussian_splatting.utils.graphics_utils import getProjectionMatrix2, getWorld2View2
    from simple_knn._C import distCUDA2
    assert GaussianRasterizer is R.GaussianRasterizer and GaussianRasterizationSettings is R.GaussianRasterizationSettings
    assert callable(render) and callable(render_with_custom_resolution) and callable(distCUDA2)
    assert GaussianRasterizationSettings._fields == (
        "image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix",
        "projmatrix_raw", "sh_degree", "campos", "prefiltered", "debug")
    P = getProjectionMatrix2(znear=0.01, zfar=100.0, fx=500.0, fy=500.0, cx=320.0, cy=240.0, W=640, H=480)
    assert P.shape == (4, 4) and float(P[3, 2]) == 1.0
    assert getWorld2View2.__module__ == "lvdgs.graphics_utils"


@pytest.mark.skipif(not os.path.isdir("/root/reference/utils"), reason="reference checkout not present on this box")
def test_reference_camera_imports_through_the_shim(monkeypatch):
    """utils/camera_utils.py:4 imports gaussian_splatting.utils.graphics_utils; with the shim on the path the
    reference's own Camera class loads and produces the same matrices as the mirror."""
    import importlib
    import torch
    monkeypatch.syspath_prepend(DROPIN)
    monkeypatch.syspath_prepend("/root/reference")
    monkeypatch.setattr(sys, "dont_write_bytecode", True)
    for m in [k for k in sys.modules if k == "utils" or k.startswith("utils.")]:
        monkeypatch.delitem(sys.modules, m)
    ref_cam = importlib.import_module("utils.camera_utils")
    from lvdgs.camera_utils import Camera
    from lvdgs.graphics_utils import getProjectionMatrix2
    from lvdgs.pose_utils import SE3_exp
    proj = getProjectionMatrix2(znear=0.01, zfar=100.0, fx=700.0, fy=710.0, cx=600.0, cy=180.0, W=1226, H=370).transpose(0, 1)
    args = (0, torch.rand(3, 8, 8), None, None, torch.eye(4), proj, 700.0, 710.0, 600.0, 180.0, 1.4, 0.5, 370, 1226)
    a, b = ref_cam.Camera(*args, device="cpu"), Camera(*args, device="cpu")
    T = SE3_exp(torch.tensor([0.1, -0.2, 0.3, 0.05, 0.02, -0.04]))
    a.update_RT(T[:3, :3], T[:3, 3])
    b.update_RT(T[:3, :3], T[:3, 3])
    for name in ("world_view_transform", "full_proj_transform", "camera_center"):
        assert torch.allclose(getattr(a, name), getattr(b, name), atol=1e-6), name
    assert sorted(n for n, _ in a.named_parameters()) == sorted(n for n, _ in b.named_parameters())
    for m in [k for k in sys.modules if k == "utils" or k.startswith("utils.")]:
        monkeypatch.delitem(sys.modules, m)


@pytest.mark.skipif(not os.path.isdir("/root/reference/utils"), reason="reference checkout not present on this box")
def test_reference_backend_constructs_and_resets_on_the_shim(monkeypatch):
    """utils/slam_backend.py imports render / l1_loss / ssim / getProjectionMatrix2 from gaussian_splatting (:10-12);
    with the shim on the path the reference's own BackEnd class loads, takes the KITTI-07 hyper-parameters and
    drives GaussianModel.prune_points through BackEnd.reset (:79-92).  utils/init_pose.py (MASt3R + cv2, out of
    scope and not installed) is replaced by an empty module for the import."""
    import importlib
    import json
    import types
    import torch
    monkeypatch.syspath_prepend(DROPIN)
    monkeypatch.syspath_prepend("/root/reference")
    monkeypatch.setattr(sys, "dont_write_bytecode", True)
    for m in [k for k in sys.modules if k == "utils" or k.startswith("utils.")]:
        monkeypatch.delitem(sys.modules, m)
    stub = types.ModuleType("utils.init_pose")
    stub.save_depth_comparison = lambda *a, **k: None
    monkeypatch.setitem(sys.modules, "utils.init_pose", stub)
    backend = importlib.import_module("utils.slam_backend")
    import lvdgs.gaussian_renderer
    import lvdgs.loss_utils
    assert backend.render is lvdgs.gaussian_renderer.render
    assert backend.ssim is lvdgs.loss_utils.ssim and backend.l1_loss is lvdgs.loss_utils.l1_loss
    cfg = json.load(open(os.path.join(ROOT, "tests", "golden", "config_07.json")))
    cfg["Training"]["monocular"] = cfg["Dataset"]["sensor_type"] == "monocular"  # set by the absent slam.py entry point
    be = backend.BackEnd(cfg)
    be.cameras_extent = 6.0
    be.set_hyperparams()
    assert be.window_size == 8 and be.gaussian_extent == 6.0 and be.init_gaussian_extent == 180.0

    from gaussian_splatting.scene.gaussian_model import GaussianModel
    g = GaussianModel.from_activated(torch.rand(9, 3), torch.rand(9, 3) + 0.1, torch.nn.functional.normalize(torch.randn(9, 4)),
                                     torch.rand(9, 1), colors=torch.rand(9, 3), device="cpu")
    g.init_lr(6.0)
    g.training_setup(cfg["opt_params"])
    be.gaussians = g

    class _Q:
        def empty(self):
            return True
    be.backend_queue = _Q()
    be.reset()
    assert g.get_xyz.shape[0] == 0 and be.iteration_count == 0 and be.current_window == []
    for m in [k for k in sys.modules if k == "utils" or k.startswith("utils.")]:
        monkeypatch.delitem(sys.modules, m, raising=False)
