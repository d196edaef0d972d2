"""``render(viewpoint, gaussians, pipeline_params, background)`` on the CPU with the dense float64 autograd
formulation of tests/ref_torch.py (test infrastructure; toy sizes only).

Same call signature and the same seven-key dict as ``gaussian_splatting.gaussian_renderer.render`` (reference call
sites utils/slam_backend.py:98,184,277; keys :110-116), so it can stand in for the HIP renderer wherever only the loop
AROUND the renderer is under test: inside the reference's own ``BackEnd.map`` when the loop fixtures are generated
(tests/golden/make_loop_golden.py), inside ``lvdgs.backend_map.map_window`` when they are replayed on the CPU, and in
the world-size-2 gloo tests.  Pose gradients come from autograd through ``SE3_exp(tau) @ [R|T]``.
"""
import math

import torch

import ref_torch


def dense_render(viewpoint, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None, mask=None):
    f64 = lambda t: t.double()
    xyz = pc.get_xyz
    N = xyz.shape[0]
    if N == 0:
        return None
    H, W = int(viewpoint.image_height), int(viewpoint.image_width)
    if viewpoint.cam_trans_delta is None:   # a frame the front end has cleaned (Camera.clean): rendered for evaluation only
        tau = torch.zeros(6, dtype=torch.float64)
    else:
        tau = torch.cat([viewpoint.cam_trans_delta, viewpoint.cam_rot_delta]).double()
    view, proj, campos = ref_torch.camera_matrices(f64(viewpoint.R), f64(viewpoint.T), tau, f64(viewpoint.projection_matrix))
    screenspace_points = torch.zeros(N, 3, dtype=xyz.dtype, requires_grad=True)
    out = ref_torch.render_dense(
        f64(xyz), f64(pc.get_opacity), H, W, math.tan(viewpoint.FoVx * 0.5), math.tan(viewpoint.FoVy * 0.5), f64(bg_color),
        view, proj, campos, scales=f64(pc.get_scaling), rotations=f64(pc.get_rotation), shs=f64(pc.get_features),
        sh_degree=pc.active_sh_degree, scale_modifier=scaling_modifier, ndc_offset=screenspace_points)
    radii = out["radii"].to(torch.int32)
    return {"render": out["color"].float(), "viewspace_points": screenspace_points, "visibility_filter": radii > 0,
            "radii": radii, "depth": out["depth"].float(), "opacity": out["opacity"].float(),
            "n_touched": out["n_touched"].to(torch.int32)}
