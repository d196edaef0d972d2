"""lvdgs: MI355X-native differentiable 3D-Gaussian rasterizer + pose/map optimiser
step for the LVD-GS tracking / mapping loops.

Import as ``lvdgs`` (see ``/lvdgs.py``).  Submodules:

  rasterizer         GaussianRasterizationSettings / GaussianRasterizer (autograd boundary)
  gaussian_renderer  render() / render_with_custom_resolution() facade
  graphics_utils     getProjectionMatrix2 / getWorld2View2 / focal2fov
  camera_utils       Camera
  pose_utils         SE(3) retraction (update_pose)
  slam_utils         tracking / mapping losses
  synthetic          seeded benchmark scenes
  backend_map        the back end's mapping iteration, its views sharded over ranks (two RCCL collectives per iteration)
  slam_loops         map initialisation and per-frame pose tracking loops
  loss_utils         fused L1 + SSIM and masked depth losses; fused_loss: photometric losses
  gaussian_model     the Gaussian map (parameters, Adam groups, seeding, densify / prune)

Nothing here falls back to a CPU implementation: the HIP library
(``lib/liblvdgs.so``) must be present for any render call.
"""
__version__ = "0.1.0"
