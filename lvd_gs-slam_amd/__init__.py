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
  window_shard       mapping-window keyframe sharding over ranks (RCCL)

Nothing here falls back to a CPU implementation: the HIP library
(``lib/liblvdgs.so``) must be present for any render call.
"""
__version__ = "0.1.0"
