"""The slice of the `gaussian_splatting` package the hot path's callers import
(utils/slam_backend.py:10-12, utils/slam_frontend.py:28-29, utils/camera_utils.py:4)."""
