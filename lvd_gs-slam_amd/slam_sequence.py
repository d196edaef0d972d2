"""The front end's and the back end's loops run as ONE sequence, in one process: ``SlamSequence``.

The reference runs ``FrontEnd.run`` (utils/slam_frontend.py:1740-1899) and ``BackEnd.run`` (utils/slam_backend.py:485-609) as two
OS processes that exchange messages over queues ("init", "keyframe", "sync_backend", "color_refinement").  The queues, the GUI and
the MASt3R / GroundingDINO models are out of scope; what IS the path is the order in which the loops
of this package are called on one another's results as a sequence advances, with the map growing and shrinking under them:

    frame 0      FrontEnd.initialize (:1702-1738)      pose := ground truth, ``add_new_keyframe(init=True)`` -> depth map
                 BackEnd "init" (:514-528)             reset, ``extend_from_pcd_seq(init=True)``, ``initialize_map``
    frame i      FrontEnd.tracking (:1416-1536)        pose := previous frame's (the "no estimate" branch, :1460-1462), ``track_frame``
                 keyframe test (:1822-1846)            ``is_keyframe`` / the covisibility rule while the window fills
      keyframe:  ``add_to_window`` (:1621-1674), ``add_new_keyframe`` (:1268-1414) -> depth map
                 BackEnd "keyframe" (:530-601)         ``extend_from_pcd_seq``, a fresh keyframe Adam, ``map(iters)``, ``map(prune=True)``
      else:      ``cleanup`` (:1697-1700)
      idle:      BackEnd's free-running branch (:487-499): ``map(window)``; every 10 iterations ``map(prune=True, iters=10)`` + push
    end          ``eval_ate`` (:1773-1776), ``eval_rendering`` before / after ``color_refinement`` (utils/eval_utils_0806.py:172-437)

This class is that order, statement for statement, as method calls: the "messages" are direct calls; ``push_to_frontend`` /
``sync_backend`` hand the front end a detached COPY of the map (``clone_obj``, utils/multiprocessing_utils.py:21-31) together with the
visibility rows of the window -- the front end tracks against the map as of the last push, which is what keeps its per-Gaussian
visibility vectors the size of the map it renders while the back end densifies and prunes the live one.  It is the schedule of the reference's
``single_thread`` mode (the front end waits for every keyframe's mapping) plus a FIXED number of the back end's free-running
iterations per tracked frame (``idle_map_iters``; in the reference that number is whatever the wall clock allows).

Stand-ins for what is out of scope, all injectable:
  * the pose initialisation: the previous frame's pose -- the reference's own branch for "MASt3R returned the identity";
  * ``keyframe_depth``: the depth map a new keyframe seeds Gaussians from.  Default: the keyframe's mono depth under the valid-pixel
    mask, i.e. ``add_new_keyframe``'s own statements for the first keyframe (:1364-1382) applied to every keyframe -- for later
    keyframes the reference blends rendered and MASt3R depth patch by patch (``process_depth``, utils/depth_utils.py, needs MASt3R);
  * ``dataset.static_mask(idx)`` for the GroundingDINO + SAM masks (``dynamic_masker.get_static_mask_for_gaussian_init``).

``render_fn`` / ``view_loss_fn`` / ``refine_loss_fn`` / ``knn_fn`` default to the HIP paths; the CPU tests pass the dense float64
renderer and the loss oracle so that the same harness runs at toy size without a GPU.
"""
import time
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn.functional as F

from . import _lib
from .gaussian_renderer import render
from .graphics_utils import getProjectionMatrix2
from .keyframe_utils import add_to_window, covisibility, is_keyframe
from .slam_loops import color_refinement, initialize_map, track_frame


def expand_dynamic_mask(dynamic_mask, kernel_size=5):
    """``FrontEnd._expand_dynamic_mask`` (utils/slam_frontend.py:1260-1266): dilation by a ``kernel_size`` square of ones
    (cv2.dilate there; a max-pool with the border left out of the maximum here -- the same set)."""
    m = dynamic_mask.to(torch.float32)[None, None]
    return F.max_pool2d(m, kernel_size, stride=1, padding=kernel_size // 2)[0, 0] > 0.5


def clone_map(gaussians):
    """``clone_obj(self.gaussians)`` of ``push_to_frontend`` (utils/slam_backend.py:478; utils/multiprocessing_utils.py:21-31): the same
    object with every tensor attribute a detached copy (no gradients flow into it, no optimiser comes with it)."""
    import copy
    snap = copy.copy(gaussians)
    for name, value in vars(gaussians).items():
        if torch.is_tensor(value):
            setattr(snap, name, value.detach().clone())
    snap.optimizer = None
    return snap


def make_backend(config, gaussians, pipeline_params, background, cameras_extent=6.0):
    """A BackEnd-shaped object: the attributes ``BackEnd.__init__`` / ``set_hyperparams`` set (utils/slam_backend.py:21-72)."""
    T = config["Training"]
    opt = config["opt_params"]
    return SimpleNamespace(
        config=config, gaussians=gaussians, pipeline_params=pipeline_params, background=background,
        opt_params=SimpleNamespace(**opt) if isinstance(opt, dict) else opt, cameras_extent=cameras_extent, live_mode=False,
        monocular=T["monocular"], iteration_count=0, last_sent=0, occ_aware_visibility={}, viewpoints={}, current_window=[],
        initialized=not T["monocular"], keyframe_optimizers=None, theta=0,
        init_itr_num=T["init_itr_num"], init_gaussian_update=T["init_gaussian_update"], init_gaussian_reset=T["init_gaussian_reset"],
        init_gaussian_th=T["init_gaussian_th"], init_gaussian_extent=cameras_extent * T["init_gaussian_extent"],
        mapping_itr_num=T["mapping_itr_num"], gaussian_update_every=T["gaussian_update_every"],
        gaussian_update_offset=T["gaussian_update_offset"], gaussian_th=T["gaussian_th"],
        gaussian_extent=cameras_extent * T["gaussian_extent"], gaussian_reset=T["gaussian_reset"], size_threshold=T["size_threshold"],
        window_size=T["window_size"], single_thread=config["Dataset"].get("single_thread", False))


def make_keyframe_optimizer(config, viewpoints, current_window, frames_to_optimize):
    """The Adam the back end builds when a keyframe arrives (utils/slam_backend.py:545-598): pose deltas of the newest
    ``frames_to_optimize`` window keyframes at half the tracking learning rates, exposure of every window keyframe; frame 0 fixed."""
    lr = config["Training"]["lr"]
    groups = []
    for cam_idx, kf in enumerate(current_window):
        if kf == 0:
            continue
        vp = viewpoints[kf]
        if cam_idx < frames_to_optimize:
            groups.append({"params": [vp.cam_rot_delta], "lr": lr["cam_rot_delta"] * 0.5, "name": "rot_{}".format(vp.uid)})
            groups.append({"params": [vp.cam_trans_delta], "lr": lr["cam_trans_delta"] * 0.5, "name": "trans_{}".format(vp.uid)})
        groups.append({"params": [vp.exposure_a], "lr": 0.01, "name": "exposure_a_{}".format(vp.uid)})
        groups.append({"params": [vp.exposure_b], "lr": 0.01, "name": "exposure_b_{}".format(vp.uid)})
    return torch.optim.Adam(groups) if groups else None


class SlamSequence:
    """See the module's docstring.  ``run()`` processes the whole dataset and returns the record ``summary()`` builds."""

    def __init__(self, config, dataset, gaussians, pipeline_params, background, *, fused="auto", render_fn=render, view_loss_fn=None,
                 refine_loss_fn=None, keyframe_depth=None, idle_map_iters=0, camera_cls=None, cameras_extent=6.0, on_event=None,
                 group=None, aux_group=None, bands_ok=None):
        from .backend_map import map_window
        if camera_cls is None:
            from .camera_utils import Camera as camera_cls
        self.config, self.dataset, self.gaussians = config, dataset, gaussians
        self.pipeline_params, self.background = pipeline_params, background
        self.fused, self.render_fn, self.view_loss_fn, self.refine_loss_fn = fused, render_fn, view_loss_fn, refine_loss_fn
        self.keyframe_depth = keyframe_depth if keyframe_depth is not None else self.default_keyframe_depth
        self.idle_map_iters, self.camera_cls, self.on_event = int(idle_map_iters), camera_cls, on_event
        self._map_window = map_window
        # several ranks (torch.distributed initialised): every rank runs the whole sequence -- tracking, map initialisation and colour
        # refinement are one view per iteration: replicas -- and the mapping windows' views are sharded over
        # `group` (backend_map.map_window); the replicas stay bit-identical, so every rank ends with the same map and trajectory
        self.group, self.aux_group, self.bands_ok = group, aux_group, bands_ok
        T = config["Training"]
        self.monocular = T["monocular"]
        self.tracking_itr_num, self.kf_interval, self.window_size = T["tracking_itr_num"], T["kf_interval"], T["window_size"]
        self.single_thread = T.get("single_thread", False)
        df = config.get("dynamic_filtering", {})
        self.enable_dynamic_filtering = df.get("enabled", True) and hasattr(dataset, "static_mask")
        self.filter_initialization = df.get("filter_initialization", True)
        # front-end state (FrontEnd.__init__, utils/slam_frontend.py:1184-1213)
        self.initialized = False
        self.kf_indices, self.current_window, self.occ_aware_visibility, self.cameras = [], [], {}, {}
        self.median_depth, self.theta = None, 0
        self.frontend_gaussians = gaussians      # the front end's copy of the map as of the last push (``_sync_backend``)
        self.backend = make_backend(config, gaussians, pipeline_params, background, cameras_extent)
        self.projection_matrix = getProjectionMatrix2(znear=0.01, zfar=100.0, fx=dataset.fx, fy=dataset.fy, cx=dataset.cx, cy=dataset.cy,
                                                      W=dataset.width, H=dataset.height).transpose(0, 1).to(device=dataset.device)
        self.counts = dict(frames=0, keyframes=0, tracking_iterations=0, mapping_iterations=0, init_iterations=0, prune_passes=0,
                           refinement_iterations=0)
        self.gaussian_counts = []      # (event, N) whenever the map's size may have changed
        self.seconds = dict(tracking=0.0, mapping=0.0, init=0.0, seeding=0.0, refinement=0.0, other=0.0)
        self.window_log = []           # the window after every keyframe
        self.frame_log = []            # per tracked frame: tracking iterations, the keyframe test's inputs and its outcome
        self.batched_sizes = []        # the map's size at every mapping call that went through the batched window (MapWindowBatch)

    # ------------------------------------------------------------------ helpers
    def _n(self):
        return int(self.gaussians.get_xyz.shape[0])

    def _note(self, event):
        self.gaussian_counts.append((event, self._n()))
        if self.on_event is not None:
            self.on_event(event, self)

    def _sync(self):
        dev = self.gaussians.get_xyz.device
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)

    def _timed(self, key):
        seq = self

        class _T:
            def __enter__(self):
                seq._sync()
                self.t = time.perf_counter()

            def __exit__(self, *exc):
                seq._sync()
                seq.seconds[key] += time.perf_counter() - self.t
                return False
        return _T()

    def _map(self, window, **kw):
        be = self.backend
        it0, n0 = be.iteration_count, self._n()
        if self.view_loss_fn is not None:
            kw["view_loss_fn"] = self.view_loss_fn
        if self.group is not None or self.aux_group is not None or self.bands_ok is not None:
            kw.update(group=self.group, aux_group=self.aux_group, bands_ok=self.bands_ok)
        runs0 = getattr(getattr(be, "_lvdgs_window_batch", None), "runs", 0)
        out = self._map_window(be, window, render_fn=self.render_fn, fused=self.fused is not False, **kw)
        if getattr(getattr(be, "_lvdgs_window_batch", None), "runs", 0) > runs0:
            self.batched_sizes.append(n0)
        if kw.get("prune"):
            self.counts["prune_passes"] += 1
        else:
            self.counts["mapping_iterations"] += be.iteration_count - it0
        if self._n() != n0:
            self._note("prune" if kw.get("prune") else "densify")
        return out

    # ------------------------------------------------------------------ front end
    def new_viewpoint(self, idx):
        """``Camera.init_from_dataset`` + ``compute_grad_mask`` (utils/slam_frontend.py:1794-1798)."""
        vp = self.camera_cls.init_from_dataset(self.dataset, idx, self.projection_matrix)
        vp.compute_grad_mask(self.config)
        return vp

    def _attach_masks(self, viewpoint, idx, first):
        """What ``add_new_keyframe`` / ``tracking`` store on the viewpoint from the detector's mask (:1306-1329, :1419-1436)."""
        static = self.dataset.static_mask(idx)
        if static is None:
            return None
        dev = viewpoint.original_image.device
        static = static.to(dev).bool()
        viewpoint.static_mask, viewpoint.dynamic_mask = static, ~static
        return static

    def default_keyframe_depth(self, viewpoint, render_pkg, valid_rgb, init):
        """``add_new_keyframe``'s monocular statements for the first keyframe (:1364-1382): the mono depth, zero where invalid."""
        d = torch.from_numpy(np.asarray(viewpoint.mono_depth)).clone().unsqueeze(0)
        d[~valid_rgb.cpu()] = 0
        return d[0].numpy()

    def add_new_keyframe(self, cur_frame_idx, render_pkg=None, init=False):
        """utils/slam_frontend.py:1268-1414 without the detector and MASt3R: rotation to the last keyframe, the valid-pixel mask
        (non-black pixels minus the dilated dynamic mask: 9 x 9 on frame 0, 7 x 7 after), the depth map the seeds come from."""
        thr = self.config["Training"]["rgb_boundary_threshold"]
        viewpoint = self.cameras[cur_frame_idx]
        if len(self.kf_indices) > 0:
            R_last = self.cameras[self.kf_indices[-1]].R.to(torch.float32)
            R_diff = R_last.T @ viewpoint.R.to(torch.float32)
            self.theta = torch.rad2deg(torch.acos(((torch.trace(R_diff) - 1) / 2).clamp(-1.0, 1.0)))
        self.kf_indices.append(cur_frame_idx)
        gt_img = viewpoint.original_image
        valid_rgb = (gt_img.sum(dim=0) > thr)[None]
        if self.enable_dynamic_filtering and (not init or self.filter_initialization):
            static = self._attach_masks(viewpoint, cur_frame_idx, cur_frame_idx == 0)
            if static is not None:
                expanded_dynamic = expand_dynamic_mask(viewpoint.dynamic_mask, 9 if cur_frame_idx == 0 else 7)
                viewpoint.expanded_dynamic_mask, viewpoint.expanded_static_mask = expanded_dynamic, ~expanded_dynamic
                valid_rgb = valid_rgb & viewpoint.expanded_static_mask[None]
        if not self.monocular and getattr(viewpoint, "depth", None) is not None:   # (:1409-1414)
            d = torch.from_numpy(np.asarray(viewpoint.depth)).clone().unsqueeze(0)
            d[~valid_rgb.cpu()] = 0
            return d[0].numpy()
        return self.keyframe_depth(viewpoint, render_pkg, valid_rgb, init)

    def initialize(self, cur_frame_idx, viewpoint):
        """FrontEnd.initialize (:1702-1738) + the back end's "init" message (utils/slam_backend.py:514-528)."""
        self.initialized = not self.monocular
        self.kf_indices, self.occ_aware_visibility, self.current_window = [], {}, []
        viewpoint.update_RT(viewpoint.R_gt, viewpoint.T_gt)
        depth_map = self.add_new_keyframe(cur_frame_idx, init=True)
        be = self.backend
        # BackEnd.reset (:80-92)
        be.iteration_count, be.occ_aware_visibility, be.viewpoints, be.current_window = 0, {}, {}, []
        be.initialized, be.keyframe_optimizers = not be.monocular, None
        if self._n():
            self.gaussians.prune_points(self.gaussians.unique_kfIDs >= 0)
        be.viewpoints[cur_frame_idx] = viewpoint
        with self._timed("seeding"):
            self.gaussians.extend_from_pcd_seq(viewpoint, kf_id=cur_frame_idx, init=True, scale=2.0, depthmap=depth_map)
        self._note("seed")
        with self._timed("init"):
            initialize_map(be, cur_frame_idx, viewpoint, render_fn=self.render_fn, fused=self.fused)
        self.counts["init_iterations"] += be.init_itr_num
        self._note("initialize_map")
        self._sync_backend()
        self.current_window.append(cur_frame_idx)
        self.counts["keyframes"] += 1
        self.window_log.append(list(self.current_window))

    def _sync_backend(self):
        """``push_to_frontend`` + ``sync_backend`` (utils/slam_backend.py:470-480, utils/slam_frontend.py:1688-1695): a detached copy
        of the map and the window's visibility rows (the keyframes are shared objects here: their poses need no copying back)."""
        self.backend.last_sent = 0
        self.frontend_gaussians = clone_map(self.gaussians)
        self.occ_aware_visibility = dict(self.backend.occ_aware_visibility)

    def tracking(self, cur_frame_idx, viewpoint):
        """FrontEnd.tracking (:1416-1536) with the previous frame's pose as the initial estimate."""
        if self.enable_dynamic_filtering:
            self._attach_masks(viewpoint, cur_frame_idx, False)
        prev = self.cameras[cur_frame_idx - 1]
        viewpoint.update_RT(prev.R, prev.T)
        with self._timed("tracking"):
            render_pkg, median_depth, its = track_frame(viewpoint, self.frontend_gaussians, self.config, self.pipeline_params, self.background,
                                                        tracking_itr_num=self.tracking_itr_num, render_fn=self.render_fn, fused=self.fused)
        self.median_depth = median_depth
        self.counts["tracking_iterations"] += int(its)
        self._last_tracking_iterations = int(its)
        return render_pkg

    def handle_keyframe(self, cur_frame_idx, viewpoint, depth_map):
        """The back end's "keyframe" message (utils/slam_backend.py:530-601)."""
        be, cfg = self.backend, self.config
        be.viewpoints[cur_frame_idx] = viewpoint
        be.current_window = self.current_window
        be.theta = self.theta
        with self._timed("seeding"):
            self.gaussians.extend_from_pcd_seq(viewpoint, kf_id=cur_frame_idx, init=False, scale=2.0, depthmap=depth_map)
        self._note("seed")
        frames_to_optimize = cfg["Training"]["pose_window"]
        iter_per_kf = be.mapping_itr_num if be.single_thread else cfg["Training"]["mapping_itr_nosingle"]
        if not be.initialized:
            if len(self.current_window) == cfg["Training"]["window_size"]:
                frames_to_optimize = cfg["Training"]["window_size"] - 1
                iter_per_kf = 50 if be.live_mode else cfg["Training"].get("initial_ba_itr_num", 300)
            else:
                iter_per_kf = be.mapping_itr_num
        be.keyframe_optimizers = make_keyframe_optimizer(cfg, be.viewpoints, self.current_window, frames_to_optimize)
        with self._timed("mapping"):
            self._map(self.current_window, iters=iter_per_kf, up_pose=True)
            self._map(self.current_window, prune=True)
        self._sync_backend()

    def idle_mapping(self, iterations):
        """The back end between messages (utils/slam_backend.py:487-499)."""
        be = self.backend
        if len(self.current_window) == 0 or be.single_thread:
            return
        with self._timed("mapping"):
            for _ in range(iterations):
                self._map(self.current_window)
                if be.last_sent >= 10:
                    self._map(self.current_window, prune=True, iters=10)
                    self._sync_backend()

    def step(self, cur_frame_idx):
        """One pass of FrontEnd.run's body for frame ``cur_frame_idx`` (:1794-1893).  Returns True for a keyframe."""
        viewpoint = self.new_viewpoint(cur_frame_idx)
        self.cameras[cur_frame_idx] = viewpoint
        self.counts["frames"] += 1
        if cur_frame_idx == 0 or not self.current_window:
            self.initialize(cur_frame_idx, viewpoint)
            return True
        self.initialized = self.initialized or (len(self.current_window) == self.window_size)
        render_pkg = self.tracking(cur_frame_idx, viewpoint)
        last_keyframe_idx = self.current_window[0]
        check_time = (cur_frame_idx - last_keyframe_idx) >= self.kf_interval
        curr_visibility = (render_pkg["n_touched"] > 0).long()
        create_kf = is_keyframe(self.config, self.cameras, cur_frame_idx, last_keyframe_idx, curr_visibility, self.occ_aware_visibility,
                                self.median_depth)
        inter, union, _, _ = covisibility(curr_visibility, self.occ_aware_visibility[last_keyframe_idx])
        point_ratio = inter / union if union else float("nan")
        if len(self.current_window) < self.window_size:
            create_kf = check_time and point_ratio < self.config["Training"]["kf_overlap"]
        if self.single_thread:
            create_kf = check_time and create_kf
        self.frame_log.append(dict(frame=cur_frame_idx, tracking_iterations=self._last_tracking_iterations, median_depth=float(self.median_depth),
                                   covisibility=point_ratio, visible=int(curr_visibility.count_nonzero()), keyframe=bool(create_kf),
                                   gaussians_tracked_against=int(self.frontend_gaussians.get_xyz.shape[0])))
        if create_kf:
            self.current_window, removed = add_to_window(self.config, self.cameras, cur_frame_idx, curr_visibility,
                                                         self.occ_aware_visibility, self.current_window, initialized=self.initialized)
            depth_map = self.add_new_keyframe(cur_frame_idx, render_pkg=render_pkg, init=False)
            self.handle_keyframe(cur_frame_idx, viewpoint, depth_map)
            self.counts["keyframes"] += 1
            self.window_log.append(list(self.current_window))
        else:
            viewpoint.clean()                      # FrontEnd.cleanup (:1697-1700): the pose stays, the images go
        self.idle_mapping(self.idle_map_iters)
        return create_kf

    def run(self, n_frames=None):
        t0 = time.perf_counter()
        n = len(self.dataset) if n_frames is None else min(n_frames, len(self.dataset))
        with _lib.quiet_gc():
            for idx in range(n):
                self.step(idx)
        self._sync()
        self.seconds["wall"] = time.perf_counter() - t0
        return self

    # ------------------------------------------------------------------ after the sequence
    def refine(self, iterations):
        """The "color_refinement" message (utils/slam_backend.py:510-512)."""
        kw = {} if self.refine_loss_fn is None else {"loss_fn": self.refine_loss_fn}
        with self._timed("refinement"):
            color_refinement(self.backend, iteration_total=iterations, render_fn=self.render_fn, fused=self.fused, **kw)
        self.counts["refinement_iterations"] += iterations
        self._sync_backend()

    def eval_ate(self):
        """``eval_ate`` over the keyframes (utils/eval_utils_0806.py:101-169); None with fewer than three."""
        from .eval_utils import eval_ate
        return eval_ate(self.cameras, self.kf_indices, monocular=self.monocular)

    def pose_errors(self):
        """Distance between estimated and ground-truth camera centres, frame by frame, WITHOUT any alignment (frame 0 starts at its
        ground-truth pose): what ``eval_ate``'s similarity alignment -- with scale for monocular runs -- can hide, e.g. a tracker that
        follows only a fraction of the motion.  -> {frame: error}."""
        out = {}
        for i, cam in self.cameras.items():
            c = lambda R, T: -(R.detach().double().cpu().T @ T.detach().double().cpu())
            out[i] = float((c(cam.R, cam.T) - c(cam.R_gt, cam.T_gt)).norm())
        return out

    def eval_rendering(self, metrics_fn=None):
        """``eval_rendering``'s numbers (utils/eval_utils_0806.py:172-306): every NON-keyframe frame rendered from its tracked pose
        against the dataset's image; means of PSNR (and of whatever else ``metrics_fn`` returns: default ``eval_utils.frame_metrics``,
        which needs the GPU for SSIM)."""
        if metrics_fn is None:
            from .eval_utils import frame_metrics as metrics_fn
        rows = []
        for idx in range(0, len(self.cameras) - 1):      # (`end_idx = len(frames) - 1`, :184)
            if idx in self.kf_indices:
                continue
            frame = self.cameras[idx]
            gt_image = self.dataset[idx][0]
            with torch.no_grad():
                pkg = self.render_fn(frame, self.frontend_gaussians, self.pipeline_params, self.background)
            static = getattr(frame, "expanded_static_mask", None)
            if static is None:
                static = getattr(frame, "static_mask", None)
            rows.append(metrics_fn(pkg["render"], gt_image.to(pkg["render"].device), static, self.background))
        if not rows:
            return {}
        return {k: float(np.mean([r[k] for r in rows if r.get(k) is not None])) for k in rows[0] if any(r.get(k) is not None for r in rows)}

    def summary(self):
        c, s = dict(self.counts), dict(self.seconds)
        ns = [n for _, n in self.gaussian_counts]
        loops = s.get("tracking", 0.0) + s.get("mapping", 0.0)
        its = c["tracking_iterations"] + c["mapping_iterations"]
        return dict(**c, gaussians_first=ns[0] if ns else 0, gaussians_last=self._n(), gaussians_max=max(ns) if ns else 0,
                    size_changes_by_densification=sum(1 for e, _ in self.gaussian_counts if e == "densify"),
                    size_changes_by_pruning=sum(1 for e, _ in self.gaussian_counts if e == "prune"),
                    seconds={k: round(v, 4) for k, v in s.items()},
                    frames_per_s=None if not s.get("wall") else round(c["frames"] / s["wall"], 3),
                    tracking_plus_mapping_iterations_per_s=None if not loops else round(its / loops, 2),
                    tracking_iterations_per_s=None if not s.get("tracking") else round(c["tracking_iterations"] / s["tracking"], 2),
                    mapping_iterations_per_s=None if not s.get("mapping") else round(c["mapping_iterations"] / s["mapping"], 2))
