"""world_size-2 gloo test (CPU) of the mapping-window sharding: the all-reduced gradients and the
merged bookkeeping equal the single-process sum over all keyframes.  The render here is the dense
float64 autograd formulation from tests/ref_torch.py (the HIP path needs a GPU; the collective
logic is what is under test)."""
import os
import sys
from types import SimpleNamespace

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _setup_paths():
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import lvdgs  # noqa: F401


def _make_problem():
    _setup_paths()
    import ref_torch
    from lvdgs import synthetic
    W, H, N = 32, 32, 60
    g = synthetic.make_gaussians(N, W, H, seed=5, r_min=1.0, r_max=4.0, z_min=1.0, z_max=5.0)
    model = SimpleNamespace()
    model.params = [torch.nn.Parameter(g[k].clone()) for k in ("means3D", "scales", "rotations", "opacities", "colors")]
    model.get_xyz = model.params[0]
    views = []
    for i in range(5):
        cam = synthetic.make_camera(W, H, pose_seed=i + 1)
        cam.weights = synthetic.make_image_grads(W, H, i)
        cam.exposure_a = torch.nn.Parameter(torch.zeros(1))
        views.append(cam)

    def render_fn(cam):
        m3, sc, rot, op, col = model.params
        out = ref_torch.render_dense(m3.double(), op.double(), H, W, cam.tanfovx, cam.tanfovy, torch.zeros(3, dtype=torch.float64),
                                     cam.world_view_transform.double(), cam.full_proj_transform.double(),
                                     cam.camera_center.double(), scales=sc.double(), rotations=rot.double(),
                                     colors_precomp=col.double())
        vsp = torch.zeros(N, 3, requires_grad=True)
        pix = out["means2D_pix"]
        # route the pixel-mean gradient into a leaf the way render() exposes viewspace_points
        hook_scale = torch.tensor([0.5 * W, 0.5 * H], dtype=torch.float64)
        pix.register_hook(lambda gr: vsp.__setattr__("grad", torch.cat([(gr * hook_scale).float(), torch.zeros(N, 1)], 1)) or gr)
        return {"render": out["color"], "depth": out["depth"], "opacity": out["opacity"], "radii": out["radii"].int(),
                "visibility_filter": out["radii"] > 0, "n_touched": out["n_touched"].int(), "viewspace_points": vsp}

    def loss_fn(cam, pkg):
        gc, gd, go = cam.weights
        return ((pkg["render"] * gc.double()).sum() + (pkg["depth"] * gd.double()).sum() + (pkg["opacity"] * go.double()).sum()).float()

    def extra():
        s = model.params[1]
        return 10 * torch.abs(s - s.mean(dim=1, keepdim=True)).mean()

    return model, views, render_fn, loss_fn, extra


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        _setup_paths()
        from lvdgs import window_shard as ws
        torch.manual_seed(0)
        model, views, render_fn, loss_fn, extra = _make_problem()
        bucket = ws.GradientBucket(model.params)
        out = ws.sharded_map_iteration(render_fn, loss_fn, views, model, bucket, extra_loss_fn=extra, window_size=3)
        grads = [p.grad.clone() for p in model.params]
        assert out["my_views"] == [i for i in range(5) if i % world == rank]
        # every rank draws the same "random" keyframes
        assert ws.shared_random_views(7, 2, iteration=11) == ws.shared_random_views(7, 2, iteration=11)
        # keyframe parameter broadcast: owners write a marker, everyone must see it
        for i, v in enumerate(views):
            if ws.owner_of(i, world) == rank:
                v.exposure_a.data.fill_(100.0 + i)
        ws.broadcast_keyframe_params(views, names=("exposure_a",))
        assert [float(v.exposure_a) for v in views] == [100.0 + i for i in range(5)]
        q.put((rank, [g.numpy() for g in grads], {k: v.numpy() for k, v in out.items() if torch.is_tensor(v)}))
    finally:
        dist.destroy_process_group()


def test_sharded_iteration_equals_single_process_sum():
    import numpy as np
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference: all five views + the regulariser
    _setup_paths()
    from lvdgs import window_shard as ws
    model, views, render_fn, loss_fn, extra = _make_problem()
    ref = ws.sharded_map_iteration(render_fn, loss_fn, views, model, ws.GradientBucket(model.params), extra_loss_fn=extra,
                                   window_size=3)
    ref_grads = [p.grad.numpy() for p in model.params]
    for rank, grads, out in results:
        for a, b in zip(grads, ref_grads):
            np.testing.assert_allclose(a, b, rtol=2e-5, atol=1e-6 * np.abs(b).max())
        np.testing.assert_array_equal(out["radii_max"], ref["radii_max"].numpy())
        np.testing.assert_array_equal(out["n_touched_gt0"], ref["n_touched_gt0"].numpy())
        np.testing.assert_allclose(out["visibility_count"], ref["visibility_count"].numpy())
        np.testing.assert_allclose(out["viewspace_grad_norm_sum"], ref["viewspace_grad_norm_sum"].numpy(), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(out["loss"], ref["loss"].numpy(), rtol=1e-5)
    # both ranks hold bit-identical gradients (replicas stay in lock-step)
    for a, b in zip(results[0][1], results[1][1]):
        np.testing.assert_array_equal(a, b)


def test_bucket_is_noop_without_process_group():
    _setup_paths()
    from lvdgs import window_shard as ws
    p = torch.nn.Parameter(torch.ones(4))
    p.grad = torch.full((4,), 2.0)
    ws.GradientBucket([p]).all_reduce()
    assert torch.equal(p.grad, torch.full((4,), 2.0))
    assert ws.owner_of(9, 8) == 1 and ws.shared_random_views(0, 2, 3) == []
