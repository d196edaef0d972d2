"""gloo tests (CPU, world sizes 2, 3 and 4) of the sharded mapping loop ``lvdgs.backend_map.map_window``: four iterations on
the toy scene (a densification in the second), then the pruning pass and one more iteration -- on several ranks and in
one process -- with the views dealt whole (six views on two or three ranks), cut into bands where they do not divide
evenly (four ranks), and all of them cut into one band per rank (``SPLIT_POLICY = "all"``).

Checked: (1) the two replicas end bit-identical (parameters, Adam moments, keyframe poses and exposures, bookkeeping),
which is what lets the ranks go on without ever broadcasting parameters; (2) they agree with the single-process run
(same random keyframes) to float rounding -- sums are formed in a different order; (3) the schedule of Gaussian counts
(densify / prune) is identical.  The renderer is the dense CPU one (the HIP path needs a GPU; the collectives and the
bookkeeping are what is under test)."""
import hashlib
import os
import sys
import types

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
ITERS = 4


def _paths():
    for p in (os.path.join(ROOT, "oracle"), ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import lvdgs  # noqa: F401


def _half_blind(render_fn):
    """A renderer whose views each see only half of the Gaussians (by parity of index + keyframe id) as far as the
    bookkeeping can tell: radii, visibility and n_touched of the other half are zeroed.  With it the views of different
    ranks see different subsets -- what the opacity reset of the non-visible (reference utils/slam_backend.py:367-370)
    and the per-view visibility rows need to get right across ranks."""
    def render(viewpoint, G, pipe, bg, **kw):
        pkg = render_fn(viewpoint, G, pipe, bg, **kw)
        keep = (torch.arange(pkg["radii"].shape[0]) + int(viewpoint.uid)) % 2 == 0
        pkg = dict(pkg)
        pkg["radii"] = torch.where(keep, pkg["radii"], torch.zeros_like(pkg["radii"]))
        pkg["n_touched"] = torch.where(keep, pkg["n_touched"], torch.zeros_like(pkg["n_touched"]))
        pkg["visibility_filter"] = pkg["radii"] > 0
        return pkg
    return render


def _run(group_world, policy="leftover", half_blind=False, real_window=False, aux_group=None, sharded=False):
    """One run of ITERS iterations + the pruning pass; returns a dict of numpy results.  ``real_window``: the reference's
    window instead of the fixtures' -- 8 keyframes + 2 random older ones of 12 (configs/mono/KITTI/base_config.yaml:37,
    utils/slam_backend.py:275), two iterations, the first of them through the opacity reset of the non-visible."""
    _paths()
    import test_loop_golden as tl
    from dense_render import dense_render
    from loop_scene import build_scene, loop_config
    from lvdgs import backend_map as bm
    bm.SPLIT_POLICY = policy
    if half_blind:
        dense_render = _half_blind(dense_render)
    cfg = loop_config()
    if half_blind:   # reach the opacity reset of the non-visible inside the run (iteration 3; densification at 2 and 6)
        cfg["Training"]["gaussian_reset"] = 3
    if real_window:
        cfg["Training"].update(window_size=8, gaussian_reset=1, gaussian_update_every=1000, gaussian_update_offset=999)
    sc = build_scene("cpu", n_cameras=12, window=list(range(11, 3, -1))) if real_window else build_scene("cpu")
    be = tl._backend(sc, cfg)
    be.initialized = True
    be.shard_aux_group = aux_group
    be.shard_optimizer = sharded      # (several ranks: the Gaussian Adam as reduce-scatter -> step of a share -> all-gather)
    for i, cam in enumerate(sc["cameras"]):
        be.viewpoints[i] = cam
    window = sc["window"]
    if real_window:
        # LVD-GS's default configuration (dynamic_filtering.enabled, utils/slam_frontend.py:1218,1429-1433): EVERY keyframe carries a
        # static mask, so the eight window views take the L1 + SSIM + masked-depth branch (not a sum over pixels: whole views only)
        # and only the two random older views, scored by get_loss_mapping whatever they carry, can be cut into bands
        gen = torch.Generator().manual_seed(31)
        for i, cam in enumerate(sc["cameras"]):
            H, W = int(cam.image_height), int(cam.image_width)
            m = torch.ones(H, W, dtype=torch.bool)
            y0, x0 = int(torch.randint(0, H - 16, (1,), generator=gen)), int(torch.randint(0, W - 20, (1,), generator=gen))
            m[y0:y0 + 16, x0:x0 + 20] = False
            cam.static_mask = m
    be.current_window = window
    be.keyframe_optimizers = sc["make_keyframe_optimizer"](be.viewpoints, window, cfg)
    counts = []   # Gaussians at every optimiser step (taken where map_window is about to step: the sharded step is not optimizer.step())
    if group_world == 1:
        # single process: draw the random keyframes the way the ranks do, so that the runs are comparable
        keyed = bm.random_view_indices
        bm.random_view_indices = lambda n, k, it, world, seed=0: keyed(n, k, it, 2, seed)
    stats = {"before_steps": lambda backend: counts.append(int(backend.gaussians.get_xyz.shape[0]))}
    try:
        bm.map_window(be, window, iters=2 if real_window else ITERS, render_fn=dense_render, view_loss_fn=tl._cpu_view_loss, stats=stats, bands_ok=True)
        n_mid = be.gaussians.get_xyz.shape[0]
        if real_window:
            raise _Done
        bm.map_window(be, window, prune=True, render_fn=dense_render, view_loss_fn=tl._cpu_view_loss, bands_ok=True)
        # one more iteration after the pruning pass: its (unreduced) gradients must have been dropped with the
        # replaced parameters, or carried consistently
        bm.map_window(be, window, iters=1, render_fn=dense_render, view_loss_fn=tl._cpu_view_loss, bands_ok=True, stats=stats)
    except _Done:
        pass
    finally:
        if group_world == 1:
            bm.random_view_indices = keyed
    G = be.gaussians
    out = {k: v.detach().numpy().copy() for k, v in G._params_by_name().items()}
    for gp in G.optimizer.param_groups:
        st = G.optimizer.state.get(gp["params"][0], {})
        if "exp_avg" in st:
            out["m_" + gp["name"]] = st["exp_avg"].numpy().copy()
            out["v_" + gp["name"]] = st["exp_avg_sq"].numpy().copy()
    out.update(max_radii2D=G.max_radii2D.numpy().copy(), accum=G.xyz_gradient_accum.numpy().copy(), denom=G.denom.numpy().copy(),
               n_obs=G.n_obs.numpy().copy(), kf_ids=G.unique_kfIDs.numpy().copy(), counts=np.array(counts), n_mid=np.array(n_mid))
    out["pieces_first_iteration"] = np.array([[v, r0, r1] for v, r0, r1 in stats["iterations"][0]["pieces"]], dtype=np.int64).reshape(-1, 3)
    for i, cam in enumerate(sc["cameras"]):
        out[f"R{i}"], out[f"T{i}"] = cam.R.numpy().copy(), cam.T.numpy().copy()
        out[f"exp{i}"] = np.array([float(cam.exposure_a.detach()), float(cam.exposure_b.detach())])
    for kf in window:
        out[f"occ{kf}"] = be.occ_aware_visibility[kf].numpy().copy()
    out["views_per_iteration"] = np.array([len(r["views"]) for r in stats["iterations"]])
    out["rows_per_iteration"] = np.array([sum(r1 - r0 for _, r0, r1 in r["pieces"]) for r in stats["iterations"]])
    return out


class _Done(Exception):
    pass


_PER_RANK = ("views_per_iteration", "rows_per_iteration", "pieces_first_iteration")


def _digest(res):
    h = hashlib.sha256()
    for k in sorted(res):
        if k not in _PER_RANK:
            h.update(k.encode())
            h.update(np.ascontiguousarray(res[k]).tobytes())
    return h.hexdigest()


def _worker(rank, world, port, q, policy, half_blind, real_window=False, sharded=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if world >= 8:
        torch.set_num_threads(1)   # eight ranks on the test box's eight cores
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)   # the ranks' global generators differ on purpose: nothing may depend on them
        # the small MAX collectives on a communicator of their own (map_window's aux_group), in the runs with >= 4 ranks
        aux = dist.new_group() if world >= 4 else None
        res = _run(world, policy, half_blind, real_window, aux, sharded)
        q.put((rank, _digest(res), res))
    finally:
        dist.destroy_process_group()


_SINGLE = {}


def _single_process(half_blind):
    if half_blind not in _SINGLE:
        torch.manual_seed(7)
        _SINGLE[half_blind] = _run(1, "leftover", half_blind)
    return _SINGLE[half_blind]


# (two ranks, every view in bands: the GPU twin of this test.  sharded: the Gaussian Adam as reduce-scatter -> every rank steps its
# share -> all-gather, through a densification, the opacity reset of the non-visible and the pruning pass)
@pytest.mark.parametrize("world,policy,half_blind,sharded", [(4, "leftover", False, False), (3, "all", True, False), (2, "leftover", False, True), (3, "all", True, True)])
def test_ranks_stay_bit_identical_and_match_the_single_process_run(world, policy, half_blind, sharded):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + 7 * world + (3 if policy == "all" else 0) + (1 if half_blind else 0) + (40 if sharded else 0)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, policy, half_blind, False, sharded)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=900) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    _, d0, r0 = results[0]
    # (1) replicas: bit-identical, everything
    for _, dk, rk in results[1:]:
        for k in r0:
            if k not in _PER_RANK:
                np.testing.assert_array_equal(r0[k], rk[k], err_msg=k)
        assert d0 == dk
    # the six views of an iteration (4 window + 2 random), three tile rows each: the same number of rows on every rank
    # where 18 divides evenly, whole views where six do
    rows = np.stack([r["rows_per_iteration"] for _, _, r in results])
    assert (rows.sum(0) == 18).all() and rows.max() - rows.min() <= (3 if policy == "all" else (1 if world == 4 else 0)), rows
    if policy == "leftover" and world in (2, 3):
        assert all((r["views_per_iteration"] == 6 // world).all() for _, _, r in results)
    else:
        assert max(r["views_per_iteration"].max() for _, _, r in results) > rows.max() // 3   # bands: more pieces than whole views' worth
    # (2) single process, same random keyframes
    ref = _single_process(half_blind)
    np.testing.assert_array_equal(ref["counts"], r0["counts"])
    assert int(ref["n_mid"]) == int(r0["n_mid"]) and len(set(ref["counts"].tolist())) > 1   # a densification happened
    for k in ref:
        if k in _PER_RANK + ("counts", "n_mid"):
            continue
        a, b = np.asarray(r0[k], np.float64), np.asarray(ref[k], np.float64)
        assert a.shape == b.shape, k
        if a.size:
            tol = 5e-4 * np.abs(b) + 5e-5 * max(np.abs(b).max(), 1e-30)
            assert (np.abs(a - b) <= tol).all(), (k, np.abs(a - b).max(), np.abs(b).max())


def test_the_reference_window_of_ten_views_on_eight_ranks():
    """The target shape of the sharded iteration, executed (not only planned): 8 window keyframes, every one with a static mask (the
    reference's default configuration: the masked branch of the loss, whole views only) + 2 random older ones on
    EIGHT ranks -- every rank one whole keyframe, the two other views in bands of tile rows -- for two iterations, the first
    of them through the opacity reset of the non-visible (the "seen by any view" statistic across ranks).  Replicas end
    bit-identical and equal to the single-process run to float rounding."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33100 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, "leftover", False, True)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=1500) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    _, d0, r0 = results[0]
    for _, dk, rk in results[1:]:
        for k in r0:
            if k not in _PER_RANK:
                np.testing.assert_array_equal(r0[k], rk[k], err_msg=k)
        assert d0 == dk
    # the plan, as executed: rank r holds window keyframe r whole; the two random views' 2 x 3 tile rows are dealt as bands
    pieces = {rank: res["pieces_first_iteration"].tolist() for rank, _, res in results}
    for rank in range(world):
        assert [rank, 0, 3] in pieces[rank], (rank, pieces[rank])
    bands = sorted(tuple(p) for ps in pieces.values() for p in ps if p[0] >= 8)
    assert sorted(r1 - r0 for _, r0, r1 in bands) and sum(r1 - r0 for _, r0, r1 in bands) == 6
    for v in (8, 9):
        cover = sorted((r0, r1) for w, r0, r1 in bands if w == v)
        assert cover[0][0] == 0 and cover[-1][1] == 3 and all(a[1] == b[0] for a, b in zip(cover, cover[1:]))
    assert len({rank for rank, ps in pieces.items() if any(p[0] >= 8 for p in ps)}) >= 4   # the bands are spread over the ranks
    torch.manual_seed(7)
    ref = _run(1, "leftover", False, True)
    assert (ref["views_per_iteration"] == 10).all()
    for k in ref:
        if k in _PER_RANK + ("counts", "n_mid"):
            continue
        a, b = np.asarray(r0[k], np.float64), np.asarray(ref[k], np.float64)
        assert a.shape == b.shape, k
        if a.size:
            tol = 5e-4 * np.abs(b) + 5e-5 * max(np.abs(b).max(), 1e-30)
            assert (np.abs(a - b) <= tol).all(), (k, np.abs(a - b).max(), np.abs(b).max())


def _or_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        _paths()
        from lvdgs.backend_map import _max_bytes
        mine = torch.tensor([[1, 0, 0, 1, 0], [0, 0, 0, 0, 1]] if rank == 0 else [[0, 1, 1, 0, 0], [0, 0, 1, 0, 0]], dtype=torch.uint8)
        q.put((rank, _max_bytes(mine).tolist()))
    finally:
        dist.destroy_process_group()


def test_flags_reduce_as_a_bytewise_or():
    """ADVICE (round 2): flags packed four to an int32 word and reduced with MAX lost the flags of the rank whose word was
    smaller ([1,0,0,1] against [0,1,1,0] gave [1,0,0,1]).  The rows are reduced as bytes."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31600 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_or_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0] == got[1] == [[1, 1, 1, 1, 0], [0, 0, 1, 0, 1]]


def _preflight_worker(rank, world, port, q, fail):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lvdgs import backend_map as bm
    aux = dist.new_group(backend="gloo")
    rep = bm.collective_preflight(torch.device("cpu"), None, aux, sharded_adam=True, fail=fail)
    # whatever the preflight decided, the flags still reduce as a byte-wise OR (through int32 when uint8 "failed")
    flags = torch.zeros(3, 9, dtype=torch.uint8)
    flags[rank % 3, rank::world] = 1
    work = bm._max_bytes(flags, aux if rep["use_aux_group"] else None, async_op=True)
    work.wait()
    sync = bm._max_bytes((flags * 0 + (rank == 0)).to(torch.uint8), None)
    q.put((rank, rep, flags.tolist(), sync.tolist(), bm.FLAGS_AS_INT32))
    dist.destroy_process_group()


@pytest.mark.parametrize("fail", ["", "u8_max", "aux_u8_max,aux_i32_max", "all_gather", "i32_max"])
def test_collective_preflight_and_its_fallbacks(fail):
    """backend_map.collective_preflight on two gloo ranks: every operation green without injected failures; with one, the fallback the
    docstring names, taken by BOTH ranks (the verdicts are MIN-reduced), and the flag reduction still a byte-wise OR."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29650 + (os.getpid() % 300) + 7 * len(fail)
    procs = [ctx.Process(target=_preflight_worker, args=(r, world, port, q, fail)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    reps = [g[1] for g in got]
    assert reps[0] == reps[1]                      # same verdicts, same fallbacks on every rank
    rep = reps[0]
    failed = set(x for x in fail.split(",") if x)
    assert set(rep["ops"]) == {"f32_sum", "i32_max", "u8_max", "aux_i32_max", "aux_u8_max", "reduce_scatter", "all_gather"}
    assert {k for k, v in rep["ops"].items() if v != "ok"} == failed
    assert rep["flags_as_int32"] == bool(failed & {"u8_max", "aux_u8_max"}) == got[0][4] == got[1][4]
    assert rep["use_aux_group"] == ("aux_i32_max" not in failed)
    assert rep["use_sharded_adam"] == (not failed & {"reduce_scatter", "all_gather"})
    assert (rep["fatal"] is not None) == bool(failed & {"f32_sum", "i32_max"})
    want = torch.zeros(3, 9, dtype=torch.uint8)
    for r in range(world):
        want[r % 3, r::world] = 1
    for g in got:
        assert g[2] == want.tolist() and g[3] == torch.ones(3, 9, dtype=torch.uint8).tolist()


def test_piece_plan_and_random_choice():
    _paths()
    from lvdgs import backend_map as bm
    rows = [24] * 10
    # ten views on eight ranks: the eight window keyframes whole (keyframe i -> rank i), the two others in quarter bands,
    # one band per rank: 1.25 views of rows everywhere, every row of every view exactly once
    for it in range(8):
        ps = bm.plan_pieces(rows, 8, it)
        assert [p for p in ps if p[0] < 8] == [(v, 0, 24, v) for v in range(8)]
        load = [sum(r1 - r0 for _, r0, r1, o in ps if o == r) for r in range(8)]
        assert load == [30] * 8
        for v in range(10):
            cover = sorted((r0, r1) for w, r0, r1, _ in ps if w == v)
            assert cover[0][0] == 0 and cover[-1][1] == 24 and all(a[1] == b[0] for a, b in zip(cover, cover[1:]))
        assert len({o for v, _, _, o in ps if v >= 8}) == 8
    assert len({bm.plan_pieces(rows, 8, it)[8][3] for it in range(8)}) == 8    # the bands rotate over the ranks
    # other world sizes: rows per rank as equal as whole tile rows allow
    for world in (2, 3, 4, 5, 6, 7, 10, 16):
        ps = bm.plan_pieces(rows, world, 1)
        load = [sum(r1 - r0 for _, r0, r1, o in ps if o == r) for r in range(world)]
        assert sum(load) == 240 and max(load) - min(load) <= 2, (world, load)
    # a view that cannot be rendered in bands is dealt whole; whole-view dealing is the round-2 behaviour
    ps = bm.plan_pieces(rows, 8, 0, splittable=[True] * 9 + [False])
    assert (9, 0, 24) in [p[:3] for p in ps]
    assert bm.assign_views(8, 2, 8, 0)[:8] == list(range(8)) and sorted(bm.assign_views(8, 2, 10, 3)) == list(range(10))
    assert bm.assign_views(8, 2, 1, 5) == [0] * 10
    assert [p[:3] for p in bm.plan_pieces(rows, 1)] == [(v, 0, 24) for v in range(10)]
    # all ranks draw the same random keyframes, different ones from iteration to iteration
    assert bm.random_view_indices(7, 2, 11, 4) == bm.random_view_indices(7, 2, 11, 4)
    assert len({tuple(bm.random_view_indices(7, 2, it, 4)) for it in range(20)}) > 5
    assert bm.random_view_indices(0, 2, 3, 2) == [] and len(bm.random_view_indices(1, 2, 3, 2)) == 1


def test_piece_plans_cover_every_row_once_whatever_the_views():
    """Random windows: views of different heights, some not splittable, every policy and world size: each view's rows are
    covered exactly once by non-empty pieces owned by existing ranks, unsplittable views stay whole, the plan is a pure
    function of its arguments, and with splittable views of one size no rank carries more than a view's worth above the mean."""
    _paths()
    from lvdgs import backend_map as bm
    rng = np.random.default_rng(5)
    for trial in range(300):
        V = int(rng.integers(1, 13))
        rows = [int(rng.integers(1, 70)) for _ in range(V)]
        world = int(rng.integers(1, 12))
        it = int(rng.integers(0, 1000))
        splittable = [bool(rng.random() < 0.8) for _ in range(V)]
        policy = str(rng.choice(["leftover", "all", "none"]))
        ps = bm.plan_pieces(rows, world, it, splittable, policy)
        assert ps == bm.plan_pieces(rows, world, it, splittable, policy)
        for v in range(V):
            cover = sorted((r0, r1) for w, r0, r1, _ in ps if w == v)
            assert cover and cover[0][0] == 0 and cover[-1][1] == rows[v], (rows, world, policy, ps)
            assert all(a[1] == b[0] for a, b in zip(cover, cover[1:])) and all(r1 > r0 for r0, r1 in cover)
            if world > 1 and (not splittable[v] or policy == "none"):
                assert len(cover) == 1
        assert all(0 <= o < max(world, 1) for _, _, _, o in ps)
        if world > 1 and policy == "leftover" and all(splittable) and len(set(rows)) == 1:
            load = [sum(r1 - r0 for _, r0, r1, o in ps if o == r) for r in range(world)]
            assert max(load) - sum(load) / world <= 1.0 + 1e-9, (rows, world, load)


def test_flat_reducer_repacks_and_handles_aliasing():
    """ADVICE (round 1): a bucket that captured Parameter objects went on reducing stale tensors after densify / prune
    replaced them, and a .grad left pointing into the flat buffer aliased the next pack.  The reducer takes the live
    tensors at every call and copes with gradients that are views of its own buffer."""
    _paths()
    from lvdgs.backend_map import FlatReducer
    red = FlatReducer()
    a, b = torch.arange(6.0), torch.arange(4.0) + 10
    out = red.sum_floats([a, None, b], [6, 3, 4], torch.device("cpu"))
    assert torch.equal(out[0], a) and not out[1].any() and torch.equal(out[2], b)
    # gradients now alias the bucket (as after map_window assigns p.grad = view); accumulate into them and re-reduce
    ga, gb = out[0], out[2]
    ga += 1.0
    out2 = red.sum_floats([ga, None, gb], [6, 3, 4], torch.device("cpu"))
    assert torch.equal(out2[0], a + 1.0) and torch.equal(out2[2], b)
    # sizes change (densification): the bucket is re-planned, views of the old layout that overlap are copied out first
    shifted = out2[2][:3]
    out3 = red.sum_floats([shifted, torch.ones(5)], [3, 5], torch.device("cpu"))
    assert torch.equal(out3[0], b[:3]) and torch.equal(out3[1], torch.ones(5))
    ints = red.max_ints([torch.tensor([1, 5, 2], dtype=torch.int32), torch.tensor([7], dtype=torch.int32)], torch.device("cpu"))
    assert ints[0].tolist() == [1, 5, 2] and ints[1].tolist() == [7]


# ---- ShardedAdam across a change of the map's size BETWEEN map_window calls (the reference extends the map on every keyframe) ----
def _run_extended(sharded):
    """Two iterations, an extension of the map by 37 Gaussians (GaussianModel.extend_from_pcd, what extend_from_pcd_seq ends in),
    an optimiser step outside map_window (initialize_map / color_refinement do that), two more iterations."""
    _paths()
    import test_loop_golden as tl
    from dense_render import dense_render
    from loop_scene import build_scene, loop_config
    from lvdgs import backend_map as bm
    bm.SPLIT_POLICY = "leftover"
    cfg = loop_config()
    cfg["Training"].update(gaussian_update_every=1000, gaussian_update_offset=999, gaussian_reset=1000)   # no densification inside the run
    sc = build_scene("cpu")
    be = tl._backend(sc, cfg)
    be.initialized = True
    be.shard_optimizer = sharded
    for i, cam in enumerate(sc["cameras"]):
        be.viewpoints[i] = cam
    window = sc["window"]
    be.current_window = window
    be.keyframe_optimizers = sc["make_keyframe_optimizer"](be.viewpoints, window, cfg)
    kw = dict(render_fn=dense_render, view_loss_fn=tl._cpu_view_loss, bands_ok=True)
    bm.map_window(be, window, iters=2, **kw)
    G = be.gaussians
    sharder = getattr(be, "_lvdgs_sharder", None)
    assert (sharder is not None) == sharded and (sharder is None or not sharder.stale)   # the moments left map_window whole
    gen = torch.Generator().manual_seed(5)
    n_new, K = 37, G._features_rest.shape[1] + 1
    src = torch.randint(0, G.get_xyz.shape[0], (n_new,), generator=gen)
    xyz = G.get_xyz.detach()[src] + 0.01 * torch.randn(n_new, 3, generator=gen)
    feats = torch.rand(n_new, 3, K, generator=gen) * 0.2
    G.extend_from_pcd(xyz, feats, G._scaling.detach()[src].clone(), G._rotation.detach()[src].clone(), G._opacity.detach()[src].clone(), kf_id=window[0])
    # a step outside the loop on every rank alike (gradients: a fixed function of the parameters)
    for p in G.parameters():
        p.grad = 1e-3 * torch.sin(p.detach() * 3.0)
    G.optimizer.step()
    G.optimizer.zero_grad(set_to_none=True)
    bm.map_window(be, window, iters=2, **kw)
    out = {k: v.detach().numpy().copy() for k, v in G._params_by_name().items()}
    for gp in G.optimizer.param_groups:
        st = G.optimizer.state.get(gp["params"][0], {})
        out["m_" + gp["name"]] = st["exp_avg"].numpy().copy()
        out["v_" + gp["name"]] = st["exp_avg_sq"].numpy().copy()
    return out


def _worker_extended(rank, world, port, q, sharded):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)
        q.put((rank, sharded, _run_extended(sharded)))
    finally:
        dist.destroy_process_group()


def test_sharded_adam_moments_survive_an_extension_of_the_map_between_calls():
    """The Adam moments of the sharded step against the replicated step's over two map_window calls with an extension of the map
    and an outside optimiser step in between: a share boundary moved by the extension must not let a rank step elements on
    moments another rank last updated (round-4 advisor finding).  Replicas bit-identical, moments equal to rounding."""
    ctx = mp.get_context("spawn")
    res = {}
    for sharded in (False, True):
        q = ctx.Queue()
        port = 29500 + (os.getpid() % 2000) + 300 + (11 if sharded else 0)
        procs = [ctx.Process(target=_worker_extended, args=(r, 2, port, q, sharded)) for r in range(2)]
        for p in procs:
            p.start()
        got = sorted([q.get(timeout=600) for _ in range(2)], key=lambda r: r[0])
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        for k in got[0][2]:
            np.testing.assert_array_equal(got[0][2][k], got[1][2][k], err_msg=f"replicas differ: {k} (sharded={sharded})")
        res[sharded] = got[0][2]
    for k, ref in res[False].items():
        a, b = np.asarray(res[True][k], np.float64), np.asarray(ref, np.float64)
        assert a.shape == b.shape, k
        if a.size:
            tol = 1e-5 * np.abs(b) + 1e-6 * max(np.abs(b).max(), 1e-30)
            assert (np.abs(a - b) <= tol).all(), (k, float(np.abs(a - b).max()), float(np.abs(b).max()))
