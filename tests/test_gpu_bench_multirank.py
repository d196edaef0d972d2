"""``bench.py --gpus N`` executed with more than one rank BEFORE the driver's scaling run does it: the command the driver launches
(``python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P bench.py --gpus 2 --steps K
--warmup W``) as a child process, with ``LVDGS_BENCH_BACKEND=gloo`` so that the two ranks can share the box's one GPU (collectives
through the host: a functional run, not a benchmark result).  Covered: the default window (every keyframe with a static mask -- the
reference's default configuration), ``--no-masks``, and the sharded Gaussian Adam (``LVDGS_BENCH_SHARDED_ADAM=1``).  The JSON line must
parse and carry the per-phase timings, the same-step-on-one-GPU anchor and no notes (nothing around the timed region went wrong)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(extra_args=(), extra_env=None, steps=3, warmup=1):
    env = dict(os.environ)
    env.update(LVDGS_BENCH_BACKEND="gloo", LVDGS_BENCH_WORKLOAD="kitti07_geom", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update(extra_env or {})
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    port = 29500 + (os.getpid() % 1500) + 17 * len(extra_args) + (5 if extra_env else 0)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", str(steps), "--warmup", str(warmup), *extra_args]
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]   # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def _check(out, masked, expect_notes=False):
    cfg = out["config"]
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["unit"] == "iters/s" and out["scaling"] == "strong"
    assert out["value"] > 0 and abs(out["value"] - 10 * 1e3 / out["ms_per_step"]) <= 1e-2 * out["value"]   # 10 views per iteration
    assert cfg["workload"] == "kitti07_geom" and cfg["views_per_step"] == 10 and cfg["window_keyframes_carry_static_mask"] is masked
    assert (cfg["notes"] is None) is not expect_notes, cfg["notes"]
    ph = cfg["phases_us_per_step"]
    assert set(ph) >= {"views", "statistics", "collectives", "bookkeeping", "optimizer_steps"} and all(v >= 0 for v in ph.values())
    assert cfg["comm_us_per_step"] == ph["collectives"]
    assert cfg["same_step_on_one_gpu_iters_per_s"] > 0 and cfg["same_step_on_one_gpu_value"] > 0
    assert "functional check" in out["collective_backend"]
    comm = cfg["comm"]
    assert comm["world"] == 2 and comm["fatal"] is None and [r[0] for r in comm["ranks_seen"]] == [0, 1] and comm["backend"] == "gloo"
    assert out["scaling_anchor_value"] == cfg["same_step_on_one_gpu_value"]   # the 1-GPU point of the curve, top level
    other = cfg["value_without_static_masks" if masked else "value_with_static_masks"]
    assert other is not None and other > 0
    assert out["steady_state"]["value"] > 0 and "blend_bwd" in out["kernels_us_per_step"]
    return out


def test_bench_two_ranks_default_window_with_static_masks():
    out = _check(_launch(), masked=True)
    assert "masked_loss" in out["kernels_us_per_step"]   # the L1 + SSIM + masked-depth branch ran on rank 0's keyframes


def test_bench_two_ranks_without_masks():
    out = _check(_launch(("--no-masks",)), masked=False)
    assert "masked_loss" not in out["kernels_us_per_step"]


def test_bench_two_ranks_sharded_adam():
    _check(_launch((), {"LVDGS_BENCH_SHARDED_ADAM": "1"}), masked=True)


def test_bench_two_ranks_survive_a_failing_uint8_reduction_and_a_failing_second_communicator():
    """The preflight's fallbacks (backend_map.collective_preflight): a uint8 MAX that fails -> the flags travel as int32; an auxiliary
    communicator that fails -> the MAX collectives on the main one; the line is printed all the same and says what happened."""
    out = _check(_launch((), {"LVDGS_PREFLIGHT_FAIL": "u8_max,aux_i32_max"}), masked=True, expect_notes=True)
    comm = out["config"]["comm"]
    assert comm["flags_as_int32"] is True and comm["use_aux_group"] is False
    assert "failed on purpose" in comm["ops"]["u8_max"] and comm["ops"]["f32_sum"] == "ok" and comm["ops"]["i32_max"] == "ok"
    assert any("int32" in n for n in out["config"]["notes"]) and any("auxiliary" in n for n in out["config"]["notes"])


def test_bench_two_ranks_sharded_adam_falls_back_when_reduce_scatter_fails():
    out = _check(_launch((), {"LVDGS_BENCH_SHARDED_ADAM": "1", "LVDGS_PREFLIGHT_FAIL": "reduce_scatter"}), masked=True, expect_notes=True)
    assert out["config"]["comm"]["use_sharded_adam"] is False
