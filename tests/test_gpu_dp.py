"""GPU, two processes: the Trainer's view-sharded data-parallel step end to end -- per-rank K1..K17, fp32
accumulation, the gradient exchange, one Adam on every rank -- must leave BOTH replicas bit-identical to a single process that
takes the same two views per step.  The exchange runs over gloo here (both ranks share the one GPU of the test box, which
RCCL refuses); on a multi-GPU node the same code path runs over RCCL (bench.py --gpus N)."""
import os
import signal
import socket
import subprocess
import sys

import numpy as np
import pytest

from webdgs_amd import ops
from webdgs_amd.trainer import Trainer

import dp_common
from harness import assert_bits_equal

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


_BIND_RACE = ("EADDRINUSE", "address already in use", "Address already in use")


def _run_child(cmd, env, timeout):
    """Runs a child in its own process group; on timeout the WHOLE group is killed, so no rank survives holding the GPU."""
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out, err = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, signal.SIGKILL)
        out, err = p.communicate()
        pytest.fail(f"child timed out after {timeout}s (process group killed)\n{out[-2000:]}\n{err[-4000:]}")
    return subprocess.CompletedProcess(cmd, p.returncode, out, err)


def _run_with_fresh_port(make_cmd_env, timeout=600):
    """Launches a rendezvous-based child.  The ONLY failure that is retried (once, on another port) is the rendezvous losing the
    race for its port between probing and binding, recognised by the bind error in stderr; anything else -- a mismatch, a HIP
    error, a signal, a non-zero exit during teardown -- is returned as it is, and the caller asserts on the exit code."""
    last = None
    for _ in range(2):
        cmd, env = make_cmd_env(_free_port())
        last = _run_child(cmd, env, timeout)
        if last.returncode == 0 or not any(sig in last.stderr for sig in _BIND_RACE):
            break
    return last


def _verdict(r):
    return f"exit code {r.returncode}\n--- stdout tail ---\n{r.stdout[-2000:]}\n--- stderr tail ---\n{r.stderr[-4000:]}"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("use_cb", [False, True])
def test_two_rank_training_equals_single_process_batch_of_two(hip_device, tmp_path, use_cb):
    steps = 7
    env = dict(os.environ, WDGS_DIST_BACKEND="gloo", WDGS_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0", WDGS_DP_TEST_WARMUP="1" if use_cb else "0")
    r = _run_with_fresh_port(lambda port: ([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                                            "--master-port", str(port), os.path.join(HERE, "dp_worker.py"), str(tmp_path), str(steps), "1" if use_cb else "0"], env))
    assert r.returncode == 0, _verdict(r)
    ranks = [np.load(os.path.join(tmp_path, f"rank{i}.npz")) for i in range(2)]
    assert_bits_equal(ranks[0]["gaussians"], ranks[1]["gaussians"], "replica gaussians")
    assert_bits_equal(ranks[0]["sh"], ranks[1]["sh"], "replica sh")

    cfg, g, sh, cameras, images = dp_common.dataset(hip_device)
    t = Trainer(hip_device, seed=11, world_size=1, rank=0, views_per_rank=2, use_command_buffers=use_cb)
    t.setDensifyPruneConfig(dict(schedule=dict(enabled=False)))
    t.setPointCloud(ops.createPointCloud(hip_device, g, sh, cfg.sh_deg))
    t.setDataset(cameras, images)
    t.start()
    for i, ids in enumerate(dp_common.view_schedule(steps, 2)):
        if i == 2 and use_cb:
            t.warmupCommandBuffers()  # (training steps on [v, v] for every view: the workers take them at the same point)
        t.step(ids)
    hip_device.synchronize()
    assert_bits_equal(t.pointCloud.gaussian_3d_buffer.read(np.uint32), ranks[0]["gaussians"], "2 ranks x 1 view vs 1 rank x 2 views: gaussians")
    assert_bits_equal(t.pointCloud.sh_buffer.read(np.uint32), ranks[0]["sh"], "2 ranks x 1 view vs 1 rank x 2 views: sh")
    # optimizer state: sliced during the run (each rank trains the Gaussians it owns), identical everywhere after the gather
    for k, b in t.optimizer.getStateBuffers().items():
        single = b.read(np.uint32)
        assert_bits_equal(ranks[0]["state_" + k], ranks[1]["state_" + k], f"state {k}: rank 0 vs rank 1 after syncOptimizerState")
        assert_bits_equal(ranks[0]["state_" + k], single, f"state {k}: 2 ranks vs single process")
    for r in range(2):
        first, count = (int(x) for x in ranks[r]["own"])
        assert count > 0
        own = slice(first * 12, (first + count) * 12)
        assert np.array_equal(ranks[r]["stale_pos"][own], ranks[r]["state_optPosBuffer"][own]), "the owned slice was current before the gather"
        assert not np.array_equal(ranks[r]["stale_pos"], ranks[r]["state_optPosBuffer"]), "the other slice was stale before the gather"
    assert int(ranks[0]["iteration"][0]) == t.optimizer.getIteration() == steps + (5 if use_cb else 0)
    assert not np.array_equal(ranks[0]["gaussians"], g.reshape(ranks[0]["gaussians"].shape)), "training did not move the parameters"


def test_two_ranks_two_views_each_equals_the_oracle_trainer(hip_device, orc, tmp_path):
    """c4's shape in small: several views per rank per global step.  Per-rank fp32 sums in view order, one sliced exchange, one Adam:
    both replicas must equal the oracle trainer's batched step (world = 2) bit for bit, point cloud and gathered optimizer state."""
    from oracle import oracle_trainer
    steps, vpr = 5, 2
    env = dict(os.environ, WDGS_DIST_BACKEND="gloo", WDGS_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = _run_with_fresh_port(lambda port: ([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                                            "--master-port", str(port), os.path.join(HERE, "dp_worker.py"), str(tmp_path), str(steps), "1", str(vpr)], env))
    assert r.returncode == 0, _verdict(r)
    ranks = [np.load(os.path.join(tmp_path, f"rank{i}.npz")) for i in range(2)]
    cfg, g, sh, cameras, images = dp_common.dataset(hip_device)
    imgs = [im["texture"].read(np.uint8).reshape(cfg.height, cfg.width, 4) for im in images]
    o = oracle_trainer.OracleTrainer(g, sh, cfg.sh_deg, [c["camera"] for c in cameras], imgs, densify=dict(schedule=dict(enabled=False)))
    for ids in dp_common.view_schedule(steps, 2, vpr):
        o.step(ids, world=2)
    n = o.num_points
    for i in range(2):
        assert_bits_equal(ranks[i]["gaussians"].reshape(-1, 6)[:n], o.g, f"rank {i} gaussians vs oracle trainer (2 ranks x 2 views)")
        assert_bits_equal(ranks[i]["sh"].reshape(-1, 24)[:n], o.sh, f"rank {i} sh vs oracle trainer")
        for k, (ok, width) in dict(optPosBuffer=("opt_pos", 12), optRotBuffer=("opt_rot", 12), optScaleBuffer=("opt_scale", 12),
                                   optOpacityBuffer=("opt_opacity", 3), paramSH=("param_sh", 48), stateSH=("state_sh", 96)).items():
            assert_bits_equal(ranks[i]["state_" + k].view(np.float32).reshape(-1, width)[:n], o.state[ok], f"rank {i} state {k} vs oracle trainer")


def test_rccl_backend_through_the_trainer(tmp_path):
    """The `nccl` backend itself (RCCL), which the multi-GPU bench uses, exercised through the Trainer on this one GPU with a
    process group of size one (see tests/nccl_worker.py)."""
    r = _run_with_fresh_port(lambda port: ([sys.executable, os.path.join(HERE, "nccl_worker.py")],
                                           dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))))
    # both: the worker's own verdict AND a clean exit (teardown of the Trainer, the device, the process group and the interpreter)
    assert r.returncode == 0 and "RCCL_PATH_OK" in r.stdout, _verdict(r)


def test_bench_self_launches_two_ranks(tmp_path):
    """`python bench.py --gpus 2` from a bare shell (no launcher, no WORLD_SIZE) starts its two ranks itself,
    prints ONE JSON line carrying the like-for-like single-GPU base of the same batched step, the ranks an all-reduce of ones really
    saw and the exchange time, and exits 0.  gloo + one shared GPU here (the 2-rank rehearsal); RCCL on a multi-GPU node."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(WDGS_DIST_BACKEND="gloo", WDGS_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0", WDGS_BENCH_WATCHDOG="500")
    root = os.path.dirname(HERE)
    r = _run_child([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--config", "c2", "--steps", "3", "--warmup", "1", "--views", "4",
                    "--views-per-rank", "2", "--min-seconds", "0.05"], env, 560)
    assert r.returncode == 0, _verdict(r)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, _verdict(r)
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["n_ranks_seen"] == 2 and out["self_launched"] is True
    assert out["config"]["views_per_rank"] == 2 and out["config"]["global_batch_views"] == 4
    base = out["single_gpu_same_step"]
    assert base["views_per_step"] == 2 and base["views_per_s"] > 0 and base["ms_per_step"] > 0
    assert abs(out["scaling_efficiency"] - out["value"] / (2 * base["views_per_s"])) < 1e-3
    assert out["exchange"]["ms_per_step"] > 0 and 0 < out["exchange"]["frac_of_step"] < 1.0  # (not double-counted: ADVICE r2)
    assert out["timed_blocks"]["blocks"] >= 1 and out["steps"] == 3
    assert out["roofline"]["kernel"] and out["roofline"]["hbm_frac"] is not None
    # per-view and per-step kernels are reported apart: a batched run's view-batched K1 / K17 and its optimizer pass are per STEP
    per_view, per_step = out["kernel_ms_per_view"], out["kernel_ms_per_step"]
    assert "backward_rasterize" in per_view and "rasterize" in per_view and not (set(per_view) & set(per_step))
    assert "project_count_views" in per_step and "geometry_backward_views" in per_step and "project_count_views" not in per_view
    assert out["batched_step"] is None, "the N > 1 line's like-for-like base is `single_gpu_same_step`"


def test_single_gpu_bench_line_carries_the_batched_step_base(tmp_path):
    """the N = 1 record also holds the 8-views-per-step rate -- the like-for-like base of the 1 -> 8 GPU curve -- so a
    ratio of two SCALE records cannot be taken against the (per view slower) one-view step by accident."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", WDGS_BENCH_WATCHDOG="500")
    root = os.path.dirname(HERE)
    r = _run_child([sys.executable, os.path.join(root, "bench.py"), "--config", "c2", "--steps", "6", "--warmup", "2", "--min-seconds", "0.05", "--sustained-steps", "0",
                    "--no-cpu-baseline"], env, 560)
    assert r.returncode == 0, _verdict(r)
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    b = out["batched_step"]
    assert out["n_gpus"] == 1 and out["config"]["global_batch_views"] == 1
    assert b["views_per_step"] == 8 and b["lanes"] == 3 and b["views_per_s"] > 0 and b["ms_per_step"] > 0 and b["dataset_views"] == 64
    assert abs(b["views_per_s"] - 8 / (b["ms_per_step"] / 1e3)) / b["views_per_s"] < 2e-3
    assert "geometry_backward_adam" in out["kernel_ms_per_view"] and out["kernel_ms_per_step"] == {}, "the one-view step has no per-step kernels"


def test_a_metric_pass_that_overflows_on_one_rank_voids_the_densify_event_on_every_rank(hip_device, tmp_path):
    """the metric views of a densify event are sharded over the ranks, so one rank alone may overflow a metric pass's tile-entry list.
    Its counts are then worthless -- and were that rank to bail out while its peer rebuilds the cloud, the replicas would hold clouds of different
    sizes and the next exchange would hang.  The ranks agree (one summed word) before any count is exchanged: the event at iteration 3 is void on
    both, rank 1 enlarges its lists, the event at 6 rebuilds the cloud on both, and the replicas stay identical."""
    steps = 7
    env = dict(os.environ, WDGS_DIST_BACKEND="gloo", WDGS_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0", WDGS_DP_TEST_METRIC_OVERFLOW="1")
    r = _run_with_fresh_port(lambda port: ([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                                            "--master-port", str(port), os.path.join(HERE, "dp_worker.py"), str(tmp_path), str(steps), "0"], env))
    assert r.returncode == 0, _verdict(r)
    ranks = [np.load(os.path.join(tmp_path, f"rank{i}.npz")) for i in range(2)]
    for k in ranks:
        assert list(k["events"]) == [0, 0, 0, 0, 0, 6, 6], f"no rebuild at iteration 3, one at 6: {list(k['events'])}"
    assert int(ranks[0]["grown"][0]) == 0 and int(ranks[1]["grown"][0]) == 1, "only the rank whose metric pass overflowed enlarged its lists"
    assert int(ranks[0]["points"][0]) == int(ranks[1]["points"][0]) != 6000, "both rebuilt the cloud, to the same size"
    assert_bits_equal(ranks[0]["gaussians"], ranks[1]["gaussians"], "replica gaussians")
    assert_bits_equal(ranks[0]["sh"], ranks[1]["sh"], "replica sh")
    for k in ("optPosBuffer", "optRotBuffer", "optScaleBuffer", "optOpacityBuffer", "paramSH", "stateSH"):
        assert_bits_equal(ranks[0]["state_" + k], ranks[1]["state_" + k], f"replica state {k}")


def test_single_gpu_bench_line_carries_the_full_run(tmp_path):
    """the driver-visible record holds the run BEYOND the first two densify events -- the sustained leg carried on to
    --full-run-steps (default: the reference's 10 000) with the rate, the point count, the tile entries and the longest tile list per window
    of 1 000 iterations.  Here c2 for 1 200 iterations (windows end at 1 000 and 1 200)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", WDGS_BENCH_WATCHDOG="500")
    root = os.path.dirname(HERE)
    r = _run_child([sys.executable, os.path.join(root, "bench.py"), "--config", "c2", "--steps", "6", "--warmup", "2", "--min-seconds", "0.05", "--sustained-steps", "620",
                    "--full-run-steps", "1200", "--no-cpu-baseline", "--no-batched-step"], env, 560)
    assert r.returncode == 0, _verdict(r)
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    s, f = out["sustained"], out["full_run"]
    assert s["crosses_iterations"][1] == 620 and s["densify_events"] == 2 and out["c3_as_written_iters_per_s"] == s["iters_per_s_overall"]
    assert f["iterations"] == 1200 - s["crosses_iterations"][0] and f["densify_events"] == 8 and out["c3_full_run_iters_per_s"] == f["iters_per_s_overall"] > 0
    assert [w["to_iteration"] for w in f["windows"]] == [1000, 1200]
    for w in f["windows"]:
        assert w["iters_per_s"] > 0 and w["points"] > 0 and w["tile_entries_E"] > 0 and 0 < w["longest_tile_list"] <= w["tile_entries_E"]
    assert f["final_points"] == f["windows"][-1]["points"]


def _visible_gpus() -> int:
    import torch
    return torch.cuda.device_count()


@pytest.mark.parametrize("transport", ["torch", "capi"])
def test_two_ranks_over_rccl_equal_the_oracle_trainer(hip_device, orc, tmp_path, transport):
    """the REAL sliced exchange on more than one RCCL rank -- in-place reduce_scatter_tensor / all_gather_into_tensor on
    aliased views (torch transport) and ncclReduceScatter / ncclAllGather inside the library's own communicator (capi transport), the
    rank * slice offsets of exchange_gradients / adam_repack_f32(first, count) / apply_rows -- one rank per GPU.  Needs two GPUs: skipped
    on the one-GPU test box, runs as it is on a multi-GPU node.  Both replicas must equal the oracle trainer's batched step bit for bit."""
    if _visible_gpus() < 2:
        pytest.skip("needs 2 GPUs (one RCCL rank per GPU)")
    from oracle import oracle_trainer
    steps, vpr = 5, 2
    env = {k: v for k, v in os.environ.items() if k not in ("WDGS_DIST_BACKEND", "WDGS_FORCE_DEVICE")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", WDGS_COMM="capi" if transport == "capi" else "")
    r = _run_with_fresh_port(lambda port: ([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                                            "--master-port", str(port), os.path.join(HERE, "dp_worker.py"), str(tmp_path), str(steps), "1", str(vpr)], env))
    assert r.returncode == 0, _verdict(r)
    ranks = [np.load(os.path.join(tmp_path, f"rank{i}.npz")) for i in range(2)]
    cfg, g, sh, cameras, images = dp_common.dataset(hip_device)
    imgs = [im["texture"].read(np.uint8).reshape(cfg.height, cfg.width, 4) for im in images]
    o = oracle_trainer.OracleTrainer(g, sh, cfg.sh_deg, [c["camera"] for c in cameras], imgs, densify=dict(schedule=dict(enabled=False)))
    for ids in dp_common.view_schedule(steps, 2, vpr):
        o.step(ids, world=2)
    n = o.num_points
    for i in range(2):
        assert_bits_equal(ranks[i]["gaussians"].reshape(-1, 6)[:n], o.g, f"rank {i} gaussians vs oracle trainer over RCCL ({transport})")
        assert_bits_equal(ranks[i]["sh"].reshape(-1, 24)[:n], o.sh, f"rank {i} sh vs oracle trainer over RCCL ({transport})")
        assert_bits_equal(ranks[i]["state_optPosBuffer"].view(np.float32).reshape(-1, 12)[:n], o.state["opt_pos"], f"rank {i} gathered position state ({transport})")


def test_a_rank_with_an_empty_slice_over_rccl(tmp_path):
    """4 ranks over 2+ GPUs is not possible (one rank per GPU), so the empty-slice case -- slice_points rounds up to 64, a small cloud
    leaves the last rank nothing to own -- runs with world = 2 and 40 Gaussians: rank 0 owns 64 >= 40, rank 1 owns none, and must still
    take part in every collective and end with the same cloud."""
    if _visible_gpus() < 2:
        pytest.skip("needs 2 GPUs (one RCCL rank per GPU)")
    env = {k: v for k, v in os.environ.items() if k not in ("WDGS_DIST_BACKEND", "WDGS_FORCE_DEVICE")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", WDGS_DP_TEST_POINTS="40")
    r = _run_with_fresh_port(lambda port: ([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                                            "--master-port", str(port), os.path.join(HERE, "dp_worker.py"), str(tmp_path), "4", "1", "2"], env))
    assert r.returncode == 0, _verdict(r)
    ranks = [np.load(os.path.join(tmp_path, f"rank{i}.npz")) for i in range(2)]
    assert int(ranks[1]["own"][1]) == 0 and int(ranks[0]["own"][1]) == 40
    assert_bits_equal(ranks[0]["gaussians"], ranks[1]["gaussians"], "replicas with an empty slice on rank 1")
    assert_bits_equal(ranks[0]["sh"], ranks[1]["sh"], "replica SH rows with an empty slice on rank 1")


def test_a_rank_with_an_empty_slice(tmp_path):
    """The same over gloo on the one GPU of the test box: 40 Gaussians, world = 2 -> rank 0 owns all of them (a slice is rounded up to
    64), rank 1 owns none: its Adam launch is empty, it publishes no rows, takes part in every collective and ends with the same cloud."""
    env = dict(os.environ, WDGS_DIST_BACKEND="gloo", WDGS_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0", WDGS_DP_TEST_POINTS="40")
    r = _run_with_fresh_port(lambda port: ([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                                            "--master-port", str(port), os.path.join(HERE, "dp_worker.py"), str(tmp_path), "4", "1", "2"], env))
    assert r.returncode == 0, _verdict(r)
    ranks = [np.load(os.path.join(tmp_path, f"rank{i}.npz")) for i in range(2)]
    assert int(ranks[1]["own"][1]) == 0 and int(ranks[0]["own"][1]) == 40
    assert_bits_equal(ranks[0]["gaussians"], ranks[1]["gaussians"], "replicas with an empty slice on rank 1")
    assert_bits_equal(ranks[0]["sh"], ranks[1]["sh"], "replica SH rows with an empty slice on rank 1")
    for k in ("optPosBuffer", "stateSH"):
        assert_bits_equal(ranks[0]["state_" + k], ranks[1]["state_" + k], f"gathered state {k} with an empty slice on rank 1")


def test_one_rank_overflows_both_grow_their_lists_and_stay_replicas(tmp_path):
    """Two ranks over gloo on the test box's one GPU; rank 0's view needs more tile entries than the lists the library sized (2^20), rank 1's view sees
    nothing.  The overflow word travels with the exchange, so BOTH skip the step; rank 0 reports its own overflow, rank 1 the step skipped on every
    rank; both Trainers grow their lists and go on -- and the replicas stay bit-identical, with the same count of iterations."""
    env = dict(os.environ, WDGS_DIST_BACKEND="gloo", WDGS_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0", WDGS_DP_TEST_OVERFLOW="1")
    r = _run_with_fresh_port(lambda port: ([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                                            "--master-port", str(port), os.path.join(HERE, "dp_worker.py"), str(tmp_path), "8", "1", "1"], env))
    assert r.returncode == 0, _verdict(r)
    ranks = [np.load(os.path.join(tmp_path, f"rank{i}.npz")) for i in range(2)]
    assert int(ranks[0]["grown"][0]) >= 1 and int(ranks[1]["grown"][0]) >= 1, "both ranks grew their lists"
    assert int(ranks[0]["cap"][0]) > (1 << 20) and int(ranks[1]["cap"][0]) > (1 << 20)
    assert int(ranks[0]["host_iteration"][0]) == int(ranks[1]["host_iteration"][0]) and int(ranks[0]["iteration"][0]) == int(ranks[1]["iteration"][0])
    assert_bits_equal(ranks[0]["gaussians"], ranks[1]["gaussians"], "replicas after one rank's overflow")
    assert_bits_equal(ranks[0]["sh"], ranks[1]["sh"], "replica SH rows after one rank's overflow")
    for k in ("optPosBuffer", "stateSH"):
        assert_bits_equal(ranks[0]["state_" + k], ranks[1]["state_" + k], f"gathered state {k} after one rank's overflow")
    g0 = synth_initial_gaussians()
    assert (ranks[0]["gaussians"][: g0.size] != g0).any(), "training went on after the growth"


def synth_initial_gaussians():
    from webdgs_amd import synth
    return synth.make_gaussians(synth.SceneConfig(2, 6000, 512, 384, 1, 550.0, 0.2, "few-large-splats"))[0].reshape(-1)
