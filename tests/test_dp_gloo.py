"""CPU, world_size 2 over gloo: the view-sharded data-parallel protocol of webdgs_amd.parallel.

Ranks compute per-view gradients (the oracle stands in for the GPU kernels here), accumulate them locally in fp32,
all-reduce, and must end with exactly the single-process sum; replicas must be bit-identical."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from webdgs_amd import parallel, synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _view_gradients(view):
    """fp32 [N,14] gradient block + visibility of one view, from the oracle (fp16 GaussianGradient unpacked as the HIP
    accumulate kernel does: pos3, opacity, rot4, log-sigma3, rgb3)."""
    from oracle import oracle as orc
    cfg = synth.SceneConfig(9, 1500, 96, 64, 1, 110.0, 0.02, "dp")
    g, sh = synth.make_gaussians(cfg)
    cams = synth.circle_cameras(cfg, 4)
    st, ti = synth.render_settings(cfg), synth.tile_info(cfg.width, cfg.height, 0)
    tg, tsh = synth.make_target_scene(g, sh)
    target = orc.forward(tg, tsh, cams[view], st, ti)["rgba8"]
    state = orc.unpack(g, sh)
    r = orc.train_step(g.copy(), sh.copy(), state, cams[view], st, ti, target)
    h = r["gradients"].view(np.float16).astype(np.float32).reshape(-1, 16)
    block = np.concatenate([h[:, 0:3], h[:, 3:4], h[:, 4:8], h[:, 8:11], h[:, 12:15]], axis=1)
    vis = (r["tile_counts"] > 0)
    block[~vis] = 0
    return block, vis.astype(np.int32)


def _worker(rank, world, port, views, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, w, _ = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    mine = parallel.shard_views(views, rank, world)
    acc, vis = None, None
    for v in mine:
        b, m = _view_gradients(v)
        acc = b.copy() if acc is None else acc + b
        vis = m.copy() if vis is None else vis + m
    g = torch.from_numpy(acc.reshape(-1).copy())
    c = torch.from_numpy(vis.copy())
    parallel.allreduce_gradients(g, c)
    counts = torch.from_numpy(np.full(5, rank + 1, np.int32))
    parallel.allreduce_counts(counts)
    parallel.barrier()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), g=g.numpy(), c=c.numpy(), counts=counts.numpy())
    dist.destroy_process_group()


def test_shard_views_partitions_the_batch():
    for world in (1, 2, 3, 8):
        views = list(range(17))
        parts = [parallel.shard_views(views, r, world) for r in range(world)]
        assert sorted(sum(parts, [])) == views
        assert max(map(len, parts)) - min(map(len, parts)) <= 1


@pytest.mark.timeout(600)
def test_two_rank_gradient_allreduce_matches_single_process(tmp_path):
    views = [0, 2, 3, 1]
    port = _free_port()
    mp.spawn(_worker, args=(2, port, views, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    # replicas are bit-identical
    assert np.array_equal(r0["g"].view(np.uint32), r1["g"].view(np.uint32)) and np.array_equal(r0["c"], r1["c"])
    assert np.array_equal(r0["counts"], np.full(5, 3, np.int32))
    # and equal the single-process result with the same per-rank association: (v0 + v3) + (v2 + v1)
    b = [_view_gradients(v) for v in views]
    rank_sum = [b[0][0] + b[2][0], b[1][0] + b[3][0]]
    ref = (rank_sum[0] + rank_sum[1]).reshape(-1)
    assert np.array_equal(r0["g"].view(np.uint32), ref.view(np.uint32))
    assert np.array_equal(r0["c"], b[0][1] + b[1][1] + b[2][1] + b[3][1])
    assert (r0["c"] > 0).sum() > 100


# ----------------------------------------------------------------------------- the sliced exchange, views_per_rank > 1
def _sliced_worker(rank, world, port, steps, out_dir):
    """One rank of the sliced protocol (parallel.py): 2 views per rank per step, fp32 accumulation in view order, reduce-scatter
    (all-reduce on gloo), Adam on the OWNED slice only, all-gather of the re-packed rows, apply.  The oracle stands in for the HIP
    kernels; partitioning and sequencing are the product's (parallel.shard_views / slice_points / owned_range)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from oracle import oracle as orc
    parallel.init_from_env(backend="gloo")
    cfg, g, sh, cams, imgs, st, ti = _dp_scene()
    g, sh = g.copy(), sh.copy()
    state = orc.unpack(g, sh)
    n = g.shape[0]
    sl = parallel.slice_points(n, world)
    first, count = parallel.owned_range(n, world, rank)
    for ids in steps:
        acc = np.zeros((world * sl, 14), np.float32)
        vis = np.zeros(world * sl, np.int32)
        for k, v in enumerate(parallel.shard_views(ids, rank, world)):
            fw = orc.view_gradients(g, sh, cams[v], st, ti, imgs[v])
            m = fw["tile_counts"] > 0
            gf = orc.unpack_gradients_f32(fw["gradients"])
            gf[~m] = 0
            if k == 0:
                acc[:n] = gf
                vis[:n] = m
            else:
                acc[:n][m] += gf[m]
                vis[:n] += m
        ta, tv = torch.from_numpy(acc.reshape(-1)), torch.from_numpy(vis)
        dist.all_reduce(ta)  # (gloo has no reduce-scatter; only the owned slice is consumed below)
        dist.all_reduce(tv)
        own_vis = np.zeros(n, np.uint32)
        own_vis[first:first + count] = vis[first:first + count]
        orc.adam_f32(orc.ADAM_DEFAULT, own_vis, np.ascontiguousarray(acc[:n]), state)  # Adam touches the owned slice only
        g2, sh2 = g.copy(), sh.copy()
        orc.repack(state, g2, sh2)
        rows = np.zeros((world * sl, 8), np.uint32)
        rows[first:first + count, 0:6] = g2[first:first + count]
        rows[first:first + count, 6] = sh2[first:first + count, 0]
        rows[first:first + count, 7] = sh2[first:first + count, 1] & 0xFFFF
        tr = torch.from_numpy(rows.view(np.int32).reshape(-1))
        parts = [tr[i * sl * 8:(i + 1) * sl * 8] for i in range(world)]
        dist.all_gather(parts, parts[rank].clone())
        g[:] = rows[:n, 0:6]  # apply (own slice included: it holds the same values)
        sh[:, 0] = rows[:n, 6]
        sh[:, 1] = (sh[:, 1] & 0xFFFF0000) | (rows[:n, 7] & 0xFFFF)
    np.savez(os.path.join(out_dir, f"sliced{rank}.npz"), g=g, sh=sh, pos=state["opt_pos"], own=np.array([first, count]))
    dist.destroy_process_group()


def _dp_scene():
    from oracle import oracle as orc
    cfg = synth.SceneConfig(9, 1500, 96, 64, 1, 110.0, 0.02, "dp")
    g, sh = synth.make_gaussians(cfg)
    cams = synth.circle_cameras(cfg, 4)
    st, ti = synth.render_settings(cfg), synth.tile_info(cfg.width, cfg.height, 0)
    tg, tsh = synth.make_target_scene(g, sh)
    imgs = [orc.forward(tg, tsh, cams[v], st, ti)["rgba8"] for v in range(4)]
    return cfg, g, sh, cams, imgs, st, ti


def test_slices_partition_the_gaussians():
    for n in (0, 1, 63, 64, 65, 1500, 1_000_000):
        for world in (1, 2, 3, 8):
            sl = parallel.slice_points(n, world)
            assert sl % 64 == 0 and sl * world >= n
            owned = [parallel.owned_range(n, world, r) for r in range(world)]
            assert sum(c for _, c in owned) == n
            pos = 0
            for f, c in owned:
                assert f == min(pos, n) and c <= sl
                pos += sl


@pytest.mark.timeout(600)
def test_two_ranks_two_views_each_sliced_adam_equals_the_oracle_trainer(tmp_path):
    """views_per_rank = 2 on 2 ranks with the sliced Adam: replicas end identical to each other and to the oracle trainer's
    batched step (per-rank fp32 sums in view order, rank sums added), and each rank's optimizer state is current exactly on the
    slice it owns."""
    from oracle import oracle_trainer
    steps = [[0, 1, 2, 3], [3, 3, 1, 0], [2, 0, 0, 1]]
    port = _free_port()
    mp.spawn(_sliced_worker, args=(2, port, steps, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "sliced0.npz"), np.load(tmp_path / "sliced1.npz")
    assert np.array_equal(r0["g"], r1["g"]) and np.array_equal(r0["sh"], r1["sh"])
    cfg, g, sh, cams, imgs, _, _ = _dp_scene()
    o = oracle_trainer.OracleTrainer(g, sh, cfg.sh_deg, list(cams), imgs, densify=dict(schedule=dict(enabled=False)))
    for ids in steps:
        o.step(ids, world=2)
    assert np.array_equal(r0["g"], o.g) and np.array_equal(r0["sh"], o.sh)
    assert not np.array_equal(o.g, g)
    for r in (r0, r1):
        f, c = (int(x) for x in r["own"])
        assert np.array_equal(r["pos"][f:f + c].view(np.uint32), o.state["opt_pos"][f:f + c].view(np.uint32))
    f1, c1 = (int(x) for x in r1["own"])
    assert not np.array_equal(r0["pos"][f1:f1 + c1].view(np.uint32), o.state["opt_pos"][f1:f1 + c1].view(np.uint32)), "rank 0 does not train rank 1's slice"


def test_only_an_aliasing_refusal_switches_the_exchange_to_staging():
    """the in-place collectives fall back to a staging tensor ONLY for an error that says the arguments alias; anything
    else -- transport errors, timeouts, the unforeseen -- is re-raised, so no rank ever changes its call sequence alone."""
    ex = object.__new__(parallel.TorchExchange)
    for text in ("NCCL error: unhandled system error", "watchdog caught collective operation timeout", "something nobody has seen before", "CUDA error: invalid argument",
                 "Socket closed", ""):
        ex._in_place = True
        with pytest.raises(RuntimeError):
            ex._out_of_place(RuntimeError(text))
        assert ex._in_place
    for text in ("output tensor must not alias the input tensor", "Tensors overlap in memory", "in-place reduce_scatter is not supported"):
        ex._in_place = True
        ex._out_of_place(RuntimeError(text))
        assert not ex._in_place
