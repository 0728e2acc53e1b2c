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
