"""Worker of tests/test_gpu_dp.py::test_rccl_backend_through_the_trainer: a real `nccl` (RCCL) process group of size one, with the
Trainer's sliced data-parallel exchange forced on through BOTH transports (torch.distributed and the library's own wdgs_comm), so
that every collective of the multi-GPU path -- reduce-scatter of the gradient block and the visibility counts, the guard word's
all-reduce, the all-gather of the re-packed rows, broadcast, the counts all-reduce, float64 MAX of the timing, barrier, teardown --
is issued against RCCL on this stack, interleaved with the recorded command buffers.  With one rank the collectives are
identities: each run must equal the same run without collectives, bit for bit."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from webdgs_amd import ops, parallel  # noqa: E402
from webdgs_amd.trainer import Trainer  # noqa: E402

import dp_common  # noqa: E402


def run(dev, data, exchange):
    cfg, g, sh, cameras, images = data
    t = Trainer(dev, seed=11, world_size=1, rank=0, views_per_rank=2, exchange=exchange)
    t.setDensifyPruneConfig(dict(schedule=dict(enabled=False)))
    t.setPointCloud(ops.createPointCloud(dev, g, sh, cfg.sh_deg))
    t.setDataset(cameras, images)
    t.start()
    for ids in dp_common.view_schedule(7, 2):
        t.step(ids)
    dev.synchronize()
    assert (("apply",) in t._cmd_cache) == (exchange is not None), "sliced step: views | exchange | adam | all-gather | apply"
    if exchange is not None:
        t._state_sliced = True
        t.world_size = 1
        # (state gather with one rank: every broadcast is an identity; goes through the backend all the same)
        n = t.pointCloud.num_points
        bufs = t.optimizer.getStateBuffers()
        exchange.broadcast(bufs["optPosBuffer"].ptr, 48 * n, 0)
        exchange.allreduce_counts(t.backwardPass.getMetricCountsBuffer().ptr, n)
        t.optimizer.stateChanged()
    dev.synchronize()
    out = (t.pointCloud.gaussian_3d_buffer.read(np.uint32).copy(), t.pointCloud.sh_buffer.read(np.uint32).copy(),
           t.optimizer.getStateBuffers()["optPosBuffer"].read(np.uint32).copy(), t.optimizer.getStateBuffers()["stateSH"].read(np.uint32).copy())
    t.destroy()  # command buffers (HIP graphs) and ops go before the device and the process group
    return out


def main():
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    dev = ops.HipDevice(0)
    data = dp_common.dataset(dev)
    plain = run(dev, data, None)
    # the two transports of the sliced exchange, forced to issue their collectives in this world of one
    via_torch = run(dev, data, parallel.TorchExchange(dev, force=True))
    capi = parallel.CapiExchange(dev)
    via_capi = run(dev, data, capi)
    capi.destroy()
    parallel.barrier()
    tt = torch.tensor([1.25], dtype=torch.float64, device=dev.torch_device)
    with parallel.collective_stream(dev.torch_device):
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    same = lambda a, b: all(np.array_equal(x, y) for x, y in zip(a, b))  # noqa: E731
    ok = same(plain, via_torch) and same(plain, via_capi) and float(tt.item()) == 1.25
    with parallel.collective_stream(dev.torch_device):
        dist.barrier()
    # deterministic teardown: dataset buffers, the library device (drains the stream), then the process group, then exit
    del data, tt
    dev.destroy()
    torch.cuda.synchronize()
    dist.destroy_process_group()
    print("RCCL_PATH_OK" if ok else "RCCL_PATH_MISMATCH", flush=True)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
