"""Worker of tests/test_gpu_dp.py::test_rccl_backend_through_the_trainer: a real `nccl` (RCCL) process group of size one, with the
Trainer's data-parallel exchange forced on, so that every collective of the multi-GPU path -- fp32 view of the gradient block,
int32 visibility, float64 MAX of the timing, barrier, teardown -- is issued against RCCL on this stack, interleaved with the
recorded command buffers.  With one rank the reductions are identities: the run must equal the same run without collectives."""
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
    t = Trainer(dev, seed=11, world_size=1, rank=0, views_per_rank=2)
    t.setDensifyPruneConfig(dict(schedule=dict(enabled=False)))
    t.setPointCloud(ops.createPointCloud(dev, g, sh, cfg.sh_deg))
    t.setDataset(cameras, images)
    t.start()
    if exchange:
        def allreduce():  # Trainer._allreduce without its world_size == 1 shortcut
            n = t.pointCloud.num_points
            gr = t._dp_grad.tensor().view(torch.float32)[: parallel.GRAD_FLOATS * n]
            vis = t._dp_visible.tensor()[:n]
            dev.torch_stream.synchronize()
            dist.all_reduce(gr, op=dist.ReduceOp.SUM)
            dist.all_reduce(vis, op=dist.ReduceOp.SUM)
            torch.cuda.synchronize(dev.torch_device)
        t._allreduce = allreduce
    for ids in dp_common.view_schedule(7, 2):
        t.step(ids)
    dev.synchronize()
    out = t.pointCloud.gaussian_3d_buffer.read(np.uint32).copy()
    t.destroy()  # command buffers (HIP graphs) and ops go before the device and the process group
    return out


def main():
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    dev = ops.HipDevice(0)
    data = dp_common.dataset(dev)
    with_rccl = run(dev, data, True)
    parallel.barrier()
    tt = torch.tensor([1.25], dtype=torch.float64, device=dev.torch_device)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    plain = run(dev, data, False)
    ok = bool(np.array_equal(with_rccl, plain)) and float(tt.item()) == 1.25
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
