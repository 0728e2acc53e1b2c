"""Worker of tests/test_gpu_dp.py: one rank of a 2-process data-parallel Trainer run (launched by torch.distributed.run)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from webdgs_amd import ops, parallel  # noqa: E402
from webdgs_amd.trainer import Trainer  # noqa: E402

import dp_common  # noqa: E402


def main():
    out_dir, steps, use_cb = sys.argv[1], int(sys.argv[2]), sys.argv[3] == "1"
    vpr = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    rank, world, _ = parallel.init_from_env()
    dev = ops.HipDevice(int(os.environ.get("WDGS_FORCE_DEVICE", os.environ.get("LOCAL_RANK", "0"))))
    cfg, g, sh, cameras, images = dp_common.dataset(dev)
    t = Trainer(dev, seed=11, world_size=world, rank=rank, views_per_rank=vpr, use_command_buffers=use_cb)
    t.setDensifyPruneConfig(dict(schedule=dict(enabled=False)))
    t.setPointCloud(ops.createPointCloud(dev, g, sh, cfg.sh_deg))
    t.setDataset(cameras, images)
    t.start()
    import torch
    for i, ids in enumerate(dp_common.view_schedule(steps, world, vpr)):
        if i == 3:  # ranks in lock-step from here on: the condition under which an unfenced exchange was overtaken by Adam
            torch.cuda.synchronize()
            parallel.barrier()
        if i == 2 and os.environ.get("WDGS_DP_TEST_WARMUP") == "1":
            # by now the ranks have recorded DIFFERENT views; the warm-up must still take the same number of (collective) steps on each
            t.warmupCommandBuffers()
        t.step(ids)
    dev.synchronize()
    own_first, own_count = parallel.owned_range(t.pointCloud.num_points, world, rank)
    stale = t.optimizer.getStateBuffers()["optPosBuffer"].read(np.uint32).copy()  # before the gather: only the own slice is current
    t.syncOptimizerState()
    dev.synchronize()
    st = {k: b.read(np.uint32) for k, b in t.optimizer.getStateBuffers().items()}
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), gaussians=t.pointCloud.gaussian_3d_buffer.read(np.uint32),
             sh=t.pointCloud.sh_buffer.read(np.uint32), iteration=np.array([t.optimizer.getIteration()]), own=np.array([own_first, own_count]),
             stale_pos=stale, **{"state_" + k: v for k, v in st.items()})
    parallel.barrier()
    t.destroy()
    dev.destroy()
    parallel.shutdown()


if __name__ == "__main__":
    main()
