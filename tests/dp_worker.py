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
    overflow = os.environ.get("WDGS_DP_TEST_OVERFLOW") == "1"
    if overflow:
        # few, large splats on 768 tiles: the lists the library sizes for this cloud (2^20 entries) overflow under view 0; view 1 looks at nothing.
        # Rank 0 always draws view 0, rank 1 view 1: only ONE rank overflows, both must skip the step, grow their lists and stay replicas.
        import warnings
        from webdgs_amd import synth
        warnings.simplefilter("always")
        cfg = synth.SceneConfig(2, 6000, 512, 384, 1, 550.0, 0.2, "few-large-splats")
        g, sh = synth.make_gaussians(cfg)
        away = np.eye(4, dtype=np.float64)
        away[2, 3] = -100.0  # the camera 100 units down the z axis, past the whole scene
        cams = [synth.circle_cameras(cfg, 1)[0], synth.camera_block(away, cfg.width, cfg.height, cfg.fy)]
        black = np.zeros((cfg.height, cfg.width, 4), np.uint8)
        cameras = [dict(camera=c, width=cfg.width, height=cfg.height) for c in cams]
        images = [dict(texture=dev.bufferFrom(black), width=cfg.width, height=cfg.height) for _ in cams]
    else:
        cfg, g, sh, cameras, images = dp_common.dataset(dev)
    t = Trainer(dev, seed=11, world_size=world, rank=rank, views_per_rank=vpr, use_command_buffers=use_cb)
    metric_overflow = os.environ.get("WDGS_DP_TEST_METRIC_OVERFLOW") == "1"
    if metric_overflow:
        # densify events at iterations 3 and 6; rank 1 builds its metric passes (first use: the event at 3) around lists of 4 096 entries, which
        # the half-resolution metric views overflow: the event must be void on BOTH ranks, rank 1 enlarges its lists, the event at 6 rebuilds the cloud
        import warnings
        warnings.simplefilter("always")
        t.setDensifyPruneConfig(dict(schedule=dict(enabled=True, warmupIterations=3, interval=3, stopIterations=100), metricViews=4, cloneThresholdCount=3,
                                     splitScaleThreshold=0.03, pruneOpacity=0.2, maxNewPointsPerStep=300))
    else:
        t.setDensifyPruneConfig(dict(schedule=dict(enabled=False)))
    t.setPointCloud(ops.createPointCloud(dev, g, sh, cfg.sh_deg))
    t.setDataset(cameras, images)
    t.start()
    import torch
    schedule = [[r % 2 for r in range(world * vpr)] for _ in range(steps)] if overflow else dp_common.view_schedule(steps, world, vpr)
    grown, events = [], []
    if overflow or metric_overflow:
        import warnings
        _show = warnings.showwarning
        warnings.showwarning = lambda message, *a, **k: (grown.append(str(message)), _show(message, *a, **k))
    for i, ids in enumerate(schedule):
        if i == 3:  # ranks in lock-step from here on: the condition under which an unfenced exchange was overtaken by Adam
            torch.cuda.synchronize()
            parallel.barrier()
        if i == 2 and os.environ.get("WDGS_DP_TEST_WARMUP") == "1":
            # by now the ranks have recorded DIFFERENT views; the warm-up must still take the same number of (collective) steps on each
            t.warmupCommandBuffers()
        if metric_overflow and i == 2 and rank == 1:
            t._grown_tile_entries = 4096   # what the metric passes, built inside this step's densify event, size their lists with
        t.step(ids)
        events.append(t.getLastDensifyPruneIteration() or 0)
    dev.synchronize()
    own_first, own_count = parallel.owned_range(t.pointCloud.num_points, world, rank)
    stale = t.optimizer.getStateBuffers()["optPosBuffer"].read(np.uint32).copy()  # before the gather: only the own slice is current
    t.syncOptimizerState()
    dev.synchronize()
    st = {k: b.read(np.uint32) for k, b in t.optimizer.getStateBuffers().items()}
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), gaussians=t.pointCloud.gaussian_3d_buffer.read(np.uint32),
             sh=t.pointCloud.sh_buffer.read(np.uint32), iteration=np.array([t.optimizer.getIteration()]), own=np.array([own_first, own_count]),
             stale_pos=stale, grown=np.array([sum("tile-entry lists grown" in m for m in grown)]), host_iteration=np.array([t.getIteration()]),
             cap=np.array([int(t.forwardPass.getResources()["maxTileEntries"])]), events=np.array(events), points=np.array([t.getPointCount()]), **{"state_" + k: v for k, v in st.items()})
    parallel.barrier()
    t.destroy()
    dev.destroy()
    parallel.shutdown()


if __name__ == "__main__":
    main()
