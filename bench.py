#!/usr/bin/env python3
"""bench.py -- training iters/s (fwd + bwd + Adam) on BASELINE.json's headline configuration.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one Trainer.step(): per rank one training view through project -> scan -> emit -> sort -> ranges ->
composite -> loss -> backward raster -> geometry backward, then (N > 1) one RCCL all-reduce of the gradient block, then
fused Adam + re-pack.  Workload at every N: config c3 ("c3-perf": 1 M synthetic Gaussians, 1920x1080, SH degree 3;
SURVEY.md section 8(d)), 8 circle cameras, ground truth rendered by the same HIP forward from the perturbed scene; the
reference's densify schedule stays enabled with its defaults (warm-up 500 iterations), so it does not fire inside a short
run.  Weak scaling: per-GPU work is fixed (one view per rank per step); value = views processed per second by the whole job.

Prints ONE JSON line on rank 0 (contract in the task statement), with `roofline` (dominant kernel, live hipEvent
durations on the launch stream) and `cpu_baseline` (the oracle's train step on the host cores, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def algorithmic_bytes(kernel: str, n: int, v: int, e: int, p: int, t: int, k_coef: int, passes: int) -> float:
    """SURVEY.md section 8(d) per-stage algorithmic bytes for one training view."""
    table = {
        "project_count": n * (24 + 6 * k_coef) + 28 * v + 4 * n,
        "scan": 8 * n,
        "emit": 16 * v + 8 * e,
        "sort": 4 * e + 16 * passes * e,
        "tile_ranges": 4 * e + 4 * (t + 1),
        "rasterize": 4 * e + 24 * v + 12 * p,
        "loss_grad": 24 * p,
        "backward_rasterize": 4 * e + 24 * v + 24 * p + 40 * v,
        "geometry_backward": 96 * n,
        "adam_repack": 4 * n + 416 * v + 100 * n,
    }
    return float(table.get(kernel, 0))


def kernel_group(name: str) -> str:
    if name.startswith("sort_") or name.startswith("scan_"):
        return "sort" if name.startswith("sort_") else "scan"
    if name.startswith("tile_ranges"):
        return "tile_ranges"
    if name.startswith("adam_repack"):
        return "adam_repack"
    return name


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c3", help="c3 (headline), c3-small, c2, c1 -- only c3 is the BASELINE metric")
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-points", type=int, default=0, help="0 = full workload")
    args = ap.parse_args()

    import torch
    from webdgs_amd import ops, parallel, synth
    from webdgs_amd.trainer import Trainer

    rank, world, local_rank = parallel.init_from_env()
    if world != args.gpus:
        if rank == 0:
            print(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}: launch with torch.distributed.run --nproc-per-node {args.gpus}", file=sys.stderr)
        if world == 1 and args.gpus > 1:
            sys.exit(2)
    # WDGS_FORCE_DEVICE (with WDGS_DIST_BACKEND=gloo): rehearse N > 1 on a single-GPU box; never set by the driver
    dev = ops.HipDevice(int(os.environ.get("WDGS_FORCE_DEVICE", local_rank)))

    cfg = synth.CONFIGS[args.config]
    g, sh = synth.make_gaussians(cfg)
    tg, tsh = synth.make_target_scene(g, sh)
    cams = synth.circle_cameras(cfg, args.views)

    # ground-truth images: HIP forward of the perturbed scene (oracle-free product path)
    tpc = ops.createPointCloud(dev, tg, tsh, cfg.sh_deg)
    tcam = dev.createBuffer(272)
    tfw = ops.TiledForwardPass(dev, tpc, tcam, dict(viewportWidth=cfg.width, viewportHeight=cfg.height, renderMode="gaussian"))
    trs = ops.TiledRasterizer(dict(device=dev, forwardPass=tfw, format="rgba8unorm"))
    images, cameras = [], []
    for i in range(args.views):
        tcam.write(cams[i])
        tfw.encode(None)
        trs.encode(None, cfg.width, cfg.height)
        dev.synchronize()
        img = trs.getOutputTextureView().read(np.uint8)
        images.append(dict(texture=dev.bufferFrom(img), width=cfg.width, height=cfg.height))
        cameras.append(dict(camera=cams[i], width=cfg.width, height=cfg.height))
    gt0 = trs.getOutputTextureView().read(np.uint8).reshape(cfg.height, cfg.width, 4) if False else None
    trs.destroy(); tfw.destroy()
    del tpc

    trainer = Trainer(dev, seed=1234, world_size=world, rank=rank, views_per_rank=1)
    trainer.setPointCloud(ops.createPointCloud(dev, g, sh, cfg.sh_deg))
    trainer.setDataset(cameras, images)
    trainer.setMaxIterations(10 ** 9)
    trainer.start()

    for _ in range(args.warmup):
        trainer.step()
    trainer.warmupCommandBuffers()  # every view's command buffer recorded before the clock starts (set-up, like pipeline creation)
    stats = trainer.forwardPass.check()  # raises on tile-entry overflow

    # ---- timed region: recorded command buffers (HIP graphs), no per-kernel events
    parallel.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        trainer.step()
    torch.cuda.synchronize()
    parallel.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev.torch_device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- per-kernel durations: the same K steps again, launched eagerly with a hipEvent pair around every kernel on the
    # launch stream (events cannot bracket single kernels inside a replayed graph).  Kernels, grids and data are identical.
    trainer.use_command_buffers = False
    trainer._invalidate_command_buffers()
    trainer.step()
    dev.setProfiling(True)
    dev.kernelTimes(reset=True)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        trainer.step()
    torch.cuda.synchronize()
    eager_elapsed = time.perf_counter() - t1
    dev.setProfiling(False)
    ktimes = dev.kernelTimes()

    stats = trainer.forwardPass.check()
    e_entries, v_visible = int(stats[0]), int(stats[1])
    n, p_pix, tiles = cfg.num_points, cfg.width * cfg.height, cfg.total_tiles
    k_coef = (cfg.sh_deg + 1) ** 2
    key_bits = 16 + max(1, tiles).bit_length()
    passes = (key_bits + 7) // 8
    pairs = int(trainer.rasterizer.getNContribTextureView().read(np.uint32).astype(np.uint64).sum())

    # ---- roofline for the dominant kernel (group sort_* / scan_* launches into their stage)
    groups: dict = {}
    for name, (launches, ms) in ktimes.items():
        gname = kernel_group(name)
        gl, gms = groups.get(gname, (0, 0.0))
        groups[gname] = (gl + launches, gms + ms)
    per_step = {k: v[1] / max(1, args.steps) for k, v in groups.items()}
    dom = max(per_step, key=per_step.get) if per_step else None
    roofline = None
    if dom:
        dur_s = per_step[dom] / 1e3
        abytes = algorithmic_bytes(dom, n, v_visible, e_entries, p_pix, tiles, k_coef, passes)
        achieved = abytes / dur_s / 1e9 if dur_s > 0 else 0.0
        roofline = dict(bound="hbm", kernel=dom, achieved=round(achieved, 2), peak=8000.0, unit="GB/s", frac=round(achieved / 8000.0, 5), traffic=None,
                        avg_ms_per_step=round(per_step[dom], 4), algorithmic_bytes=abytes,
                        note="K14/K16 are fp32-VALU-issue bound (SURVEY 8(d)); see valu_* for the binding resource")
        # HBM traffic of that kernel from the committed PMC passes (scripts/pmc.sh -> profiles/*_pmc.json; rocprofv3 cannot be
        # run from inside this process): FETCH_SIZE x2 (gfx950) + WRITE_SIZE, per launch, same workload and kernels.
        try:
            import glob
            pmcs = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")))
            if pmcs and args.config == "c3":
                pk = json.load(open(pmcs[-1]))["kernels"]
                launches = {"sort": passes}.get(dom, 1)
                names = {"sort": ("sort_hist", "sort_scan_rows", "sort_scan_base", "sort_scatter"), "scan": ("scan_reduce", "scan_block_sums", "scan_downsweep")}.get(dom, (dom,))
                tr = sum(pk[nm]["hbm_bytes_corrected"] for nm in names if nm in pk and "hbm_bytes_corrected" in pk[nm]) * launches
                if tr > 0:
                    roofline["traffic"] = round(tr)
                    roofline["traffic_source"] = os.path.basename(pmcs[-1])
        except Exception as exc:  # a missing or malformed profile must never break the bench line
            roofline["traffic_note"] = f"no PMC profile: {exc}"
        fwd_flops = 256.0 * e_entries * 23
        bwd_flops = 256.0 * e_entries * 12 + pairs * 60.0
        if dom in ("rasterize", "backward_rasterize"):
            fl = fwd_flops if dom == "rasterize" else bwd_flops
            roofline.update(valu_achieved_tflops=round(fl / dur_s / 1e12, 3), valu_peak_tflops=157.3, valu_frac=round(fl / dur_s / 157.3e12, 5))

    # every stage against the HBM roof (algorithmic bytes / live duration): the streaming stages are the ones it binds
    hbm_by_stage = {}
    for k, ms in per_step.items():
        ab = algorithmic_bytes(k, n, v_visible, e_entries, p_pix, tiles, k_coef, passes)
        if ab > 0 and ms > 0:
            hbm_by_stage[k] = dict(GBps=round(ab / (ms / 1e3) / 1e9, 1), frac=round(ab / (ms / 1e3) / 8.0e12, 4))

    ms_per_step = elapsed / args.steps * 1e3
    value = world * args.steps / elapsed

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_baseline = run_cpu_baseline(cfg, g, sh, cams[0], args.cpu_baseline_points)

    if rank == 0:
        out = {
            "metric": "training iters/sec (fwd+bwd+Adam), 1M Gaussians @1080p SH3" if args.config == "c3" else f"training iters/sec (fwd+bwd+Adam), {cfg.name}",
            "value": round(value, 3), "unit": "iters/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "eager_profiled_ms_per_step": round(eager_elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{cfg.name}: {n} Gaussians, {cfg.width}x{cfg.height}, SH deg {cfg.sh_deg}, fwd+bwd+Adam per view, {args.views} circle views",
                       "global_batch_views": world, "parallelism": f"dp{world} (views sharded, RCCL all-reduce of 60 B/Gaussian)" if world > 1 else "single GPU",
                       "tile_entries_E": e_entries, "visible_V": v_visible, "contributing_pairs_C_upper": pairs,
                       "submission": "recorded command buffers (HIP graphs) re-submitted per view; per-step host sync as in the reference",
                       "densify_schedule": "reference defaults (warm-up 500): not reached in this run",
                       "iter_definition": "one training view (fwd+bwd); a global step = n_gpus views + 1 gradient all-reduce + 1 Adam"},
            "kernel_ms_per_step": {k: round(v, 4) for k, v in sorted(per_step.items(), key=lambda kv: -kv[1])},
            "roofline": roofline, "hbm_roofline_by_stage": hbm_by_stage, "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(out), flush=True)
    if world > 1:  # orderly shutdown: every rank has finished its collectives before any communicator is torn down
        import torch.distributed as dist
        parallel.barrier()
        torch.cuda.synchronize()
        dist.destroy_process_group()


def run_cpu_baseline(cfg, g, sh, cam, points: int) -> dict:
    """Times the oracle (the reference's path restated on the CPU, OpenMP over Gaussians/tiles/rows) on this box's host cores."""
    from oracle import oracle as orc
    from webdgs_amd import synth
    cores = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    if points and points < cfg.num_points:
        g, sh = g[:points].copy(), sh[:points].copy()
    else:
        g, sh = g.copy(), sh.copy()
    st, ti = synth.render_settings(cfg), synth.tile_info(cfg.width, cfg.height, 0)
    tg, tsh = synth.make_target_scene(g, sh)
    target = orc.forward(tg, tsh, cam, st, ti)["rgba8"]
    state = orc.unpack(g, sh)
    t0 = time.perf_counter()
    iters = 0
    while True:
        orc.train_step(g, sh, state, cam, st, ti, target)
        iters += 1
        if time.perf_counter() - t0 > 10.0 or iters >= 5:
            break
    dt = time.perf_counter() - t0
    frac = g.shape[0] / cfg.num_points
    return dict(value=round(iters / dt, 4), unit="iters/s", cores=cores, kind="port",
                sample=f"{iters} full training iteration(s) of {cfg.name} with {g.shape[0]} Gaussians ({frac:.0%} of the workload) on {cores} host threads (oracle = CPU restatement of the reference WGSL; the reference itself cannot run without a browser)",
                ms_per_step=round(dt / iters * 1e3, 1))


if __name__ == "__main__":
    main()
