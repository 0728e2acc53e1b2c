#!/usr/bin/env python3
"""bench.py -- training views/s (fwd + bwd + Adam) on BASELINE.json's headline configuration.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload: config c3 ("c3-perf": 1 M synthetic Gaussians, 1920x1080, SH degree 3), ground truth rendered by the
same HIP forward from the perturbed scene.  A training view = project -> scan -> emit -> sort -> ranges -> composite -> loss ->
backward raster -> geometry backward.

  N = 1   BASELINE config c3, the reference's own step: ONE view + fused Adam + re-pack per Trainer.step() (trainer.ts:568-660);
          8 circle cameras.  `value` = steps (= views) per second.
  N > 1   BASELINE config c4's shape: 64 circle cameras, `--views-per-rank` (default 8) views per rank per global step -- 64 global
          views at N = 8 -- summed locally in fp32, then ONE exchange per global step: reduce-scatter of the 60 B/Gaussian gradient
          block, Adam + re-pack on the owned 1/N slice, all-gather of the re-packed 32 B rows (webdgs_amd/parallel.py).  Weak
          scaling: the per-rank work (8 views) is fixed as N grows.  `value` = views per second of the whole job.  A rank deals its
          views to `--lanes` device lanes (default 3) so one view's bandwidth-bound stages run beside another's rasterization kernels.
          (`--views-per-rank 8` at N = 1 gives the single-GPU rate of the same 8-view step, the like-for-like base of the curve.)

The timed region is K Trainer.step() calls between a barrier + torch.cuda.synchronize() on both sides, max over ranks.  Submission
(`--pipeline-depth`, default 2): a step awaits the PREVIOUS step's completion ticket, so the host submits step k+1 while step k runs; every
one of the K steps has finished when the clock stops.  The reference awaits inside every step (trainer.ts:639-645); that form is timed right
after as `ms_per_step_awaiting_every_step` (and is what `--pipeline-depth 1` makes the headline).  After the timed region:
an eager pass with hipEvents around every launch on the launch stream (per-kernel averages, roofline of the dominant kernel), at
N = 1 a `sustained` leg (620 steps at the reference's densify defaults, crossing the first two densify events) and the
`cpu_baseline` (the oracle's train step built -O3 -march=native on this box's host cores).  ONE JSON line on rank 0.
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
        # K17 + K18 + K19 in one pass (single-view step): the packed gradient is written (32 N) but not read back
        "geometry_backward_adam": 96 * n + 4 * n + 416 * v + 100 * n - 32 * n,
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


def head_commit() -> str:
    try:
        import subprocess
        return subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], text=True, stderr=subprocess.DEVNULL).strip()
    except Exception:
        return ""


def make_dataset(dev, cfg, tg, tsh, cams):
    """Ground-truth images: HIP forward of the perturbed scene (oracle-free product path); resident rgba8 buffers."""
    from webdgs_amd import ops
    tpc = ops.createPointCloud(dev, tg, tsh, cfg.sh_deg)
    tcam = dev.createBuffer(272)
    tfw = ops.TiledForwardPass(dev, tpc, tcam, dict(viewportWidth=cfg.width, viewportHeight=cfg.height, renderMode="gaussian"))
    trs = ops.TiledRasterizer(dict(device=dev, forwardPass=tfw, format="rgba8unorm"))
    images, cameras = [], []
    for i in range(len(cams)):
        tcam.write(cams[i])
        tfw.encode(None)
        trs.encode(None, cfg.width, cfg.height)
        dev.synchronize()
        images.append(dict(texture=dev.bufferFrom(trs.getOutputTextureView().read(np.uint8)), width=cfg.width, height=cfg.height))
        cameras.append(dict(camera=cams[i], width=cfg.width, height=cfg.height))
    trs.destroy()
    tfw.destroy()
    return cameras, images


def run_sustained(dev, cfg, g, sh, cameras, images, steps: int, pipeline_depth: int = 1) -> dict:
    """BASELINE config c3 as written -- "full train loop with densify/prune schedule": a fresh Trainer at the reference's densify
    defaults (warm-up 500, every 100, 10 metric views at half resolution, <= 5000 new points per event), `steps` iterations
    crossing the first densify events.  Wall clock around every step(), densify events timed separately."""
    from webdgs_amd import ops
    from webdgs_amd.trainer import Trainer
    t = Trainer(dev, seed=99, pipeline_depth=pipeline_depth)
    t.setPointCloud(ops.createPointCloud(dev, g, sh, cfg.sh_deg))
    t.setDataset(cameras, images)
    t.setMaxIterations(10 ** 9)
    t.start()
    for _ in range(3):
        t.step()
    t.warmupCommandBuffers()
    dev.synchronize()
    start_it = t.getIteration()
    plain, events, sizes = [], [], [t.getPointCount()]
    t_all = time.perf_counter()
    while t.getIteration() < steps:
        before = t.getLastDensifyPruneIteration()
        t0 = time.perf_counter()
        t.step()
        dt = time.perf_counter() - t0
        if t.getLastDensifyPruneIteration() != before:
            events.append(dt)
            sizes.append(t.getPointCount())
        else:
            plain.append(dt)
    t.drain()
    dev.synchronize()
    total = time.perf_counter() - t_all
    n_steps = t.getIteration() - start_it
    # the steps right after a rebuild re-record their command buffers: count them with the event that caused them
    med = float(np.median(plain)) if plain else 0.0
    rerecord = float(sum(d - med for d in plain if d > 4.0 * med))
    out = dict(steps=n_steps, crosses_iterations=[start_it, t.getIteration()], iters_per_s_overall=round(n_steps / total, 2),
               iters_per_s_steady=round(1.0 / med, 2) if med > 0 else None, ms_per_step_median=round(med * 1e3, 4), densify_events=len(events),
               ms_per_densify_event=round((sum(events) + rerecord) / max(1, len(events)) * 1e3, 2) if events else None,
               ms_per_densify_event_excluding_rerecording=round(float(np.mean(events)) * 1e3, 2) if events else None,
               points=sizes, pipeline_depth=pipeline_depth, schedule="reference defaults: warm-up 500, interval 100, 10 metric views at 1/2 resolution, maxNewPointsPerStep 5000")
    t.destroy()
    return out


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c3", help="c3 (headline), c3-small, c2, c1, c5 -- only c3 is the BASELINE metric")
    ap.add_argument("--views", type=int, default=0, help="training views in the dataset (default: 8 at N = 1, 64 at N > 1)")
    ap.add_argument("--pipeline-depth", type=int, default=2, help="1: every step awaits its own completion (trainer.ts:639-645); 2: a step awaits the previous "
                    "one, so the host submits step k+1 while step k runs (the same K steps, all finished inside the timed region)")
    ap.add_argument("--lanes", type=int, default=0, help="device lanes a batched step deals its views to (0 = the Trainer's default; 1 = no overlap)")
    ap.add_argument("--views-per-rank", type=int, default=0, help="views per rank per global step (default: 1 at N = 1, 8 at N > 1)")
    ap.add_argument("--sustained-steps", type=int, default=620, help="N = 1: length of the densify-inclusive leg (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the eager per-kernel pass (roofline becomes null)")
    ap.add_argument("--cpu-baseline-points", type=int, default=0, help="0 = full workload")
    args = ap.parse_args()

    # WDGS_BENCH_WATCHDOG=<seconds>: if the run is still going after that long, every thread's Python stack goes to stderr and the
    # process exits -- a hung collective then names itself instead of running into the caller's timeout
    if os.environ.get("WDGS_BENCH_WATCHDOG"):
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["WDGS_BENCH_WATCHDOG"]), exit=True)

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
    vpr = args.views_per_rank or (1 if world == 1 else 8)
    n_dataset = args.views or (8 if world == 1 and vpr == 1 else 64)

    cfg = synth.CONFIGS[args.config]
    g, sh = synth.make_gaussians(cfg)
    tg, tsh = synth.make_target_scene(g, sh)
    cams = synth.circle_cameras(cfg, n_dataset)
    cameras, images = make_dataset(dev, cfg, tg, tsh, cams)

    trainer = Trainer(dev, seed=1234, world_size=world, rank=rank, views_per_rank=vpr, overlap_views=args.lanes or None, pipeline_depth=args.pipeline_depth)
    lanes = trainer._op_sets
    trainer.setPointCloud(ops.createPointCloud(dev, g, sh, cfg.sh_deg))
    trainer.setDataset(cameras, images)
    trainer.setMaxIterations(10 ** 9)
    if vpr > 1 or world > 1:  # the headline leg of a batched run measures the step itself; densify is the N = 1 `sustained` leg
        trainer.setDensifyPruneConfig(dict(schedule=dict(enabled=False)))
    trainer.start()

    for _ in range(args.warmup):
        trainer.step()
    trainer.warmupCommandBuffers()  # every view's command buffer recorded before the clock starts (set-up, like pipeline creation)
    stats = trainer.forwardPass.check()  # raises on tile-entry overflow

    # ---- timed region: recorded command buffers (HIP graphs), no per-kernel events.  Device time of the same region from two
    # events on the kernels' own stream (it is torch's current stream: ops.HipDevice).
    trainer.exchange_timing = world > 1
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    parallel.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record(dev.torch_stream)
    for _ in range(args.steps):
        trainer.step()
    trainer.drain()  # (pipeline depth 2: the last step's own await, with its deferred error check)
    ev1.record(dev.torch_stream)
    torch.cuda.synchronize()
    parallel.barrier()
    elapsed = time.perf_counter() - t0
    device_ms = ev0.elapsed_time(ev1)
    # the same K steps with the reference's own await inside every step (depth 1), for comparison; not the headline
    awaited_ms = None
    if trainer.pipeline_depth > 1:
        trainer.pipeline_depth = 1
        torch.cuda.synchronize()
        parallel.barrier()
        t_a = time.perf_counter()
        for _ in range(args.steps):
            trainer.step()
        torch.cuda.synchronize()
        parallel.barrier()
        awaited_ms = (time.perf_counter() - t_a) / args.steps * 1e3
        trainer.pipeline_depth = args.pipeline_depth
    exchange_ms = trainer.exchangeMilliseconds() if world > 1 else 0.0
    trainer.exchange_timing = False
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([elapsed, exchange_ms], dtype=torch.float64, device=dev.torch_device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed, exchange_ms = float(tt[0].item()), float(tt[1].item())

    # ---- per-kernel durations: the same K steps again, launched eagerly with a hipEvent pair around every kernel on the
    # launch stream (events cannot bracket single kernels inside a replayed graph).  Kernels, grids and data are identical.
    ktimes, eager_elapsed = {}, 0.0
    if not args.no_profile:
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
        if not ktimes:
            raise RuntimeError("bench: the per-kernel leg recorded no kernel (roofline would be empty)")

    stats = trainer.forwardPass.check()
    e_entries, v_visible = int(stats[0]), int(stats[1])
    n, p_pix, tiles = cfg.num_points, cfg.width * cfg.height, cfg.total_tiles
    k_coef = (cfg.sh_deg + 1) ** 2
    key_bits = 16 + max(1, tiles).bit_length()
    passes = (key_bits + 7) // 8
    pairs = int(trainer.rasterizer.getNContribTextureView().read(np.uint32).astype(np.uint64).sum())
    views_per_step = world * vpr

    # ---- roofline for the dominant kernel (group sort_* / scan_* launches into their stage); durations are PER VIEW
    groups: dict = {}
    for name, (launches, ms) in ktimes.items():
        gname = kernel_group(name)
        gl, gms = groups.get(gname, (0, 0.0))
        groups[gname] = (gl + launches, gms + ms)
    per_view_kernels = ("project_count", "scan", "emit", "sort", "tile_ranges", "rasterize", "loss_grad", "backward_rasterize", "geometry_backward", "geometry_backward_adam",
                        "store_gradients", "accumulate_gradients", "guard_accumulate")
    per_step = {k: v[1] / max(1, args.steps) / (vpr if k in per_view_kernels else 1) for k, v in groups.items()}
    dom = max(per_step, key=per_step.get) if per_step else None
    roofline = None
    if dom:
        dur_s = per_step[dom] / 1e3
        abytes = algorithmic_bytes(dom, n, v_visible, e_entries, p_pix, tiles, k_coef, passes)
        achieved = abytes / dur_s / 1e9 if dur_s > 0 else 0.0
        roofline = dict(bound="hbm", kernel=dom, achieved=round(achieved, 2), peak=8000.0, unit="GB/s", frac=round(achieved / 8000.0, 5), traffic=None,
                        avg_ms_per_launch=round(per_step[dom], 4), algorithmic_bytes=abytes,
                        note="K14/K16 are fp32-VALU-issue bound (SURVEY 8(d)); the binding resource is valu_issue_frac (counter-based, offline)")
        # HBM traffic and VALU issue of that kernel come from PMC passes (scripts/pmc.sh -> profiles/*_pmc.json), which rocprofv3
        # collects OFFLINE in runs of their own: the figures are labelled with the profile and the commit they were taken at, and are
        # dropped when the profile does not carry the kernel or the workload.
        try:
            import glob
            pmcs = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")))
            if pmcs and args.config == "c3":
                prof = json.load(open(pmcs[-1]))
                pk = prof["kernels"]
                launches = {"sort": passes}.get(dom, 1)
                names = {"sort": ("sort_hist", "sort_scan_rows", "sort_scan_base", "sort_scatter"), "scan": ("scan_reduce", "scan_block_sums", "scan_downsweep")}.get(dom, (dom,))
                tr = sum(pk[nm]["hbm_bytes"] if "hbm_bytes" in pk[nm] else pk[nm].get("hbm_bytes_corrected", 0) for nm in names if nm in pk) * launches
                if tr > 0:
                    roofline["traffic"] = round(tr)
                    roofline["traffic_source"] = f"offline PMC profile {os.path.basename(pmcs[-1])}" + (f" @ {prof['head']}" if prof.get("head") else "")
                if dom in pk and "SQ_INSTS_VALU" in pk[dom]:  # wave-instructions x 2 issue cycles / (CUs x 4 SIMDs) / clock, vs the live duration
                    issue_s = pk[dom]["SQ_INSTS_VALU"] * 2.0 / (256 * 4) / 2.4e9
                    roofline["valu_issue_frac"] = round(issue_s / dur_s, 4)
                    roofline["valu_insts_per_launch"] = round(pk[dom]["SQ_INSTS_VALU"])
        except Exception as exc:  # a missing or malformed profile must never break the bench line
            roofline["traffic_note"] = f"no PMC profile: {exc}"

    # every stage against the HBM roof (algorithmic bytes / live duration): the streaming stages are the ones it binds
    hbm_by_stage = {}
    for k, ms in per_step.items():
        ab = algorithmic_bytes(k, n, v_visible, e_entries, p_pix, tiles, k_coef, passes)
        if ab > 0 and ms > 0:
            hbm_by_stage[k] = dict(GBps=round(ab / (ms / 1e3) / 1e9, 1), frac=round(ab / (ms / 1e3) / 8.0e12, 4))

    ms_per_step = elapsed / args.steps * 1e3
    value = views_per_step * args.steps / elapsed

    sustained = cpu_baseline = None
    if rank == 0 and world == 1:
        trainer.destroy()
        if args.sustained_steps > 0 and args.config in ("c3", "c3-small", "c2"):
            sustained = run_sustained(dev, cfg, g, sh, cameras, images, args.sustained_steps, args.pipeline_depth)
        if not args.no_cpu_baseline:
            cpu_baseline = run_cpu_baseline(cfg, g, sh, cams[0], args.cpu_baseline_points)

    if rank == 0:
        exch = None
        if world > 1:
            sl = parallel.slice_points(n, world)
            exch = dict(transport=trainer.exchange.name, ms_per_step=round(exchange_ms / args.steps, 4), frac_of_step=round(exchange_ms / args.steps / ms_per_step, 4),
                        bytes_sent_per_rank_per_step=int((world - 1) * sl * (60 + 32)),
                        note="device time between the events that bracket the two collectives of a step (waiting for the slowest rank included)")
        out = {
            "metric": "training iters/sec (fwd+bwd+Adam), 1M Gaussians @1080p SH3" if args.config == "c3" else f"training iters/sec (fwd+bwd+Adam), {cfg.name}",
            "value": round(value, 3), "unit": "iters/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "device_ms_per_step": round(device_ms / args.steps, 4),
            "ms_per_step_awaiting_every_step": round(awaited_ms, 4) if awaited_ms is not None else None,
            "eager_profiled_ms_per_step": round(eager_elapsed / args.steps * 1e3, 4) if ktimes else None,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "head": head_commit(),
            "config": {"workload": f"{cfg.name}: {n} Gaussians, {cfg.width}x{cfg.height}, SH deg {cfg.sh_deg}, fwd+bwd per view, {n_dataset} circle views"
                                   + (" (BASELINE c3: the reference's one-view step)" if views_per_step == 1 else f" (BASELINE c4 shape: {vpr} views per rank per global step)"),
                       "views_per_rank": vpr, "global_batch_views": views_per_step, "lanes": lanes,
                       "parallelism": (f"dp{world}: views sharded; per global step one reduce-scatter (60 B/Gaussian) -> Adam on the owned 1/{world} slice -> "
                                       f"all-gather (32 B/Gaussian) over RCCL") if world > 1 else "single GPU",
                       "tile_entries_E": e_entries, "visible_V": v_visible, "contributing_pairs_C_upper": pairs,
                       "pipeline_depth": args.pipeline_depth,
                       "submission": "recorded command buffers (HIP graphs), one per view, re-submitted; " +
                                     ("every step awaits its own completion, as the reference does" if args.pipeline_depth <= 1 else
                                      "a step awaits the PREVIOUS step's completion ticket (the host submits step k+1 while step k runs); all K steps "
                                      "finish inside the timed region; `ms_per_step_awaiting_every_step` is the reference's own await-per-step"),
                       "densify_schedule": "reference defaults (warm-up 500): not reached in the timed region; see `sustained`" if views_per_step == 1 else "disabled in this leg",
                       "iter_definition": "value counts training VIEWS (fwd+bwd) per second; a step = views_per_rank x n_gpus views + 1 exchange + 1 Adam"},
            "kernel_ms_per_view": {k: round(v, 4) for k, v in sorted(per_step.items(), key=lambda kv: -kv[1])},
            "roofline": roofline, "hbm_roofline_by_stage": hbm_by_stage, "exchange": exch, "sustained": sustained, "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(out), flush=True)
    # orderly shutdown: every rank has finished its collectives; ops and command buffers, then the device, then the process group
    parallel.barrier()
    trainer.destroy()
    if getattr(trainer, "exchange", None) is not None:
        trainer.exchange.destroy()
    dev.destroy()
    parallel.shutdown()


def run_cpu_baseline(cfg, g, sh, cam, points: int) -> dict:
    """Times the oracle (the reference's path restated on the CPU, OpenMP over Gaussians/tiles/rows) on this box's host cores."""
    from oracle import oracle as orc
    from webdgs_amd import synth
    native = orc.use_native_build()  # -O3 -march=native, built here, on the box that times it
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
                build="g++ -O3 -march=native -fopenmp" if native else "g++ -O3 -fopenmp (portable build: the native build failed)",
                sample=f"{iters} full training iteration(s) of {cfg.name} with {g.shape[0]} Gaussians ({frac:.0%} of the workload) on {cores} host threads (oracle = CPU restatement of the reference WGSL; the reference itself cannot run without a browser)",
                ms_per_step=round(dt / iters * 1e3, 1))


if __name__ == "__main__":
    main()
