#!/usr/bin/env node
// bench.js -- bench.py's headline measurement driven by the TypeScript-side host: training iterations per second (forward + backward + Adam) on a
// BASELINE.json configuration, through bindings/ts/trainer.js over the N-API addon (no Python, no torch in the process).
//
//     node bindings/napi/bench.js [--config c3] [--steps 30] [--warmup 5] [--views-per-step 1] [--lanes 0] [--pipeline-depth 2] [--min-seconds 1]
//                                 [--views 0] [--gpus 1] [--no-profile] [--sustained-steps 0]
//
// Same workload as bench.py: the synthetic scene of SURVEY.md 8(d) (bindings/ts/synth.js generates the very bits webdgs_amd/synth.py does),
// ground truth rendered by the HIP forward from the perturbed scene, 8 circle cameras (64 for a batched step), the reference's one-view step
// (trainer.ts:568-660) or, with --views-per-step V, the batched step of BASELINE config c4.  Same timed region: blocks of exactly K
// Trainer.step() calls between device synchronisations, every block restored to the same training state (device-side snapshot), repeated
// until --min-seconds have been timed, the MEDIAN block reported; pipeline depth 2 = a step awaits the previous step's ticket, and the
// reference's own await-inside-every-step form is timed right after (`ms_per_step_awaiting_every_step`, trainer.ts:639-651 is that meter).
// --gpus N > 1: this process starts N ranks of itself as CHILD processes (one per GPU; RANK / WORLD_SIZE / LOCAL_RANK, the RCCL id through a
// file) before it has loaded the addon or touched a GPU, and relays rank 0's line.  ONE JSON line on stdout.
'use strict';
const fs = require('fs');
const os = require('os');
const path = require('path');

function parseArgs(argv) {
  const a = { config: 'c3', steps: 30, warmup: 5, viewsPerStep: 0, lanes: 0, pipelineDepth: 2, minSeconds: 1.0, views: 0, gpus: 1, profile: true, sustainedSteps: 0 };
  const names = { '--config': 'config', '--steps': 'steps', '--warmup': 'warmup', '--views-per-step': 'viewsPerStep', '--views-per-rank': 'viewsPerStep', '--lanes': 'lanes',
    '--pipeline-depth': 'pipelineDepth', '--min-seconds': 'minSeconds', '--views': 'views', '--gpus': 'gpus', '--sustained-steps': 'sustainedSteps' };
  for (let i = 0; i < argv.length; i++) {
    if (argv[i] === '--no-profile') { a.profile = false; continue; }
    const k = names[argv[i]];
    if (!k) throw new Error(`unknown flag ${argv[i]}`);
    a[k] = k === 'config' ? argv[++i] : Number(argv[++i]);
  }
  return a;
}

/** `node bench.js --gpus N` from a bare shell: N children, one rank each; rank 0's stdout is relayed; the exit code is the worst child's. */
function selfLaunch(args, argv) {
  const { spawn } = require('child_process');
  const dir = fs.mkdtempSync(path.join(os.tmpdir(), 'wdgs-bench-'));
  const env = Object.assign({}, process.env, { WORLD_SIZE: String(args.gpus), WDGS_RENDEZVOUS: path.join(dir, 'rccl-id'), WDGS_BENCH_SELF_LAUNCHED: '1' });
  if (!env.HSA_ENABLE_IPC_MODE_LEGACY) env.HSA_ENABLE_IPC_MODE_LEGACY = '0';
  let left = args.gpus, worst = 0;
  const kids = [];
  for (let r = 0; r < args.gpus; r++) {
    const kid = spawn(process.execPath, [__filename].concat(argv), { env: Object.assign({}, env, { RANK: String(r), LOCAL_RANK: String(r) }), stdio: ['ignore', r === 0 ? 'inherit' : 'ignore', 'inherit'] });
    kids.push(kid);
    kid.on('exit', (code, signal) => {
      if (code !== 0) { worst = worst || code || 1; console.error(`[bench.js] rank ${r} exited with ${signal || code}`); for (const k of kids) if (k !== kid && k.exitCode === null) k.kill('SIGTERM'); }
      if (--left === 0) { try { fs.rmdirSync(dir, { recursive: true }); } catch (_e) { /* scratch */ } process.exit(worst); }
    });
  }
}

/** mulberry32: a seeded view-draw stream whose state can be saved and restored (the blocks of the timed region replay the same draws). */
function seededRandom(seed) {
  const r = () => { r.state = (r.state + 0x6d2b79f5) | 0; let t = Math.imul(r.state ^ (r.state >>> 15), 1 | r.state); t = (t + Math.imul(t ^ (t >>> 7), 61 | t)) ^ t; return ((t ^ (t >>> 14)) >>> 0) / 4294967296; };
  r.state = seed | 0;
  return r;
}

async function main() {
  const argv = process.argv.slice(2), args = parseArgs(argv);
  if (args.gpus > 1 && !process.env.WORLD_SIZE) { selfLaunch(args, argv); return; }
  // WDGS_BENCH_SELFTEST=1: what a rank does in the launcher's CPU test (tests/test_bench_launcher.py) -- no addon, no GPU: the ranks meet through
  // files beside the rendezvous path, rank 0 prints one line.  WDGS_BENCH_SELFTEST=fail makes the last rank exit non-zero.
  if (process.env.WDGS_BENCH_SELFTEST) {
    const world = parseInt(process.env.WORLD_SIZE || '1', 10), rank = parseInt(process.env.RANK || '0', 10);
    if (process.env.WDGS_BENCH_SELFTEST === 'fail' && rank === world - 1) { console.error('selftest: this rank fails on purpose'); process.exit(7); }
    const base = process.env.WDGS_RENDEZVOUS || path.join(os.tmpdir(), 'wdgs-selftest');
    fs.writeFileSync(`${base}.rank${rank}`, String(process.pid));
    const deadline = Date.now() + 20000;
    let seen = 0;
    while (Date.now() < deadline) {
      seen = 0; for (let r = 0; r < world; r++) if (fs.existsSync(`${base}.rank${r}`)) seen++;
      if (seen === world) break;
      Atomics.wait(new Int32Array(new SharedArrayBuffer(4)), 0, 0, 20);
    }
    if (rank === 0) console.log(JSON.stringify({ selftest: true, n_gpus: world, n_ranks_seen: seen, self_launched: process.env.WDGS_BENCH_SELF_LAUNCHED === '1', argv }));
    process.exit(seen === world ? 0 : 3);
  }

  const hip = require(path.join(__dirname, '..', 'ts', 'webdgs_hip.js'));
  const parallel = require(path.join(__dirname, '..', 'ts', 'parallel.js'));
  const synth = require(path.join(__dirname, '..', 'ts', 'synth.js'));
  const { Trainer } = require(path.join(__dirname, '..', 'ts', 'trainer.js'));
  const now = () => { const t = process.hrtime(); return t[0] + t[1] * 1e-9; };
  const addon = hip.addon;

  const { rank, world, localRank } = parallel.envRanks();
  if (world !== args.gpus) throw new Error(`WORLD_SIZE=${world} but --gpus ${args.gpus}`);
  const dev = new hip.HipDevice(process.env.WDGS_FORCE_DEVICE !== undefined ? Number(process.env.WDGS_FORCE_DEVICE) : localRank);
  const exchange = parallel.defaultExchange(dev);
  const vpr = args.viewsPerStep || (world === 1 ? 1 : 8);
  const nDataset = args.views || (world === 1 && vpr === 1 ? 8 : 64);
  const cfg = synth.CONFIGS[args.config];
  if (!cfg) throw new Error(`unknown config ${args.config}`);

  // a word summed over the ranks + a host wait: the barrier of the timed region
  const barrierWord = dev.createBuffer({ size: 4 });
  const barrier = () => { if (world > 1) exchange.allreduceCounts(barrierWord, 1); dev.synchronize(); };

  // ---- scene and ground truth (HIP forward of the perturbed scene: oracle-free, resident rgba8 buffers)
  const tGen = now();
  const scene = synth.makeGaussians(cfg), target = synth.makeTargetScene(scene.gaussians, scene.sh);
  const cams = synth.circleCameras(cfg, nDataset);
  const genSeconds = now() - tGen;
  const upload = (words) => { const b = dev.createBuffer({ size: words.byteLength }); dev.queue.writeBuffer(b, 0, words); return b; };
  const cloudOf = (s) => ({ type: 'full', num_points: cfg.num_points, sh_deg: cfg.sh_deg, gaussian_3d_buffer: upload(s.gaussians), sh_buffer: upload(s.sh) });
  const tpc = cloudOf(target), tcam = dev.createBuffer({ size: 272 });
  const tfw = new hip.TiledForwardPass(dev, tpc, tcam, { viewportWidth: cfg.width, viewportHeight: cfg.height, renderMode: 'gaussian' });
  const trs = new hip.TiledRasterizer({ device: dev, forwardPass: tfw, format: 'rgba8unorm' });
  const cameras = [], images = [];
  for (const cam of cams) {
    dev.queue.writeBuffer(tcam, 0, cam);
    tfw.encode(null); trs.encode(null, cfg.width, cfg.height);
    const img = dev.createBuffer({ size: 4 * cfg.width * cfg.height });
    dev.createCommandEncoder().copyBufferToBuffer(trs.getOutputTextureView(), 0, img, 0, 4 * cfg.width * cfg.height);
    dev.synchronize();
    images.push({ texture: img, width: cfg.width, height: cfg.height });
    cameras.push({ camera: cam, width: cfg.width, height: cfg.height });
  }
  trs.destroy(); tfw.destroy(); tpc.gaussian_3d_buffer.destroy(); tpc.sh_buffer.destroy(); tcam.destroy();

  const random = seededRandom(1234);
  const trainer = new Trainer(dev, undefined, { random, viewsPerStep: vpr, lanes: args.lanes, pipelineDepth: args.pipelineDepth, worldSize: world, rank, exchange });
  const cloud = cloudOf(scene);
  trainer.setPointCloud(cloud);
  trainer.setDataset(cameras, images);
  trainer.setMaxIterations(1e9);
  // the headline leg measures the step itself (bench.py times the densify-inclusive loop in its own `sustained` leg)
  trainer.setDensifyPruneConfig({ schedule: { enabled: false } });
  trainer.start();
  for (let i = 0; i < args.warmup; i++) await trainer.step();
  await trainer.warmupCommandBuffers();   // every view's command buffer recorded before the clock starts
  trainer.drain();
  let stats = trainer.forwardPass.check();   // throws on tile-entry overflow

  // ---- snapshot: everything a block of steps changes, in device copies, so every block times the SAME K steps (a training scene drifts)
  trainer.flushPointCloud();
  const live = Object.assign({}, trainer.optimizer.getStateBuffers(), { gaussians: trainer.pointCloud.gaussian_3d_buffer, sh: trainer.pointCloud.sh_buffer });
  const copies = {};
  for (const k of Object.keys(live)) { copies[k] = dev.createBuffer({ size: live[k].size }); dev.createCommandEncoder().copyBufferToBuffer(live[k], 0, copies[k], 0, live[k].size); }
  const snap = { iteration: trainer.iteration, rng: random.state, optIteration: trainer.optimizer.getIteration() };
  const restore = () => {
    trainer.drain();
    trainer.flushPointCloud();
    for (const k of Object.keys(live)) addon.copyBufferToBuffer(dev.handle, live[k].ptr, copies[k].ptr, live[k].size);
    trainer.optimizer.stateChanged();                                            // the compact training copy is reloaded from the restored arrays
    if (trainer.deferredSH) trainer.optimizer.setDeferredSH(trainer.pointCloud, true);   // ... and the compact SH-DC halves from the restored rows
    trainer.optimizer.advanceIteration((snap.optIteration - trainer.optimizer.getIteration()) >>> 0);
    trainer.iteration = snap.iteration; random.state = snap.rng;
  };

  const oneBlock = async () => {
    barrier();
    const t0 = now();
    for (let i = 0; i < args.steps; i++) await trainer.step();
    trainer.drain();   // (pipeline depth 2: the last step's own await, with its deferred error check)
    barrier();
    return now() - t0;
  };
  const blocks = [await oneBlock()];
  // every rank must take the same number of blocks: rank 0 decides, the count travels through the barrier word's buffer
  let more = args.minSeconds <= 0 ? 0 : Math.min(400, Math.max(0, Math.ceil(args.minSeconds / Math.max(blocks[0], 1e-6)) - 1));
  if (world > 1) {
    const w = dev.createBuffer({ size: 4 });
    dev.queue.writeBuffer(w, 0, new Uint32Array([rank === 0 ? more : 0]));
    exchange.allreduceCounts(w, 1);
    more = new Uint32Array(dev.readBuffer(w, 4))[0];
    w.destroy();
  }
  for (let b = 0; b < more; b++) { restore(); blocks.push(await oneBlock()); }
  const sorted = blocks.slice().sort((x, y) => x - y), elapsed = sorted[(sorted.length - 1) >> 1];

  // the same K steps with the reference's own await inside every step (depth 1), for comparison; not the headline
  let awaitedMs = null;
  if (trainer.pipelineDepth > 1) {
    restore();
    trainer.pipelineDepth = 1;
    barrier();
    const t0 = now();
    for (let i = 0; i < args.steps; i++) await trainer.step();
    barrier();
    awaitedMs = (now() - t0) / args.steps * 1e3;
    trainer.pipelineDepth = args.pipelineDepth;
  }
  const emaItersPerSec = trainer.getItersPerSec();   // the reference's own meter (trainer.ts:647-651), over the awaited steps

  // ---- per-kernel durations: the same K steps launched eagerly with a hipEvent pair around every kernel
  let kernelMs = null;
  if (args.profile && world === 1) {
    restore();
    trainer.useCommandBuffers = false; trainer.invalidateCommandBuffers();
    await trainer.step();
    dev.setProfiling(true); dev.kernelTimes(true);
    for (let i = 0; i < args.steps; i++) await trainer.step();
    dev.synchronize();
    dev.setProfiling(false);
    const kt = dev.kernelTimes(false);
    kernelMs = {};
    for (const k of Object.keys(kt)) kernelMs[k] = Math.round(kt[k].totalMs / args.steps * 1e4) / 1e4;
    trainer.useCommandBuffers = true;
  }
  stats = trainer.forwardPass.check();

  // ---- BASELINE c3 "as written" through this host (bench.py: run_sustained): a fresh Trainer at the reference's densify defaults, wall clock around
  // every step() of a run that crosses the first densify events (--sustained-steps 608 = bench.py's default leg)
  let sustained = null;
  if (args.sustainedSteps > 0 && world === 1 && vpr === 1) {
    const st = new Trainer(dev, undefined, { random: seededRandom(99), pipelineDepth: args.pipelineDepth });
    st.setPointCloud(cloudOf(scene));
    st.setDataset(cameras, images);
    st.setMaxIterations(1e9);
    st.start();
    for (let i = 0; i < 3; i++) await st.step();
    await st.warmupCommandBuffers();
    st.drain(); dev.synchronize();
    const startIt = st.getIteration(), plain = [], events = [], sizes = [st.getPointCount()];
    const tAll = now();
    while (st.getIteration() < args.sustainedSteps) {
      const before = st.getLastDensifyPruneIteration(), t0 = now();
      await st.step();
      const dt = now() - t0;
      if (st.getLastDensifyPruneIteration() !== before) { events.push(dt); sizes.push(st.getPointCount()); } else plain.push(dt);
    }
    st.drain(); dev.synchronize();
    const total = now() - tAll, n = st.getIteration() - startIt;
    const sortedPlain = plain.slice().sort((a, b) => a - b), med = sortedPlain.length ? sortedPlain[Math.floor(sortedPlain.length / 2)] : 0;
    const rerecord = plain.reduce((acc, d) => acc + (d > 4 * med ? d - med : 0), 0);
    const r2 = (x) => Math.round(x * 100) / 100;
    sustained = { steps: n, crosses_iterations: [startIt, st.getIteration()], iters_per_s_overall: r2(n / total), iters_per_s_steady: med > 0 ? r2(1 / med) : null,
      ms_per_step_median: Math.round(med * 1e7) / 1e4, densify_events: events.length,
      ms_per_densify_event: events.length ? r2((events.reduce((a, b) => a + b, 0) + rerecord) / events.length * 1e3) : null, points: sizes, pipeline_depth: args.pipelineDepth,
      schedule: 'reference defaults: warm-up 500, interval 100, 10 metric views at 1/2 resolution, maxNewPointsPerStep 5000' };
    const sc = st.pointCloud; st.destroy(); sc.gaussian_3d_buffer.destroy(); sc.sh_buffer.destroy();
  }

  const msPerStep = elapsed / args.steps * 1e3, viewsPerStep = world * vpr;
  if (rank === 0) {
    const r4 = (x) => Math.round(x * 1e4) / 1e4;
    console.log(JSON.stringify({
      metric: args.config === 'c3' ? 'training iters/sec (fwd+bwd+Adam), 1M Gaussians @1080p SH3' : `training iters/sec (fwd+bwd+Adam), ${cfg.name}`,
      value: Math.round(viewsPerStep * args.steps / elapsed * 1e3) / 1e3, unit: 'iters/s', n_gpus: world, steps: args.steps, warmup: args.warmup,
      ms_per_step: r4(msPerStep), ms_per_step_awaiting_every_step: awaitedMs === null ? null : r4(awaitedMs), trainer_ema_iters_per_s: r4(emaItersPerSec),
      timed_blocks: { blocks: blocks.length, steps_per_block: args.steps, seconds_timed: r4(blocks.reduce((s, x) => s + x, 0)), reported: 'median block',
        ms_per_step_min: r4(sorted[0] / args.steps * 1e3), ms_per_step_max: r4(sorted[sorted.length - 1] / args.steps * 1e3) },
      higher_is_better: true, scaling: 'weak', vs_baseline: null, dtype: 'f32', data: 'synthetic',
      host: `node ${process.version}: bindings/ts/trainer.js over the N-API addon (bindings/napi/webdgs_napi.node) over libwebdgs_hip.so`,
      self_launched: process.env.WDGS_BENCH_SELF_LAUNCHED === '1', scene_generation_s: r4(genSeconds),
      config: { workload: `${cfg.name}: ${cfg.num_points} Gaussians, ${cfg.width}x${cfg.height}, SH deg ${cfg.sh_deg}, fwd+bwd per view, ${nDataset} circle views` +
          (viewsPerStep === 1 ? " (BASELINE c3: the reference's one-view step)" : ` (BASELINE c4 shape: ${vpr} views per rank per global step)`),
        views_per_rank: vpr, global_batch_views: viewsPerStep, lanes: trainer.lanes, pipeline_depth: args.pipelineDepth, tile_entries_E: stats.totalTileEntries,
        visible_V: stats.visibleCount, parallelism: world > 1 ? `dp${world}: views sharded; ${exchange.name}` : 'single GPU', densify_schedule: 'disabled in this leg' },
      kernel_ms_per_step: kernelMs, sustained, c3_as_written_iters_per_s: sustained ? sustained.iters_per_s_overall : null,
    }));
  }
  barrier();
  const last = trainer.pointCloud;
  trainer.destroy();
  exchange.destroy();
  last.gaussian_3d_buffer.destroy(); last.sh_buffer.destroy();
  for (const k of Object.keys(copies)) copies[k].destroy();
  for (const im of images) im.texture.destroy();
  barrierWord.destroy();
  dev.destroy();
}

main().catch((e) => { console.error(e); process.exit(1); });
