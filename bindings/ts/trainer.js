'use strict';
/*
 * trainer.js (+ trainer.d.ts) -- the reference's `Trainer` (src/trainer.ts:42-769) rewritten over the HIP operator classes of
 * webdgs_hip.js: same public surface (trainer.ts:177-566), same sequencing -- step() 568-660: pick a view, forward, rasterize,
 * backward, Adam, submit, await, it/s EMA, scheduled densify; runDensifyPruneMultiView() 373-497; applyPointCloudSwap() 201-237.
 *
 * Differences from the reference, all host-side (DESIGN.md): textures are linear rgba8 buffers; every training view owns a
 * resident 272-byte camera block (the reference rewrites one uniform buffer per step); a view's encodes are recorded once into a
 * replayable command buffer (HIP graph) instead of being re-encoded every step; the rasterizer's grid follows the current
 * viewport (SURVEY Q19); every metric view renders with its own camera because uploads are stream-ordered (Q12); the swap a densify
 * requests is applied inside the step that produced it (the reference defers it to the next animation frame, main.ts:587-593).
 * The view draws go through `this.random` (default Math.random, as trainer.ts:573 and 394), so a test can fix them.
 *
 * New, with no counterpart in the reference (whose step is one view on one GPU) and at the level of the Python host
 * (webdgs_amd/trainer.py): `viewsPerStep` > 1 -- a BATCHED step: K1 of all the step's views in one launch, the views' scan ... K16 dealt
 * to device lanes, K17 of all views in one launch into the step's fp32 gradient block, one Adam; `worldSize` > 1 -- view-sharded data
 * parallelism through parallel.js (reduce-scatter -> Adam on the owned slice -> all-gather, BASELINE config c4); `pipelineDepth` 2 --
 * step() resolves once the PREVIOUS step has finished (tickets: queue.mark / queue.wait), so the device never idles across the step
 * boundary; the single-view step runs K17 + Adam + re-pack as ONE kernel (Optimizer.stepWithGeometry).  Results are bit-identical to the
 * one-lane, depth-1, unfused forms (tests/test_gpu_napi.py compares this host with the Python host byte for byte).
 */
const hip = require('./webdgs_hip.js');
const parallel = require('./parallel.js');

const { mat4Inverse, projectionMatrix, cameraBlockFor } = require('./camera-math.js');

const DEFAULT_DENSIFY = {   // trainer.ts:147-164
  schedule: { enabled: true, warmupIterations: 500, interval: 100, stopIterations: 15000 },
  metricViews: 10, metricDownscale: 2, metricThreshold: 0.5, maxBufferBytes: 128 * 1024 * 1024, maxNewPointsPerStep: 5000,
  pruneOpacity: 0.01, cloneThresholdCount: 500, splitScaleThreshold: 1.0,
};

class Trainer {
  /** options: random, useCommandBuffers, maxTileEntries, reusePasses, deferredSH, fuseGeometryAdam, keepGradients, pipelineDepth (1..4),
   *  viewsPerStep (views per rank per global step), lanes (device lanes of a batched step; default 3), batchViews (view-batched K1 / K17;
   *  default on), worldSize / rank / exchange (parallel.js; default: from the environment). */
  constructor(device, trainingConfig, options) {
    const o = options || {};
    this.device = device;
    this.trainingConfig = Object.assign({ lambda_l1: 0.8, lambda_l2: 0.0, lambda_dssim: 0.2 }, trainingConfig || {});  // trainer.ts:100-104
    this.optimizerHyperparameters = Object.assign({}, hip.DEFAULT_ADAM_HYPERPARAMETERS);
    this.random = o.random || Math.random;
    this.useCommandBuffers = o.useCommandBuffers !== false;
    this.maxTileEntries = o.maxTileEntries || 0;
    this.grownTileEntries = 0;   // (sizing left to the library only: what an overflow has made of the lists, stepViews)
    this.reusePasses = o.reusePasses !== false;   // applyPointCloudSwap resizes the passes instead of rebuilding them
    // Adam writes the trained SH-DC halves to a compact array that K1 reads instead of 6 bytes into every 96-byte SH row (Optimizer.setDeferredSH);
    // the rows are flushed at hand-over points (flushPointCloud; host reads and forward passes built on the cloud follow by themselves).
    this.deferredSH = o.deferredSH !== false;
    this.fuseGeometryAdam = o.fuseGeometryAdam !== false;   // the single-view step runs K17, Adam and the re-pack as one kernel
    this.keepGradients = !!o.keepGradients;                 // ... and still fills backwardPass.getGradientsBuffer() (nothing in the trainer reads it)
    this.gradientOutputApplied = null;
    this.pipelineDepth = Math.max(1, Math.min(Math.floor(o.pipelineDepth || 1), 4));
    this.tickets = [];
    this.worldSize = Math.max(1, Math.floor(o.worldSize || 1)); this.rank = Math.floor(o.rank || 0);
    this.viewsPerRank = Math.max(1, Math.floor(o.viewsPerStep || o.viewsPerRank || 1));
    this.exchange = o.exchange || new parallel.Exchange();
    if (this.worldSize > 1 && (this.exchange.worldSize !== this.worldSize || this.exchange.rank !== this.rank)) {
      throw new Error(`exchange is rank ${this.exchange.rank} of ${this.exchange.worldSize}, trainer is rank ${this.rank} of ${this.worldSize}`);
    }
    // sliced step (reduce-scatter / owned-slice Adam / all-gather): any real exchange, also a forced one in a world of one
    this.sliced = this.worldSize > 1 || !!this.exchange.force;
    const lanes = o.lanes === undefined || o.lanes === null || o.lanes === 0 ? Trainer.DEFAULT_LANES : o.lanes;
    this.lanes = Math.max(1, Math.min(Math.floor(lanes), this.viewsPerRank, hip.MAX_LANES));
    this.batchViews = this.viewsPerRank > 1 && o.batchViews !== false;
    this.opSets = this.batchViews ? Math.min(this.viewsPerRank, hip.MAX_BATCH_VIEWS) : this.lanes;
    this.moreOpSets = [];   // [forwardPass, rasterizer, backwardPass] of op sets 1.. (set 0 is the three below)
    // The metric views of a densify event are independent until normalizeMetricCounts (integer atomics: any order gives the same bits): they are
    // dealt to `metricLanes` op sets, each on a device lane of its own, all adding into set 0's counts (TiledBackwardPass.setMetricCountsTarget).
    this.metricLanes = Math.max(1, Math.min(Math.floor(o.metricLanes || Trainer.DEFAULT_LANES), hip.MAX_LANES));
    this.longLists = null;      // long tile lists (csrc/longlist.h): null = the library's defaults; { threshold, maxItems, maxRows } for the passes this trainer builds
    this.moreMetricSets = [];   // [forwardPass, rasterizer, metricsPass, target, cameraBuffer] of metric lanes 1..
    this.dpGrad = null; this.dpVisible = null; this.dpRows = null; this.dpFlag = null; this.stateSliced = false;
    this.dcWords = null;
    this.forwardPass = null; this.rasterizer = null; this.backwardPass = null; this.optimizer = null; this.pointCloud = null;
    this.metricsForwardPass = null; this.metricsRasterizer = null; this.metricsPass = null;
    this.metricsViewportWidth = 0; this.metricsViewportHeight = 0; this.metricsTarget = null;
    this.metricsCameraBuffer = device.createBuffer({ size: 272, label: 'metrics camera uniform' });
    this.densifyPruneConfig = JSON.parse(JSON.stringify(DEFAULT_DENSIFY));
    this.densifyPrune = new hip.DensifyPrunePass(device, this.densifyOpConfig());
    this.isTraining = false; this.iteration = 0; this.maxIterations = 10000; this.stepItersPerSec = 0; this.stepMs = 0;
    this.lastDensifyPruneIteration = null; this.lastViewportWidth = 1; this.lastViewportHeight = 1; this.pendingPointCloudSwap = null;
    this.trainCameras = []; this.images = []; this.cameraBuffers = [];
    this.commandBuffers = new Map(); this.eagerSteps = 0;
  }

  densifyOpConfig() {
    const c = this.densifyPruneConfig;
    return { strategy: 'gpu_rebuild', numViews: c.metricViews, maxBufferBytes: c.maxBufferBytes, maxNewPointsPerStep: c.maxNewPointsPerStep,
      pruneThreshold: c.pruneOpacity, cloneThreshold: c.cloneThresholdCount, splitThreshold: c.splitScaleThreshold };
  }

  // ---- trainer.ts:177-247
  setPointCloud(pointCloud) { this.applyPointCloudSwap({ pointCloud }); }
  requestPointCloudSwap(pointCloud, optimizerInitialState) { this.pendingPointCloudSwap = { pointCloud, optimizerInitialState }; }
  consumePointCloudSwapRequest() { const r = this.pendingPointCloudSwap; this.pendingPointCloudSwap = null; return r; }
  requestResizeTo(numPoints) {
    if (!this.pointCloud) return;
    this.requestPointCloudSwap(hip.allocatePointCloudLike(this.device, this.pointCloud, { numPoints }));
  }
  destroyMoreOpSets() { for (const set of this.moreOpSets) for (const op of set) op.destroy(); this.moreOpSets = []; }
  destroyMoreMetricSets() { for (const set of this.moreMetricSets) for (const op of set) op.destroy(); this.moreMetricSets = []; }
  forwardPasses() { return [this.forwardPass, this.metricsForwardPass].concat(this.moreOpSets.map((m) => m[0]), this.moreMetricSets.map((m) => m[0])).filter((p) => p); }
  applyPointCloudSwap(request) {   // trainer.ts:201-237
    this.drain();
    this.synchronize();
    const oldParams = this.optimizer ? this.optimizer.getHyperparameters() : null;
    this.invalidateCommandBuffers();
    if (this.optimizer) { this.optimizer.destroy(); this.optimizer = null; }
    const old = this.pointCloud;
    this.pointCloud = request.pointCloud;
    // The reference destroys every pass and constructs new ones; the passes here can follow a cloud of another size
    // (setPointCloud: buffers reused, or re-allocated with headroom), so only the optimizer -- which adopts the rebuilt state -- is new.
    const passes = [this.forwardPass, this.backwardPass, this.metricsForwardPass, this.metricsPass];
    for (const m of this.moreOpSets.concat(this.moreMetricSets)) passes.push(m[0], m[2]);
    const kept = this.reusePasses && old && !this.recreateBackward && passes.filter((p) => p).every((p) => p.setPointCloud(this.pointCloud));
    if (!kept) {
      for (const name of ['forwardPass', 'rasterizer', 'backwardPass', 'metricsForwardPass', 'metricsRasterizer', 'metricsPass']) {
        if (this[name]) this[name].destroy();
        this[name] = null;
      }
      this.destroyMoreOpSets();
      this.destroyMoreMetricSets();
      this.gradientOutputApplied = null;
    }
    this.optimizer = new hip.Optimizer(this.device, this.pointCloud, oldParams || this.optimizerHyperparameters, request.optimizerInitialState);
    this.optimizerHyperparameters = this.optimizer.getHyperparameters();
    this.dcWords = this.deferredSH ? this.optimizer.setDeferredSH(this.pointCloud, true) : null;
    for (const fw of this.forwardPasses()) fw.setDcSource(this.dcWords);
    if (old && old !== this.pointCloud) { old.gaussian_3d_buffer.destroy(); old.sh_buffer.destroy(); }
    for (const name of ['dpGrad', 'dpVisible', 'dpRows', 'dpFlag']) { if (this[name]) this[name].destroy(); this[name] = null; }
    this.stateSliced = false;
    this.ensurePipelines(this.lastViewportWidth, this.lastViewportHeight);
  }
  /** Brings pointCloud.sh_buffer up to date with what has been trained (deferred SH writes).  Host reads of the buffer and forward passes built
   *  on the cloud (a Viewer's) follow by themselves; call this before a device-side reader of the raw rows that is neither. */
  flushPointCloud() { if (this.optimizer && this.pointCloud) this.optimizer.flushSH(this.pointCloud); }
  /** cameras[i] pairs with images[i] (trainer.ts:575-577).  Accepted: ready entries { camera: Float32Array(68), width, height } /
   *  { texture: HipBuffer, width, height }, or the loaders' own shapes -- CameraData (loaders.js; the 68-float block is then built for the
   *  image size as Camera.set_preset + update_buffer do, trainer.ts:583-586) and LoadedImage (images.js: { bitmap, width, height }). */
  setDataset(cameras, images) {
    this.drain();
    for (const t of this.ownedTextures || []) t.destroy();   // (the ones a previous call uploaded itself; a caller's textures stay the caller's)
    this.ownedTextures = [];
    const imgs = images.map((im) => {
      if (im.texture) return im;
      const tex = this.device.createBuffer({ size: 4 * im.width * im.height, label: 'gt image' });
      this.device.queue.writeBuffer(tex, 0, im.bitmap);
      this.ownedTextures.push(tex);
      return Object.assign({}, im, { texture: tex });
    });
    const cams = cameras.map((c, i) => (c.camera ? c : Object.assign({}, c, { camera: require('./loaders.js').cameraUniforms(c, imgs[i].width, imgs[i].height) })));
    this.trainCameras = cams; this.images = imgs;
    for (const b of this.cameraBuffers) b.destroy();
    this.cameraBuffers = this.trainCameras.map((c) => {   // one resident 272-byte block per view (the reference rewrites a single uniform buffer)
      const b = this.device.createBuffer({ size: 272, label: 'camera uniform' });
      this.device.queue.writeBuffer(b, 0, c.camera);
      return b;
    });
    this.invalidateCommandBuffers();
  }
  getTrainingConfig() { return Object.assign({}, this.trainingConfig); }
  setTrainingConfig(next) { Object.assign(this.trainingConfig, next); this.invalidateCommandBuffers(); this.recreateBackward = true; }
  getOptimizerHyperparameters() { return this.optimizer ? this.optimizer.getHyperparameters() : Object.assign({}, this.optimizerHyperparameters); }
  setOptimizerHyperparameters(next) {
    Object.assign(this.optimizerHyperparameters, next); this.invalidateCommandBuffers();
    if (this.optimizer) this.optimizer.setHyperparameters(next);
  }
  setDensifyPruneConfig(next) {
    const schedule = Object.assign({}, this.densifyPruneConfig.schedule, next.schedule || {});
    this.densifyPruneConfig = Object.assign({}, this.densifyPruneConfig, next, { schedule });
    this.densifyPrune.setConfig(this.densifyOpConfig());
  }
  start() {   // trainer.ts:499-511
    if (!this.pointCloud || this.trainCameras.length === 0) { console.log('Cannot start training: Missing point cloud or dataset.'); return; }
    this.isTraining = true; this.iteration = 0; this.stepItersPerSec = 0; this.stepMs = 0; this.lastDensifyPruneIteration = null;
    // the lanes this trainer will use come into being now (the library creates a lane's stream and event at its first use, ~5 ms each: otherwise
    // the first densify event, which is the first to touch the metric lanes, pays for them)
    for (let k = 1; k < Math.max(this.lanes, this.metricLanes); k++) { this.device.laneOrder(k, 0); this.device.laneOrder(0, k); }
  }
  stop() { this.isTraining = false; if (this.device.handle !== null && this.tickets.length) this.drain(); }
  getIsTraining() { return this.isTraining; }
  setMaxIterations(n) { this.maxIterations = Math.max(1, Math.floor(n)); }
  getMaxIterations() { return this.maxIterations; }
  getIteration() { return this.iteration; }
  getPointCount() { return this.pointCloud ? this.pointCloud.num_points : 0; }
  getLastStepMs() { return this.stepMs; }
  getItersPerSec() { return this.stepItersPerSec; }
  getLastDensifyPruneIteration() { return this.lastDensifyPruneIteration; }
  getNextDensifyPruneIteration() {   // trainer.ts:546-566
    const s = this.densifyPruneConfig.schedule;
    if (!s.enabled) return null;
    const warmup = s.warmupIterations, interval = Math.max(1, s.interval), stop = s.stopIterations, i = this.iteration;
    if (i >= stop) return null;
    if (i < warmup) return Math.min(warmup, stop);
    const next = warmup + Math.ceil((i + 1 - warmup) / interval) * interval;
    return next <= stop ? next : null;
  }

  /** Recorded kernels bake pointers, viewport, hyper-parameters and loss weights: any change drops the recordings (after the steps still in
   *  flight, which replay them, have finished). */
  invalidateCommandBuffers() {
    let deferred = null;
    if (this.tickets.length && this.device.handle !== null) {
      this.tickets = [];
      try { this.synchronize(); } catch (e) { deferred = e; }
    }
    for (const c of this.commandBuffers.values()) c.destroy();
    this.commandBuffers.clear();
    this.eagerSteps = 0;
    if (deferred) throw deferred;
  }

  /** A forward pass of this trainer: the deferred SH-DC source, and the long-list settings if any were given (this.longLists; null = the library's defaults). */
  newForwardPass(cam, w, h) {
    const fw = new hip.TiledForwardPass(this.device, this.pointCloud, cam, { viewportWidth: w, viewportHeight: h, renderMode: 'gaussian', maxTileEntries: this.tileEntries() });
    fw.setDcSource(this.dcWords);
    if (this.longLists) fw.setLongLists(this.longLists.threshold === undefined ? 2048 : this.longLists.threshold, this.longLists.maxItems || 0, this.longLists.maxRows || 0);
    return fw;
  }

  /** Long tile lists (csrc/longlist.h) work in scratch of a fixed size; a frame whose long tiles find no room is composited the ordinary way -- correct, but
   *  as slow as its longest list.  The work's header says what the last frame wanted: looked at where the host waits anyway (a densify event), and every
   *  pass is given room for 1.5 x that -- up to longLists.maxItemsCap / maxRowsCap (8 192 chunk slots: 33 000 entries of long tiles; 65 536 rows).  A
   *  frame that wants more than the caps is FULL of long tiles (a dense cloud at a small viewport): the path is not for it (longlist.h: ll_frame_on) and
   *  the scratch is left alone.  (Command buffers recorded against the old scratch are dropped.) */
  growLongLists() {
    const ll = this.longLists || {};
    const capItems = ll.maxItemsCap === undefined ? 8192 : ll.maxItemsCap, capRows = ll.maxRowsCap === undefined ? 65536 : ll.maxRowsCap;
    let haveItems = 0, haveRows = 0, needItems = 0, needRows = 0;
    for (const fw of this.forwardPasses()) {
      const st = fw.longListStats();
      if (!st.threshold) continue;
      if (st.stalled) console.warn('a long-list task gave up waiting (code 0x' + st.stalled.toString(16) + '): the frame\'s long tiles are not to be trusted');
      haveItems = Math.max(haveItems, st.maxItems); haveRows = Math.max(haveRows, st.maxRows);
      if (st.itemsWanted <= capItems) { needItems = Math.max(needItems, st.itemsWanted); needRows = Math.max(needRows, st.rowsWanted); }
    }
    const items = needItems > haveItems ? Math.max(haveItems, Math.min(Math.floor(needItems * 1.5), capItems)) : haveItems;
    const rows = needRows > haveRows ? Math.max(haveRows, Math.min(Math.floor(needRows * 1.5), capRows)) : haveRows;
    if (items === haveItems && rows === haveRows) return;
    this.longLists = Object.assign({}, ll, { maxItems: items, maxRows: rows });
    console.warn('long-list scratch enlarged to ' + this.longLists.maxItems + ' chunk slots and ' + this.longLists.maxRows + ' rows');
    this.invalidateCommandBuffers();
    for (const fw of this.forwardPasses()) fw.setLongLists(this.longLists.threshold === undefined ? 2048 : this.longLists.threshold, this.longLists.maxItems, this.longLists.maxRows);
  }

  newOpSet(w, h) {
    const fw = this.newForwardPass(this.cameraBuffers.length ? this.cameraBuffers[0] : this.metricsCameraBuffer, w, h);
    return [fw, new hip.TiledRasterizer({ device: this.device, forwardPass: fw, format: 'rgba8unorm' }),
      new hip.TiledBackwardPass(this.device, this.pointCloud, { viewportWidth: w, viewportHeight: h, trainingConfig: this.trainingConfig })];
  }

  ensurePipelines(width, height) {   // trainer.ts:662-692 (the rasterizer follows the viewport here: SURVEY Q19)
    const w = Math.max(1, Math.floor(width)), h = Math.max(1, Math.floor(height));
    if (w !== this.lastViewportWidth || h !== this.lastViewportHeight) this.invalidateCommandBuffers();
    this.lastViewportWidth = w; this.lastViewportHeight = h;
    const cam = this.cameraBuffers.length ? this.cameraBuffers[0] : this.metricsCameraBuffer;
    if (!this.forwardPass) {
      this.forwardPass = this.newForwardPass(cam, w, h);
    } else this.forwardPass.setViewport(w, h);
    if (!this.rasterizer) this.rasterizer = new hip.TiledRasterizer({ device: this.device, forwardPass: this.forwardPass, format: 'rgba8unorm' });
    if (!this.backwardPass || this.recreateBackward) {
      if (this.backwardPass) this.backwardPass.destroy();
      this.backwardPass = new hip.TiledBackwardPass(this.device, this.pointCloud, { viewportWidth: w, viewportHeight: h, trainingConfig: this.trainingConfig });
      for (const m of this.moreOpSets) { m[2].destroy(); m[2] = new hip.TiledBackwardPass(this.device, this.pointCloud, { viewportWidth: w, viewportHeight: h, trainingConfig: this.trainingConfig }); }
      this.recreateBackward = false; this.gradientOutputApplied = null;
    } else this.backwardPass.setViewport(w, h);
    for (const m of this.moreOpSets) { m[0].setViewport(w, h); m[2].setViewport(w, h); }
    while (this.moreOpSets.length < this.opSets - 1) this.moreOpSets.push(this.newOpSet(w, h));
    // a step whose tile-entry list overflowed is skipped on the device and reported by the next synchronize()
    if (this.optimizer && this.worldSize * this.viewsPerRank === 1) this.optimizer.setGuard(this.forwardPass.getStatsBuffer(), 8);
  }

  ensureMetricsPipelines(baseWidth, baseHeight) {   // trainer.ts:330-371
    const down = Math.max(1, Math.floor(this.densifyPruneConfig.metricDownscale));
    const w = Math.max(1, Math.floor(baseWidth / down)), h = Math.max(1, Math.floor(baseHeight / down));
    if (this.metricsForwardPass && this.metricsViewportWidth === w && this.metricsViewportHeight === h) return { width: w, height: h };
    for (const name of ['metricsForwardPass', 'metricsRasterizer', 'metricsPass']) { if (this[name]) this[name].destroy(); this[name] = null; }
    this.destroyMoreMetricSets();
    if (this.metricsTarget) this.metricsTarget.destroy();
    this.metricsViewportWidth = w; this.metricsViewportHeight = h;
    this.metricsForwardPass = this.newForwardPass(this.metricsCameraBuffer, w, h);
    this.metricsRasterizer = new hip.TiledRasterizer({ device: this.device, forwardPass: this.metricsForwardPass, format: 'rgba8unorm' });
    this.metricsPass = new hip.TiledBackwardPass(this.device, this.pointCloud, { viewportWidth: w, viewportHeight: h, trainingConfig: this.trainingConfig });
    this.metricsTarget = this.device.createBuffer({ size: 4 * w * h, label: 'metrics-gt-downsampled' });
    return { width: w, height: h };
  }

  /** [forwardPass, rasterizer, metricsPass, downsampled-GT buffer, camera buffer] of metric lane k; sets 1.. are built on first use. */
  metricSet(k) {
    if (k === 0) return [this.metricsForwardPass, this.metricsRasterizer, this.metricsPass, this.metricsTarget, this.metricsCameraBuffer];
    const w = this.metricsViewportWidth, h = this.metricsViewportHeight;
    while (this.moreMetricSets.length < k) {
      const cam = this.device.createBuffer({ size: 272, label: 'metrics camera uniform' });
      const fw = this.newForwardPass(cam, w, h);
      this.moreMetricSets.push([fw, new hip.TiledRasterizer({ device: this.device, forwardPass: fw, format: 'rgba8unorm' }),
        new hip.TiledBackwardPass(this.device, this.pointCloud, { viewportWidth: w, viewportHeight: h, trainingConfig: this.trainingConfig }),
        this.device.createBuffer({ size: 4 * w * h, label: 'metrics-gt-downsampled' }), cam]);
    }
    return this.moreMetricSets[k - 1];
  }

  opsOf(opSet) { return opSet > 0 ? this.moreOpSets[opSet - 1] : [this.forwardPass, this.rasterizer, this.backwardPass]; }

  /** trainer.ts:606-628 for one view on one op set.  geometry = false: the view ends with K16 (K17 follows separately: fused with Adam, or batched);
   *  projected = true: K1 ran for all the views of the step at once (hip.projectViews), scan / emit / sort remain. */
  encodeView(encoder, index, opSet, geometry, projected) {
    const ops = this.opsOf(opSet || 0), forwardPass = ops[0], rasterizer = ops[1], backwardPass = ops[2];
    const image = this.images[index], cam = this.cameraBuffers[index];
    forwardPass.setCameraBuffer(cam);
    if (projected) forwardPass.encodeProjected(encoder); else forwardPass.encode(encoder);
    rasterizer.encode(encoder, image.width, image.height);
    const res = { splatBuffer: forwardPass.getResources().splatBuffer, tileOffsetsBuffer: rasterizer.getTileOffsetsBuffer(),
      tileIndicesBuffer: forwardPass.getSortedIndicesBuffer(), cameraBuffer: cam,
      alphaTexture: rasterizer.getAlphaTextureView(), nContribTexture: rasterizer.getNContribTextureView() };
    if (geometry !== false) backwardPass.encode(encoder, rasterizer.getOutputTextureView(), image.texture, res);
    else backwardPass.encodeRaster(encoder, rasterizer.getOutputTextureView(), image.texture, res);
  }

  /** Submits the command buffer recorded under `key`; the first time, `encode(encoder)` is encoded -- eagerly while the pipelines still
   *  allocate on first use, into a recorded command buffer (HIP graph) afterwards.  True if it was replayed. */
  run(key, encode) {
    let cmd = this.commandBuffers.get(key);
    if (cmd) { this.device.queue.submit([cmd]); return true; }
    const record = this.useCommandBuffers && this.eagerSteps >= 1;
    const encoder = this.device.createCommandEncoder({ label: 'trainer-' + key, record });
    try {
      encode(encoder);
      cmd = encoder.finish();
    } catch (e) {
      encoder.abort();
      throw e;
    }
    if (record) this.commandBuffers.set(key, cmd);
    this.device.queue.submit([cmd]);
    return false;
  }

  /** Records every view's command buffers up front (the first pass over a dataset does this anyway; calling it before a timed region keeps
   *  recording out of it).  The number of steps taken is a function of the dataset size ONLY: under data parallelism every step is a
   *  collective, so all ranks must take the same number of them. */
  async warmupCommandBuffers() {
    if (!this.useCommandBuffers || !this.isTraining || !this.pointCloud) return 0;
    const nViews = this.worldSize * this.viewsPerRank;
    let taken = 0;
    for (let v = 0; v < this.trainCameras.length; v++) {
      for (let r = 0; r < (v === 0 ? 2 : 1); r++) { await this.stepViews(new Array(nViews).fill(v)); taken++; }
    }
    return taken;
  }

  /** One training iteration (trainer.ts:568-660): the views are drawn with this.random, as the reference picks Math.random() per step. */
  async step() { return this.stepViews(undefined); }

  /** step() on a given global batch of views (worldSize * viewsPerStep indices; with worldSize > 1 each rank takes its shard).
   *  Tile-entry capacity: with maxTileEntries left at 0 the forward passes size their entry lists from the cloud (30 entries per Gaussian, at
   *  least 2^20); a cloud that training has thinned out and whose survivors have grown can outrun that (c3 does, after ~3 000 iterations of
   *  the default schedule).  The reference truncates such a list silently; the library skips the step on the device and reports it.  The
   *  Trainer then doubles the lists (growTileEntryCapacity), warns, and training goes on -- the step or two that were skipped are lost
   *  iterations.  A capacity the caller pinned is never touched: the error is the caller's. */
  async stepViews(viewIds) {
    try {
      await this.stepViewsOnce(viewIds);
    } catch (e) {
      if (!(e && e.code === 'WDGS_E_CAPACITY' && this.growTileEntryCapacity(e))) throw e;
    }
  }

  /** maxTileEntries for a new forward pass: the caller's, or what growTileEntryCapacity has arrived at (0 = the library's own sizing). */
  tileEntries() { return this.maxTileEntries || this.grownTileEntries; }

  growTileEntryCapacity(error) {
    if (this.maxTileEntries !== 0 || this.device.handle === null) return false;
    const mine = this.ownOverflow(error) || [];   // (a report about other owners' passes only never gets here: wait / synchronize)
    let now = Math.max(this.grownTileEntries, 1 << 20);
    for (const fw of this.forwardPasses()) now = Math.max(now, fw.getResources().maxTileEntries);
    const next = Math.min(Math.max(2 * now, mine.length ? Math.floor(Math.max.apply(null, mine) * 1.5) : 0), 0xFFFFF000);
    if (next <= now) return false;
    console.warn(`tile-entry lists grown from ${now} to ${next} entries after an overflow (${error.message}); the step that overflowed was skipped`);
    this.grownTileEntries = next;
    this.tickets = [];
    try { this.device.synchronize(); } catch (_e) { /* a step still in flight overflowed as well */ }
    this.invalidateCommandBuffers();
    // forward passes own the lists: every pass set is rebuilt around lists of the new size (as a cloud the passes cannot follow rebuilds them)
    for (const name of ['forwardPass', 'rasterizer', 'backwardPass', 'metricsForwardPass', 'metricsRasterizer', 'metricsPass']) {
      if (this[name]) this[name].destroy();
      this[name] = null;
    }
    this.destroyMoreOpSets();
    this.destroyMoreMetricSets();
    this.gradientOutputApplied = null;
    this.ensurePipelines(this.lastViewportWidth, this.lastViewportHeight);
    for (const fw of this.forwardPasses()) fw.setDcSource(this.dcWords);
    return true;
  }

  async stepViewsOnce(viewIds) {
    if (!this.isTraining || !this.pointCloud) return;
    const t0 = process.hrtime();
    const nViews = this.worldSize * this.viewsPerRank;
    if (!viewIds) { viewIds = []; for (let i = 0; i < nViews; i++) viewIds.push(Math.floor(this.random() * this.trainCameras.length)); }
    const mine = parallel.shardViews(viewIds, this.rank, this.worldSize);
    const image = this.images[mine[0]];
    this.ensurePipelines(image.width, image.height);

    const s = this.densifyPruneConfig.schedule;   // trainer.ts:593-601: checked on iteration + 1
    const nextIteration = this.iteration + 1, warmup = s.warmupIterations, interval = Math.max(1, s.interval), stop = s.stopIterations;
    const shouldDensify = s.enabled && nextIteration >= warmup && nextIteration <= stop && (nextIteration === warmup || (nextIteration - warmup) % interval === 0);

    try {
      if (nViews === 1) this.stepSingleView(mine[0]); else this.stepBatched(mine);
    } catch (e) {
      // a failed encode must not leave the stream in capture mode or half-recorded command buffers behind
      if (this.device.handle !== null) hip.addon.encoderAbort(this.device.handle);
      try { this.invalidateCommandBuffers(); } catch (_deferred) { /* this step's error stays the one raised */ }
      this.tickets = [];
      throw e;
    }
    try {
      await this.finishStep();
    } catch (e) {
      this.tickets = [];
      throw e;
    }

    this.iteration += 1;
    const dt = process.hrtime(t0);
    this.stepMs = dt[0] * 1e3 + dt[1] / 1e6;
    const inst = this.stepMs > 0 ? 1000 / this.stepMs : 0;
    this.stepItersPerSec = this.stepItersPerSec === 0 ? inst : this.stepItersPerSec * 0.9 + inst * 0.1;   // trainer.ts:647-651
    if (shouldDensify) {
      this.drain();
      this.growLongLists();
      await this.runDensifyPruneMultiView();
      const req = this.consumePointCloudSwapRequest();
      if (req) this.applyPointCloudSwap(req);
    }
    if (this.iteration >= this.maxIterations) this.stop();
  }

  /** The entries needed by THIS trainer's passes among those a capacity report names (csrc/api.hip: deferred_checks names every pass that overflowed),
   *  [] if it names only other owners' passes, null if it names none (a step skipped on every rank). */
  ownOverflow(error) {
    const named = [], re = /(\d+) entries needed, max_tile_entries = \d+ \(forward pass (0x[0-9a-fA-F]+)\)/g, text = String(error && error.message);
    for (let m = re.exec(text); m; m = re.exec(text)) named.push([Number(m[1]), BigInt(m[2])]);
    if (!named.length) return null;
    const own = this.forwardPasses().map((fw) => BigInt(fw.handle));
    return named.filter((n) => own.some((h) => h === n[1])).map((n) => n[0]);
  }
  /** true (after saying so once) for a capacity report about passes this trainer does not own -- a Viewer rendering the same cloud on this device: the
   *  report is device-wide, whoever waits first gets it, and it is the pass's owner who has to enlarge its lists. */
  notOurs(error) {
    if (!(error && error.code === 'WDGS_E_CAPACITY')) return false;
    const mine = this.ownOverflow(error);
    if (mine === null || mine.length) return false;
    this.device.capacityReports.post(error);   // (for the passes' owner: it looks there at its own next wait)
    if (!this.foreignOverflowWarned) console.warn(`a forward pass that is not this trainer's overflowed its tile-entry lists (${error.message}); its owner has to enlarge them`);
    this.foreignOverflowWarned = true;
    return true;
  }
  wait(ticket) { try { this.device.queue.wait(ticket); } catch (e) { if (!this.notOurs(e)) throw e; } this.reportsLeftForUs(); }
  synchronize() { try { this.device.synchronize(); } catch (e) { if (!this.notOurs(e)) throw e; } this.reportsLeftForUs(); }
  /** A report about THIS trainer's passes that another owner's wait consumed (a Viewer reading its frame): raised here, as if this wait had got it. */
  reportsLeftForUs() {
    if (!this.device.capacityReports.pending.length) return;
    const e = this.device.capacityReports.take(this.forwardPasses().map((fw) => fw.handle));
    if (e) throw e;
  }

  /** `await onSubmittedWorkDone()` (trainer.ts:639-645) + the deferred capacity check.  Depth 1: this step's own completion, through the
   *  Promise, as the reference awaits it.  Depth d > 1: a ticket for this step is kept and the step d - 1 submissions ago is awaited. */
  async finishStep() {
    if (this.pipelineDepth <= 1) {
      await this.device.queue.onSubmittedWorkDone();
      this.synchronize();   // deferred device-side checks (tile-entry overflow) surface here as a thrown Error
      return;
    }
    this.tickets.push(this.device.queue.mark());
    while (this.tickets.length >= this.pipelineDepth) this.wait(this.tickets.shift());
  }
  /** Awaits every step still in flight (a no-op at pipelineDepth 1). */
  drain() { const t = this.tickets; this.tickets = []; for (const ticket of t) this.wait(ticket); }

  applyGradientOutput() {
    const want = this.keepGradients || !this.fuseGeometryAdam;
    if (this.gradientOutputApplied === want) return;
    if (this.gradientOutputApplied !== null) this.invalidateCommandBuffers();   // the recorded fused step baked the old choice
    this.backwardPass.setGradientOutput(want);
    this.gradientOutputApplied = want;
  }

  /** The reference's step (trainer.ts:603-645): one view, Adam straight from the packed fp16 gradients. */
  stepSingleView(view) {
    this.applyGradientOutput();
    const tileCounts = this.forwardPass.getResources().tileCountsBuffer;
    const replayed = this.run('step/' + view, (encoder) => {
      if (this.fuseGeometryAdam) {   // K1..K16, then K17 + Adam + re-pack in one pass over the Gaussians
        this.encodeView(encoder, view, 0, false, false);
        this.optimizer.stepWithGeometry(encoder, this.pointCloud, this.backwardPass, this.cameraBuffers[view], tileCounts);
      } else {
        this.encodeView(encoder, view, 0, true, false);
        this.optimizer.step(encoder, this.pointCloud, this.backwardPass.getGradientsBuffer(), tileCounts);
      }
    });
    if (replayed) this.optimizer.advanceIteration(1);
    else if (!this.useCommandBuffers || this.eagerSteps < 1) this.eagerSteps += 1;
  }

  /** [views -> fp32 block] -> exchange -> [Adam on the owned slice] -> all-gather -> [apply the other ranks' rows]. */
  stepBatched(mine) {
    const n = this.pointCloud.num_points, w = this.worldSize, sl = parallel.slicePoints(n, w), dev = this.device;
    if (!this.dpGrad) {   // (allocated before any recording is opened; sized world * slice so the collectives run in place)
      this.dpGrad = dev.createBuffer({ size: 4 * parallel.GRAD_FLOATS * w * sl, label: 'dp-grad-f32' });
      this.dpVisible = dev.createBuffer({ size: 4 * w * sl, label: 'dp-visible' });
      this.dpFlag = dev.createBuffer({ size: 16, label: 'dp-guard' });
      this.dpRows = this.sliced ? dev.createBuffer({ size: 32 * w * sl, label: 'dp-repacked-rows' }) : null;
      this.optimizer.setGuard(this.dpFlag, 0);
    }
    const own = parallel.ownedRange(n, w, this.rank);
    const eagerBefore = this.eagerSteps;
    if (this.batchViews && mine.length <= this.opSets) this.viewsBatched(mine); else this.viewsOneByOne(mine);
    this.exchange.exchangeGradients(this.dpGrad, this.dpVisible, this.dpFlag, sl);
    if (this.run('adam', (encoder) => this.optimizer.stepF32Range(encoder, this.pointCloud, this.dpGrad, this.dpVisible, own.first, own.count, this.dpRows))) {
      this.optimizer.advanceIteration(1);
    }
    if (this.sliced) {
      this.exchange.allgatherRows(this.dpRows, sl);
      this.run('apply', (_encoder) => this.optimizer.applyRepackedRows(this.dpRows, own.first, own.count, this.dpFlag, this.pointCloud));
      this.stateSliced = w > 1;
    }
    if (!this.useCommandBuffers || eagerBefore < 1) this.eagerSteps += 1;
  }

  /** [K1 of all the views, one launch] -> per view, dealt to the lanes: scan, emit, sort, composite, loss, backward raster (one recorded command
   *  buffer per (view, place in the batch)) -> [K17 of all the views, one launch, into the step's fp32 block].  The batched launches run on a
   *  lane of their own; each view waits only for the projection, K17 for every view. */
  viewsBatched(mine) {
    const dev = this.device, L = this.lanes;
    const sets = mine.map((_v, k) => this.opsOf(k));
    const cams = mine.map((v) => this.cameraBuffers[v]);
    const lanes = L > 1 && this.useCommandBuffers && mine.every((v, k) => this.commandBuffers.has(`viewp/${v}/${k}`));
    const U = L < hip.MAX_LANES ? L : 0;   // the lane of the batched launches
    const joinFrom = U ? hip.MAX_LANES : L;
    try {
      if (lanes) for (let s = 1; s < joinFrom; s++) dev.laneOrder(s, 0);   // every lane starts behind whatever lane 0 holds
      if (lanes) dev.selectLane(U);
      hip.projectViews(sets.map((s) => s[0]), cams, this.pointCloud);
      if (lanes) dev.laneMark(U, 0);
      mine.forEach((v, k) => {
        if (lanes) { dev.laneWaitMark(k % L, 0); dev.selectLane(k % L); }
        this.run(`viewp/${v}/${k}`, (encoder) => this.encodeView(encoder, v, k, false, true));
        if (lanes) dev.laneOrder(U, k % L);   // K17 follows every view
      });
      if (lanes) dev.selectLane(U);
      hip.geometryViews(sets.map((s) => s[2]), cams, sets.map((s) => s[0]), this.dpGrad, this.dpVisible, this.dpFlag, this.pointCloud, false, false);
    } finally {
      if (lanes) {
        hip.addon.encoderAbort(dev.handle);   // (a no-op unless an encode above failed mid-recording)
        dev.selectLane(0);
        for (let s = 1; s < joinFrom; s++) dev.laneOrder(0, s);   // join: the exchange and the optimizer step follow every lane
      }
    }
  }

  /** Per view K1..K16 recorded, K17 eager per view and ordered across the lanes (one op set per lane): the form for batches larger than
   *  MAX_BATCH_VIEWS or with batchViews off. */
  viewsOneByOne(mine) {
    const dev = this.device, L = Math.min(this.lanes, this.opSets);
    const lanes = L > 1 && this.useCommandBuffers && mine.every((v, k) => this.commandBuffers.has(`view/${v}/${k % L}`));
    try {
      if (lanes) for (let s = 1; s < L; s++) dev.laneOrder(s, 0);
      mine.forEach((v, k) => {
        const s = k % L, ops = this.opsOf(s);
        if (lanes) dev.selectLane(s);
        this.run(`view/${v}/${s}`, (encoder) => this.encodeView(encoder, v, s, false, false));
        if (lanes && k > 0) dev.laneOrder(s, (k - 1) % L);   // the fp32 block is filled in view order
        ops[2].encodeGeometry(null, this.cameraBuffers[v], { sums: this.dpGrad, visible: this.dpVisible, first: k === 0,
          tileCounts: ops[0].getResources().tileCountsBuffer, guard: this.dpFlag, stats: ops[0].getStatsBuffer() });
      });
    } finally {
      if (lanes) {
        hip.addon.encoderAbort(dev.handle);
        dev.selectLane(0);
        for (let s = 1; s < L; s++) dev.laneOrder(0, s);
      }
    }
  }

  /** Brings every rank's optimizer state up to date: after sliced steps a rank holds current (param, m, v) only for the Gaussians it owns; each
   *  owner broadcasts its slice of the six state arrays.  A no-op on one rank.  Called before a densify rebuild. */
  syncOptimizerState() {
    if (!this.stateSliced || this.worldSize <= 1) return;
    const n = this.pointCloud.num_points, w = this.worldSize;
    const bufs = this.optimizer.getStateBuffers();
    const rows = { optPosBuffer: 48, optRotBuffer: 48, optScaleBuffer: 48, optOpacityBuffer: 12, paramSH: 192, stateSH: 384 };
    for (let root = 0; root < w; root++) {
      const own = parallel.ownedRange(n, w, root);
      for (const k of Object.keys(rows)) this.exchange.broadcast(bufs[k].ptr + BigInt(own.first * rows[k]), own.count * rows[k], root);
    }
    this.optimizer.stateChanged();
    this.stateSliced = false;
  }

  /** The capacity report of this rank's metric forward passes (their sticky words are consumed), or null.  Synchronises. */
  metricOverflow() {
    let found = null;
    for (const fw of [this.metricsForwardPass].concat(this.moreMetricSets.map((m) => m[0]))) {
      if (!fw) continue;
      try { fw.check(); } catch (e) { if (e && e.code === 'WDGS_E_CAPACITY') found = found || e; else throw e; }
    }
    return found;
  }

  /** true on every rank if `flag` is true on any: one u32 summed over the ranks on the device and read back (only inside a densify event). */
  agree(flag) {
    if (this.worldSize <= 1) return !!flag;
    if (!this.agreeWord) this.agreeWord = this.device.createBuffer({ size: 4, label: 'agreement word' });
    this.device.queue.writeBuffer(this.agreeWord, 0, new Uint32Array([flag ? 1 : 0]));
    this.exchange.allreduceCounts(this.agreeWord, 1);
    return new Uint32Array(this.agreeWord.read(4))[0] !== 0;   // (read: an ArrayBuffer)
  }

  /** trainer.ts:373-497 */
  async runDensifyPruneMultiView() {
    if (!this.pointCloud || !this.optimizer || this.trainCameras.length === 0 || this.images.length === 0) return;
    const baseW = this.lastViewportWidth, baseH = this.lastViewportHeight;
    const m = this.ensureMetricsPipelines(baseW, baseH), mW = m.width, mH = m.height;
    const c = this.densifyPruneConfig;
    const viewsTarget = Math.max(1, Math.floor(c.metricViews));
    const encoder = this.device.createCommandEncoder({ label: 'densify-prune multiview metrics' });
    const counts = this.metricsPass.getMetricCountsBuffer();
    encoder.clearBuffer(counts);
    const dev = this.device, L = this.metricLanes;
    let usedViews = 0, taken = 0;
    try {
      for (let attempt = 0; attempt < viewsTarget * 4 && usedViews < viewsTarget; attempt++) {
        const idx = Math.floor(this.random() * this.trainCameras.length);
        const camData = this.trainCameras[idx], image = this.images[idx];
        if (!camData || !image) continue;
        if (image.width !== baseW || image.height !== baseH) continue;
        // every rank walks the same view list; the work is sharded round-robin and the counts are all-reduced below
        const take = (usedViews % this.worldSize) === this.rank;
        usedViews++;
        if (!take) continue;
        const k = taken % L;   // this rank's views in turn on its metric lanes; every lane's pass adds into set 0's counts
        taken++;
        const set = this.metricSet(k), fw = set[0], rast = set[1], mpass = set[2], target = set[3], cam = set[4];
        if (k > 0) {
          if (taken <= L) { mpass.setMetricCountsTarget(counts); dev.laneOrder(k, 0); }   // the lane's first view of this event: behind the clear
          dev.selectLane(k);
        }
        dev.queue.writeBuffer(cam, 0, cameraBlockFor(camData.camera, mW, mH));
        fw.encode(encoder);
        rast.encode(encoder, mW, mH);
        hip.downsampleRGBA8(dev, image.texture, baseW, baseH, target, mW, mH);
        mpass.computeMetricMap(encoder, rast.getOutputTextureView(), target, { threshold: c.metricThreshold });
        mpass.computeMetricCounts(encoder, { splatBuffer: fw.getResources().splatBuffer, tileOffsetsBuffer: rast.getTileOffsetsBuffer(),
          tileIndicesBuffer: fw.getSortedIndicesBuffer(), nContribTexture: rast.getNContribTextureView() }, { clear: false });
        if (k > 0) dev.selectLane(0);
      }
    } finally {
      dev.selectLane(0);
      for (let k = 1; k < Math.min(L, taken); k++) dev.laneOrder(0, k);   // join: normalize / prepare / the exchange follow every lane
    }
    if (usedViews === 0) return;
    if (this.worldSize > 1) {
      // A metric pass whose tile-entry list overflowed has counted a truncated view: the event is void -- on EVERY rank (the views are sharded, so one
      // rank alone may overflow; were it to bail out while its peers rebuild the cloud, the replicas would part and the next exchange hang).  Each rank
      // looks at its own metric passes, the flags are summed over the ranks, all skip the event together before any count has been exchanged; a rank
      // that overflowed throws on the way out, which makes step() enlarge its lists.
      const overflow = this.metricOverflow();
      if (this.agree(overflow !== null)) {
        if (overflow !== null) throw overflow;
        return;
      }
      this.exchange.allreduceCounts(this.metricsPass.getMetricCountsBuffer(), this.pointCloud.num_points);   // u32 sum, in place
    }
    this.metricsPass.normalizeMetricCounts(encoder, { divisor: usedViews });
    this.densifyPrune.ensureSize(this.pointCloud.num_points);
    const prepared = this.densifyPrune.encodePrepare(encoder, { pointCloud: this.pointCloud, metricCountsBuffer: this.metricsPass.getMetricCountsBuffer() });
    this.device.queue.submit([encoder.finish()]);
    await this.device.queue.onSubmittedWorkDone();
    const outTotal = this.densifyPrune.readTotal();   // the one 4-byte read-back (trainer.ts:440-458)
    const inN = this.pointCloud.num_points;
    const outN = Math.min(outTotal, prepared.maxOutPoints);
    if (outN === 0 || outN === inN) return;
    this.syncOptimizerState();   // every rank rebuilds the whole cloud, so every rank needs the whole state
    this.flushPointCloud();      // the rebuild copies the cloud's SH rows: bring the deferred DC halves in first
    const outPointCloud = hip.allocatePointCloudLike(this.device, this.pointCloud, { numPoints: outN });
    const outOptimizerState = hip.allocateOptimizerStateBuffers(this.device, outN);
    const scatterEncoder = this.device.createCommandEncoder({ label: 'densify-prune scatter' });
    this.densifyPrune.encodeScatter(scatterEncoder, { pointCloud: this.pointCloud, optimizerState: this.optimizer.getStateBuffers(),
      outOffsetBuffer: prepared.outOffsetBuffer, outNumPoints: outN, resetNewOptimizerState: true }, { outPointCloud, outOptimizerState });
    this.device.queue.submit([scatterEncoder.finish()]);
    await this.device.queue.onSubmittedWorkDone();
    this.requestPointCloudSwap(outPointCloud, { iteration: this.optimizer.getIteration(), buffers: outOptimizerState });
    this.lastDensifyPruneIteration = this.iteration;
  }

  /** Deterministic teardown: command buffers, ops, the buffers this trainer allocated (the device and the exchange belong to the caller). */
  destroy() {
    if (this.device.handle !== null) {
      hip.addon.encoderAbort(this.device.handle);
      this.tickets = [];
      try { this.device.synchronize(); } catch (_e) { /* a deferred report about a step of a trainer that is going away */ }
    }
    this.invalidateCommandBuffers();
    for (const name of ['forwardPass', 'rasterizer', 'backwardPass', 'metricsForwardPass', 'metricsRasterizer', 'metricsPass', 'optimizer', 'densifyPrune']) {
      if (this[name]) this[name].destroy();
      this[name] = null;
    }
    this.destroyMoreOpSets();
    this.destroyMoreMetricSets();
    for (const b of this.cameraBuffers) b.destroy();
    this.cameraBuffers = [];
    for (const t of this.ownedTextures || []) t.destroy();
    this.ownedTextures = [];
    for (const name of ['dpGrad', 'dpVisible', 'dpRows', 'dpFlag', 'metricsTarget', 'agreeWord']) { if (this[name]) this[name].destroy(); this[name] = null; }
    this.metricsCameraBuffer.destroy();
    this.isTraining = false;
  }
}
Trainer.DEFAULT_LANES = 3;

module.exports = { Trainer, cameraBlockFor, mat4Inverse, projectionMatrix, DEFAULT_DENSIFY };
