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
 */
const hip = require('./webdgs_hip.js');

/** wgpu-matrix 3.2.0 mat4.inverse (the reference's dependency; cofactor expansion in binary64 on the Float32Array's values). */
function mat4Inverse(m) {
  const m00 = m[0], m01 = m[1], m02 = m[2], m03 = m[3], m10 = m[4], m11 = m[5], m12 = m[6], m13 = m[7];
  const m20 = m[8], m21 = m[9], m22 = m[10], m23 = m[11], m30 = m[12], m31 = m[13], m32 = m[14], m33 = m[15];
  const tmp0 = m22 * m33, tmp1 = m32 * m23, tmp2 = m12 * m33, tmp3 = m32 * m13, tmp4 = m12 * m23, tmp5 = m22 * m13;
  const tmp6 = m02 * m33, tmp7 = m32 * m03, tmp8 = m02 * m23, tmp9 = m22 * m03, tmp10 = m02 * m13, tmp11 = m12 * m03;
  const tmp12 = m20 * m31, tmp13 = m30 * m21, tmp14 = m10 * m31, tmp15 = m30 * m11, tmp16 = m10 * m21, tmp17 = m20 * m11;
  const tmp18 = m00 * m31, tmp19 = m30 * m01, tmp20 = m00 * m21, tmp21 = m20 * m01, tmp22 = m00 * m11, tmp23 = m10 * m01;
  const t0 = (tmp0 * m11 + tmp3 * m21 + tmp4 * m31) - (tmp1 * m11 + tmp2 * m21 + tmp5 * m31);
  const t1 = (tmp1 * m01 + tmp6 * m21 + tmp9 * m31) - (tmp0 * m01 + tmp7 * m21 + tmp8 * m31);
  const t2 = (tmp2 * m01 + tmp7 * m11 + tmp10 * m31) - (tmp3 * m01 + tmp6 * m11 + tmp11 * m31);
  const t3 = (tmp5 * m01 + tmp8 * m11 + tmp11 * m21) - (tmp4 * m01 + tmp9 * m11 + tmp10 * m21);
  const d = 1 / (m00 * t0 + m10 * t1 + m20 * t2 + m30 * t3);
  const o = new Float32Array(16);
  o[0] = d * t0; o[1] = d * t1; o[2] = d * t2; o[3] = d * t3;
  o[4] = d * ((tmp1 * m10 + tmp2 * m20 + tmp5 * m30) - (tmp0 * m10 + tmp3 * m20 + tmp4 * m30));
  o[5] = d * ((tmp0 * m00 + tmp7 * m20 + tmp8 * m30) - (tmp1 * m00 + tmp6 * m20 + tmp9 * m30));
  o[6] = d * ((tmp3 * m00 + tmp6 * m10 + tmp11 * m30) - (tmp2 * m00 + tmp7 * m10 + tmp10 * m30));
  o[7] = d * ((tmp4 * m00 + tmp9 * m10 + tmp10 * m20) - (tmp5 * m00 + tmp8 * m10 + tmp11 * m20));
  o[8] = d * ((tmp12 * m13 + tmp15 * m23 + tmp16 * m33) - (tmp13 * m13 + tmp14 * m23 + tmp17 * m33));
  o[9] = d * ((tmp13 * m03 + tmp18 * m23 + tmp21 * m33) - (tmp12 * m03 + tmp19 * m23 + tmp20 * m33));
  o[10] = d * ((tmp14 * m03 + tmp19 * m13 + tmp22 * m33) - (tmp15 * m03 + tmp18 * m13 + tmp23 * m33));
  o[11] = d * ((tmp17 * m03 + tmp20 * m13 + tmp23 * m23) - (tmp16 * m03 + tmp21 * m13 + tmp22 * m23));
  o[12] = d * ((tmp14 * m22 + tmp17 * m32 + tmp13 * m12) - (tmp16 * m32 + tmp12 * m12 + tmp15 * m22));
  o[13] = d * ((tmp20 * m32 + tmp12 * m02 + tmp19 * m22) - (tmp18 * m22 + tmp21 * m32 + tmp13 * m02));
  o[14] = d * ((tmp18 * m12 + tmp23 * m32 + tmp15 * m02) - (tmp22 * m32 + tmp14 * m02 + tmp19 * m12));
  o[15] = d * ((tmp22 * m22 + tmp16 * m02 + tmp21 * m12) - (tmp20 * m12 + tmp23 * m22 + tmp17 * m02));
  return o;
}

/** get_projection_matrix (src/camera/camera.ts:29-56), column-major after its transpose. */
function projectionMatrix(znear, zfar, fovX, fovY) {
  const tanY = Math.tan(fovY / 2), tanX = Math.tan(fovX / 2);
  const top = tanY * znear, right = tanX * znear;
  const p = new Float32Array(16);
  p[0] = 2 * znear / (2 * right);
  p[5] = -2 * znear / (2 * top);
  p[10] = zfar / (zfar - znear);
  p[11] = 1;
  p[14] = -(zfar * znear) / (zfar - znear);
  return p;
}

/** Camera.set_preset + on_update_canvas + update_buffer (camera.ts:138-205) for a view given as its 68-float block, on a canvas of
 *  width x height: the pose is kept, fovY = 2 atan(height_view / (2 fy_view)), focal = 0.5 height / tan(fovY / 2). */
function cameraBlockFor(block, width, height) {
  const fovY = 2 * Math.atan(block[65] / (2 * block[67]));
  const focal = 0.5 * height / Math.tan(fovY * 0.5);
  const fovX = 2 * Math.atan(width / (2 * focal));
  const out = new Float32Array(68);
  out.set(block.subarray(0, 16), 0);
  out.set(projectionMatrix(0.01, 100, fovX, fovY), 32);
  out.set(mat4Inverse(out.subarray(0, 16)), 16);
  out.set(mat4Inverse(out.subarray(32, 48)), 48);
  out[64] = width; out[65] = height; out[66] = focal; out[67] = focal;
  return out;
}

const DEFAULT_DENSIFY = {   // trainer.ts:147-164
  schedule: { enabled: true, warmupIterations: 500, interval: 100, stopIterations: 15000 },
  metricViews: 10, metricDownscale: 2, metricThreshold: 0.5, maxBufferBytes: 128 * 1024 * 1024, maxNewPointsPerStep: 5000,
  pruneOpacity: 0.01, cloneThresholdCount: 500, splitScaleThreshold: 1.0,
};

class Trainer {
  constructor(device, trainingConfig, options) {
    this.device = device;
    this.trainingConfig = Object.assign({ lambda_l1: 0.8, lambda_l2: 0.0, lambda_dssim: 0.2 }, trainingConfig || {});  // trainer.ts:100-104
    this.optimizerHyperparameters = Object.assign({}, hip.DEFAULT_ADAM_HYPERPARAMETERS);
    this.random = (options && options.random) || Math.random;
    this.useCommandBuffers = !(options && options.useCommandBuffers === false);
    this.maxTileEntries = (options && options.maxTileEntries) || 0;
    this.reusePasses = !(options && options.reusePasses === false);   // applyPointCloudSwap resizes the passes instead of rebuilding them
    // Adam writes the trained SH-DC halves to a compact array that K1 reads instead of 6 bytes into every 96-byte SH row (Optimizer.setDeferredSH);
    // the rows are flushed at hand-over points (flushPointCloud).  Results are identical either way.
    this.deferredSH = !(options && options.deferredSH === false);
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
  applyPointCloudSwap(request) {   // trainer.ts:201-237
    this.device.synchronize();
    const oldParams = this.optimizer ? this.optimizer.getHyperparameters() : null;
    this.invalidateCommandBuffers();
    if (this.optimizer) { this.optimizer.destroy(); this.optimizer = null; }
    const old = this.pointCloud;
    this.pointCloud = request.pointCloud;
    // The reference destroys every pass and constructs new ones; the passes here can follow a cloud of another size
    // (setPointCloud: buffers reused, or re-allocated with headroom), so only the optimizer -- which adopts the rebuilt state -- is new.
    const passes = [this.forwardPass, this.backwardPass, this.metricsForwardPass, this.metricsPass].filter((p) => p);
    const kept = this.reusePasses && old && !this.recreateBackward && passes.every((p) => p.setPointCloud(this.pointCloud));
    if (!kept) {
      for (const name of ['forwardPass', 'rasterizer', 'backwardPass', 'metricsForwardPass', 'metricsRasterizer', 'metricsPass']) {
        if (this[name]) this[name].destroy();
        this[name] = null;
      }
    }
    this.optimizer = new hip.Optimizer(this.device, this.pointCloud, oldParams || this.optimizerHyperparameters, request.optimizerInitialState);
    this.optimizerHyperparameters = this.optimizer.getHyperparameters();
    this.dcWords = this.deferredSH ? this.optimizer.setDeferredSH(this.pointCloud, true) : null;
    for (const fw of [this.forwardPass, this.metricsForwardPass]) if (fw) fw.setDcSource(this.dcWords);
    if (old && old !== this.pointCloud) { old.gaussian_3d_buffer.destroy(); old.sh_buffer.destroy(); }
    this.ensurePipelines(this.lastViewportWidth, this.lastViewportHeight);
  }
  /** Brings pointCloud.sh_buffer up to date with what has been trained (deferred SH writes): call before reading the cloud's SH buffer on the
   *  host, exporting it, or rendering it with a forward pass that is not this trainer's. */
  flushPointCloud() { if (this.optimizer && this.pointCloud) this.optimizer.flushSH(this.pointCloud); }
  /** cameras[i] pairs with images[i] (trainer.ts:575-577): { camera: Float32Array(68), width, height } and { texture: HipBuffer, width, height }. */
  setDataset(cameras, images) {
    this.trainCameras = cameras.slice(); this.images = images.slice();
    for (const b of this.cameraBuffers) b.destroy();
    this.cameraBuffers = this.trainCameras.map((c) => {
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
  }
  stop() { this.isTraining = false; }
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

  invalidateCommandBuffers() {
    for (const c of this.commandBuffers.values()) c.destroy();
    this.commandBuffers.clear();
    this.eagerSteps = 0;
  }

  ensurePipelines(width, height) {   // trainer.ts:662-692 (the rasterizer follows the viewport here: SURVEY Q19)
    const w = Math.max(1, Math.floor(width)), h = Math.max(1, Math.floor(height));
    if (w !== this.lastViewportWidth || h !== this.lastViewportHeight) this.invalidateCommandBuffers();
    this.lastViewportWidth = w; this.lastViewportHeight = h;
    const cam = this.cameraBuffers.length ? this.cameraBuffers[0] : this.metricsCameraBuffer;
    if (!this.forwardPass) {
      this.forwardPass = new hip.TiledForwardPass(this.device, this.pointCloud, cam, { viewportWidth: w, viewportHeight: h, renderMode: 'gaussian', maxTileEntries: this.maxTileEntries });
      this.forwardPass.setDcSource(this.dcWords);
    } else this.forwardPass.setViewport(w, h);
    if (!this.rasterizer) this.rasterizer = new hip.TiledRasterizer({ device: this.device, forwardPass: this.forwardPass, format: 'rgba8unorm' });
    if (!this.backwardPass || this.recreateBackward) {
      if (this.backwardPass) this.backwardPass.destroy();
      this.backwardPass = new hip.TiledBackwardPass(this.device, this.pointCloud, { viewportWidth: w, viewportHeight: h, trainingConfig: this.trainingConfig });
      this.recreateBackward = false;
    } else this.backwardPass.setViewport(w, h);
    // a step whose tile-entry list overflowed is skipped on the device and reported by the next synchronize()
    this.optimizer.setGuard(this.forwardPass.getStatsBuffer(), 8);
  }

  ensureMetricsPipelines(baseWidth, baseHeight) {   // trainer.ts:330-371
    const down = Math.max(1, Math.floor(this.densifyPruneConfig.metricDownscale));
    const w = Math.max(1, Math.floor(baseWidth / down)), h = Math.max(1, Math.floor(baseHeight / down));
    if (this.metricsForwardPass && this.metricsViewportWidth === w && this.metricsViewportHeight === h) return { width: w, height: h };
    for (const name of ['metricsForwardPass', 'metricsRasterizer', 'metricsPass']) { if (this[name]) this[name].destroy(); this[name] = null; }
    if (this.metricsTarget) this.metricsTarget.destroy();
    this.metricsViewportWidth = w; this.metricsViewportHeight = h;
    this.metricsForwardPass = new hip.TiledForwardPass(this.device, this.pointCloud, this.metricsCameraBuffer, { viewportWidth: w, viewportHeight: h, renderMode: 'gaussian', maxTileEntries: this.maxTileEntries });
    this.metricsForwardPass.setDcSource(this.dcWords);
    this.metricsRasterizer = new hip.TiledRasterizer({ device: this.device, forwardPass: this.metricsForwardPass, format: 'rgba8unorm' });
    this.metricsPass = new hip.TiledBackwardPass(this.device, this.pointCloud, { viewportWidth: w, viewportHeight: h, trainingConfig: this.trainingConfig });
    this.metricsTarget = this.device.createBuffer({ size: 4 * w * h, label: 'metrics-gt-downsampled' });
    return { width: w, height: h };
  }

  encodeView(encoder, index) {   // trainer.ts:606-628
    const image = this.images[index], cam = this.cameraBuffers[index];
    this.forwardPass.setCameraBuffer(cam);
    this.forwardPass.encode(encoder);
    this.rasterizer.encode(encoder, image.width, image.height);
    this.backwardPass.encode(encoder, this.rasterizer.getOutputTextureView(), image.texture, {
      splatBuffer: this.forwardPass.getResources().splatBuffer, tileOffsetsBuffer: this.rasterizer.getTileOffsetsBuffer(),
      tileIndicesBuffer: this.forwardPass.getSortedIndicesBuffer(), cameraBuffer: cam,
      alphaTexture: this.rasterizer.getAlphaTextureView(), nContribTexture: this.rasterizer.getNContribTextureView() });
    this.optimizer.step(encoder, this.pointCloud, this.backwardPass.getGradientsBuffer(), this.forwardPass.getResources().tileCountsBuffer);
  }

  /** One training iteration (trainer.ts:568-660). */
  async step() {
    if (!this.isTraining || !this.pointCloud) return;
    const stepStart = Date.now() + 0;
    const t0 = process.hrtime();
    const idx = Math.floor(this.random() * this.trainCameras.length);
    const image = this.images[idx];
    this.ensurePipelines(image.width, image.height);

    const s = this.densifyPruneConfig.schedule;   // trainer.ts:593-601: checked on iteration + 1
    const nextIteration = this.iteration + 1, warmup = s.warmupIterations, interval = Math.max(1, s.interval), stop = s.stopIterations;
    const shouldDensify = s.enabled && nextIteration >= warmup && nextIteration <= stop && (nextIteration === warmup || (nextIteration - warmup) % interval === 0);

    let cmd = this.commandBuffers.get(idx);
    if (cmd) {
      this.device.queue.submit([cmd]);
      this.optimizer.advanceIteration(1);
    } else {
      // the first step runs eagerly (textures are allocated on first use); afterwards each view is recorded once and replayed
      const record = this.useCommandBuffers && this.eagerSteps >= 1;
      const encoder = this.device.createCommandEncoder({ label: 'trainer-step', record });
      try {
        this.encodeView(encoder, idx);
        cmd = encoder.finish();
      } catch (e) {
        encoder.abort();
        this.invalidateCommandBuffers();
        throw e;
      }
      this.device.queue.submit([cmd]);
      if (record) this.commandBuffers.set(idx, cmd); else this.eagerSteps += 1;
    }
    await this.device.queue.onSubmittedWorkDone();
    this.device.synchronize();   // deferred device-side checks (tile-entry overflow) surface here as a thrown Error

    this.iteration += 1;
    const dt = process.hrtime(t0);
    this.stepMs = dt[0] * 1e3 + dt[1] / 1e6;
    const inst = this.stepMs > 0 ? 1000 / this.stepMs : 0;
    this.stepItersPerSec = this.stepItersPerSec === 0 ? inst : this.stepItersPerSec * 0.9 + inst * 0.1;   // trainer.ts:647-651
    if (shouldDensify) {
      await this.runDensifyPruneMultiView();
      const req = this.consumePointCloudSwapRequest();
      if (req) this.applyPointCloudSwap(req);
    }
    if (this.iteration >= this.maxIterations) this.stop();
    void stepStart;
  }

  /** trainer.ts:373-497 */
  async runDensifyPruneMultiView() {
    if (!this.pointCloud || !this.optimizer || this.trainCameras.length === 0 || this.images.length === 0) return;
    const baseW = this.lastViewportWidth, baseH = this.lastViewportHeight;
    const m = this.ensureMetricsPipelines(baseW, baseH), mW = m.width, mH = m.height;
    const c = this.densifyPruneConfig;
    const viewsTarget = Math.max(1, Math.floor(c.metricViews));
    const encoder = this.device.createCommandEncoder({ label: 'densify-prune multiview metrics' });
    encoder.clearBuffer(this.metricsPass.getMetricCountsBuffer());
    let usedViews = 0;
    for (let attempt = 0; attempt < viewsTarget * 4 && usedViews < viewsTarget; attempt++) {
      const idx = Math.floor(this.random() * this.trainCameras.length);
      const camData = this.trainCameras[idx], image = this.images[idx];
      if (!camData || !image) continue;
      if (image.width !== baseW || image.height !== baseH) continue;
      this.device.queue.writeBuffer(this.metricsCameraBuffer, 0, cameraBlockFor(camData.camera, mW, mH));
      this.metricsForwardPass.encode(encoder);
      this.metricsRasterizer.encode(encoder, mW, mH);
      hip.downsampleRGBA8(this.device, image.texture, baseW, baseH, this.metricsTarget, mW, mH);
      this.metricsPass.computeMetricMap(encoder, this.metricsRasterizer.getOutputTextureView(), this.metricsTarget, { threshold: c.metricThreshold });
      this.metricsPass.computeMetricCounts(encoder, { splatBuffer: this.metricsForwardPass.getResources().splatBuffer,
        tileOffsetsBuffer: this.metricsRasterizer.getTileOffsetsBuffer(), tileIndicesBuffer: this.metricsForwardPass.getSortedIndicesBuffer(),
        nContribTexture: this.metricsRasterizer.getNContribTextureView() }, { clear: false });
      usedViews++;
    }
    if (usedViews === 0) return;
    this.metricsPass.normalizeMetricCounts(encoder, { divisor: usedViews });
    this.densifyPrune.ensureSize(this.pointCloud.num_points);
    const prepared = this.densifyPrune.encodePrepare(encoder, { pointCloud: this.pointCloud, metricCountsBuffer: this.metricsPass.getMetricCountsBuffer() });
    this.device.queue.submit([encoder.finish()]);
    await this.device.queue.onSubmittedWorkDone();
    const outTotal = this.densifyPrune.readTotal();   // the one 4-byte read-back (trainer.ts:440-458)
    const inN = this.pointCloud.num_points;
    const outN = Math.min(outTotal, prepared.maxOutPoints);
    if (outN === 0 || outN === inN) return;
    this.flushPointCloud();   // the rebuild copies the cloud's SH rows: bring the deferred DC halves in first
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

  /** Deterministic teardown: command buffers, ops, the buffers this trainer allocated (the device belongs to the caller). */
  destroy() {
    this.device.synchronize();
    this.invalidateCommandBuffers();
    for (const name of ['forwardPass', 'rasterizer', 'backwardPass', 'metricsForwardPass', 'metricsRasterizer', 'metricsPass', 'optimizer', 'densifyPrune']) {
      if (this[name]) this[name].destroy();
      this[name] = null;
    }
    for (const b of this.cameraBuffers) b.destroy();
    this.cameraBuffers = [];
    if (this.metricsTarget) this.metricsTarget.destroy();
    this.metricsCameraBuffer.destroy();
    this.isTraining = false;
  }
}

module.exports = { Trainer, cameraBlockFor, mat4Inverse, projectionMatrix, DEFAULT_DENSIFY };
