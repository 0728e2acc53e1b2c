'use strict';
/*
 * webdgs_hip.js (+ webdgs_hip.d.ts) -- drop-in module for the reference's operator layer (src/renderers/tiled-forward-pass.ts,
 * tiled-rasterizer.ts, tiled-backward-pass.ts, optimizer.ts, densify-prune.ts, src/sort/sort_dynamic.ts, src/prefix/prefix.ts,
 * src/utils/allocate-pointcloud.ts) backed by the N-API addon over libwebdgs_hip.so.
 *
 * Same class names, constructor shapes and method names as the reference; GPUDevice / GPUBuffer / GPUTextureView /
 * GPUCommandEncoder / GPUCommandBuffer become HipDevice / HipBuffer / HipEncoder / HipCommandBuffer.  A src/trainer.ts that imports
 * these instead of the WebGPU classes needs no other change than the import lines (INTEGRATION.md); bindings/ts/trainer.js is that
 * trainer, already rewritten.  Shipped as JavaScript with hand-written typings, the way an npm package reaches a TypeScript host:
 * there is no tsc in this repository's image, but plain CommonJS RUNS here (node 12: no `?.`, no `??`), so the module is executed
 * end to end on the GPU by tests/test_gpu_napi.py instead of merely being written.
 */
const path = require('path');
const addon = require(path.join(__dirname, '..', 'napi', 'webdgs_napi.node'));

const dflt = (v, d) => (v === undefined || v === null ? d : v);

class HipBuffer {                        // GPUBuffer
  constructor(device, ptr, size, handle) {
    this.device = device; this.ptr = ptr; this.size = size; this.handle = handle; this.destroyed = false;
    // called before the buffer's CONTENT is handed to the host (HipDevice.readBuffer / readBufferAsync) or copied by copyBufferToBuffer: a producer
    // that keeps part of it elsewhere brings it up to date first (the optimizer's deferred SH writes and compact training copy)
    this.beforeRead = null;
  }
  /** mapAsync + getMappedRange in one synchronous call: the buffer's bytes as an ArrayBuffer. */
  read(byteLength) { return this.device.readBuffer(this, byteLength); }
  destroy() {
    if (this.destroyed) return;
    this.destroyed = true;
    if (this.handle !== undefined) addon.bufferDestroy(this.handle);
  }
}

/** GPUCommandBuffer backed by an instantiated HIP graph; unlike WebGPU's it may be submitted again (all sizes are read on the device). */
class HipCommandBuffer {
  constructor(device, handle) { this.device = device; this.handle = handle; }
  destroy() { if (this.handle !== null) { addon.commandBufferDestroy(this.handle); this.handle = null; } }
}

/** GPUCommandEncoder.  Default (eager): encodes go to the HIP stream as they are made; `record: true` captures them into a graph. */
class HipEncoder {
  constructor(device, label, record) {
    this.device = device; this.label = label || ''; this.record = !!record; this.open = false;
    if (this.record) { addon.encoderBegin(device.handle); this.open = true; }
  }
  clearBuffer(buffer) { addon.bufferClear(this.device.handle, buffer.ptr, buffer.size); }
  /** encoder.copyBufferToBuffer (trainer.ts:445): device to device, stream-ordered, recordable. */
  copyBufferToBuffer(src, srcOffset, dst, dstOffset, size) {
    if (srcOffset + size > src.size || dstOffset + size > dst.size) throw new RangeError('copyBufferToBuffer: range exceeds a buffer');
    if (src.beforeRead) src.beforeRead();
    addon.copyBufferToBuffer(this.device.handle, dst.ptr + BigInt(dstOffset), src.ptr + BigInt(srcOffset), size);
  }
  finish() {
    if (!this.record) return new HipCommandBuffer(this.device, null);
    this.open = false;
    return new HipCommandBuffer(this.device, addon.encoderFinish(this.device.handle));
  }
  /** Drops an unfinished recording (an encode threw): the device goes back to eager mode.  No-op otherwise. */
  abort() { if (this.open) { this.open = false; addon.encoderAbort(this.device.handle); } }
}

/** Capacity reports that reached the wrong owner.  The library reports a truncated tile-entry list at the next host wait on the DEVICE, to whoever waits
 *  (csrc/api.hip: deferred_checks consumes every pass's word and names the passes); a Trainer and a Viewer that share a device each handle the reports
 *  that name their own passes -- and leave the others here, where the passes' owner looks at its own next wait. */
class CapacityReports {
  constructor(keep) { this.pending = []; this.keep = keep || 16; }
  static passesNamed(error) {
    const out = [], re = /\(forward pass (0x[0-9a-fA-F]+)\)/g, text = String(error && error.message);
    for (let m = re.exec(text); m; m = re.exec(text)) out.push(BigInt(m[1]));
    return out;
  }
  post(error) { this.pending.push(error); if (this.pending.length > this.keep) this.pending.splice(0, this.pending.length - this.keep); }
  /** The oldest pending report that names one of `ownHandles` (removed), or null. */
  take(ownHandles) {
    const own = ownHandles.map((h) => BigInt(h));
    for (let i = 0; i < this.pending.length; i++) {
      if (CapacityReports.passesNamed(this.pending[i]).some((h) => own.some((o) => o === h))) return this.pending.splice(i, 1)[0];
    }
    return null;
  }
}

class HipDevice {                        // GPUDevice + GPUQueue
  constructor(ordinal) {
    this.handle = addon.deviceCreate(ordinal || 0);
    this.capacityReports = new CapacityReports();
    const self = this;
    this.queue = {
      submit(cmds) { for (const c of cmds) if (c.handle !== null) addon.queueSubmit(self.handle, c.handle); },
      onSubmittedWorkDone() { return addon.queueOnSubmittedWorkDone(self.handle); },   // Promise, resolved from the HIP runtime thread
      mark() { return addon.queueMark(self.handle); },                                  // a ticket for the work submitted so far ...
      wait(ticket) { addon.queueWait(self.handle, ticket); },                           // ... awaited later (blocking; raises deferred capacity errors)
      writeBuffer(buffer, offset, data) { addon.copyToDevice(self.handle, buffer.ptr + BigInt(offset), data); },
    };
  }
  createBuffer(desc) {
    const b = addon.bufferCreate(this.handle, desc.size);
    return new HipBuffer(this, b.ptr, desc.size, b.handle);
  }
  createCommandEncoder(desc) { return new HipEncoder(this, desc && desc.label, desc && desc.record); }
  view(ptr, size) { return new HipBuffer(this, ptr, size, undefined); }
  readBuffer(buffer, byteLength) {
    if (buffer.beforeRead) buffer.beforeRead();
    return addon.copyToHost(this.handle, buffer.ptr, byteLength === undefined ? buffer.size : byteLength);
  }
  /** Pinned host memory (wdgs_host_alloc) as an ArrayBuffer: the destination of readBufferAsync. */
  createPinnedArrayBuffer(byteLength) { return addon.hostAlloc(byteLength); }
  /** mapAsync(READ) counterpart (trainer.ts:455-458): queues the copy and resolves once the stream reaches it. */
  readBufferAsync(buffer, offset, pinned, byteLength) {
    if (buffer.handle === undefined) throw new Error('readBufferAsync needs a buffer created by createBuffer');
    if (buffer.beforeRead) buffer.beforeRead();
    addon.bufferReadAsync(this.handle, buffer.handle, offset, pinned, byteLength);
    return addon.queueOnSubmittedWorkDone(this.handle).then(() => pinned);
  }
  /** Blocking variant of onSubmittedWorkDone; also raises deferred capacity errors (code WDGS_E_CAPACITY). */
  synchronize() { addon.deviceSynchronize(this.handle); }
  /** device.limits as far as memory goes (trainer.ts:147): { free, total, cached } in bytes; cached = what the library's allocation cache holds. */
  memoryInfo() { return addon.deviceMemoryInfo(this.handle); }
  selectLane(lane) { addon.deviceSelectLane(this.handle, lane); }            // include/webdgs.h "Lanes"
  laneOrder(waiter, signal) { addon.deviceLaneOrder(this.handle, waiter, signal); }
  laneMark(lane, mark) { addon.deviceLaneMark(this.handle, lane, mark); }          // remembers the current end of `lane` in mark `mark` ...
  laneWaitMark(lane, mark) { addon.deviceLaneWaitMark(this.handle, lane, mark); }  // ... for lanes that wait for it later
  /** Per-kernel hipEvent timing (wdgs_device_set_profiling): kernelTimes() -> { name: { launches, totalMs } } after a synchronize. */
  setProfiling(enabled) { addon.deviceKernelTimes(this.handle, enabled ? 1 : 0); }
  kernelTimes(reset) { const t = addon.deviceKernelTimes(this.handle, 3); if (reset) addon.deviceKernelTimes(this.handle, 2); return t; }
  destroy() { if (this.handle !== null) { addon.encoderAbort(this.handle); addon.deviceDestroy(this.handle); this.handle = null; } }
}

/** allocatePointCloudLike (src/utils/allocate-pointcloud.ts:8-44): zeroed buffers of the template's layout for `numPoints`. */
function allocatePointCloudLike(device, template, options) {
  const n = Math.max(1, Math.floor(options.numPoints));
  const tp = Math.max(1, template.num_points);
  const bpg = Math.max(1, Math.floor(template.gaussian_3d_buffer.size / tp));
  const bps = Math.max(1, Math.floor(template.sh_buffer.size / tp));
  return { type: template.type || 'normal', num_points: n, sh_deg: template.sh_deg || 0,
    gaussian_3d_buffer: device.createBuffer({ size: n * bpg }), sh_buffer: device.createBuffer({ size: n * bps }) };
}

class PrefixScanner {                    // src/prefix/prefix.ts:26-43
  constructor(maxElements, device) {
    const s = addon.prefixScanner(0, device.handle, maxElements);
    this.device = device; this.handle = s.handle; this.max_elements = maxElements;
    this.input_buffer = device.view(s.inputBuffer, 4 * maxElements);
    this.output_buffer = device.view(s.outputBuffer, 4 * maxElements);
  }
  set_count(count) { addon.prefixScanner(1, this.handle, count); return { num_workgroups: Math.ceil(count / 4096) }; }
  scan(_encoder) { addon.prefixScanner(2, this.handle, 0); }
  destroy() { if (this.handle !== null) { addon.prefixScanner(3, this.handle, 0); this.handle = null; } }
}
/** get_prefix_scanner(maxElements, device): PrefixScanner (prefix.ts:140) */
function get_prefix_scanner(maxElements, device) { return new PrefixScanner(maxElements, device); }

class DynamicSortStuff {                 // src/sort/sort_dynamic.ts:9-24 -- the element count is read on the device from statsBuffer[0]
  constructor(maxCapacity, device, statsBuffer) {
    const s = addon.dynamicSorter(0, device.handle, maxCapacity, statsBuffer.ptr);
    this.device = device; this.handle = s.handle; this.capacity = s.capacity; this.final_out_index = 0;
    this.ping_pong = [
      { sort_depths_buffer: device.view(s.keys0, 4 * s.capacity), sort_indices_buffer: device.view(s.values0, 4 * s.capacity) },
      { sort_depths_buffer: device.view(s.keys1, 4 * s.capacity), sort_indices_buffer: device.view(s.values1, 4 * s.capacity) }];
  }
  sort(_encoder, keyBits) { this.final_out_index = addon.dynamicSorter(1, this.handle, dflt(keyBits, 32), 0); }
  destroy() { if (this.handle !== null) { addon.dynamicSorter(2, this.handle, 0, 0); this.handle = null; } }
}
/** get_dynamic_sorter(maxCapacity, device, statsBuffer): DynamicSortStuff (sort_dynamic.ts:252) */
function get_dynamic_sorter(maxCapacity, device, statsBuffer) { return new DynamicSortStuff(maxCapacity, device, statsBuffer); }

class TiledForwardPass {                 // tiled-forward-pass.ts:62
  constructor(device, pointCloud, cameraBuffer, config) {
    this.device = device; this.pointCloud = pointCloud; this.cameraBuffer = cameraBuffer; this.destroyed = false;
    this.handle = addon.tiledForwardCreate(device.handle, {
      numPoints: pointCloud.num_points, shDeg: pointCloud.sh_deg || 0, viewportWidth: config.viewportWidth, viewportHeight: config.viewportHeight,
      gaussianScale: dflt(config.gaussianScale, 1.0), pointSizePx: dflt(config.pointSizePx, 3.0), maxSplatRadiusPx: dflt(config.maxSplatRadiusPx, 128.0),
      renderMode: dflt(config.renderMode, 'gaussian') === 'gaussian' ? 1 : 0, maxTileEntries: config.maxTileEntries || 0, compatCaps: config.compatCaps ? 1 : 0,
    });
  }
  get nativeHandle() { return this.handle; }
  /** K1 takes the SH-DC halves from the optimizer's compact array (Optimizer.setDeferredSH) instead of the cloud's rows; null restores the rows. */
  setDcSource(dcWords) { this.dcSource = dcWords || null; addon.tiledForwardSetDcSource(this.handle, dcWords ? dcWords.ptr : null); }
  /** A cloud that is being trained with deferred SH writes carries the optimizer's compact array (pointCloud.dcWords, Optimizer.setDeferredSH):
   *  every forward pass built on that cloud -- the trainer's, a Viewer's -- takes the SH-DC halves from it without the host knowing. */
  syncDcSource() { const want = this.pointCloud.dcWords || null; if (want !== (this.dcSource || null)) this.setDcSource(want); }
  encode(_encoder, options) {
    this.syncDcSource();
    addon.tiledForwardEncode(this.handle, this.pointCloud.gaussian_3d_buffer.ptr, this.pointCloud.sh_buffer.ptr, this.cameraBuffer.ptr, options && options.skipSort ? 1 : 0);
  }
  /** The rest of encode (scan, emit, sort) for a pass whose K1 ran through projectViews (view-batched step; no reference counterpart). */
  encodeProjected(_encoder) { addon.tiledForwardEncodeProjected(this.handle); }
  isProjected() { return addon.tiledForwardIsProjected(this.handle) !== 0; }
  setCameraBuffer(buffer) { this.cameraBuffer = buffer; }
  /** Adopts a point cloud of another size (wdgs_tiled_forward_resize) instead of destroy + construct; false if the SH degree differs. */
  setPointCloud(pointCloud) {
    if ((pointCloud.sh_deg || 0) !== (this.pointCloud.sh_deg || 0)) return false;
    addon.tiledForwardResize(this.handle, pointCloud.num_points); this.pointCloud = pointCloud; return true;
  }
  setRenderMode(mode) { addon.tiledForwardSet(this.handle, 0, mode === 'gaussian' ? 1 : 0); }
  setPointSize(value) { addon.tiledForwardSet(this.handle, 1, value); }
  setGaussianScale(value) { addon.tiledForwardSet(this.handle, 2, value); }
  setViewport(width, height) { addon.tiledForwardSetViewport(this.handle, width, height); }
  getResources() {
    const r = addon.tiledForwardGetResources(this.handle); const n = Math.max(1, this.pointCloud.num_points); const d = this.device;
    return { splatBuffer: d.view(r.splatBuffer, 24 * n), tileKeysBuffer: d.view(r.tileKeysBuffer, 4 * r.maxTileEntries), tileIndicesBuffer: d.view(r.tileIndicesBuffer, 4 * r.maxTileEntries),
      tileOffsetsBuffer: d.view(r.tileOffsetsBuffer, 4 * n), tileCountsBuffer: d.view(r.tileCountsBuffer, 4 * n), statsBuffer: d.view(r.statsBuffer, 16),
      numTilesX: r.numTilesX, numTilesY: r.numTilesY, totalTiles: r.totalTiles, maxTileEntries: r.maxTileEntries };
  }
  getSortedIndicesBuffer() { return this.getResources().tileIndicesBuffer; }
  getSortedKeysBuffer() { return this.getResources().tileKeysBuffer; }
  getTileOffsetsBuffer() { return this.getResources().tileOffsetsBuffer; }
  getStatsBuffer() { return this.getResources().statsBuffer; }
  /** Synchronises; throws (code WDGS_E_CAPACITY) if an encode since the last check overflowed maxTileEntries. */
  check() { return addon.tiledForwardCheck(this.handle); }
  /** Long tile lists (include/webdgs.h: wdgs_tiled_forward_set_long_lists): tiles with more than `threshold` entries get per-pixel lists (0: off). */
  setLongLists(threshold, maxItems, maxRows) { addon.tiledForwardSetLongLists(this.handle, threshold, maxItems || 0, maxRows || 0); }
  /** The last frame's long-list work: what it wanted and what the pass has room for (synchronises). */
  longListStats() { return addon.tiledForwardLongListStats(this.handle); }
  destroy() { if (this.destroyed) return; this.destroyed = true; addon.tiledForwardDestroy(this.handle); this.handle = null; }   // (a call after destroy meets the library's null check, not freed memory)
}

class TiledRasterizer {                  // tiled-rasterizer.ts:34
  constructor(config) {
    this.device = config.device; this.destroyed = false; this.w = 0; this.h = 0;
    this.handle = addon.tiledRasterizerCreate(config.device.handle, config.forwardPass.nativeHandle);
  }
  encode(_encoder, width, height) { addon.tiledRasterizerEncode(this.handle, width, height); this.w = width; this.h = height; }
  getOutputTextureView() { return this.device.view(addon.tiledRasterizerGet(this.handle, 0), 4 * this.w * this.h); }      // throws before the first encode
  getAlphaTextureView() { return this.device.view(addon.tiledRasterizerGet(this.handle, 1), 4 * this.w * this.h); }
  getNContribTextureView() { return this.device.view(addon.tiledRasterizerGet(this.handle, 2), 4 * this.w * this.h); }
  getTileOffsetsBuffer() { return this.device.view(addon.tiledRasterizerGet(this.handle, 3), 4 * (Math.ceil(this.w / 16) * Math.ceil(this.h / 16) + 1)); }
  /** blitToTexture(encoder, targetView, clearColor?) (tiled-rasterizer.ts:333-357): `target` is an rgba8 image buffer; it may carry its own
   *  `width` / `height` (a canvas of another size: the blit is a bilinear resample), otherwise it has the rasterizer's size.  The blit covers
   *  the whole target, so the reference's clear colour never shows and is accepted only for signature compatibility. */
  blitToTexture(_encoder, target, _clearColor) { addon.tiledRasterizerBlit(this.handle, target.ptr, dflt(target.width, this.w), dflt(target.height, this.h)); }
  destroy() { if (this.destroyed) return; this.destroyed = true; addon.tiledRasterizerDestroy(this.handle); this.handle = null; }
}

const resourcePtrs = (r) => ({ splatBuffer: r.splatBuffer.ptr, tileOffsetsBuffer: r.tileOffsetsBuffer.ptr, tileIndicesBuffer: r.tileIndicesBuffer.ptr,
  cameraBuffer: r.cameraBuffer ? r.cameraBuffer.ptr : null, alphaTexture: r.alphaTexture ? r.alphaTexture.ptr : null, nContribTexture: r.nContribTexture.ptr });

class TiledBackwardPass {                // tiled-backward-pass.ts:71
  constructor(device, pointCloud, config) {
    const t = config.trainingConfig;
    this.device = device; this.pointCloud = pointCloud; this.destroyed = false; this.w = config.viewportWidth; this.h = config.viewportHeight;
    this.handle = addon.tiledBackwardCreate(device.handle, { numPoints: pointCloud.num_points, shDeg: pointCloud.sh_deg || 0, viewportWidth: config.viewportWidth,
      viewportHeight: config.viewportHeight, lambda_l1: t.lambda_l1, lambda_l2: t.lambda_l2, lambda_dssim: t.lambda_dssim, c1: dflt(t.c1, 0.0001), c2: dflt(t.c2, 0.0009),
      maxSplatRadiusPx: dflt(config.maxSplatRadiusPx, 128.0) });
    this.trainingConfig = { lambda_l1: t.lambda_l1, lambda_l2: t.lambda_l2, lambda_dssim: t.lambda_dssim, c1: dflt(t.c1, 0.0001), c2: dflt(t.c2, 0.0009) };
  }
  encode(_encoder, predictedTexture, targetTexture, r, _options) {   // (TiledBackwardPassOptions is an empty interface in the reference)
    addon.tiledBackwardEncode(this.handle, predictedTexture.ptr, targetTexture.ptr, resourcePtrs(r), this.pointCloud.gaussian_3d_buffer.ptr);
  }
  /** First half of encode (K15 loss gradient, clear, K16 backward raster): wdgs_tiled_backward_encode_raster. */
  encodeRaster(_encoder, predictedTexture, targetTexture, r) { addon.tiledBackwardEncodeRaster(this.handle, predictedTexture.ptr, targetTexture.ptr, resourcePtrs(r)); }
  /** Second half (K17).  `accumulate` (a batched step) = { sums, visible, tileCounts, guard, stats, first }: K17 also adds the view's gradient to the
   *  step's fp32 block and folds the forward pass's overflow word (stats + 8 bytes) into the guard word. */
  encodeGeometry(_encoder, cameraBuffer, accumulate) {
    const a = accumulate ? { sums: accumulate.sums.ptr, visible: accumulate.visible.ptr, tileCounts: accumulate.tileCounts.ptr, guard: accumulate.guard.ptr,
      overflowWord: accumulate.stats.ptr + 8n, first: accumulate.first ? 1 : 0 } : null;
    addon.tiledBackwardEncodeGeometry(this.handle, cameraBuffer.ptr, this.pointCloud.gaussian_3d_buffer.ptr, a);
  }
  /** Whether Optimizer.stepWithGeometry also writes K17's packed gradient to getGradientsBuffer() (default: yes, as the reference's K17 does). */
  setGradientOutput(enabled) { this.gradientOutput = !!enabled; addon.tiledBackwardSetGradientOutput(this.handle, enabled ? 1 : 0); }
  /** computeMetricCounts of this pass adds into `counts` (another pass's getMetricCountsBuffer()) instead of its own array; null restores its own.
   *  Several passes can then take the metric views of one densify event on different lanes (integer atomics: any order gives the same bits). */
  setMetricCountsTarget(counts) { this.metricTarget = counts || null; addon.tiledBackwardSetMetricCountsTarget(this.handle, counts ? counts.ptr : null); }
  /** setTrainingConfig(next) (tiled-backward-pass.ts:812-830): loss weights of the next encode. */
  setTrainingConfig(next) {
    this.trainingConfig = Object.assign({ lambda_l1: 0.8, lambda_l2: 0.0, lambda_dssim: 0.2, c1: 0.0001, c2: 0.0009 }, this.trainingConfig || {}, next || {});
    addon.tiledBackwardMetric(this.handle, 5, this.trainingConfig, 0, 0);
  }
  /** See TiledForwardPass.setPointCloud (wdgs_tiled_backward_resize). */
  setPointCloud(pointCloud) {
    if ((pointCloud.sh_deg || 0) !== (this.pointCloud.sh_deg || 0)) return false;
    addon.tiledBackwardResize(this.handle, pointCloud.num_points); this.pointCloud = pointCloud; return true;
  }
  computeLossOnly(_encoder, predicted, target) { addon.tiledBackwardMetric(this.handle, 0, predicted.ptr, target.ptr, 0); }
  computeMetricMap(_encoder, predicted, target, options) { addon.tiledBackwardMetric(this.handle, 1, predicted.ptr, target.ptr, dflt(options && options.threshold, 0.5)); }
  computeMetricCounts(_encoder, r, options) {
    addon.tiledBackwardMetric(this.handle, 2, resourcePtrs(r), (options && options.numInstances) || Math.floor(r.tileIndicesBuffer.size / 4), options && options.clear === false ? 0 : 1);
  }
  normalizeMetricCounts(_encoder, options) { addon.tiledBackwardMetric(this.handle, 3, Math.max(1, Math.floor(options.divisor)), 0, 0); }
  setViewport(width, height) { addon.tiledBackwardMetric(this.handle, 4, width, height, 0); this.w = width; this.h = height; }
  getGradientsBuffer() { return this.device.view(addon.tiledBackwardGet(this.handle, 0), 32 * Math.max(1, this.pointCloud.num_points)); }
  getMetricCountsBuffer() { return this.device.view(addon.tiledBackwardGet(this.handle, 1), 4 * Math.max(1, this.pointCloud.num_points)); }
  getLossTextureView() { return this.device.view(addon.tiledBackwardGet(this.handle, 2), 16 * this.w * this.h); }
  getMetricMapTextureView() { return this.device.view(addon.tiledBackwardGet(this.handle, 3), 4 * this.w * this.h); }
  getMetricMapTexture() { return this.getMetricMapTextureView(); }   // (texture and view are the same r32uint image buffer here)
  destroy() { if (this.destroyed) return; this.destroyed = true; addon.tiledBackwardDestroy(this.handle); this.handle = null; }
}

const DEFAULT_ADAM_HYPERPARAMETERS = { lr_pos: 0.00016, lr_color: 0.0025, lr_opacity: 0.05, lr_scale: 0.005, lr_rot: 0.001, beta1: 0.9, beta2: 0.999, epsilon: 1e-8 };  // adam-config.ts:12-21
const STATE_KEYS = ['optPosBuffer', 'optRotBuffer', 'optScaleBuffer', 'optOpacityBuffer', 'paramSH', 'stateSH'];
const statePtrs = (s) => { const o = {}; for (const k of STATE_KEYS) o[k] = s[k].ptr; return o; };

/** allocateOptimizerStateBuffers (optimizer.ts:27-38). */
function allocateOptimizerStateBuffers(device, numPoints) {
  const sizes = addon.optimizerStateSizes(Math.max(1, numPoints));
  const o = {};
  STATE_KEYS.forEach((k, i) => { o[k] = device.createBuffer({ size: sizes[i] }); });
  return o;
}

class Optimizer {                        // optimizer.ts:40
  /** `initialState` = { iteration, buffers } is ADOPTED as-is (OptimizerInitialState, optimizer.ts:22-25, 81-88). */
  constructor(device, pointCloud, params, initialState) {
    this.device = device; this.pointCloud = pointCloud; this.destroyed = false;
    this.buffers = initialState && initialState.buffers ? initialState.buffers : null;
    this.handle = this.buffers
      ? addon.optimizerCreateWithState(device.handle, pointCloud.num_points, pointCloud.gaussian_3d_buffer.ptr, pointCloud.sh_buffer.ptr, statePtrs(this.buffers), 0,
                                       initialState.iteration || 0)
      : addon.optimizerCreate(device.handle, pointCloud.num_points, pointCloud.gaussian_3d_buffer.ptr, pointCloud.sh_buffer.ptr);
    if (params) addon.optimizerHyperparameters(this.handle, params);
  }
  getIteration() { return addon.optimizerGetIteration(this.handle); }
  getHyperparameters() { return addon.optimizerHyperparameters(this.handle, null); }
  setHyperparameters(next) { addon.optimizerHyperparameters(this.handle, next); }
  /** Brings the SH-DC rows of paramSH / stateSH up to date before handing the arrays out (see include/webdgs.h). */
  getStateBuffers() {
    const s = addon.optimizerState(this.handle, 0);
    let o = this.buffers;
    if (!o) {
      const sizes = addon.optimizerStateSizes(Math.max(1, this.pointCloud.num_points));
      o = {};
      STATE_KEYS.forEach((k, i) => { o[k] = this.device.view(s[k], sizes[i]); });
    }
    // position, log-scale and SH-DC {param, m, v} are trained in a compact copy: a handle kept across steps is brought up to date whenever its
    // content is read through the host (the reference's GPUBuffers are live)
    const self = this;
    for (const k of STATE_KEYS) if (!o[k].beforeRead) o[k].beforeRead = () => { if (!self.destroyed) addon.optimizerState(self.handle, 0); };
    return o;
  }
  step(_encoder, coefficients, gradientsBuffer, tileCountsBuffer) {
    addon.optimizerStep(this.handle, coefficients.gaussian_3d_buffer.ptr, coefficients.sh_buffer.ptr, gradientsBuffer.ptr, tileCountsBuffer.ptr);
  }
  /** Deferred SH writes (include/webdgs.h wdgs_optimizer_set_deferred_sh): on, step() writes the trained SH-DC halves to a compact array
   *  (returned: give it to every forward pass that renders the cloud, TiledForwardPass.setDcSource) instead of 6 bytes into every 96-byte
   *  row; flushSH(pointCloud) brings the rows up to date -- call it before anything reads pointCloud.sh_buffer (a host read, an export, a
   *  viewer's own forward pass, DensifyPrunePass.encodeScatter).  Off: flushes and restores the reference's write pattern. */
  setDeferredSH(pointCloud, enabled) {
    const on = enabled !== false;
    const p = addon.optimizerDeferredSH(this.handle, pointCloud.sh_buffer.ptr, on ? 1 : 0);
    this.deferredCloud = on ? pointCloud : null;
    const words = on && p !== null ? this.device.view(p, 8 * Math.max(1, this.pointCloud.num_points)) : null;
    const self = this;
    pointCloud.sh_buffer.beforeRead = on ? () => self.flushSH(pointCloud) : null;   // host reads of the rows see the trained values
    pointCloud.dcWords = words;                                                   // forward passes built on the cloud pick the halves up (syncDcSource)
    return words;
  }
  flushSH(pointCloud) { if (!this.destroyed) addon.optimizerFlushSH(this.handle, pointCloud.sh_buffer.ptr); }
  /** step() fused with K17 (wdgs_optimizer_step_with_geometry): call after backwardPass.encodeRaster for the view. */
  stepWithGeometry(_encoder, coefficients, backwardPass, cameraBuffer, tileCountsBuffer) {
    addon.optimizerStepWithGeometry(this.handle, backwardPass.handle, cameraBuffer.ptr, coefficients.gaussian_3d_buffer.ptr, coefficients.sh_buffer.ptr, tileCountsBuffer.ptr);
  }
  /** Adam + re-pack on Gaussians [first, first + count) -- the slice a data-parallel rank owns; rowsOut also receives the re-packed 32-byte rows. */
  stepF32Range(_encoder, coefficients, gradF32, visibleCounts, first, count, rowsOut) {
    addon.optimizerStepF32Range(this.handle, coefficients.gaussian_3d_buffer.ptr, coefficients.sh_buffer.ptr, gradF32.ptr, visibleCounts.ptr, first, count, rowsOut ? rowsOut.ptr : null);
  }
  /** Writes the rows the other ranks published into this replica (with deferred SH writes the gathered halves go to the compact array). */
  applyRepackedRows(rows, skipFirst, skipCount, guard, pointCloud) {
    addon.optimizerApplyRepackedRows(this.handle, rows.ptr, skipFirst, skipCount, guard ? guard.ptr : null, pointCloud.gaussian_3d_buffer.ptr, pointCloud.sh_buffer.ptr);
  }
  /** The state arrays were rewritten from outside (slices gathered from other ranks, a restored snapshot): refresh the internal copies. */
  stateChanged() { addon.optimizerStateChanged(this.handle); }
  /** While the u32 at `flagBuffer + offset` is non-zero at execution time, step() leaves every buffer untouched (tile-entry overflow). */
  setGuard(flagBuffer, offset) { addon.optimizerSetGuard(this.handle, flagBuffer ? flagBuffer.ptr + BigInt(offset || 0) : null); }
  /** Host-side iteration counter: call when a recorded command buffer containing step() is re-submitted. */
  advanceIteration(count) { addon.optimizerAdvanceIteration(this.handle, dflt(count, 1)); }
  /** Also destroys adopted state buffers, as the reference's Optimizer.destroy() does (optimizer.ts:352-362). */
  destroy() {
    if (this.destroyed) return;
    if (this.deferredCloud && !this.deferredCloud.sh_buffer.destroyed) this.setDeferredSH(this.deferredCloud, false);  // the cloud outlives its optimizer: leave its rows current
    this.destroyed = true;
    addon.optimizerDestroy(this.handle);
    this.handle = null;
    if (this.buffers) for (const k of STATE_KEYS) this.buffers[k].destroy();
  }
}

class DensifyPrunePass {                 // densify-prune.ts:75
  constructor(device, config) {
    this.device = device; this.numPoints = 0;
    this.config = Object.assign({ strategy: 'cpu_rebuild', numViews: 1, cloneThreshold: 0, pruneThreshold: 0, maxNewPointsPerStep: 0, maxBufferBytes: 128 * 1024 * 1024 }, config || {});
    this.handle = addon.densifyCreate(device.handle, this.config);
  }
  setConfig(next) { this.config = Object.assign({}, this.config, next); addon.densifySetConfig(this.handle, this.config); }
  getConfig() { return Object.assign({}, this.config); }
  wrap(p, n) {
    const d = this.device; const m = Math.max(1, n);
    return { actionBuffer: d.view(p.actionBuffer, 4 * m), outCountBuffer: d.view(p.outCountBuffer, 4 * m), outOffsetBuffer: d.view(p.outOffsetBuffer, 4 * m),
      outTotalBuffer: d.view(p.outTotalBuffer, 4), maxOutPoints: p.maxOutPoints };
  }
  encodePrepare(_encoder, inputs) {
    const n = inputs.pointCloud.num_points; this.numPoints = n;
    return this.wrap(addon.densifyEncodePrepare(this.handle, n, inputs.pointCloud.gaussian_3d_buffer.ptr, inputs.metricCountsBuffer ? inputs.metricCountsBuffer.ptr : null), n);
  }
  // ---- the stages encodePrepare is made of (densify-prune.ts:327-456), individually recordable as in the reference
  stage(stage, n, a, b) { return this.wrap(addon.densifyStage(this.handle, stage, n, dflt(a, 0), dflt(b, 0)), n); }
  ensureSize(numPoints) { this.numPoints = numPoints; this.stage(4, numPoints); }
  computeMaxOutPoints(pointCloud) { return this.stage(4, pointCloud.num_points).maxOutPoints; }
  encodeDecision(_encoder, inputs) {
    this.numPoints = inputs.pointCloud.num_points;
    const p = this.stage(0, this.numPoints, inputs.pointCloud.gaussian_3d_buffer.ptr, inputs.metricCountsBuffer ? inputs.metricCountsBuffer.ptr : null);
    return { actionBuffer: p.actionBuffer, outCountBuffer: p.outCountBuffer };
  }
  encodePrefixSum(_encoder) { return this.stage(1, this.numPoints).outOffsetBuffer; }
  encodeCapToMax(_encoder, _outOffsetBuffer, maxOutPoints) { this.stage(2, this.numPoints, Math.max(0, Math.floor(maxOutPoints))); }
  encodeTotalOut(_encoder, _outOffsetBuffer) { return this.stage(3, this.numPoints).outTotalBuffer; }
  /** The one 4-byte read-back of the densify path (trainer.ts:440-458, mapAsync on outTotalBuffer). */
  readTotal() { return addon.densifyReadTotal(this.handle); }
  encodeScatter(_encoder, inputs, outputs) {
    if (outputs.outPointCloud.num_points !== inputs.outNumPoints) throw new Error('encodeScatter: outPointCloud.num_points must equal outNumPoints');  // densify-prune.ts:478-480
    addon.densifyEncodeScatter(this.handle, inputs.pointCloud.num_points, inputs.pointCloud.gaussian_3d_buffer.ptr, inputs.pointCloud.sh_buffer.ptr,
      inputs.optimizerState ? statePtrs(inputs.optimizerState) : null, inputs.outNumPoints, inputs.resetNewOptimizerState === false ? 0 : 1,
      outputs.outPointCloud.gaussian_3d_buffer.ptr, outputs.outPointCloud.sh_buffer.ptr, outputs.outOptimizerState ? statePtrs(outputs.outOptimizerState) : null);
  }
  getActionBuffer() { return this.numPoints ? this.stage(4, this.numPoints).actionBuffer : null; }        // densify-prune.ts getters: null before ensureSize
  getOutCountBuffer() { return this.numPoints ? this.stage(4, this.numPoints).outCountBuffer : null; }
  getOutTotalBuffer() { return this.stage(4, Math.max(1, this.numPoints)).outTotalBuffer; }
  applyActions(_encoder) { throw new Error('DensifyPrunePass.applyActions is unimplemented in the reference (densify-prune.ts:680-686)'); }
  destroy() { if (this.handle !== null) { addon.densifyDestroy(this.handle); this.handle = null; } }
}

/** Bilinear blit of an rgba8 image to another size (trainer.ts:303-328: the ground-truth down-sample of the metric views). */
function downsampleRGBA8(device, src, srcW, srcH, dst, dstW, dstH) { addon.downsampleRGBA8(device.handle, src.ptr, srcW, srcH, dst.ptr, dstW, dstH); }

/** K1 of ALL the views of a batched step in one launch (wdgs_tiled_forward_project_views); follow with forwardPasses[v].encodeProjected(encoder). */
function projectViews(forwardPasses, cameraBuffers, pointCloud) {
  for (const f of forwardPasses) f.syncDcSource();
  addon.tiledForwardProjectViews(forwardPasses.map((f) => f.handle), cameraBuffers.map((c) => c.ptr), pointCloud.gaussian_3d_buffer.ptr, pointCloud.sh_buffer.ptr);
}
/** K17 of ALL the views of a batched step in one launch (wdgs_tiled_backward_encode_geometry_views): the step's fp32 gradient block, visibility counts
 *  and guard word, bit for bit what encodeGeometry(camera, { first: v === 0, ... }) per view produces. */
function geometryViews(backwardPasses, cameraBuffers, forwardPasses, sums, visible, guard, pointCloud, writeGradients, continues) {
  const res = forwardPasses.map((f) => addon.tiledForwardGetResources(f.handle));
  addon.tiledBackwardGeometryViews(backwardPasses.map((b) => b.handle), cameraBuffers.map((c) => c.ptr), res.map((r) => r.tileCountsBuffer), res.map((r) => r.statsBuffer + 8n),
    pointCloud.gaussian_3d_buffer.ptr, sums.ptr, visible.ptr, guard.ptr, writeGradients ? 1 : 0, continues ? 1 : 0);
}
/** Exact sum of squared rgb8 differences of two rgba8 images (synchronises); PSNR = 10 log10(255^2 3P / SSE). */
function imageSSE(device, a, b, numPixels) {
  const out = device.createBuffer({ size: 8 });
  addon.imageSSE(device.handle, a.ptr, b.ptr, numPixels, out.ptr);
  const v = new BigUint64Array(device.readBuffer(out, 8))[0];
  out.destroy();
  return Number(v);
}
function imagePSNR(device, a, b, numPixels) {
  const sse = imageSSE(device, a, b, numPixels);
  return sse === 0 ? Infinity : 10 * Math.log10(255 * 255 * 3 * numPixels / sse);
}

/** The C-ABI communicator (wdgs_comm_*, include/webdgs.h): RCCL queued on the device's stream by the library itself -- the transport of the
 *  data-parallel step for a host without torch.distributed.  `uniqueId` (ArrayBuffer, 128 bytes) comes from Communicator.uniqueId() on rank 0 and
 *  reaches the other ranks over any host channel (parallel.js ships it through a file). */
class Communicator {
  constructor(device, uniqueId, worldSize, rank) {
    this.device = device; this.worldSize = worldSize; this.rank = rank;
    this.handle = addon.commCreate(device.handle, uniqueId, worldSize, rank);
  }
  static uniqueId() { return addon.commUniqueId(); }
  exchangeGradients(grad, visible, flag, slicePoints) { addon.commExchangeGradients(this.handle, grad.ptr, visible.ptr, flag ? flag.ptr : null, slicePoints); }
  allgatherRows(rows, slicePoints) { addon.commAllgatherRows(this.handle, rows.ptr, slicePoints); }
  broadcast(ptr, bytes, root) { addon.commBroadcast(this.handle, ptr, bytes, root); }
  allreduceCounts(counts, count) { addon.commAllreduceCounts(this.handle, counts.ptr, count); }
  allreduceGradients(grad, visible, numPoints) { addon.commAllreduceGradients(this.handle, grad.ptr, visible.ptr, numPoints); }
  static groupStart() { addon.commGroup(0); }
  static groupEnd() { addon.commGroup(1); }
  destroy() { if (this.handle !== null) { addon.commDestroy(this.handle); this.handle = null; } }
}

const MAX_LANES = 4;         // WDGS_MAX_LANES
const MAX_BATCH_VIEWS = 16;  // WDGS_MAX_BATCH_VIEWS

module.exports = { addon, MAX_LANES, MAX_BATCH_VIEWS, projectViews, geometryViews, imageSSE, imagePSNR, Communicator, HipBuffer, HipCommandBuffer, HipEncoder, HipDevice, CapacityReports, allocatePointCloudLike, PrefixScanner, get_prefix_scanner, DynamicSortStuff,
  get_dynamic_sorter, TiledForwardPass, TiledRasterizer, TiledBackwardPass, DEFAULT_ADAM_HYPERPARAMETERS, allocateOptimizerStateBuffers, Optimizer,
  DensifyPrunePass, downsampleRGBA8 };
