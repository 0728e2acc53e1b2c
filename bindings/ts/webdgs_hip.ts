/*
 * webdgs_hip.ts -- drop-in module for the reference's operator layer (src/renderers/tiled-forward-pass.ts, tiled-rasterizer.ts,
 * tiled-backward-pass.ts, optimizer.ts, densify-prune.ts, allocate-pointcloud.ts) backed by the N-API addon over libwebdgs_hip.so.
 *
 * Same class names, constructor shapes and method names as the reference; GPUDevice / GPUBuffer / GPUTextureView /
 * GPUCommandEncoder / GPUCommandBuffer become HipDevice / HipBuffer / HipEncoder / HipCommandBuffer.  A src/trainer.ts that
 * imports these instead of the WebGPU classes needs no other change than the import lines.
 * NOT type-checked in this repository's image (no tsc); the addon underneath is compiled and exercised end to end (forward,
 * backward, Adam, recorded command buffer, Promise completion, blit, densify prepare) by bindings/napi/smoke.js on the GPU.
 */
// eslint-disable-next-line @typescript-eslint/no-var-requires
const addon = require('../napi/webdgs_napi.node');

export type RenderMode = 'gaussian' | 'pointcloud';

export class HipBuffer {
  destroyed = false;
  constructor(readonly device: HipDevice, readonly ptr: bigint, readonly size: number, private readonly handle?: bigint) {}
  destroy(): void {
    if (this.destroyed) return;
    this.destroyed = true;
    if (this.handle !== undefined) addon.bufferDestroy(this.handle);
  }
}

/** GPUCommandBuffer backed by an instantiated HIP graph; unlike WebGPU's it may be submitted again (all sizes are read on the device). */
export class HipCommandBuffer {
  constructor(readonly device: HipDevice, readonly handle: bigint | null) {}
  destroy(): void { if (this.handle !== null) addon.commandBufferDestroy(this.handle); }
}

/** GPUCommandEncoder.  Default (eager): encodes go to the HIP stream as they are made; `record: true` captures them into a graph. */
export class HipEncoder {
  constructor(readonly device: HipDevice, readonly label = '', readonly record = false) {
    if (record) addon.encoderBegin(device.handle);
  }
  clearBuffer(buffer: HipBuffer): void { addon.bufferClear(this.device.handle, buffer.ptr, buffer.size); }
  finish(): HipCommandBuffer { return new HipCommandBuffer(this.device, this.record ? addon.encoderFinish(this.device.handle) : null); }
}

export class HipDevice {
  readonly handle: bigint;
  readonly queue = {
    submit: (cmds: HipCommandBuffer[]): void => { for (const c of cmds) if (c.handle !== null) addon.queueSubmit(this.handle, c.handle); },
    onSubmittedWorkDone: (): Promise<void> => addon.queueOnSubmittedWorkDone(this.handle),   // resolved from the HIP runtime thread
    writeBuffer: (buffer: HipBuffer, offset: number, data: ArrayBufferView): void => {
      addon.copyToDevice(this.handle, buffer.ptr + BigInt(offset), data);
    },
  };
  constructor(ordinal = 0) { this.handle = addon.deviceCreate(ordinal); }
  createBuffer(desc: { size: number; label?: string }): HipBuffer {
    const b = addon.bufferCreate(this.handle, desc.size);
    return new HipBuffer(this, b.ptr, desc.size, b.handle);
  }
  createCommandEncoder(desc?: { label?: string; record?: boolean }): HipEncoder { return new HipEncoder(this, desc?.label, desc?.record ?? false); }
  view(ptr: bigint, size: number): HipBuffer { return new HipBuffer(this, ptr, size); }
  readBuffer(buffer: HipBuffer, byteLength = buffer.size): ArrayBuffer { return addon.copyToHost(this.handle, buffer.ptr, byteLength); }
  /** Blocking variant of onSubmittedWorkDone; also raises deferred capacity errors (WDGS_E_CAPACITY). */
  synchronize(): void { addon.deviceSynchronize(this.handle); }
  destroy(): void { addon.deviceDestroy(this.handle); }
}

export interface PointCloud {          // src/utils/load-pointcloud.ts:16-23
  type: 'full' | 'normal';
  num_points: number;
  sh_deg?: number;
  gaussian_3d_buffer: HipBuffer;
  sh_buffer?: HipBuffer;
}

/** allocatePointCloudLike (src/utils/allocate-pointcloud.ts:8-44): zeroed buffers of the template's layout for `numPoints`. */
export function allocatePointCloudLike(device: HipDevice, template: PointCloud, options: { numPoints: number }): PointCloud {
  const n = Math.max(0, Math.floor(options.numPoints));
  return { type: template.type, num_points: n, sh_deg: template.sh_deg,
    gaussian_3d_buffer: device.createBuffer({ size: Math.max(1, n) * 24 }), sh_buffer: device.createBuffer({ size: Math.max(1, n) * 96 }) };
}

export interface TiledForwardPassConfig {  // tiled-forward-pass.ts:24-31
  viewportWidth: number; viewportHeight: number; gaussianScale?: number; pointSizePx?: number; maxSplatRadiusPx?: number; renderMode?: RenderMode;
  maxTileEntries?: number; compatCaps?: boolean;
}

export class TiledForwardPass {          // tiled-forward-pass.ts:62
  private handle: bigint; private destroyed = false;
  constructor(private readonly device: HipDevice, private readonly pointCloud: PointCloud, private cameraBuffer: HipBuffer, config: TiledForwardPassConfig) {
    this.handle = addon.tiledForwardCreate(device.handle, {
      numPoints: pointCloud.num_points, shDeg: pointCloud.sh_deg ?? 0, viewportWidth: config.viewportWidth, viewportHeight: config.viewportHeight,
      gaussianScale: config.gaussianScale ?? 1.0, pointSizePx: config.pointSizePx ?? 3.0, maxSplatRadiusPx: config.maxSplatRadiusPx ?? 128.0,
      renderMode: (config.renderMode ?? 'gaussian') === 'gaussian' ? 1 : 0, maxTileEntries: config.maxTileEntries ?? 0, compatCaps: config.compatCaps ? 1 : 0,
    });
  }
  get nativeHandle(): bigint { return this.handle; }
  encode(_encoder: HipEncoder, options?: { skipSort?: boolean }): void {
    addon.tiledForwardEncode(this.handle, this.pointCloud.gaussian_3d_buffer.ptr, this.pointCloud.sh_buffer!.ptr, this.cameraBuffer.ptr, options?.skipSort ? 1 : 0);
  }
  setCameraBuffer(buffer: HipBuffer): void { this.cameraBuffer = buffer; }
  setRenderMode(mode: RenderMode): void { addon.tiledForwardSet(this.handle, 0, mode === 'gaussian' ? 1 : 0); }
  setPointSize(value: number): void { addon.tiledForwardSet(this.handle, 1, value); }
  setGaussianScale(value: number): void { addon.tiledForwardSet(this.handle, 2, value); }
  setViewport(width: number, height: number): void { addon.tiledForwardSetViewport(this.handle, width, height); }
  getResources() {
    const r = addon.tiledForwardGetResources(this.handle); const n = Math.max(1, this.pointCloud.num_points); const d = this.device;
    return { splatBuffer: d.view(r.splatBuffer, 24 * n), tileKeysBuffer: d.view(r.tileKeysBuffer, 4 * r.maxTileEntries), tileIndicesBuffer: d.view(r.tileIndicesBuffer, 4 * r.maxTileEntries),
      tileOffsetsBuffer: d.view(r.tileOffsetsBuffer, 4 * n), tileCountsBuffer: d.view(r.tileCountsBuffer, 4 * n), statsBuffer: d.view(r.statsBuffer, 16),
      numTilesX: r.numTilesX as number, numTilesY: r.numTilesY as number, totalTiles: r.totalTiles as number, maxTileEntries: r.maxTileEntries as number };
  }
  getSortedIndicesBuffer(): HipBuffer { return this.getResources().tileIndicesBuffer; }
  getSortedKeysBuffer(): HipBuffer { return this.getResources().tileKeysBuffer; }
  getTileOffsetsBuffer(): HipBuffer { return this.getResources().tileOffsetsBuffer; }
  getStatsBuffer(): HipBuffer { return this.getResources().statsBuffer; }
  /** Synchronises; throws (code WDGS_E_CAPACITY) if the last encode overflowed maxTileEntries. */
  check(): { totalTileEntries: number; visibleCount: number } { return addon.tiledForwardCheck(this.handle); }
  destroy(): void { if (this.destroyed) return; this.destroyed = true; addon.tiledForwardDestroy(this.handle); }
}

export class TiledRasterizer {           // tiled-rasterizer.ts:34
  private handle: bigint; private destroyed = false; private w = 0; private h = 0;
  private readonly device: HipDevice;
  constructor(config: { device: HipDevice; forwardPass: TiledForwardPass; format?: string }) {
    this.device = config.device;
    this.handle = addon.tiledRasterizerCreate(config.device.handle, config.forwardPass.nativeHandle);
  }
  encode(_encoder: HipEncoder, width: number, height: number): void { addon.tiledRasterizerEncode(this.handle, width, height); this.w = width; this.h = height; }
  getOutputTextureView(): HipBuffer { return this.device.view(addon.tiledRasterizerGet(this.handle, 0), 4 * this.w * this.h); }      // throws before first encode
  getAlphaTextureView(): HipBuffer { return this.device.view(addon.tiledRasterizerGet(this.handle, 1), 4 * this.w * this.h); }
  getNContribTextureView(): HipBuffer { return this.device.view(addon.tiledRasterizerGet(this.handle, 2), 4 * this.w * this.h); }
  getTileOffsetsBuffer(): HipBuffer { return this.device.view(addon.tiledRasterizerGet(this.handle, 3), 4 * (Math.ceil(this.w / 16) * Math.ceil(this.h / 16) + 1)); }
  /** blitToTexture(encoder, targetView): `target` is an rgba8 image buffer of width x height (default: the rasterizer's size). */
  blitToTexture(_encoder: HipEncoder, target: HipBuffer, width = this.w, height = this.h): void { addon.tiledRasterizerBlit(this.handle, target.ptr, width, height); }
  destroy(): void { if (this.destroyed) return; this.destroyed = true; addon.tiledRasterizerDestroy(this.handle); }
}

export interface TrainingConfig { lambda_l1: number; lambda_l2: number; lambda_dssim: number; c1?: number; c2?: number; }  // tiled-backward-pass.ts:19-25
export interface TiledBackwardResources {  // tiled-backward-pass.ts:40-50
  splatBuffer: HipBuffer; tileOffsetsBuffer: HipBuffer; tileIndicesBuffer: HipBuffer; cameraBuffer?: HipBuffer; alphaTexture?: HipBuffer; nContribTexture: HipBuffer;
}
const resourcePtrs = (r: TiledBackwardResources) => ({ splatBuffer: r.splatBuffer.ptr, tileOffsetsBuffer: r.tileOffsetsBuffer.ptr,
  tileIndicesBuffer: r.tileIndicesBuffer.ptr, cameraBuffer: r.cameraBuffer?.ptr ?? null, alphaTexture: r.alphaTexture?.ptr ?? null, nContribTexture: r.nContribTexture.ptr });

export class TiledBackwardPass {         // tiled-backward-pass.ts:71
  private handle: bigint; private destroyed = false; private w: number; private h: number;
  constructor(private readonly device: HipDevice, private readonly pointCloud: PointCloud,
              config: { viewportWidth: number; viewportHeight: number; trainingConfig: TrainingConfig; maxSplatRadiusPx?: number }) {
    const t = config.trainingConfig;
    this.w = config.viewportWidth; this.h = config.viewportHeight;
    this.handle = addon.tiledBackwardCreate(device.handle, { numPoints: pointCloud.num_points, shDeg: pointCloud.sh_deg ?? 0, viewportWidth: config.viewportWidth,
      viewportHeight: config.viewportHeight, lambda_l1: t.lambda_l1, lambda_l2: t.lambda_l2, lambda_dssim: t.lambda_dssim, c1: t.c1 ?? 0.0001, c2: t.c2 ?? 0.0009,
      maxSplatRadiusPx: config.maxSplatRadiusPx ?? 128.0 });
  }
  encode(_encoder: HipEncoder, predictedTexture: HipBuffer, targetTexture: HipBuffer, r: TiledBackwardResources): void {
    addon.tiledBackwardEncode(this.handle, predictedTexture.ptr, targetTexture.ptr, resourcePtrs(r), this.pointCloud.gaussian_3d_buffer.ptr);
  }
  computeLossOnly(_encoder: HipEncoder, predicted: HipBuffer, target: HipBuffer): void { addon.tiledBackwardMetric(this.handle, 0, predicted.ptr, target.ptr, 0); }
  computeMetricMap(_encoder: HipEncoder, predicted: HipBuffer, target: HipBuffer, options?: { threshold?: number }): void {
    addon.tiledBackwardMetric(this.handle, 1, predicted.ptr, target.ptr, options?.threshold ?? 0.5);
  }
  computeMetricCounts(_encoder: HipEncoder, r: TiledBackwardResources, options?: { clear?: boolean; numInstances?: number }): void {
    addon.tiledBackwardMetric(this.handle, 2, resourcePtrs(r), options?.numInstances ?? 0, options?.clear === false ? 0 : 1);
  }
  normalizeMetricCounts(_encoder: HipEncoder, options: { divisor: number }): void { addon.tiledBackwardMetric(this.handle, 3, options.divisor, 0, 0); }
  setViewport(width: number, height: number): void { addon.tiledBackwardMetric(this.handle, 4, width, height, 0); this.w = width; this.h = height; }
  getGradientsBuffer(): HipBuffer { return this.device.view(addon.tiledBackwardGet(this.handle, 0), 32 * Math.max(1, this.pointCloud.num_points)); }
  getMetricCountsBuffer(): HipBuffer { return this.device.view(addon.tiledBackwardGet(this.handle, 1), 4 * Math.max(1, this.pointCloud.num_points)); }
  getLossTextureView(): HipBuffer { return this.device.view(addon.tiledBackwardGet(this.handle, 2), 16 * this.w * this.h); }
  getMetricMapTextureView(): HipBuffer { return this.device.view(addon.tiledBackwardGet(this.handle, 3), 4 * this.w * this.h); }
  destroy(): void { if (this.destroyed) return; this.destroyed = true; addon.tiledBackwardDestroy(this.handle); }
}

export interface AdamHyperparameters { lr_pos: number; lr_color: number; lr_opacity: number; lr_scale: number; lr_rot: number; beta1: number; beta2: number; epsilon: number; }
export interface OptimizerStateBuffers {   // optimizer.ts:13-20
  optPosBuffer: HipBuffer; optRotBuffer: HipBuffer; optScaleBuffer: HipBuffer; optOpacityBuffer: HipBuffer; paramSH: HipBuffer; stateSH: HipBuffer;
}
const STATE_KEYS = ['optPosBuffer', 'optRotBuffer', 'optScaleBuffer', 'optOpacityBuffer', 'paramSH', 'stateSH'] as const;
const statePtrs = (s: OptimizerStateBuffers) => Object.fromEntries(STATE_KEYS.map((k) => [k, s[k].ptr]));

/** allocateOptimizerStateBuffers (optimizer.ts:27-38). */
export function allocateOptimizerStateBuffers(device: HipDevice, numPoints: number): OptimizerStateBuffers {
  const sizes: number[] = addon.optimizerStateSizes(Math.max(1, numPoints));
  return Object.fromEntries(STATE_KEYS.map((k, i) => [k, device.createBuffer({ size: sizes[i] })])) as unknown as OptimizerStateBuffers;
}

export class Optimizer {                 // optimizer.ts:40
  private handle: bigint; private destroyed = false;
  /** `initialState` is ADOPTED as-is (optimizer.ts:81-88); without it the state is allocated and initialised from the point cloud. */
  constructor(private readonly device: HipDevice, private readonly pointCloud: PointCloud, params?: Partial<AdamHyperparameters>,
              initialState?: { state: OptimizerStateBuffers; iteration?: number }) {
    this.handle = initialState
      ? addon.optimizerCreateWithState(device.handle, pointCloud.num_points, pointCloud.gaussian_3d_buffer.ptr, pointCloud.sh_buffer!.ptr, statePtrs(initialState.state), 0,
                                       initialState.iteration ?? 0)
      : addon.optimizerCreate(device.handle, pointCloud.num_points, pointCloud.gaussian_3d_buffer.ptr, pointCloud.sh_buffer!.ptr);
    if (params) addon.optimizerHyperparameters(this.handle, params);
  }
  getIteration(): number { return addon.optimizerGetIteration(this.handle); }
  getHyperparameters(): AdamHyperparameters { return addon.optimizerHyperparameters(this.handle, null); }
  setHyperparameters(next: Partial<AdamHyperparameters>): void { addon.optimizerHyperparameters(this.handle, next); }
  /** Brings the SH-DC rows of paramSH / stateSH up to date before handing the arrays out (see include/webdgs.h). */
  getStateBuffers(): OptimizerStateBuffers {
    const s = addon.optimizerState(this.handle, 0); const n = Math.max(1, this.pointCloud.num_points); const sizes: number[] = addon.optimizerStateSizes(n);
    return Object.fromEntries(STATE_KEYS.map((k, i) => [k, this.device.view(s[k], sizes[i])])) as unknown as OptimizerStateBuffers;
  }
  step(_encoder: HipEncoder, coefficients: PointCloud, gradientsBuffer: HipBuffer, tileCountsBuffer: HipBuffer): void {
    addon.optimizerStep(this.handle, coefficients.gaussian_3d_buffer.ptr, coefficients.sh_buffer!.ptr, gradientsBuffer.ptr, tileCountsBuffer.ptr);
  }
  /** Host-side iteration counter: call when a recorded command buffer containing step() is re-submitted. */
  advanceIteration(count = 1): void { addon.optimizerAdvanceIteration(this.handle, count); }
  destroy(): void { if (this.destroyed) return; this.destroyed = true; addon.optimizerDestroy(this.handle); }
}

export interface DensifyPruneConfig {      // densify-prune.ts:17-30
  strategy?: 'cpu_rebuild' | 'gpu_rebuild'; numViews?: number; cloneThreshold?: number; splitThreshold?: number; pruneThreshold?: number;
  maxNewPointsPerStep?: number; maxBufferBytes?: number;
}
export interface DensifyPrunePrepared {    // densify-prune.ts:42-48
  actionBuffer: HipBuffer; outCountBuffer: HipBuffer; outOffsetBuffer: HipBuffer; outTotalBuffer: HipBuffer; maxOutPoints: number;
}

export class DensifyPrunePass {          // densify-prune.ts:75
  private handle: bigint; private config: DensifyPruneConfig;
  constructor(private readonly device: HipDevice, config: DensifyPruneConfig = {}) {
    this.config = { strategy: 'cpu_rebuild', numViews: 1, cloneThreshold: 0, pruneThreshold: 0, maxNewPointsPerStep: 0, maxBufferBytes: 128 * 1024 * 1024, ...config };
    this.handle = addon.densifyCreate(device.handle, this.config);
  }
  setConfig(next: Partial<DensifyPruneConfig>): void { this.config = { ...this.config, ...next }; addon.densifySetConfig(this.handle, this.config); }
  getConfig(): DensifyPruneConfig { return { ...this.config }; }
  encodePrepare(_encoder: HipEncoder, inputs: { pointCloud: PointCloud; metricCountsBuffer?: HipBuffer }): DensifyPrunePrepared {
    const n = inputs.pointCloud.num_points; const d = this.device;
    const p = addon.densifyEncodePrepare(this.handle, n, inputs.pointCloud.gaussian_3d_buffer.ptr, inputs.metricCountsBuffer?.ptr ?? null);
    return { actionBuffer: d.view(p.actionBuffer, 4 * n), outCountBuffer: d.view(p.outCountBuffer, 4 * n), outOffsetBuffer: d.view(p.outOffsetBuffer, 4 * n),
      outTotalBuffer: d.view(p.outTotalBuffer, 4), maxOutPoints: p.maxOutPoints };
  }
  // ---- the stages encodePrepare is made of (densify-prune.ts:327-456), individually recordable as in the reference
  private numPoints = 0;
  private stage(stage: number, n: number, a: unknown = 0, b: unknown = 0): DensifyPrunePrepared {
    const p = addon.densifyStage(this.handle, stage, n, a, b); const d = this.device; const m = Math.max(1, n);
    return { actionBuffer: d.view(p.actionBuffer, 4 * m), outCountBuffer: d.view(p.outCountBuffer, 4 * m), outOffsetBuffer: d.view(p.outOffsetBuffer, 4 * m),
      outTotalBuffer: d.view(p.outTotalBuffer, 4), maxOutPoints: p.maxOutPoints };
  }
  ensureSize(numPoints: number): void { this.numPoints = numPoints; this.stage(4, numPoints); }
  computeMaxOutPoints(pointCloud: PointCloud): number { return this.stage(4, pointCloud.num_points).maxOutPoints; }
  encodeDecision(_encoder: HipEncoder, inputs: { pointCloud: PointCloud; metricCountsBuffer?: HipBuffer }): { actionBuffer: HipBuffer; outCountBuffer: HipBuffer } {
    this.numPoints = inputs.pointCloud.num_points;
    const p = this.stage(0, this.numPoints, inputs.pointCloud.gaussian_3d_buffer.ptr, inputs.metricCountsBuffer?.ptr ?? null);
    return { actionBuffer: p.actionBuffer, outCountBuffer: p.outCountBuffer };
  }
  encodePrefixSum(_encoder: HipEncoder): HipBuffer { return this.stage(1, this.numPoints).outOffsetBuffer; }
  encodeCapToMax(_encoder: HipEncoder, _outOffsetBuffer: HipBuffer, maxOutPoints: number): void { this.stage(2, this.numPoints, Math.max(0, Math.floor(maxOutPoints))); }
  encodeTotalOut(_encoder: HipEncoder, _outOffsetBuffer?: HipBuffer): HipBuffer { return this.stage(3, this.numPoints).outTotalBuffer; }
  /** The one 4-byte read-back of the densify path (trainer.ts:440-458, mapAsync on outTotalBuffer). */
  readTotal(): number { return addon.densifyReadTotal(this.handle); }
  encodeScatter(_encoder: HipEncoder,
                inputs: { pointCloud: PointCloud; optimizerState?: OptimizerStateBuffers; outOffsetBuffer: HipBuffer; outNumPoints: number; resetNewOptimizerState?: boolean },
                outputs: { outPointCloud: PointCloud; outOptimizerState?: OptimizerStateBuffers }): void {
    if (outputs.outPointCloud.num_points !== inputs.outNumPoints) throw new Error('encodeScatter: outPointCloud.num_points must equal outNumPoints');  // densify-prune.ts:478-480
    addon.densifyEncodeScatter(this.handle, inputs.pointCloud.num_points, inputs.pointCloud.gaussian_3d_buffer.ptr, inputs.pointCloud.sh_buffer!.ptr,
      inputs.optimizerState ? statePtrs(inputs.optimizerState) : null, inputs.outNumPoints, inputs.resetNewOptimizerState === false ? 0 : 1,
      outputs.outPointCloud.gaussian_3d_buffer.ptr, outputs.outPointCloud.sh_buffer!.ptr, outputs.outOptimizerState ? statePtrs(outputs.outOptimizerState) : null);
  }
  applyActions(): never { throw new Error('DensifyPrunePass.applyActions is unimplemented in the reference (densify-prune.ts:680-686)'); }
  destroy(): void { addon.densifyDestroy(this.handle); }
}

/** Bilinear blit of an rgba8 image to another size (trainer.ts:303-328: the ground-truth down-sample of the metric views). */
export function downsampleRGBA8(device: HipDevice, src: HipBuffer, srcW: number, srcH: number, dst: HipBuffer, dstW: number, dstH: number): void {
  addon.downsampleRGBA8(device.handle, src.ptr, srcW, srcH, dst.ptr, dstW, dstH);
}
