/*
 * webdgs_hip.ts -- drop-in module for the reference's operator layer (src/renderers/tiled-forward-pass.ts,
 * tiled-rasterizer.ts, tiled-backward-pass.ts, optimizer.ts) backed by the N-API addon over libwebdgs_hip.so.
 *
 * Same class names, constructor shapes and method names as the reference; GPUDevice / GPUBuffer / GPUTextureView /
 * GPUCommandEncoder become HipDevice / HipBuffer / HipEncoder.  A src/trainer.ts that imports these instead of the WebGPU
 * classes needs no other change than the import lines and `device.queue.onSubmittedWorkDone()` (same name here).
 * NOT type-checked in this repository's image (no tsc); the addon underneath is compiled and smoke-tested with node.
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

export class HipEncoder {
  constructor(readonly device: HipDevice, readonly label = '') {}
  finish(): HipEncoder { return this; }
}

export class HipDevice {
  readonly handle: bigint;
  readonly queue = {
    submit: (_cmds: HipEncoder[]): void => { /* work is already on the stream, in encode order */ },
    onSubmittedWorkDone: (): Promise<void> => new Promise((resolve, reject) => {
      try { addon.deviceSynchronize(this.handle); resolve(); } catch (e) { reject(e); }
    }),
    writeBuffer: (buffer: HipBuffer, offset: number, data: ArrayBufferView): void => {
      addon.copyToDevice(this.handle, buffer.ptr + BigInt(offset), data);
    },
  };
  constructor(ordinal = 0) { this.handle = addon.deviceCreate(ordinal); }
  createBuffer(desc: { size: number; label?: string }): HipBuffer {
    const b = addon.bufferCreate(this.handle, desc.size);
    return new HipBuffer(this, b.ptr, desc.size, b.handle);
  }
  createCommandEncoder(desc?: { label?: string }): HipEncoder { return new HipEncoder(this, desc?.label); }
  view(ptr: bigint, size: number): HipBuffer { return new HipBuffer(this, ptr, size); }
  readBuffer(buffer: HipBuffer, byteLength = buffer.size): ArrayBuffer { return addon.copyToHost(this.handle, buffer.ptr, byteLength); }
  destroy(): void { addon.deviceDestroy(this.handle); }
}

export interface PointCloud {          // src/utils/load-pointcloud.ts:16-23
  type: 'full' | 'normal';
  num_points: number;
  sh_deg?: number;
  gaussian_3d_buffer: HipBuffer;
  sh_buffer?: HipBuffer;
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
  destroy(): void { if (this.destroyed) return; this.destroyed = true; addon.tiledRasterizerDestroy(this.handle); }
}

export interface TrainingConfig { lambda_l1: number; lambda_l2: number; lambda_dssim: number; c1?: number; c2?: number; }  // tiled-backward-pass.ts:19-25
export interface TiledBackwardResources {  // tiled-backward-pass.ts:40-50
  splatBuffer: HipBuffer; tileOffsetsBuffer: HipBuffer; tileIndicesBuffer: HipBuffer; cameraBuffer: HipBuffer; alphaTexture: HipBuffer; nContribTexture: HipBuffer;
}

export class TiledBackwardPass {         // tiled-backward-pass.ts:71
  private handle: bigint; private destroyed = false;
  constructor(private readonly device: HipDevice, private readonly pointCloud: PointCloud,
              config: { viewportWidth: number; viewportHeight: number; trainingConfig: TrainingConfig; maxSplatRadiusPx?: number }) {
    const t = config.trainingConfig;
    this.handle = addon.tiledBackwardCreate(device.handle, { numPoints: pointCloud.num_points, shDeg: pointCloud.sh_deg ?? 0, viewportWidth: config.viewportWidth,
      viewportHeight: config.viewportHeight, lambda_l1: t.lambda_l1, lambda_l2: t.lambda_l2, lambda_dssim: t.lambda_dssim, c1: t.c1 ?? 0.0001, c2: t.c2 ?? 0.0009,
      maxSplatRadiusPx: config.maxSplatRadiusPx ?? 128.0 });
  }
  encode(_encoder: HipEncoder, predictedTexture: HipBuffer, targetTexture: HipBuffer, r: TiledBackwardResources): void {
    addon.tiledBackwardEncode(this.handle, predictedTexture.ptr, targetTexture.ptr, { splatBuffer: r.splatBuffer.ptr, tileOffsetsBuffer: r.tileOffsetsBuffer.ptr,
      tileIndicesBuffer: r.tileIndicesBuffer.ptr, cameraBuffer: r.cameraBuffer.ptr, alphaTexture: r.alphaTexture.ptr, nContribTexture: r.nContribTexture.ptr },
      this.pointCloud.gaussian_3d_buffer.ptr);
  }
  getGradientsBuffer(): HipBuffer { return this.device.view(addon.tiledBackwardGradients(this.handle), 32 * Math.max(1, this.pointCloud.num_points)); }
  destroy(): void { if (this.destroyed) return; this.destroyed = true; addon.tiledBackwardDestroy(this.handle); }
}

export class Optimizer {                 // optimizer.ts:40
  private handle: bigint; private destroyed = false;
  constructor(device: HipDevice, pointCloud: PointCloud) {
    this.handle = addon.optimizerCreate(device.handle, pointCloud.num_points, pointCloud.gaussian_3d_buffer.ptr, pointCloud.sh_buffer!.ptr);
  }
  getIteration(): number { return addon.optimizerGetIteration(this.handle); }
  step(_encoder: HipEncoder, coefficients: PointCloud, gradientsBuffer: HipBuffer, tileCountsBuffer: HipBuffer): void {
    addon.optimizerStep(this.handle, coefficients.gaussian_3d_buffer.ptr, coefficients.sh_buffer!.ptr, gradientsBuffer.ptr, tileCountsBuffer.ptr);
  }
  destroy(): void { if (this.destroyed) return; this.destroyed = true; addon.optimizerDestroy(this.handle); }
}
