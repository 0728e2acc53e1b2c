// Typings of webdgs_hip.js: the reference's operator classes (src/renderers/*.ts, src/sort/sort_dynamic.ts, src/prefix/prefix.ts,
// src/utils/allocate-pointcloud.ts) with GPUDevice / GPUBuffer / GPUTextureView / GPUCommandEncoder replaced by Hip* handle types.
export type RenderMode = 'gaussian' | 'pointcloud';

export class HipBuffer {
  readonly device: HipDevice; readonly ptr: bigint; readonly size: number; destroyed: boolean;
  destroy(): void;
}
export class HipCommandBuffer { readonly device: HipDevice; destroy(): void; }
export class HipEncoder {
  readonly device: HipDevice; readonly label: string; readonly record: boolean;
  clearBuffer(buffer: HipBuffer): void;
  finish(): HipCommandBuffer;
  abort(): void;
}
export class HipDevice {
  constructor(ordinal?: number);
  readonly queue: {
    submit(cmds: HipCommandBuffer[]): void;
    onSubmittedWorkDone(): Promise<void>;
    /** include/webdgs.h wdgs_queue_mark / wdgs_queue_wait: keep the completion of step k and await it after submitting step k+1. */
    mark(): number;
    wait(ticket: number): void;
    writeBuffer(buffer: HipBuffer, offset: number, data: ArrayBufferView | ArrayBuffer): void;
  };
  createBuffer(desc: { size: number; label?: string }): HipBuffer;
  createCommandEncoder(desc?: { label?: string; record?: boolean }): HipEncoder;
  view(ptr: bigint, size: number): HipBuffer;
  readBuffer(buffer: HipBuffer, byteLength?: number): ArrayBuffer;
  createPinnedArrayBuffer(byteLength: number): ArrayBuffer;
  readBufferAsync(buffer: HipBuffer, offset: number, pinned: ArrayBuffer, byteLength: number): Promise<ArrayBuffer>;
  synchronize(): void;
  /** Lanes (include/webdgs.h): lane 0 is the device's stream, 1..3 internal ones; work on different lanes may overlap. */
  selectLane(lane: number): void;
  laneOrder(waiterLane: number, signalLane: number): void;
  destroy(): void;
}

export interface PointCloud {          // src/utils/load-pointcloud.ts:16-23
  type: 'full' | 'normal'; num_points: number; sh_deg?: number; gaussian_3d_buffer: HipBuffer; sh_buffer?: HipBuffer;
}
export function allocatePointCloudLike(device: HipDevice, template: PointCloud, options: { numPoints: number }): PointCloud;

export class PrefixScanner {             // src/prefix/prefix.ts:26-43
  readonly input_buffer: HipBuffer; readonly output_buffer: HipBuffer; readonly max_elements: number;
  set_count(count: number): { num_workgroups: number };
  scan(encoder: HipEncoder | null): void;
  destroy(): void;
}
export function get_prefix_scanner(maxElements: number, device: HipDevice): PrefixScanner;
export class DynamicSortStuff {          // src/sort/sort_dynamic.ts:9-24
  readonly capacity: number; final_out_index: number;
  readonly ping_pong: { sort_depths_buffer: HipBuffer; sort_indices_buffer: HipBuffer }[];
  sort(encoder: HipEncoder | null, keyBits?: number): void;
  destroy(): void;
}
export function get_dynamic_sorter(maxCapacity: number, device: HipDevice, statsBuffer: HipBuffer): DynamicSortStuff;

export interface TiledForwardPassConfig {  // tiled-forward-pass.ts:24-31
  viewportWidth: number; viewportHeight: number; gaussianScale?: number; pointSizePx?: number; maxSplatRadiusPx?: number; renderMode?: RenderMode;
  maxTileEntries?: number; compatCaps?: boolean;
}
export interface TiledForwardResources {   // tiled-forward-pass.ts:33-46
  splatBuffer: HipBuffer; tileKeysBuffer: HipBuffer; tileIndicesBuffer: HipBuffer; tileOffsetsBuffer: HipBuffer; tileCountsBuffer: HipBuffer;
  statsBuffer: HipBuffer; numTilesX: number; numTilesY: number; totalTiles: number; maxTileEntries: number;
}
export class TiledForwardPass {
  constructor(device: HipDevice, pointCloud: PointCloud, cameraBuffer: HipBuffer, config: TiledForwardPassConfig);
  readonly nativeHandle: bigint;
  encode(encoder: HipEncoder | null, options?: { skipSort?: boolean }): void;
  setCameraBuffer(buffer: HipBuffer): void;
  /** Resize instead of rebuild (include/webdgs.h wdgs_tiled_forward_resize); false if the SH degree differs. */
  setPointCloud(pointCloud: PointCloud): boolean;
  /** K1 takes the SH-DC halves from the optimizer's compact array (Optimizer.setDeferredSH); null restores the rows. */
  setDcSource(dcWords: HipBuffer | null): void;
  setRenderMode(mode: RenderMode): void; setPointSize(value: number): void; setGaussianScale(value: number): void; setViewport(width: number, height: number): void;
  getResources(): TiledForwardResources;
  getSortedIndicesBuffer(): HipBuffer; getSortedKeysBuffer(): HipBuffer; getTileOffsetsBuffer(): HipBuffer; getStatsBuffer(): HipBuffer;
  check(): { totalTileEntries: number; visibleCount: number };
  destroy(): void;
}
export class TiledRasterizer {
  constructor(config: { device: HipDevice; forwardPass: TiledForwardPass; format?: string });
  encode(encoder: HipEncoder | null, width: number, height: number): void;
  getOutputTextureView(): HipBuffer; getAlphaTextureView(): HipBuffer; getNContribTextureView(): HipBuffer; getTileOffsetsBuffer(): HipBuffer;
  /** `target` may carry its own `width` / `height` (a canvas of another size); the clear colour is accepted for signature compatibility only. */
  blitToTexture(encoder: HipEncoder | null, target: HipBuffer & { width?: number; height?: number }, clearColor?: { r: number; g: number; b: number; a: number }): void;
  destroy(): void;
}
export interface TrainingConfig { lambda_l1: number; lambda_l2: number; lambda_dssim: number; c1?: number; c2?: number; }  // tiled-backward-pass.ts:19-25
export interface TiledBackwardResources {  // tiled-backward-pass.ts:40-50
  splatBuffer: HipBuffer; tileOffsetsBuffer: HipBuffer; tileIndicesBuffer: HipBuffer; cameraBuffer?: HipBuffer; alphaTexture?: HipBuffer; nContribTexture: HipBuffer;
}
export class TiledBackwardPass {
  constructor(device: HipDevice, pointCloud: PointCloud, config: { viewportWidth: number; viewportHeight: number; trainingConfig: TrainingConfig; maxSplatRadiusPx?: number });
  encode(encoder: HipEncoder | null, predictedTexture: HipBuffer, targetTexture: HipBuffer, resources: TiledBackwardResources, options?: {}): void;
  setTrainingConfig(next: Partial<TrainingConfig>): void;
  getMetricMapTexture(): HipBuffer;
  computeLossOnly(encoder: HipEncoder | null, predicted: HipBuffer, target: HipBuffer): void;
  computeMetricMap(encoder: HipEncoder | null, predicted: HipBuffer, target: HipBuffer, options?: { threshold?: number }): void;
  computeMetricCounts(encoder: HipEncoder | null, resources: TiledBackwardResources, options?: { clear?: boolean; numInstances?: number }): void;
  normalizeMetricCounts(encoder: HipEncoder | null, options: { divisor: number }): void;
  setViewport(width: number, height: number): void;
  /** Resize instead of rebuild (include/webdgs.h wdgs_tiled_backward_resize); false if the SH degree differs. */
  setPointCloud(pointCloud: PointCloud): boolean;
  getGradientsBuffer(): HipBuffer; getMetricCountsBuffer(): HipBuffer; getLossTextureView(): HipBuffer; getMetricMapTextureView(): HipBuffer;
  destroy(): void;
}
export interface AdamHyperparameters { lr_pos: number; lr_color: number; lr_opacity: number; lr_scale: number; lr_rot: number; beta1: number; beta2: number; epsilon: number; }
export const DEFAULT_ADAM_HYPERPARAMETERS: AdamHyperparameters;
export interface OptimizerStateBuffers {   // optimizer.ts:13-20
  optPosBuffer: HipBuffer; optRotBuffer: HipBuffer; optScaleBuffer: HipBuffer; optOpacityBuffer: HipBuffer; paramSH: HipBuffer; stateSH: HipBuffer;
}
export interface OptimizerInitialState { iteration: number; buffers: OptimizerStateBuffers; }   // optimizer.ts:22-25
export function allocateOptimizerStateBuffers(device: HipDevice, numPoints: number): OptimizerStateBuffers;
export class Optimizer {
  constructor(device: HipDevice, pointCloud: PointCloud, params?: Partial<AdamHyperparameters>, initialState?: OptimizerInitialState);
  getIteration(): number; getHyperparameters(): AdamHyperparameters; setHyperparameters(next: Partial<AdamHyperparameters>): void;
  getStateBuffers(): OptimizerStateBuffers;
  step(encoder: HipEncoder | null, coefficients: PointCloud, gradientsBuffer: HipBuffer, tileCountsBuffer: HipBuffer): void;
  /** Deferred SH writes (include/webdgs.h wdgs_optimizer_set_deferred_sh); no reference counterpart. */
  setDeferredSH(pointCloud: PointCloud, enabled?: boolean): HipBuffer | null;
  flushSH(pointCloud: PointCloud): void;
  setGuard(flagBuffer: HipBuffer | null, offset?: number): void;
  advanceIteration(count?: number): void;
  destroy(): void;
}
export interface DensifyPruneConfig {      // densify-prune.ts:17-30
  strategy?: 'cpu_rebuild' | 'gpu_rebuild'; numViews?: number; cloneThreshold?: number; splitThreshold?: number; pruneThreshold?: number;
  maxNewPointsPerStep?: number; maxBufferBytes?: number;
}
export interface DensifyPrunePrepared {    // densify-prune.ts:42-48
  actionBuffer: HipBuffer; outCountBuffer: HipBuffer; outOffsetBuffer: HipBuffer; outTotalBuffer: HipBuffer; maxOutPoints: number;
}
export class DensifyPrunePass {
  constructor(device: HipDevice, config?: DensifyPruneConfig);
  setConfig(next: Partial<DensifyPruneConfig>): void; getConfig(): DensifyPruneConfig;
  ensureSize(numPoints: number): void; computeMaxOutPoints(pointCloud: PointCloud): number;
  encodeDecision(encoder: HipEncoder | null, inputs: { pointCloud: PointCloud; metricCountsBuffer?: HipBuffer }): { actionBuffer: HipBuffer; outCountBuffer: HipBuffer };
  encodePrefixSum(encoder: HipEncoder | null): HipBuffer;
  encodeCapToMax(encoder: HipEncoder | null, outOffsetBuffer: HipBuffer, maxOutPoints: number): void;
  encodeTotalOut(encoder: HipEncoder | null, outOffsetBuffer: HipBuffer | null): HipBuffer;
  getActionBuffer(): HipBuffer | null;
  getOutCountBuffer(): HipBuffer | null;
  getOutTotalBuffer(): HipBuffer;
  encodePrepare(encoder: HipEncoder | null, inputs: { pointCloud: PointCloud; metricCountsBuffer?: HipBuffer }): DensifyPrunePrepared;
  readTotal(): number;
  encodeScatter(encoder: HipEncoder | null,
                inputs: { pointCloud: PointCloud; optimizerState?: OptimizerStateBuffers; outOffsetBuffer: HipBuffer; outNumPoints: number; resetNewOptimizerState?: boolean },
                outputs: { outPointCloud: PointCloud; outOptimizerState?: OptimizerStateBuffers }): void;
  applyActions(encoder: HipEncoder | null): never;
  destroy(): void;
}
export function downsampleRGBA8(device: HipDevice, src: HipBuffer, srcW: number, srcH: number, dst: HipBuffer, dstW: number, dstH: number): void;
