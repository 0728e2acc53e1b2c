// Typings of webdgs_hip.js: the reference's operator classes (src/renderers/*.ts, src/sort/sort_dynamic.ts, src/prefix/prefix.ts,
// src/utils/allocate-pointcloud.ts) with GPUDevice / GPUBuffer / GPUTextureView / GPUCommandEncoder replaced by Hip* handle types.
export type RenderMode = 'gaussian' | 'pointcloud';

export class HipBuffer {
  readonly device: HipDevice; readonly ptr: bigint; readonly size: number; destroyed: boolean;
  /** Called before the buffer's content is handed to the host or copied by copyBufferToBuffer (deferred SH writes, the compact training copy). */
  beforeRead: (() => void) | null;
  read(byteLength?: number): ArrayBuffer;
  destroy(): void;
}
export class HipCommandBuffer { readonly device: HipDevice; destroy(): void; }
export class HipEncoder {
  readonly device: HipDevice; readonly label: string; readonly record: boolean;
  clearBuffer(buffer: HipBuffer): void;
  copyBufferToBuffer(src: HipBuffer, srcOffset: number, dst: HipBuffer, dstOffset: number, size: number): void;
  finish(): HipCommandBuffer;
  abort(): void;
}
/** Capacity reports that reached the wrong owner (a Trainer and a Viewer sharing a device): left here for the owner of the passes they name. */
export class CapacityReports {
  constructor(keep?: number);
  pending: Error[];
  static passesNamed(error: Error): bigint[];
  post(error: Error): void;
  take(ownHandles: Array<bigint | number>): Error | null;
}
export class HipDevice {
  constructor(ordinal?: number);
  readonly capacityReports: CapacityReports;
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
  /** device.limits as far as memory goes (trainer.ts:147): bytes; `cached` is what the library's allocation cache holds. */
  memoryInfo(): { free: number; total: number; cached: number };
  /** Lanes (include/webdgs.h): lane 0 is the device's stream, 1..3 internal ones; work on different lanes may overlap. */
  selectLane(lane: number): void;
  laneOrder(waiterLane: number, signalLane: number): void;
  laneMark(lane: number, mark: number): void;
  laneWaitMark(lane: number, mark: number): void;
  /** Per-kernel hipEvent timing (wdgs_device_set_profiling / wdgs_device_get_kernel_times). */
  setProfiling(enabled: boolean): void;
  kernelTimes(reset?: boolean): { [kernel: string]: { launches: number; totalMs: number } };
  destroy(): void;
}
export const MAX_LANES: number;
export const MAX_BATCH_VIEWS: number;

export interface PointCloud {          // src/utils/load-pointcloud.ts:16-23
  type: 'full' | 'normal'; num_points: number; sh_deg?: number; gaussian_3d_buffer: HipBuffer; sh_buffer?: HipBuffer;
  /** Set while an Optimizer trains this cloud with deferred SH writes: the compact SH-DC array forward passes built on the cloud read. */
  dcWords?: HipBuffer | null;
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
  /** Follows pointCloud.dcWords (called by encode / projectViews): a pass built on a cloud in training renders the trained colours by itself. */
  syncDcSource(): void;
  /** The rest of encode (scan, emit, sort) after projectViews ran K1 for this pass (view-batched step; no reference counterpart). */
  encodeProjected(encoder: HipEncoder | null): void;
  isProjected(): boolean;
  setRenderMode(mode: RenderMode): void; setPointSize(value: number): void; setGaussianScale(value: number): void; setViewport(width: number, height: number): void;
  getResources(): TiledForwardResources;
  getSortedIndicesBuffer(): HipBuffer; getSortedKeysBuffer(): HipBuffer; getTileOffsetsBuffer(): HipBuffer; getStatsBuffer(): HipBuffer;
  check(): { totalTileEntries: number; visibleCount: number };
  setLongLists(threshold: number, maxItems?: number, maxRows?: number): void;
  longListStats(): { blocksWanted: number; itemsWanted: number; forwardQueue: number; backwardQueue: number; rowsUsed: number; rowsWanted: number; stalled: number; maxItems: number; maxBlocks: number; maxRows: number; threshold: number };
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
  /** The two halves of encode (a batched step; the fused single-view step): K15 + clear + K16, then K17. */
  encodeRaster(encoder: HipEncoder | null, predictedTexture: HipBuffer, targetTexture: HipBuffer, resources: TiledBackwardResources): void;
  encodeGeometry(encoder: HipEncoder | null, cameraBuffer: HipBuffer,
                 accumulate?: { sums: HipBuffer; visible: HipBuffer; tileCounts: HipBuffer; guard: HipBuffer; stats: HipBuffer; first: boolean } | null): void;
  /** Whether Optimizer.stepWithGeometry also writes K17's packed gradient to getGradientsBuffer() (default true, as the reference's K17 does). */
  setGradientOutput(enabled: boolean): void;
  /** computeMetricCounts adds into `counts` (another pass's metric counts) instead of this pass's own array; null restores its own. */
  setMetricCountsTarget(counts: HipBuffer | null): void;
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
  /** step() fused with K17 (wdgs_optimizer_step_with_geometry): after backwardPass.encodeRaster for the view. */
  stepWithGeometry(encoder: HipEncoder | null, coefficients: PointCloud, backwardPass: TiledBackwardPass, cameraBuffer: HipBuffer, tileCountsBuffer: HipBuffer): void;
  /** Data-parallel step (SURVEY 8(e)): Adam + re-pack on the owned slice; rowsOut receives the re-packed 32-byte rows. */
  stepF32Range(encoder: HipEncoder | null, coefficients: PointCloud, gradF32: HipBuffer, visibleCounts: HipBuffer, first: number, count: number, rowsOut?: HipBuffer | null): void;
  applyRepackedRows(rows: HipBuffer, skipFirst: number, skipCount: number, guard: HipBuffer | null, pointCloud: PointCloud): void;
  stateChanged(): void;
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
/** View-batched K1 / K17 (include/webdgs.h wdgs_tiled_forward_project_views, wdgs_tiled_backward_encode_geometry_views); no reference counterpart. */
export function projectViews(forwardPasses: TiledForwardPass[], cameraBuffers: HipBuffer[], pointCloud: PointCloud): void;
export function geometryViews(backwardPasses: TiledBackwardPass[], cameraBuffers: HipBuffer[], forwardPasses: TiledForwardPass[], sums: HipBuffer, visible: HipBuffer, guard: HipBuffer,
                              pointCloud: PointCloud, writeGradients?: boolean, continues?: boolean): void;
export function imageSSE(device: HipDevice, a: HipBuffer, b: HipBuffer, numPixels: number): number;
export function imagePSNR(device: HipDevice, a: HipBuffer, b: HipBuffer, numPixels: number): number;
/** The C-ABI communicator (wdgs_comm_*): RCCL on the device's stream, for the data-parallel step of a host without torch.distributed. */
export class Communicator {
  constructor(device: HipDevice, uniqueId: ArrayBuffer, worldSize: number, rank: number);
  readonly worldSize: number; readonly rank: number;
  static uniqueId(): ArrayBuffer;
  exchangeGradients(grad: HipBuffer, visible: HipBuffer, flag: HipBuffer | null, slicePoints: number): void;
  allgatherRows(rows: HipBuffer, slicePoints: number): void;
  broadcast(ptr: bigint, bytes: number, root: number): void;
  allreduceCounts(counts: HipBuffer, count: number): void;
  allreduceGradients(grad: HipBuffer, visible: HipBuffer, numPoints: number): void;
  static groupStart(): void; static groupEnd(): void;
  destroy(): void;
}
