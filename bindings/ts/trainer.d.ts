// Typings of trainer.js: the public surface of the reference's Trainer (src/trainer.ts:177-566).
import { AdamHyperparameters, HipBuffer, HipDevice, OptimizerInitialState, PointCloud, TrainingConfig } from './webdgs_hip';
import { CameraData } from './loaders';
import { LoadedImage } from './images';
import { Exchange } from './parallel';

export interface DensifyPruneTrainingConfig {   // trainer.ts:23-40
  schedule: { enabled: boolean; warmupIterations: number; interval: number; stopIterations: number };
  metricViews: number; metricDownscale: number; metricThreshold: number; maxBufferBytes: number; maxNewPointsPerStep: number;
  pruneOpacity: number; cloneThresholdCount: number; splitScaleThreshold: number;
}
export interface TrainingView { camera: Float32Array; width: number; height: number; }   // the 68-float CameraUniforms block of the view
export interface TrainingImage { texture: HipBuffer; width: number; height: number; }    // rgba8, row-major (LoadedImage.texture)
export interface PointCloudSwapRequest { pointCloud: PointCloud; optimizerInitialState?: OptimizerInitialState; }

export interface TrainerOptions {
  random?: () => number; useCommandBuffers?: boolean; maxTileEntries?: number; reusePasses?: boolean; deferredSH?: boolean;
  /** the single-view step runs K17 + Adam + re-pack as one kernel (default true); keepGradients also fills backwardPass.getGradientsBuffer() */
  fuseGeometryAdam?: boolean; keepGradients?: boolean;
  /** 1: every step awaits its own completion (trainer.ts:639-645); 2..4: a step awaits the one pipelineDepth - 1 submissions ago */
  pipelineDepth?: number;
  /** views per rank per global step (a batched step: BASELINE config c4), device lanes they are dealt to (default 3), view-batched K1 / K17 (default on) */
  viewsPerStep?: number; lanes?: number; batchViews?: boolean;
  /** device lanes the metric views of a densify event are dealt to (default 3; integer counts: any order gives the same bits) */
  metricLanes?: number;
  /** view-sharded data parallelism (parallel.js) */
  worldSize?: number; rank?: number; exchange?: Exchange;
}
export class Trainer {
  constructor(device: HipDevice, trainingConfig?: TrainingConfig, options?: TrainerOptions);
  /** Deferred SH writes: brings pointCloud.sh_buffer up to date for a device-side reader of the raw rows (host reads and forward passes built on the cloud follow by themselves). */
  flushPointCloud(): void;
  /** step() on a given global batch of views (worldSize * viewsPerStep indices); no reference counterpart. */
  stepViews(viewIds?: number[]): Promise<void>;
  /** maxTileEntries for a new forward pass: the caller's, or what an overflow has grown the library-sized lists to (0 = the library's sizing). */
  tileEntries(): number;
  /** Doubles the library-sized tile-entry lists after a WDGS_E_CAPACITY report and rebuilds the passes; false when the caller pinned maxTileEntries. */
  growTileEntryCapacity(error: Error): boolean;
  /** Entries needed by this trainer's passes among those a capacity report names; [] if it names only other owners' passes; null if it names none. */
  ownOverflow(error: Error): number[] | null;
  /** true (after one console.warn) for a capacity report about passes this trainer does not own (a Viewer on the same device). */
  notOurs(error: Error): boolean;
  /** device.queue.wait / device.synchronize with other owners' capacity reports filtered out. */
  wait(ticket: number): void;
  synchronize(): void;
  /** Records every view's command buffers up front; the number of steps taken depends on the dataset size only. */
  warmupCommandBuffers(): Promise<number>;
  /** Awaits every step still in flight (pipelineDepth > 1). */
  drain(): void;
  /** Data parallelism: every owner broadcasts its slice of the optimizer state (before a densify rebuild, an export). */
  syncOptimizerState(): void;
  pipelineDepth: number; keepGradients: boolean; fuseGeometryAdam: boolean; useCommandBuffers: boolean;
  /** Long tile lists (csrc/longlist.h): null = the library's defaults; set before the first step to give the passes this trainer builds other sizes. */
  longLists: { threshold?: number; maxItems?: number; maxRows?: number; maxItemsCap?: number; maxRowsCap?: number } | null;
  /** Enlarges the long-list scratch of every pass when the last frame wanted more than there is room for (called at densify events). */
  growLongLists(): void;
  readonly lanes: number; readonly viewsPerRank: number; readonly worldSize: number; readonly rank: number;
  random: () => number;
  setPointCloud(pointCloud: PointCloud): void;
  requestPointCloudSwap(pointCloud: PointCloud, optimizerInitialState?: OptimizerInitialState): void;
  consumePointCloudSwapRequest(): PointCloudSwapRequest | null;
  requestResizeTo(numPoints: number): void;
  applyPointCloudSwap(request: PointCloudSwapRequest): void;
  setDataset(cameras: (TrainingView | CameraData)[], images: (TrainingImage | LoadedImage)[]): void;
  getTrainingConfig(): TrainingConfig; setTrainingConfig(next: Partial<TrainingConfig>): void;
  getOptimizerHyperparameters(): AdamHyperparameters; setOptimizerHyperparameters(next: Partial<AdamHyperparameters>): void;
  setDensifyPruneConfig(next: Partial<DensifyPruneTrainingConfig>): void;
  start(): void; stop(): void; getIsTraining(): boolean;
  setMaxIterations(n: number): void; getMaxIterations(): number;
  getIteration(): number; getPointCount(): number; getLastStepMs(): number; getItersPerSec(): number;
  getLastDensifyPruneIteration(): number | null; getNextDensifyPruneIteration(): number | null;
  step(): Promise<void>;
  destroy(): void;
}
export function cameraBlockFor(block: Float32Array, width: number, height: number): Float32Array;
export function mat4Inverse(m: ArrayLike<number>): Float32Array;
export function projectionMatrix(znear: number, zfar: number, fovX: number, fovY: number): Float32Array;
export const DEFAULT_DENSIFY: DensifyPruneTrainingConfig;
