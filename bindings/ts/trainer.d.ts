// Typings of trainer.js: the public surface of the reference's Trainer (src/trainer.ts:177-566).
import { AdamHyperparameters, HipBuffer, HipDevice, OptimizerInitialState, PointCloud, TrainingConfig } from './webdgs_hip';

export interface DensifyPruneTrainingConfig {   // trainer.ts:23-40
  schedule: { enabled: boolean; warmupIterations: number; interval: number; stopIterations: number };
  metricViews: number; metricDownscale: number; metricThreshold: number; maxBufferBytes: number; maxNewPointsPerStep: number;
  pruneOpacity: number; cloneThresholdCount: number; splitScaleThreshold: number;
}
export interface TrainingView { camera: Float32Array; width: number; height: number; }   // the 68-float CameraUniforms block of the view
export interface TrainingImage { texture: HipBuffer; width: number; height: number; }    // rgba8, row-major (LoadedImage.texture)
export interface PointCloudSwapRequest { pointCloud: PointCloud; optimizerInitialState?: OptimizerInitialState; }

export class Trainer {
  constructor(device: HipDevice, trainingConfig?: TrainingConfig, options?: { random?: () => number; useCommandBuffers?: boolean; maxTileEntries?: number; reusePasses?: boolean; deferredSH?: boolean });
  /** Deferred SH writes: brings pointCloud.sh_buffer up to date before a host read, an export, or a foreign forward pass (no reference counterpart). */
  flushPointCloud(): void;
  random: () => number;
  setPointCloud(pointCloud: PointCloud): void;
  requestPointCloudSwap(pointCloud: PointCloud, optimizerInitialState?: OptimizerInitialState): void;
  consumePointCloudSwapRequest(): PointCloudSwapRequest | null;
  requestResizeTo(numPoints: number): void;
  applyPointCloudSwap(request: PointCloudSwapRequest): void;
  setDataset(cameras: TrainingView[], images: TrainingImage[]): void;
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
export const DEFAULT_DENSIFY: DensifyPruneTrainingConfig;
