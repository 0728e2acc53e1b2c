// Typings of viewer.js: src/viewer.ts without the browser.
import { HipBuffer, HipDevice, HipEncoder, PointCloud, RenderMode, TiledForwardPass } from './webdgs_hip';
import { Camera, CanvasLike } from './camera';
export type FrameTarget = HipBuffer & { width: number; height: number };
export class Viewer {
  constructor(device: HipDevice, context: { getCurrentTexture(): FrameTarget } | null, canvas: CanvasLike, format: string);
  readonly camera: Camera; readonly cameraControl: { update(dt: number): void };
  setPointCloud(pointCloud: PointCloud): void;
  update(dt: number): void;
  render(commandEncoder: HipEncoder | null): void;
  setRenderMode(mode: RenderMode): void;
  setGaussianScale(value: number): void;
  setPointSize(value: number): void;
  getForwardPass(): TiledForwardPass | null;
  currentTexture(): FrameTarget;
  resize(width: number, height: number): void;
  readFrame(): Uint8Array;
  savePNG(file: string): void;
  destroy(): void;
}
