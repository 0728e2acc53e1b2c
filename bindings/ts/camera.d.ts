// Typings of camera.js: src/camera/camera.ts without the browser (`canvas` is any object with width and height).
import { HipBuffer, HipDevice } from './webdgs_hip';
import { CameraData } from './loaders';
export interface CanvasLike { width: number; height: number; clientWidth?: number; clientHeight?: number; }
export function create_camera_uniform_buffer(device: HipDevice): HipBuffer;
export function load_camera_presets(file: Buffer | ArrayBuffer | string): { position: Float32Array; rotation: Float32Array }[];
export function uniformBlock(rotation: Float32Array, position: ArrayLike<number>, fovX: number, fovY: number, width: number, height: number, focal: number): Float32Array;
export class Camera {
  constructor(canvas: CanvasLike, device: HipDevice);
  readonly canvas: CanvasLike; readonly uniform_buffer: HipBuffer;
  position: Float32Array; rotation: Float32Array; look: Float32Array; up: Float32Array; right: Float32Array; uniforms: Float32Array;
  reset(): void;
  on_update_canvas(): void;
  update_buffer(): void;
  set_preset(preset: CameraData): void;
  destroy(): void;
}
