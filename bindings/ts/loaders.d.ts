// Typings of loaders.js: src/utils/plyreader.ts, load-pointcloud.ts, load-camera.ts and the camera block of src/camera/camera.ts.
import { HipBuffer, HipDevice } from './webdgs_hip';
export type Bytes = Buffer | ArrayBuffer | ArrayBufferView;
export interface CameraData {              // utils/load-camera.ts:4-19
  id: number; position?: Float32Array; rotation?: Float32Array; width?: number; height?: number; fx?: number; fy?: number; cx?: number; cy?: number;
  img_name?: string; camera_id?: number;
}
export interface LoadedPointCloud {        // PointCloud (utils/load-pointcloud.ts:16-23) + the host copy of its words
  type: 'full' | 'normal'; num_points: number; sh_deg: number; gaussians: Uint32Array; sh: Uint32Array;
  gaussian_3d_buffer?: HipBuffer; sh_buffer?: HipBuffer;
}
export const C0: number;
export function f16Bits(x: number): number;
export function f16ToNumber(bits: number): number;
export function decodeHeader(plyArrayBuffer: Bytes): [number, Record<string, string>, DataView];
export function readRawVertex(offset: number, vertexData: DataView, propertyTypes: Record<string, string>): [number, Record<string, number>];
export function nShCoeffs(sphericalHarmonicsDegree: number): number;
export function loadPly(data: Bytes, device?: HipDevice): LoadedPointCloud;
export function loadColmapBin(data: Bytes, device?: HipDevice): LoadedPointCloud;
export function loadPointCloud(file: Bytes, device: HipDevice | null | undefined): LoadedPointCloud;
export function exportPly(gaussians: Uint32Array, sh: Uint32Array, shDeg: number): Buffer;
export function loadCameraJson(data: Bytes): CameraData[];
export function loadColmapImagesBin(data: Bytes): CameraData[];
export function loadColmapCamerasBin(data: Bytes): CameraData[];
export function mergeColmap(images: CameraData[], cameras: CameraData[]): CameraData[];
export function loadCamera(fileOrFiles: { name?: string; data: Bytes } | { name?: string; data: Bytes }[]): CameraData[];
export function cameraUniforms(cam: Partial<CameraData>, width?: number, height?: number): Float32Array;
export function fromQuat(qx: number, qy: number, qz: number, qw: number): Float32Array;
