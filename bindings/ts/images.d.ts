// Typings of images.js: src/utils/load-images.ts.
import { HipBuffer, HipDevice } from './webdgs_hip';
export interface Bitmap { width: number; height: number; data: Uint8Array; }   // rgba8, rows top to bottom
export interface LoadedImage {              // utils/load-images.ts:1-8
  name: string; file: string | { name: string; data: Buffer }; bitmap: Bitmap; width: number; height: number; texture: HipBuffer | null;
}
export function decodePNG(bytes: Buffer | Uint8Array): Bitmap;
export function decodeJPEG(bytes: Buffer | Uint8Array): Bitmap;
export function encodePNG(rgba: Uint8Array, width: number, height: number): Buffer;
export function decodeImage(bytes: Buffer | Uint8Array, name?: string): Bitmap;
export function compareNames(a: string, b: string): number;
export function createTextureFromImage(device: HipDevice, image: Bitmap): HipBuffer;
export function loadImages(files: (string | { name: string; data: Buffer })[], device: HipDevice | null): LoadedImage[];
