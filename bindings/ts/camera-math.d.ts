// Typings of camera-math.js.
export function mat4Inverse(m: ArrayLike<number>): Float32Array;
export function projectionMatrix(znear: number, zfar: number, fovX: number, fovY: number): Float32Array;
export function cameraBlockFor(block: Float32Array, width: number, height: number): Float32Array;
