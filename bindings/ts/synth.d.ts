// Typings of synth.js: the deterministic synthetic scenes of SURVEY.md 8(d) (the generator webdgs_amd/synth.py implements, bit for bit).
export interface SceneConfig { config_id: number; num_points: number; width: number; height: number; sh_deg: number; fy: number; s0: number; name: string; }
export const CONFIGS: { [name: string]: SceneConfig };
export class SplitMix64 { constructor(seedHi: number, seedLo: number); next(): number; }
export function makeGaussians(cfg: SceneConfig, numPoints?: number): { gaussians: Uint32Array; sh: Uint32Array };
export function makeTargetScene(gaussians: Uint32Array, sh: Uint32Array): { gaussians: Uint32Array; sh: Uint32Array };
export function cameraBlock(viewRowMajor: ArrayLike<number>, width: number, height: number, fy: number): Float32Array;
export function identityCamera(cfg: SceneConfig): Float32Array;
export function circleCameras(cfg: SceneConfig, count: number, radius?: number, target?: number[]): Float32Array[];
