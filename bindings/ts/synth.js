'use strict';
/*
 * synth.js (+ synth.d.ts) -- the deterministic synthetic scenes of SURVEY.md section 8(d) for the TypeScript-side host ("PRNG: splitmix64 ...
 * identical in C++ and TS; the per-Gaussian draw order is part of the fixture contract").  Counterpart of webdgs_amd/synth.py: the same seed
 * (0x5EEDD650000 + config id), the same draw order (z, x, y, quaternion x4 by Box-Muller, opacity, log-sigma x3, SH DC x3, then the bands in
 * [k][rgb] order), the same roundings (binary64 arithmetic -> binary32 -> binary16, to nearest even) -- so bench.js trains the scene
 * bench.py trains.  64-bit integers are carried as two u32 halves (no BigInt in the inner loop).
 */
const { f16Bits, f16ToNumber } = require('./loaders.js');
const { mat4Inverse } = require('./camera-math.js');

const CONFIGS = {   // BASELINE.json configs (SURVEY 8(d)): id, N, W, H, SH degree, fy, s0
  c1: { config_id: 1, num_points: 10000, width: 256, height: 256, sh_deg: 0, fy: 300, s0: 0.005, name: 'c1' },
  c2: { config_id: 2, num_points: 100000, width: 640, height: 480, sh_deg: 1, fy: 550, s0: 0.003, name: 'c2' },
  c3: { config_id: 3, num_points: 1000000, width: 1920, height: 1080, sh_deg: 3, fy: 1200, s0: 0.003, name: 'c3-perf' },
  'c3-small': { config_id: 3, num_points: 1000000, width: 1920, height: 1080, sh_deg: 3, fy: 1200, s0: 0.0005, name: 'c3-small' },
  c5: { config_id: 5, num_points: 5000000, width: 3840, height: 2160, sh_deg: 3, fy: 2400, s0: 0.002, name: 'c5' },
};

/** splitmix64 as a stream of uniform doubles in [0, 1): next() = (z >>> 11) * 2^-53.  64-bit products by 16-bit limbs (every partial sum stays
 *  below 2^53); the multiplier's limbs are constants, the product's halves land in this.ph / this.pl. */
class SplitMix64 {
  constructor(seedHi, seedLo) { this.hi = seedHi >>> 0; this.lo = seedLo >>> 0; this.ph = 0; this.pl = 0; }
  mul(aH, aL, b48, b32, b16, b00) {   // low 64 bits of (aH:aL) * b
    const a48 = aH >>> 16, a32 = aH & 0xffff, a16 = aL >>> 16, a00 = aL & 0xffff;
    let c00 = a00 * b00, c16 = c00 >>> 16; c00 &= 0xffff;
    c16 += a16 * b00; let c32 = Math.floor(c16 / 65536); c16 %= 65536;
    c16 += a00 * b16; c32 += Math.floor(c16 / 65536); c16 %= 65536;
    c32 += a32 * b00; let c48 = Math.floor(c32 / 65536); c32 %= 65536;
    c32 += a16 * b16; c48 += Math.floor(c32 / 65536); c32 %= 65536;
    c32 += a00 * b32; c48 += Math.floor(c32 / 65536); c32 %= 65536;
    c48 = (c48 + a48 * b00 + a32 * b16 + a16 * b32 + a00 * b48) % 65536;
    this.ph = (c48 * 65536 + c32) >>> 0; this.pl = (c16 * 65536 + c00) >>> 0;
  }
  next() {
    const lo = this.lo + 0x7f4a7c15, carry = lo > 0xffffffff ? 1 : 0;   // state += 0x9E3779B97F4A7C15
    this.lo = lo >>> 0; this.hi = (this.hi + 0x9e3779b9 + carry) >>> 0;
    let h = this.hi, l = this.lo;
    l = (l ^ ((l >>> 30) | (h << 2))) >>> 0; h = (h ^ (h >>> 30)) >>> 0;                 // z ^= z >>> 30
    this.mul(h, l, 0xbf58, 0x476d, 0x1ce4, 0xe5b9); h = this.ph; l = this.pl;         // z *= 0xBF58476D1CE4E5B9
    l = (l ^ ((l >>> 27) | (h << 5))) >>> 0; h = (h ^ (h >>> 27)) >>> 0;                 // z ^= z >>> 27
    this.mul(h, l, 0x94d0, 0x49bb, 0x1331, 0x11eb); h = this.ph; l = this.pl;         // z *= 0x94D049BB133111EB
    l = (l ^ ((l >>> 31) | (h << 1))) >>> 0; h = (h ^ (h >>> 31)) >>> 0;                 // z ^= z >>> 31
    return (h * 2097152 + (l >>> 11)) / 9007199254740992;
  }
}

const boxMuller = (u1, u2) => Math.sqrt(-2 * Math.log1p(-u1)) * Math.cos(2 * Math.PI * u2);
const half = (v) => f16Bits(Math.fround(v));   // binary64 -> binary32 -> binary16, as the float32 staging array of synth.py does

/** { gaussians: Uint32Array(6 N), sh: Uint32Array(24 N) } for a config (optionally only its first numPoints Gaussians). */
function makeGaussians(cfg, numPoints) {
  const n = numPoints === undefined || numPoints === null ? cfg.num_points : numPoints;
  const k = (cfg.sh_deg + 1) * (cfg.sh_deg + 1), draws = 3 + 8 + 2 + 3 + 3 + 2 * 3 * (k - 1);
  const seed = 0x5eedd650000 + cfg.config_id;
  const rng = new SplitMix64(Math.floor(seed / 4294967296), seed % 4294967296);
  const g = new Uint16Array(12 * n), s = new Uint16Array(48 * n), u = new Float64Array(draws);
  const logS0 = Math.log(cfg.s0), log10 = Math.log(10);
  for (let i = 0; i < n; i++) {
    for (let d = 0; d < draws; d++) u[d] = rng.next();
    const z = 2 + 8 * u[0];
    const x = (2 * u[1] - 1) * (z * cfg.width / (2 * cfg.fy)), y = (2 * u[2] - 1) * (z * cfg.height / (2 * cfg.fy));
    const q = [boxMuller(u[3], u[4]), boxMuller(u[5], u[6]), boxMuller(u[7], u[8]), boxMuller(u[9], u[10])];
    const norm = Math.max(Math.sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]), 1e-12);
    const o = 12 * i;
    g[o] = half(x); g[o + 1] = half(y); g[o + 2] = half(z); g[o + 3] = half(boxMuller(u[11], u[12]));
    for (let j = 0; j < 4; j++) g[o + 4 + j] = half(q[j] / norm);
    for (let j = 0; j < 3; j++) g[o + 8 + j] = half(logS0 + u[13 + j] * log10);
    const so = 48 * i;
    for (let j = 0; j < 3; j++) s[so + j] = half(2 * u[16 + j] - 1);
    for (let j = 0; j < 3 * (k - 1); j++) s[so + 3 + j] = half(0.1 * boxMuller(u[19 + 2 * j], u[20 + 2 * j]));
  }
  return { gaussians: new Uint32Array(g.buffer), sh: new Uint32Array(s.buffer) };
}

/** The ground-truth variant of a scene: opacity_raw + 1 and DC + 0.2 in binary32, re-rounded to binary16 (SURVEY 8(d)). */
function makeTargetScene(gaussians, sh) {
  const g = new Uint16Array(gaussians.slice().buffer), s = new Uint16Array(sh.slice().buffer);
  const p2 = Math.fround(0.2);
  for (let i = 0; i < g.length; i += 12) g[i + 3] = half(f16ToNumber(g[i + 3]) + 1);
  for (let i = 0; i < s.length; i += 48) for (let j = 0; j < 3; j++) s[i + j] = half(f16ToNumber(s[i + j]) + p2);
  return { gaussians: new Uint32Array(g.buffer), sh: new Uint32Array(s.buffer) };
}

/** The 272-byte CameraUniforms block from a row-major world -> view matrix; fy only (fx is ignored: SURVEY Q18). */
function cameraBlock(viewRowMajor, width, height, fy) {
  const out = new Float32Array(68), znear = 0.01, zfar = 100;
  for (let c = 0; c < 4; c++) for (let r = 0; r < 4; r++) out[c * 4 + r] = viewRowMajor[r * 4 + c];
  const top = (height * 0.5) / fy * znear, right = (width * 0.5) / fy * znear;
  out[32] = 2 * znear / (2 * right); out[37] = -2 * znear / (2 * top); out[42] = zfar / (zfar - znear); out[43] = 1; out[46] = -(zfar * znear) / (zfar - znear);
  out.set(mat4Inverse(out.subarray(0, 16)), 16);
  out.set(mat4Inverse(out.subarray(32, 48)), 48);
  out[64] = width; out[65] = height; out[66] = fy; out[67] = fy;
  return out;
}
function identityCamera(cfg) { return cameraBlock([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1], cfg.width, cfg.height, cfg.fy); }

/** `count` cameras on a circle of `radius` around the origin looking at `target` (COLMAP axes: +x right, +y down, +z forward). */
function circleCameras(cfg, count, radius, target) {
  const R = radius === undefined ? 1 : radius, tgt = target || [0, 0, 6];
  const unit = (v) => { const n = Math.sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); return [v[0] / n, v[1] / n, v[2] / n]; };
  const cross = (a, b) => [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]];
  const out = [];
  for (let i = 0; i < count; i++) {
    const th = 2 * Math.PI * i / count, c = [R * Math.cos(th), R * Math.sin(th), 0];
    const f = unit([tgt[0] - c[0], tgt[1] - c[1], tgt[2] - c[2]]), xr = unit(cross([0, 1, 0], f)), yd = cross(f, xr);
    const rows = [xr, yd, f], view = new Float64Array(16);
    for (let r = 0; r < 3; r++) {
      view[r * 4] = rows[r][0]; view[r * 4 + 1] = rows[r][1]; view[r * 4 + 2] = rows[r][2];
      view[r * 4 + 3] = 0 + (-rows[r][0]) * c[0] + (-rows[r][1]) * c[1] + (-rows[r][2]) * c[2];   // (-rot) @ c accumulated from +0, as synth.py's matmul does (the sign of a zero included)
    }
    view[15] = 1;
    out.push(cameraBlock(view, cfg.width, cfg.height, cfg.fy));
  }
  return out;
}

module.exports = { CONFIGS, SplitMix64, makeGaussians, makeTargetScene, cameraBlock, identityCamera, circleCameras };
