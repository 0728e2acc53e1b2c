'use strict';
/*
 * loaders.js (+ loaders.d.ts) -- the on-disk formats either side of the hot path for the TypeScript-side host (SURVEY.md section 8(f)
 * ranks 1-2): PLY / COLMAP point clouds, COLMAP and JSON cameras, the 272-byte camera block, PLY export.  Counterpart of
 * /root/reference/src/utils/plyreader.ts, load-pointcloud.ts, load-camera.ts and src/camera/camera.ts (citations relative to src/),
 * and of webdgs_amd/loaders.py, with which it agrees bit for bit (tests/test_gpu_js_host.py, tests/test_js_host_cpu.py).
 *
 * Inputs are node Buffers / ArrayBuffers instead of browser Files; `device` is a HipDevice (webdgs_hip.js) and may be omitted, in which
 * case the functions return host data only ({ gaussians: Uint32Array, sh: Uint32Array }).  Numbers go through the same types as in the
 * browser: JS numbers are binary64; a Float16Array store rounds binary64 -> binary16 once, to nearest even (@petamoriken/float16 3.8.7 --
 * restated here as f16Bits, the package is not vendored in the reference); wgpu-matrix (3.2.0) matrices are Float32Array.  Reference quirks
 * kept: `uchar` properties are divided by 255 on read AND again as colours (SURVEY Q22); only `float` and `uchar` properties advance the
 * read offset (plyreader.ts:63-72); fx, cx, cy are ignored by the camera (Q18).
 */
const C0 = 0.28209479177387814;

const f64 = new Float64Array(1), f64hi = new Uint32Array(f64.buffer);
/** binary64 -> binary16 bit pattern, one rounding, ties to even; overflow to infinity, NaN to a quiet NaN. */
function f16Bits(x) {
  if (x !== x) return 0x7e00;
  f64[0] = x;
  const sign = (f64hi[1] >>> 16) & 0x8000;
  const a = Math.abs(x);
  if (a >= 65520) return sign | 0x7c00;           // 65520 = the tie between 65504 and 2^16: rounds to the even neighbour, infinity
  let q, bias;
  if (a < 6.103515625e-5) { q = a * 16777216; bias = 0; }   // below 2^-14: a multiple-of-2^-24 grid, no implicit bit
  else {
    const e = ((f64hi[1] >>> 20) & 0x7ff) - 1023;
    q = (a * Math.pow(2, -e) - 1) * 1024; bias = (e + 15) << 10;
  }
  let r = Math.floor(q);
  const d = q - r;
  if (d > 0.5 || (d === 0.5 && (r & 1))) r += 1;
  return sign | (bias + r);                       // a carry out of the mantissa bumps the exponent field (up to infinity) by itself
}
/** binary16 bit pattern -> number (exact). */
function f16ToNumber(h) {
  const s = h & 0x8000 ? -1 : 1, e = (h >> 10) & 31, m = h & 1023;
  if (e === 0) return s * m * 5.960464477539063e-8;
  if (e === 31) return m ? NaN : s * Infinity;
  return s * (1 + m / 1024) * Math.pow(2, e - 15);
}

function asBuffer(data) {
  if (Buffer.isBuffer(data)) return data;
  if (data instanceof ArrayBuffer) return Buffer.from(data);
  return Buffer.from(data.buffer, data.byteOffset, data.byteLength);
}

/** decodeHeader (utils/plyreader.ts:1-54) -> [vertexCount, propertyTypes (name -> type, declaration order), vertexData: DataView].  The header is the
 *  text up to the word "end_header"; the payload starts one byte after that word (the reference's own offset rule). */
function decodeHeader(plyArrayBuffer) {
  const buf = asBuffer(plyArrayBuffer);
  const h = parseHeader(buf);
  return [h.vertexCount, h.propertyTypes, new DataView(buf.buffer, buf.byteOffset + h.vertexByteOffset, buf.length - h.vertexByteOffset)];
}
/** readRawVertex (utils/plyreader.ts:56-74) -> [offset of the next vertex, { property: value }]: float as is, uchar / 255; a property of any
 *  other type is skipped WITHOUT advancing the offset. */
function readRawVertex(offset, vertexData, propertyTypes) {
  const raw = {};
  for (const name of Object.keys(propertyTypes)) {
    if (propertyTypes[name] === 'float') { raw[name] = vertexData.getFloat32(offset, true); offset += 4; }
    else if (propertyTypes[name] === 'uchar') { raw[name] = vertexData.getUint8(offset) / 255.0; offset += 1; }
  }
  return [offset, raw];
}
function parseHeader(buf) {
  const end = buf.indexOf('end_header');
  if (end < 0) throw new Error("PLY header: 'end_header' not found");
  let vertexCount = 0;
  const propertyTypes = {};
  for (const raw of buf.toString('utf8', 0, end).split('\n')) {
    const words = raw.trim().split(/\s+/);
    if (words[0] === 'element' && words[1] === 'vertex') { const n = /\d+/.exec(raw); if (n) vertexCount = parseInt(n[0], 10); }
    else if (words[0] === 'property' && words.length >= 3) propertyTypes[words[2]] = words[1];   // "property <type> <name>"
  }
  return { vertexCount, propertyTypes, vertexByteOffset: end + 'end_header'.length + 1 };
}

/** nShCoeffs (plyreader.ts:76-89). */
function nShCoeffs(deg) {
  if (deg === 0 || deg === 1 || deg === 2 || deg === 3) return (deg + 1) * (deg + 1);
  throw new Error(`Unsupported SH degree: ${deg}`);
}

function hostCloud(type, n, shDeg) {
  return { type, num_points: n, sh_deg: shDeg, gaussianHalves: new Uint16Array(12 * n), shHalves: new Uint16Array(48 * n) };
}
function normalDefaults(g, o, x, y, z) {   // load-pointcloud.ts:111-123 / 255-264: raw opacity 1, quaternion (1,0,0,0), log-sigma -5
  g[o] = f16Bits(x); g[o + 1] = f16Bits(y); g[o + 2] = f16Bits(z);
  g[o + 3] = 0x3c00; g[o + 4] = 0x3c00; g[o + 8] = 0xc500; g[o + 9] = 0xc500; g[o + 10] = 0xc500;
}
/** Host cloud -> PointCloud: with a device the buffers are HipBuffers (load-pointcloud.ts:16-23), without one the words stay on the host. */
function finishCloud(c, device) {
  const gaussians = new Uint32Array(c.gaussianHalves.buffer), sh = new Uint32Array(c.shHalves.buffer);
  const out = { type: c.type, num_points: c.num_points, sh_deg: c.sh_deg, gaussians, sh };
  if (device) {
    out.gaussian_3d_buffer = device.createBuffer({ size: Math.max(4, gaussians.byteLength), label: 'input 3d gaussians data buffer' });
    out.sh_buffer = device.createBuffer({ size: Math.max(4, sh.byteLength), label: 'sh/color data buffer' });
    if (c.num_points) { device.queue.writeBuffer(out.gaussian_3d_buffer, 0, gaussians); device.queue.writeBuffer(out.sh_buffer, 0, sh); }
  }
  return out;
}

/** loadPly (utils/load-pointcloud.ts:156-307). */
function loadPly(data, device) {
  const buf = asBuffer(data);
  const { vertexCount: n, propertyTypes, vertexByteOffset } = parseHeader(buf);
  // the rule of readRawVertex, with the per-property offsets resolved once instead of per vertex
  const fields = Object.keys(propertyTypes).filter((k) => propertyTypes[k] === 'float' || propertyTypes[k] === 'uchar');
  const isFloat = fields.map((k) => propertyTypes[k] === 'float');
  const stride = isFloat.reduce((s, f) => s + (f ? 4 : 1), 0);
  if (buf.length < vertexByteOffset + n * stride) throw new Error(`PLY payload too short: need ${n * stride} bytes after the header, have ${buf.length - vertexByteOffset}`);
  const where = {};
  let at = 0;
  fields.forEach((k, i) => { where[k] = at; at += isFloat[i] ? 4 : 1; });
  const has = (k) => Object.prototype.hasOwnProperty.call(propertyTypes, k);
  const reader = (k) => {
    if (!(k in where)) return () => undefined;
    const o = where[k];
    return propertyTypes[k] === 'float' ? (base) => buf.readFloatLE(base + o) : (base) => buf[base + o] / 255.0;
  };
  const isFull = has('rot_0') && has('scale_0');
  if (isFull) {
    const nRest = Object.keys(propertyTypes).filter((k) => k.startsWith('f_rest_')).length;
    const perColor = nRest / 3;
    const shDeg = Math.sqrt(perColor + 1) - 1;
    const numCoefs = nShCoeffs(shDeg);
    const order = ['f_dc_0', 'f_dc_1', 'f_dc_2'];
    for (let i = 0; i < perColor; i++) for (let rgb = 0; rgb < 3; rgb++) order.push(`f_rest_${rgb * perColor + i}`);
    const c = hostCloud('full', n, shDeg);
    const gRead = ['x', 'y', 'z', 'opacity', 'rot_0', 'rot_1', 'rot_2', 'rot_3', 'scale_0', 'scale_1', 'scale_2'].map(reader);
    const sRead = order.slice(0, numCoefs * 3).map(reader);
    for (let i = 0, base = vertexByteOffset; i < n; i++, base += stride) {
      const o = i * 12, so = i * 48;
      for (let k = 0; k < 11; k++) c.gaussianHalves[o + k] = f16Bits(gRead[k](base));
      for (let k = 0; k < sRead.length; k++) c.shHalves[so + k] = f16Bits(sRead[k](base));
    }
    return finishCloud(c, device);
  }
  const c = hostCloud('normal', n, 0);
  const xyz = ['x', 'y', 'z'].map(reader);
  const names = has('red') ? ['red', 'green', 'blue'] : (has('diffuse_red') ? ['diffuse_red', 'diffuse_green', 'diffuse_blue'] : null);
  const rgb = names ? names.map(reader) : null;
  for (let i = 0, base = vertexByteOffset; i < n; i++, base += stride) {
    normalDefaults(c.gaussianHalves, i * 12, xyz[0](base), xyz[1](base), xyz[2](base));
    for (let k = 0; k < 3; k++) {
      const v = rgb ? rgb[k](base) / 255.0 : 0.5;   // the second division by 255: SURVEY Q22
      c.shHalves[i * 48 + k] = f16Bits((v - 0.5) / C0);
    }
  }
  return finishCloud(c, device);
}

/** COLMAP points3D.bin (utils/load-pointcloud.ts:54-154): id u64, xyz f64 x3, rgb u8 x3, error f64, track length u64 + 8 bytes per element. */
function loadColmapBin(data, device) {
  const buf = asBuffer(data);
  const n = Number(buf.readBigUInt64LE(0));
  const c = hostCloud('normal', n, 0);
  let off = 8;
  for (let i = 0; i < n; i++) {
    off += 8;
    normalDefaults(c.gaussianHalves, i * 12, buf.readDoubleLE(off), buf.readDoubleLE(off + 8), buf.readDoubleLE(off + 16));
    off += 24;
    for (let k = 0; k < 3; k++) c.shHalves[i * 48 + k] = f16Bits((buf[off + k] / 255.0 - 0.5) / C0);
    off += 3 + 8;
    off += 8 + Number(buf.readBigUInt64LE(off)) * 8;
  }
  return finishCloud(c, device);
}

/** loadPointCloud (utils/load-pointcloud.ts:29-52): 'ply' magic, else COLMAP points3D.bin. */
function loadPointCloud(data, device) {
  const buf = asBuffer(data);
  if (buf.length >= 3 && buf[0] === 0x70 && buf[1] === 0x6c && buf[2] === 0x79) return loadPly(buf, device);
  try { return loadColmapBin(buf, device); } catch (e) { throw new Error(`Failed to load pointcloud: ${e.message}`); }
}

/** Binary little-endian 3DGS PLY of a 'full' cloud (the reference has no exporter); loadPly(exportPly(...)) returns the same fp16 words.
 *  gaussians: Uint32Array(6 N), sh: Uint32Array(24 N). */
function exportPly(gaussians, sh, shDeg) {
  const g = new Uint16Array(gaussians.buffer, gaussians.byteOffset, gaussians.length * 2), s = new Uint16Array(sh.buffer, sh.byteOffset, sh.length * 2);
  const n = g.length / 12, k = (shDeg + 1) * (shDeg + 1), perColor = k - 1;
  const names = ['x', 'y', 'z', 'nx', 'ny', 'nz', 'f_dc_0', 'f_dc_1', 'f_dc_2'];
  for (let i = 0; i < 3 * perColor; i++) names.push(`f_rest_${i}`);
  names.push('opacity', 'scale_0', 'scale_1', 'scale_2', 'rot_0', 'rot_1', 'rot_2', 'rot_3');
  const idx = {}; names.forEach((nm, i) => { idx[nm] = i; });
  const header = Buffer.from(`ply\nformat binary_little_endian 1.0\nelement vertex ${n}\n` + names.map((nm) => `property float ${nm}\n`).join('') + 'end_header\n', 'ascii');
  const body = new Float32Array(n * names.length);
  const src = [['x', 0], ['y', 1], ['z', 2], ['opacity', 3], ['rot_0', 4], ['rot_1', 5], ['rot_2', 6], ['rot_3', 7], ['scale_0', 8], ['scale_1', 9], ['scale_2', 10]];
  for (let i = 0; i < n; i++) {
    const row = i * names.length;
    for (const [nm, at] of src) body[row + idx[nm]] = f16ToNumber(g[i * 12 + at]);
    for (let c = 0; c < 3; c++) {
      body[row + idx[`f_dc_${c}`]] = f16ToNumber(s[i * 48 + c]);
      for (let j = 0; j < perColor; j++) body[row + idx[`f_rest_${c * perColor + j}`]] = f16ToNumber(s[i * 48 + (j + 1) * 3 + c]);
    }
  }
  return Buffer.concat([header, Buffer.from(body.buffer)]);
}

// ----------------------------------------------------------------------------- cameras
/** loadCameraJson (utils/load-camera.ts:138-168): `rotation` (rows of numbers) stored as a column-major mat4 of the same matrix. */
function loadCameraJson(data) {
  const j = JSON.parse(asBuffer(data).toString('utf8'));
  return (Array.isArray(j) ? j : [j]).map((c) => {
    const r = c.rotation, rot = new Float32Array(16);
    for (let col = 0; col < 3; col++) for (let row = 0; row < 3; row++) rot[col * 4 + row] = r[row][col];
    rot[15] = 1;
    return { id: c.id, img_name: c.img_name, width: c.width, height: c.height, fx: c.fx, fy: c.fy, position: Float32Array.from(c.position.slice(0, 3)), rotation: rot };
  });
}

/** wgpu-matrix 3.2.0 mat4.fromQuat on Float32Array operands: column-major, products in binary64, stored as f32. */
function fromQuat(qx, qy, qz, qw) {
  const x = Math.fround(qx), y = Math.fround(qy), z = Math.fround(qz), w = Math.fround(qw);
  const x2 = x + x, y2 = y + y, z2 = z + z;
  const xx = x * x2, yx = y * x2, yy = y * y2, zx = z * x2, zy = z * y2, zz = z * z2, wx = w * x2, wy = w * y2, wz = w * z2;
  return Float32Array.from([1 - yy - zz, yx + wz, zx - wy, 0, yx - wz, 1 - xx - zz, zy + wx, 0, zx + wy, zy - wx, 1 - xx - yy, 0, 0, 0, 0, 1]);
}

/** COLMAP images.bin (utils/load-camera.ts:171-240): world position C = -(R^T t), rotation kept as R (world -> camera). */
function loadColmapImagesBin(data) {
  const buf = asBuffer(data);
  if (buf.length < 8) return [];
  const n = Number(buf.readBigUInt64LE(0));
  const out = [];
  let off = 8;
  for (let i = 0; i < n; i++) {
    const imageId = buf.readUInt32LE(off); off += 4;
    const q = [0, 1, 2, 3, 4, 5, 6].map((k) => buf.readDoubleLE(off + 8 * k)); off += 56;   // qw qx qy qz tx ty tz
    const cameraId = buf.readUInt32LE(off); off += 4;
    const end = buf.indexOf(0, off);
    const name = buf.toString('latin1', off, end); off = end + 1;
    off += 8 + Number(buf.readBigUInt64LE(off)) * 24;
    const R = fromQuat(q[1], q[2], q[3], q[0]);
    const t = [Math.fround(q[4]), Math.fround(q[5]), Math.fround(q[6])];
    // vec3.transformMat4(T, transpose(R)): element (row r, column c) of transpose(R) is R[r * 4 + c] in R's column-major storage
    const p = new Float32Array(3);
    for (let r = 0; r < 3; r++) p[r] = -Math.fround(R[r * 4] * t[0] + R[r * 4 + 1] * t[1] + R[r * 4 + 2] * t[2] + R[r * 4 + 3]);
    out.push({ id: imageId, camera_id: cameraId, img_name: name, rotation: R, position: p });
  }
  return out;
}

/** COLMAP cameras.bin (utils/load-camera.ts:243-288): models 0 (SIMPLE_PINHOLE) and 1 (PINHOLE). */
function loadColmapCamerasBin(data) {
  const buf = asBuffer(data);
  const n = Number(buf.readBigUInt64LE(0));
  const out = [];
  let off = 8;
  for (let i = 0; i < n; i++) {
    const cameraId = buf.readUInt32LE(off), model = buf.readInt32LE(off + 4); off += 8;
    const width = Number(buf.readBigUInt64LE(off)), height = Number(buf.readBigUInt64LE(off + 8)); off += 16;
    const params = model === 0 ? 3 : (model === 1 ? 4 : -1);
    if (params < 0) throw new Error(`Unsupported COLMAP camera model ID: ${model}`);
    const v = []; for (let k = 0; k < params; k++) v.push(buf.readDoubleLE(off + 8 * k));
    off += 8 * params;
    const [fx, fy, cx, cy] = model === 0 ? [v[0], v[0], v[1], v[2]] : v;
    out.push({ id: cameraId, camera_id: cameraId, width, height, fx, fy, cx, cy });
  }
  return out;
}

/** The images.bin + cameras.bin merge of loadCamera (utils/load-camera.ts:45-72): intrinsics by camera_id, id from the image. */
function mergeColmap(images, cameras) {
  const byId = new Map(cameras.map((c) => [c.id, c]));
  return images.map((img) => (img.camera_id !== undefined && byId.has(img.camera_id) ? Object.assign({}, img, byId.get(img.camera_id), { id: img.id }) : Object.assign({}, img)));
}

/** loadCamera (utils/load-camera.ts:25-136) for named buffers: files = [{ name, data }].  A .json wins; images.bin + cameras.bin are merged; one
 *  of the two alone loads with the other half missing; anything else must look like JSON. */
function loadCamera(files) {
  const list = Array.isArray(files) ? files : [files];
  const named = (suffix) => list.find((f) => f.name && f.name.toLowerCase().endsWith(suffix));
  const json = named('.json'), images = named('images.bin'), cameras = named('cameras.bin');
  if (json) return loadCameraJson(json.data);
  if (images && cameras) return mergeColmap(loadColmapImagesBin(images.data), loadColmapCamerasBin(cameras.data));
  if (images) return loadColmapImagesBin(images.data);
  if (cameras) return loadColmapCamerasBin(cameras.data);
  if (!list.length) return [];
  const first = list[0], data = asBuffer(first.data || first);
  const head = data.toString('latin1', 0, 10).trim();
  if (head[0] === '{' || head[0] === '[') return loadCameraJson(data);
  throw new Error(`Unsupported camera file format: ${first.name || ''}`);
}

const { uniformBlock } = require('./camera.js');

/** Camera.set_preset + on_update_canvas + update_buffer (camera/camera.ts:23-56, 138-205): the 68-float block of a CameraData on a canvas of
 *  width x height (the trainer sets the canvas to the image size, trainer.ts:583-584).  Matrices are formed in binary64 from Float32Array
 *  operands and stored as f32, as wgpu-matrix does; the inverses are wgpu-matrix's cofactor formula on the stored matrices. */
function cameraUniforms(cam, width, height) {
  const w = Math.floor(width !== undefined && width !== null ? width : cam.width), h = Math.floor(height !== undefined && height !== null ? height : cam.height);
  let fovY = 45 / 180 * Math.PI;                                            // Camera defaults (camera.ts:113-136)
  if (cam.fx && cam.fy && cam.height) fovY = 2 * Math.atan(cam.height / (2 * cam.fy));
  const focal = 0.5 * h / Math.tan(fovY * 0.5);
  const fovX = 2 * Math.atan(w / (2 * focal));
  const rot = cam.rotation ? Float32Array.from(cam.rotation) : Float32Array.from([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1]);
  const pos = cam.position ? Float32Array.from(cam.position) : Float32Array.from([0, 0, 5]);
  return uniformBlock(rot, pos, fovX, fovY, w, h, focal);
}

module.exports = { C0, f16Bits, f16ToNumber, decodeHeader, readRawVertex, nShCoeffs, loadPly, loadColmapBin, loadPointCloud, exportPly, loadCameraJson, loadColmapImagesBin,
  loadColmapCamerasBin, mergeColmap, loadCamera, cameraUniforms, fromQuat };
