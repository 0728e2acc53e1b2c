'use strict';
/*
 * images.js (+ images.d.ts) -- ground-truth image ingest for the TypeScript-side host, mirroring src/utils/load-images.ts:11-56 and
 * webdgs_amd/images.py.  The reference filters a file list to .jpg / .jpeg / .png, orders it with localeCompare(numeric, base sensitivity)
 * (load-images.ts:12-17), decodes with createImageBitmap and uploads each bitmap into an rgba8unorm texture (createTextureFromImage, 42-56);
 * a file that fails to decode is logged and dropped (31-34).  Outside a browser there is no createImageBitmap: PNGs (8 bits per sample or
 * fewer, non-interlaced: grey, grey + alpha, RGB, RGBA, palette) are decoded here with node's zlib; a JPEG needs a decoder the image does not
 * hold and is dropped with that message, as any undecodable file is.  The "texture" is a width * height * 4-byte device buffer, rows top to
 * bottom -- the layout every kernel of the hot path consumes.
 */
const fs = require('fs');
const path = require('path');
const zlib = require('zlib');
const { decodeJPEG } = require('./jpeg.js');

const PNG_MAGIC = Buffer.from([0x89, 0x50, 0x4e, 0x47, 0x0d, 0x0a, 0x1a, 0x0a]);

/** PNG bytes -> { width, height, data: Uint8Array(4 W H) }. */
function decodePNG(bytes) {
  const buf = Buffer.isBuffer(bytes) ? bytes : Buffer.from(bytes);
  if (buf.length < 8 || buf.compare(PNG_MAGIC, 0, 8, 0, 8) !== 0) throw new Error('not a PNG file');
  let pos = 8, width = 0, height = 0, depth = 0, ctype = 0, interlace = 0, palette = null, trns = null;
  const idat = [];
  while (pos + 8 <= buf.length) {
    const length = buf.readUInt32BE(pos), tag = buf.toString('latin1', pos + 4, pos + 8), body = buf.subarray(pos + 8, pos + 8 + length);
    pos += 12 + length;
    if (tag === 'IHDR') { width = body.readUInt32BE(0); height = body.readUInt32BE(4); depth = body[8]; ctype = body[9]; interlace = body[12]; }
    else if (tag === 'PLTE') palette = body;
    else if (tag === 'tRNS') trns = body;
    else if (tag === 'IDAT') idat.push(body);
    else if (tag === 'IEND') break;
  }
  const packed = (depth === 1 || depth === 2 || depth === 4) && (ctype === 0 || ctype === 3);   // sub-byte samples: grey and palette only
  if (!width || interlace !== 0 || !(depth === 8 || packed)) throw new Error('unsupported PNG (need non-interlaced, at most 8 bits per sample)');
  const channels = { 0: 1, 2: 3, 3: 1, 4: 2, 6: 4 }[ctype];
  if (!channels) throw new Error(`bad PNG colour type ${ctype}`);
  const stride = packed ? Math.ceil(width * depth / 8) : width * channels, bpp = packed ? 1 : channels;
  const raw = zlib.inflateSync(Buffer.concat(idat));
  if (raw.length < height * (stride + 1)) throw new Error('PNG data too short');
  const lines = new Uint8Array(height * stride);
  for (let y = 0; y < height; y++) {   // undo the per-line filters (PNG specification, section 9)
    const ft = raw[y * (stride + 1)], src = y * (stride + 1) + 1, dst = y * stride, up = dst - stride;
    for (let x = 0; x < stride; x++) {
      const a = x >= bpp ? lines[dst + x - bpp] : 0, b = y ? lines[up + x] : 0, c = x >= bpp && y ? lines[up + x - bpp] : 0;
      let pred = 0;
      if (ft === 1) pred = a; else if (ft === 2) pred = b; else if (ft === 3) pred = (a + b) >> 1;
      else if (ft === 4) { const p = a + b - c, pa = Math.abs(p - a), pb = Math.abs(p - b), pc = Math.abs(p - c); pred = pa <= pb && pa <= pc ? a : (pb <= pc ? b : c); }
      else if (ft !== 0) throw new Error(`bad PNG filter ${ft}`);
      lines[dst + x] = (raw[src + x] + pred) & 255;
    }
  }
  const out = new Uint8Array(4 * width * height);
  const alphaOf = new Uint8Array(256).fill(255);
  if (trns && ctype === 3) alphaOf.set(trns.subarray(0, 256));
  if (ctype === 3 && !palette) throw new Error('palette PNG without PLTE');
  for (let y = 0; y < height; y++) {
    for (let x = 0; x < width; x++) {
      const o = 4 * (y * width + x);
      let r, g, b, a = 255;
      if (packed) {   // samples are packed most significant bit first; grey levels scale to 0..255
        const bit = x * depth, v = (lines[y * stride + (bit >> 3)] >> (8 - depth - (bit & 7))) & ((1 << depth) - 1);
        if (ctype === 0) { r = g = b = Math.floor(v * 255 / ((1 << depth) - 1)); } else { r = palette[3 * v]; g = palette[3 * v + 1]; b = palette[3 * v + 2]; a = alphaOf[v]; }
      } else {
        const p = y * stride + x * channels;
        if (ctype === 0) { r = g = b = lines[p]; }
        else if (ctype === 2) { r = lines[p]; g = lines[p + 1]; b = lines[p + 2]; }
        else if (ctype === 3) { const v = lines[p]; r = palette[3 * v]; g = palette[3 * v + 1]; b = palette[3 * v + 2]; a = alphaOf[v]; }
        else if (ctype === 4) { r = g = b = lines[p]; a = lines[p + 1]; }
        else { r = lines[p]; g = lines[p + 1]; b = lines[p + 2]; a = lines[p + 3]; }
      }
      out[o] = r; out[o + 1] = g; out[o + 2] = b; out[o + 3] = a;
    }
  }
  return { width, height, data: out };
}

const CRC = (() => { const t = new Uint32Array(256); for (let n = 0; n < 256; n++) { let c = n; for (let k = 0; k < 8; k++) c = c & 1 ? 0xedb88320 ^ (c >>> 1) : c >>> 1; t[n] = c >>> 0; } return t; })();
function crc32(buf) { let c = 0xffffffff; for (let i = 0; i < buf.length; i++) c = CRC[(c ^ buf[i]) & 255] ^ (c >>> 8); return (c ^ 0xffffffff) >>> 0; }
/** Minimal PNG writer (8-bit RGBA, filter 0, one IDAT) for a frame read back from the device. */
function encodePNG(rgba, width, height) {
  const raw = Buffer.alloc(height * (4 * width + 1));
  for (let y = 0; y < height; y++) Buffer.from(rgba.buffer, rgba.byteOffset + 4 * width * y, 4 * width).copy(raw, y * (4 * width + 1) + 1);
  const chunk = (tag, data) => {
    const head = Buffer.alloc(8); head.writeUInt32BE(data.length, 0); head.write(tag, 4, 'latin1');
    const tail = Buffer.alloc(4); tail.writeUInt32BE(crc32(Buffer.concat([head.subarray(4), data])), 0);
    return Buffer.concat([head, data, tail]);
  };
  const ihdr = Buffer.alloc(13); ihdr.writeUInt32BE(width, 0); ihdr.writeUInt32BE(height, 4); ihdr[8] = 8; ihdr[9] = 6;
  return Buffer.concat([PNG_MAGIC, chunk('IHDR', ihdr), chunk('IDAT', zlib.deflateSync(raw, { level: 6 })), chunk('IEND', Buffer.alloc(0))]);
}

/** PNG or JPEG bytes -> { width, height, data } (RGBA, opaque where the file has no alpha), like an ImageBitmap upload. */
function decodeImage(bytes, name) {
  const buf = Buffer.isBuffer(bytes) ? bytes : Buffer.from(bytes);
  if (buf.length >= 8 && buf.compare(PNG_MAGIC, 0, 8, 0, 8) === 0) return decodePNG(buf);
  if (buf.length >= 3 && buf[0] === 0xff && buf[1] === 0xd8) return decodeJPEG(buf);
  throw new Error(`${name || 'image'}: neither a PNG nor a JPEG file`);
}

/** Ordering of a.localeCompare(b, undefined, { numeric: true, sensitivity: 'base' }) (load-images.ts:17). */
function compareNames(a, b) { return a.localeCompare(b, undefined, { numeric: true, sensitivity: 'base' }); }

/** createTextureFromImage(device, image) (load-images.ts:42-56): rgba8 rows, top to bottom, into a device buffer. */
function createTextureFromImage(device, image) {
  const tex = device.createBuffer({ size: 4 * image.width * image.height, label: 'gt image' });
  device.queue.writeBuffer(tex, 0, image.data);
  return tex;
}

/** loadImages(files, device) (load-images.ts:11-40): `files` are paths or { name, data } entries; without a device the images stay on the host
 *  (texture null).  -> LoadedImage[]: { name, file, bitmap: { width, height, data }, width, height, texture }. */
function loadImages(files, device) {
  const entries = Array.from(files).map((f) => (typeof f === 'string' ? { name: path.basename(f), file: f } : { name: f.name, file: f, data: f.data }))
    .filter((e) => /\.(jpe?g|png)$/i.test(e.name));
  entries.sort((a, b) => compareNames(a.name, b.name));
  const out = [];
  for (const e of entries) {
    try {
      const bitmap = decodeImage(e.data || fs.readFileSync(e.file), e.name);
      out.push({ name: e.name, file: e.file, bitmap, width: bitmap.width, height: bitmap.height, texture: device ? createTextureFromImage(device, bitmap) : null });
    } catch (err) {
      console.error(`Failed to load image ${e.name}:`, err.message);   // load-images.ts:31-34: log and drop
    }
  }
  return out;
}

module.exports = { decodePNG, decodeJPEG, encodePNG, decodeImage, compareNames, createTextureFromImage, loadImages };
