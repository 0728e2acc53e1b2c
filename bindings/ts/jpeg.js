'use strict';
/*
 * jpeg.js (+ jpeg.d.ts) -- JPEG ground truth for the TypeScript-side host without a browser.  The reference ingests .jpg / .jpeg / .png through
 * createImageBitmap (src/utils/load-images.ts:11-40), i.e. through the browser's decoder -- libjpeg-turbo in Chromium.  COLMAP datasets are
 * JPEG, so a host that is to train on "identical COLMAP inputs" has to turn the same file into the same rgba8 texels.  This is a decoder for
 * the files such datasets hold -- 8-bit baseline / extended-sequential / progressive Huffman JPEG, greyscale or three components, 4:4:4 / 4:2:2 /
 * 4:2:0 / 4:4:0 sampling, restart intervals -- written to the PUBLISHED algorithms libjpeg-turbo uses by default, integer step for integer step,
 * so that the texels equal the ones the Python host gets from Pillow (which bundles libjpeg-turbo): tests/test_js_host_cpu.py compares them
 * bit for bit over samplings, qualities, odd sizes, progressive and restart-marker files.
 *   - inverse DCT: the "slow-but-accurate" integer IDCT (Loeffler, Ligtenberg & Moschytz, 13-bit constants, 2-bit pass-1 scaling: jidctint.c),
 *   - chroma: "fancy" triangle-filter upsampling (h2v1: 3/4 + 1/4; h2v2: 9/16, 3/16, 3/16, 1/16 with libjpeg-turbo's alternating rounding bias),
 *     plain replication where a component is at most two samples wide,
 *   - colour: YCbCr -> RGB from 16-bit fixed-point tables (1.40200, 0.34414, 0.71414, 1.77200), results clamped to 0..255.
 * Not handled (the file is then dropped with a message, as any undecodable file is: load-images.ts:31-34): arithmetic coding, 12-bit samples,
 * lossless and hierarchical processes, four-component (CMYK / YCCK) files.  EXIF orientation is not applied (Pillow does not apply it either).
 */

const ZIGZAG = Uint8Array.from([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57,
  50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]);

/** Canonical Huffman table -> { maxcode, valptr, mincode, values, look (8-bit first-level lookup: length << 8 | symbol) } */
function buildHuffman(counts, values) {
  const codes = [], sizes = [];
  let code = 0;
  for (let len = 1; len <= 16; len++) { for (let i = 0; i < counts[len - 1]; i++) { codes.push(code++); sizes.push(len); } code <<= 1; }
  const maxcode = new Int32Array(18).fill(-1), mincode = new Int32Array(17), valptr = new Int32Array(17);
  let p = 0;
  for (let len = 1; len <= 16; len++) {
    if (counts[len - 1]) { valptr[len] = p; mincode[len] = codes[p]; p += counts[len - 1]; maxcode[len] = codes[p - 1]; }
  }
  maxcode[17] = 0x7fffffff;
  const look = new Int32Array(256).fill(0);
  for (let i = 0; i < codes.length; i++) {
    if (sizes[i] > 8) break;
    const first = codes[i] << (8 - sizes[i]);
    for (let k = 0; k < (1 << (8 - sizes[i])); k++) look[first + k] = (sizes[i] << 8) | values[i];
  }
  return { maxcode, mincode, valptr, values, look };
}

class BitReader {
  constructor(data, pos) { this.data = data; this.pos = pos; this.bits = 0; this.count = 0; this.marker = 0; }
  fill() {   // one more byte into the bit buffer; after a marker the stream is padded with zero bits (a truncated scan decodes as zeros)
    let b = 0;
    if (!this.marker && this.pos < this.data.length) {
      b = this.data[this.pos++];
      if (b === 0xff) {
        const n = this.pos < this.data.length ? this.data[this.pos] : 0xd9;
        if (n === 0) this.pos++; else { this.marker = n; this.pos--; b = 0; }
      }
    }
    this.bits = ((this.bits << 8) | b) >>> 0; this.count += 8;
  }
  peek(n) { while (this.count < n) this.fill(); return (this.bits >>> (this.count - n)) & ((1 << n) - 1); }
  skip(n) { this.count -= n; this.bits &= this.count ? ((1 << this.count) - 1) >>> 0 : 0; }
  get(n) { if (n === 0) return 0; const v = this.peek(n); this.skip(n); return v; }
  bit() { return this.get(1); }
  /** sign-extended n-bit magnitude (the EXTEND procedure of the standard, F.2.2.1) */
  extend(n) { if (n === 0) return 0; const v = this.get(n); return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v; }
  decode(h) {
    const l = h.look[this.peek(8)];
    if (l) { this.skip(l >> 8); return l & 255; }
    let code = this.peek(8), len = 8;
    this.skip(8);
    do { code = (code << 1) | this.bit(); len++; } while (len <= 16 && code > h.maxcode[len]);
    if (len > 16) return 0;   // (a corrupt code: libjpeg warns and uses symbol 0)
    return h.values[h.valptr[len] + code - h.mincode[len]];
  }
  /** positions the reader behind an expected restart marker (byte-aligned) */
  restart() {
    this.bits = 0; this.count = 0;
    if (!this.marker) {   // the marker may not have been reached by the bit buffer yet: look for it
      while (this.pos + 1 < this.data.length && !(this.data[this.pos] === 0xff && this.data[this.pos + 1] >= 0xd0 && this.data[this.pos + 1] <= 0xd7)) this.pos++;
    }
    if (this.pos + 1 < this.data.length && this.data[this.pos] === 0xff) this.pos += 2;
    this.marker = 0;
  }
}

// ---- the slow-but-accurate integer inverse DCT (CONST_BITS = 13, PASS1_BITS = 2)
const F_0_298 = 2446, F_0_390 = 3196, F_0_541 = 4433, F_0_765 = 6270, F_0_899 = 7373, F_1_175 = 9633, F_1_501 = 12299, F_1_847 = 15137, F_1_961 = 16069,
  F_2_053 = 16819, F_2_562 = 20995, F_3_072 = 25172;
const ws = new Int32Array(64);
/** coef: Int16Array / Int32Array view of one block in natural order; quant: the component's table in natural order; out: Uint8Array(64) */
function idctBlock(coef, base, quant, out, outBase, outStride) {
  for (let c = 0; c < 8; c++) {   // pass 1: columns, results scaled up by 2^PASS1_BITS
    const i = base + c;
    if (!(coef[i + 8] | coef[i + 16] | coef[i + 24] | coef[i + 32] | coef[i + 40] | coef[i + 48] | coef[i + 56])) {
      const dc = (coef[i] * quant[c]) << 2;
      ws[c] = ws[c + 8] = ws[c + 16] = ws[c + 24] = ws[c + 32] = ws[c + 40] = ws[c + 48] = ws[c + 56] = dc;
      continue;
    }
    let z2 = coef[i + 16] * quant[c + 16], z3 = coef[i + 48] * quant[c + 48];
    let z1 = (z2 + z3) * F_0_541;
    const tmp2 = z1 + z3 * -F_1_847, tmp3 = z1 + z2 * F_0_765;
    z2 = coef[i] * quant[c]; z3 = coef[i + 32] * quant[c + 32];
    const tmp0 = (z2 + z3) << 13, tmp1 = (z2 - z3) << 13;
    const tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    let t0 = coef[i + 56] * quant[c + 56], t1 = coef[i + 40] * quant[c + 40], t2 = coef[i + 24] * quant[c + 24], t3 = coef[i + 8] * quant[c + 8];
    z1 = t0 + t3; z2 = t1 + t2; z3 = t0 + t2; let z4 = t1 + t3;
    const z5 = (z3 + z4) * F_1_175;
    t0 *= F_0_298; t1 *= F_2_053; t2 *= F_3_072; t3 *= F_1_501;
    z1 *= -F_0_899; z2 *= -F_2_562; z3 = z3 * -F_1_961 + z5; z4 = z4 * -F_0_390 + z5;
    t0 += z1 + z3; t1 += z2 + z4; t2 += z2 + z3; t3 += z1 + z4;
    ws[c] = (tmp10 + t3 + 1024) >> 11; ws[c + 56] = (tmp10 - t3 + 1024) >> 11;
    ws[c + 8] = (tmp11 + t2 + 1024) >> 11; ws[c + 48] = (tmp11 - t2 + 1024) >> 11;
    ws[c + 16] = (tmp12 + t1 + 1024) >> 11; ws[c + 40] = (tmp12 - t1 + 1024) >> 11;
    ws[c + 24] = (tmp13 + t0 + 1024) >> 11; ws[c + 32] = (tmp13 - t0 + 1024) >> 11;
  }
  // the table lookup range_limit[x & 1023] of the decoder, centred: clamp(s + 128, 0, 255) for the 10-bit two's-complement value s of x
  const lim = (x) => { const s = ((x & 1023) ^ 512) - 512 + 128; return s < 0 ? 0 : (s > 255 ? 255 : s); };
  for (let r = 0; r < 8; r++) {   // pass 2: rows, descaled by 2^(CONST_BITS + PASS1_BITS + 3)
    const w = r * 8, o = outBase + r * outStride;
    if (!(ws[w + 1] | ws[w + 2] | ws[w + 3] | ws[w + 4] | ws[w + 5] | ws[w + 6] | ws[w + 7])) {
      const dc = lim((ws[w] + 16) >> 5);
      out[o] = out[o + 1] = out[o + 2] = out[o + 3] = out[o + 4] = out[o + 5] = out[o + 6] = out[o + 7] = dc;
      continue;
    }
    let z2 = ws[w + 2], z3 = ws[w + 6];
    let z1 = (z2 + z3) * F_0_541;
    const tmp2 = z1 + z3 * -F_1_847, tmp3 = z1 + z2 * F_0_765;
    const tmp0 = (ws[w] + ws[w + 4]) << 13, tmp1 = (ws[w] - ws[w + 4]) << 13;
    const tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    let t0 = ws[w + 7], t1 = ws[w + 5], t2 = ws[w + 3], t3 = ws[w + 1];
    z1 = t0 + t3; z2 = t1 + t2; z3 = t0 + t2; let z4 = t1 + t3;
    const z5 = (z3 + z4) * F_1_175;
    t0 *= F_0_298; t1 *= F_2_053; t2 *= F_3_072; t3 *= F_1_501;
    z1 *= -F_0_899; z2 *= -F_2_562; z3 = z3 * -F_1_961 + z5; z4 = z4 * -F_0_390 + z5;
    t0 += z1 + z3; t1 += z2 + z4; t2 += z2 + z3; t3 += z1 + z4;
    const H = 131072;   // 1 << (CONST_BITS + PASS1_BITS + 3 - 1)
    out[o] = lim((tmp10 + t3 + H) >> 18); out[o + 7] = lim((tmp10 - t3 + H) >> 18);
    out[o + 1] = lim((tmp11 + t2 + H) >> 18); out[o + 6] = lim((tmp11 - t2 + H) >> 18);
    out[o + 2] = lim((tmp12 + t1 + H) >> 18); out[o + 5] = lim((tmp12 - t1 + H) >> 18);
    out[o + 3] = lim((tmp13 + t0 + H) >> 18); out[o + 4] = lim((tmp13 - t0 + H) >> 18);
  }
}

function parse(data) {
  if (data.length < 4 || data[0] !== 0xff || data[1] !== 0xd8) throw new Error('not a JPEG file');
  const img = { quant: [], dc: [], ac: [], components: null, restartInterval: 0, jfif: false, adobe: -1, progressive: false, scans: [] };
  let pos = 2;
  const u16 = (p) => (data[p] << 8) | data[p + 1];
  for (;;) {
    while (pos < data.length && data[pos] !== 0xff) pos++;   // (garbage between segments is skipped, as libjpeg does with a warning)
    while (pos < data.length && data[pos] === 0xff) pos++;
    if (pos >= data.length) break;
    const m = data[pos++];
    if (m === 0xd9) break;                       // EOI
    if (m === 0x01 || (m >= 0xd0 && m <= 0xd7)) continue;   // TEM, stray RSTn
    const len = u16(pos), end = pos + len;
    if (m === 0xdb) {                            // DQT
      let p = pos + 2;
      while (p < end) {
        const pq = data[p] >> 4, tq = data[p] & 15; p++;
        const t = new Int32Array(64);
        for (let i = 0; i < 64; i++) { t[ZIGZAG[i]] = pq ? u16(p) : data[p]; p += pq ? 2 : 1; }
        img.quant[tq] = t;
      }
    } else if (m === 0xc4) {                     // DHT
      let p = pos + 2;
      while (p < end) {
        const tc = data[p] >> 4, th = data[p] & 15; p++;
        const counts = data.subarray(p, p + 16); p += 16;
        let n = 0; for (let i = 0; i < 16; i++) n += counts[i];
        (tc ? img.ac : img.dc)[th] = buildHuffman(counts, data.subarray(p, p + n)); p += n;
      }
    } else if (m === 0xc0 || m === 0xc1 || m === 0xc2) {   // SOF0 / SOF1 / SOF2
      if (data[pos + 2] !== 8) throw new Error(`unsupported JPEG: ${data[pos + 2]}-bit samples`);
      img.progressive = m === 0xc2;
      img.height = u16(pos + 3); img.width = u16(pos + 5);
      const n = data[pos + 7];
      if (n !== 1 && n !== 3) throw new Error(`unsupported JPEG: ${n} colour components`);
      img.components = [];
      for (let i = 0; i < n; i++) { const q = pos + 8 + 3 * i; img.components.push({ id: data[q], h: data[q + 1] >> 4, v: data[q + 1] & 15, tq: data[q + 2] }); }
    } else if (m >= 0xc3 && m <= 0xcf && m !== 0xc4 && m !== 0xc8 && m !== 0xcc) {
      throw new Error('unsupported JPEG process (lossless, hierarchical or arithmetic coding)');
    } else if (m === 0xdd) img.restartInterval = u16(pos + 2);
    else if (m === 0xe0 && len >= 7 && data[pos + 2] === 0x4a && data[pos + 3] === 0x46 && data[pos + 4] === 0x49 && data[pos + 5] === 0x46 && data[pos + 6] === 0) img.jfif = true;
    else if (m === 0xee && len >= 14 && String.fromCharCode.apply(null, data.subarray(pos + 2, pos + 7)) === 'Adobe') img.adobe = data[pos + 13];
    else if (m === 0xda) {                       // SOS: the entropy-coded segment follows
      if (!img.components) throw new Error('JPEG: scan before frame header');
      const ns = data[pos + 2], comps = [];
      for (let i = 0; i < ns; i++) {
        const cid = data[pos + 3 + 2 * i], tb = data[pos + 4 + 2 * i];
        const ci = img.components.findIndex((c) => c.id === cid);
        if (ci < 0) throw new Error('JPEG: scan names an unknown component');
        comps.push({ ci, td: tb >> 4, ta: tb & 15 });
      }
      const q = pos + 3 + 2 * ns;
      const scan = { comps, ss: data[q], se: data[q + 1], ah: data[q + 2] >> 4, al: data[q + 2] & 15, start: end,
        dc: img.dc.slice(), ac: img.ac.slice(), restartInterval: img.restartInterval };
      img.scans.push(scan);
      pos = decodeScan(data, img, scan);
      continue;
    }
    pos = end;
  }
  if (!img.components || !img.scans.length) throw new Error('JPEG: no image data');
  return img;
}

function setupFrame(img) {
  if (img.blocksReady) return;
  img.hmax = Math.max.apply(null, img.components.map((c) => c.h)); img.vmax = Math.max.apply(null, img.components.map((c) => c.v));
  img.mcusX = Math.ceil(img.width / (8 * img.hmax)); img.mcusY = Math.ceil(img.height / (8 * img.vmax));
  for (const c of img.components) {
    c.bw = img.mcusX * c.h; c.bh = img.mcusY * c.v;                                  // blocks held (whole MCUs)
    c.widthBlocks = Math.ceil(Math.ceil(img.width * c.h / img.hmax) / 8);            // blocks a non-interleaved scan of the component codes
    c.heightBlocks = Math.ceil(Math.ceil(img.height * c.v / img.vmax) / 8);
    c.coef = new Int16Array(64 * c.bw * c.bh); c.pred = 0;
  }
  img.blocksReady = true;
}

/** Decodes one scan into the components' coefficient arrays (natural order); returns the position behind its entropy-coded data. */
function decodeScan(data, img, scan) {
  setupFrame(img);
  const br = new BitReader(data, scan.start);
  const comps = scan.comps.map((s) => Object.assign({ c: img.components[s.ci] }, s));
  for (const s of comps) s.c.pred = 0;
  const interleaved = comps.length > 1;
  const total = interleaved ? img.mcusX * img.mcusY : comps[0].c.widthBlocks * comps[0].c.heightBlocks;
  let eobrun = 0;
  const { ss, se, ah, al } = scan;
  const p1 = 1 << al, m1 = -1 << al;

  const baselineBlock = (s, coef, b) => {
    const t = br.decode(scan.dc[s.td]);
    s.c.pred += br.extend(t);
    coef[b] = s.c.pred;
    const h = scan.ac[s.ta];
    for (let k = 1; k < 64;) {
      const rs = br.decode(h), r = rs >> 4, sz = rs & 15;
      if (sz === 0) { if (r !== 15) break; k += 16; continue; }
      k += r;
      if (k > 63) break;
      coef[b + ZIGZAG[k]] = br.extend(sz); k++;
    }
  };
  const dcFirst = (s, coef, b) => { const t = br.decode(scan.dc[s.td]); s.c.pred += br.extend(t); coef[b] = s.c.pred * p1; };
  const dcRefine = (_s, coef, b) => { if (br.bit()) coef[b] |= p1; };
  const acFirst = (s, coef, b) => {
    if (eobrun > 0) { eobrun--; return; }
    const h = scan.ac[s.ta];
    for (let k = ss; k <= se;) {
      const rs = br.decode(h), r = rs >> 4, sz = rs & 15;
      if (sz === 0) {
        if (r < 15) { eobrun = (1 << r) - 1; if (r) eobrun += br.get(r); break; }
        k += 16; continue;
      }
      k += r;
      coef[b + ZIGZAG[k]] = br.extend(sz) * p1; k++;
    }
  };
  const acRefine = (s, coef, b) => {   // successive-approximation refinement of AC coefficients (G.1.2.3 of the standard)
    const h = scan.ac[s.ta];
    let k = ss;
    if (eobrun === 0) {
      for (; k <= se; k++) {
        const rs = br.decode(h); let r = rs >> 4; const sz = rs & 15;
        let value = 0;
        if (sz) value = br.bit() ? p1 : m1;   // (size must be 1)
        else if (r !== 15) { eobrun = 1 << r; if (r) eobrun += br.get(r); break; }
        for (; k <= se; k++) {   // skip r zero-history coefficients, refining the non-zero ones passed on the way
          const z = b + ZIGZAG[k];
          if (coef[z] !== 0) { if (br.bit() && (coef[z] & p1) === 0) coef[z] += coef[z] >= 0 ? p1 : m1; }
          else { if (--r < 0) break; }
        }
        if (value && k <= se) coef[b + ZIGZAG[k]] = value;
      }
    }
    if (eobrun > 0) {
      for (; k <= se; k++) {
        const z = b + ZIGZAG[k];
        if (coef[z] !== 0 && br.bit() && (coef[z] & p1) === 0) coef[z] += coef[z] >= 0 ? p1 : m1;
      }
      eobrun--;
    }
  };
  const blockFn = !img.progressive ? baselineBlock : (ss === 0 ? (ah === 0 ? dcFirst : dcRefine) : (ah === 0 ? acFirst : acRefine));

  for (let mcu = 0; mcu < total; mcu++) {
    if (scan.restartInterval && mcu > 0 && mcu % scan.restartInterval === 0) { br.restart(); for (const s of comps) s.c.pred = 0; eobrun = 0; }
    if (interleaved) {
      const mx = mcu % img.mcusX, my = Math.floor(mcu / img.mcusX);
      for (const s of comps) for (let v = 0; v < s.c.v; v++) for (let h = 0; h < s.c.h; h++) blockFn(s, s.c.coef, 64 * ((my * s.c.v + v) * s.c.bw + mx * s.c.h + h));
    } else {
      const s = comps[0], bx = mcu % s.c.widthBlocks, by = Math.floor(mcu / s.c.widthBlocks);
      blockFn(s, s.c.coef, 64 * (by * s.c.bw + bx));
    }
  }
  // behind the scan: the next marker (the bit reader stops in front of it)
  let pos = br.marker ? br.pos : br.pos;
  while (pos + 1 < data.length && !(data[pos] === 0xff && data[pos + 1] !== 0 && !(data[pos + 1] >= 0xd0 && data[pos + 1] <= 0xd7))) pos++;
  return pos;
}

/** JPEG bytes -> { width, height, data: Uint8Array(4 W H) } (opaque alpha). */
function decodeJPEG(bytes) {
  const data = bytes instanceof Uint8Array ? bytes : new Uint8Array(bytes);
  const img = parse(data);
  const W = img.width, H = img.height;
  // ---- dequantisation + inverse DCT: one sample plane per component, whole blocks
  for (const c of img.components) {
    const q = img.quant[c.tq];
    if (!q) throw new Error('JPEG: missing quantisation table');
    c.pw = c.bw * 8; c.ph = c.bh * 8;
    c.plane = new Uint8Array(c.pw * c.ph);
    for (let by = 0; by < c.bh; by++) for (let bx = 0; bx < c.bw; bx++) idctBlock(c.coef, 64 * (by * c.bw + bx), q, c.plane, by * 8 * c.pw + bx * 8, c.pw);
    c.dw = Math.ceil(W * c.h / img.hmax); c.dh = Math.ceil(H * c.v / img.vmax);   // the component's true ("downsampled") size
  }
  // ---- upsampling to full resolution
  const full = img.components.map((c) => {
    const hx = img.hmax / c.h, vx = img.vmax / c.v;
    if (hx === 1 && vx === 1) return { p: c.plane, stride: c.pw };
    const out = new Uint8Array(W * H + 2 * W + 4);   // (rows are written in pairs: room for one row and a sample beyond an odd size)
    const src = c.plane, sp = c.pw, dw = c.dw, dh = c.dh;
    const row = (y) => Math.min(Math.max(y, 0), dh - 1) * sp;   // rows above the first / below the last real row repeat it
    if (hx === 2 && vx === 1) {
      for (let y = 0; y < H; y++) {
        const i = row(y), o = y * W;
        if (dw > 2) {   // triangle filter: 3/4 nearer + 1/4 further sample; the first and last columns repeat the edge
          let v = src[i];
          out[o] = v; out[o + 1] = (v * 3 + src[i + 1] + 2) >> 2;
          for (let x = 1; x < dw - 1; x++) { v = src[i + x] * 3; out[o + 2 * x] = (v + src[i + x - 1] + 1) >> 2; out[o + 2 * x + 1] = (v + src[i + x + 1] + 2) >> 2; }
          v = src[i + dw - 1];
          out[o + 2 * dw - 2] = (v * 3 + src[i + dw - 2] + 1) >> 2; if (2 * dw - 1 < W) out[o + 2 * dw - 1] = v;
        } else for (let x = 0; x < W; x++) out[o + x] = src[i + (x >> 1)];
      }
    } else if (hx === 2 && vx === 2) {
      for (let y = 0; y < H; y++) {
        const i0 = row(y >> 1), i1 = row((y & 1) ? (y >> 1) + 1 : (y >> 1) - 1), o = y * W;
        if (dw > 2) {   // 3/4 nearer row + 1/4 further row, then the same horizontally, on the column sums; rounding bias alternates 8 / 7
          let cur = src[i0] * 3 + src[i1], next = src[i0 + 1] * 3 + src[i1 + 1], last;
          out[o] = (cur * 4 + 8) >> 4; out[o + 1] = (cur * 3 + next + 7) >> 4;
          last = cur; cur = next;
          for (let x = 1; x < dw - 1; x++) {
            next = src[i0 + x + 1] * 3 + src[i1 + x + 1];
            out[o + 2 * x] = (cur * 3 + last + 8) >> 4; out[o + 2 * x + 1] = (cur * 3 + next + 7) >> 4;
            last = cur; cur = next;
          }
          out[o + 2 * dw - 2] = (cur * 3 + last + 8) >> 4; if (2 * dw - 1 < W) out[o + 2 * dw - 1] = (cur * 4 + 7) >> 4;
        } else for (let x = 0; x < W; x++) out[o + x] = src[i0 + (x >> 1)];
      }
    } else if (hx === 1 && vx === 2) {
      for (let y = 0; y < H; y++) {
        const i0 = row(y >> 1), i1 = row((y & 1) ? (y >> 1) + 1 : (y >> 1) - 1), o = y * W, bias = (y & 1) ? 2 : 1;
        for (let x = 0; x < W; x++) out[o + x] = (src[i0 + x] * 3 + src[i1 + x] + bias) >> 2;
      }
    } else if (Number.isInteger(hx) && Number.isInteger(vx)) {
      for (let y = 0; y < H; y++) { const i = row(Math.floor(y / vx)), o = y * W; for (let x = 0; x < W; x++) out[o + x] = src[i + Math.floor(x / hx)]; }
    } else throw new Error('unsupported JPEG sampling factors');
    return { p: out, stride: W };
  });
  // ---- colour
  const rgba = new Uint8Array(4 * W * H);
  const clamp = (v) => (v < 0 ? 0 : (v > 255 ? 255 : v));
  if (img.components.length === 1) {
    const f = full[0];
    for (let y = 0; y < H; y++) for (let x = 0; x < W; x++) { const v = f.p[y * f.stride + x], o = 4 * (y * W + x); rgba[o] = rgba[o + 1] = rgba[o + 2] = v; rgba[o + 3] = 255; }
    return { width: W, height: H, data: rgba };
  }
  // three components: YCbCr unless the markers say RGB (an Adobe marker with transform 0, or component ids 'R', 'G', 'B' without JFIF / Adobe)
  const ids = img.components.map((c) => c.id);
  const isRGB = !img.jfif && (img.adobe === 0 || (img.adobe < 0 && ids[0] === 0x52 && ids[1] === 0x47 && ids[2] === 0x42));
  const crR = new Int32Array(256), cbB = new Int32Array(256), crG = new Int32Array(256), cbG = new Int32Array(256);
  for (let i = 0; i < 256; i++) {
    const x = i - 128;
    crR[i] = (91881 * x + 32768) >> 16; cbB[i] = (116130 * x + 32768) >> 16; crG[i] = -46802 * x; cbG[i] = -22554 * x + 32768;
  }
  const [f0, f1, f2] = full;
  for (let y = 0; y < H; y++) {
    for (let x = 0; x < W; x++) {
      const a = f0.p[y * f0.stride + x], b = f1.p[y * f1.stride + x], c = f2.p[y * f2.stride + x], o = 4 * (y * W + x);
      if (isRGB) { rgba[o] = a; rgba[o + 1] = b; rgba[o + 2] = c; }
      else { rgba[o] = clamp(a + crR[c]); rgba[o + 1] = clamp(a + ((cbG[b] + crG[c]) >> 16)); rgba[o + 2] = clamp(a + cbB[b]); }
      rgba[o + 3] = 255;
    }
  }
  return { width: W, height: H, data: rgba };
}

module.exports = { decodeJPEG };
