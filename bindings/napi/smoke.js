// node bindings/napi/smoke.js [gpu] -- loads the addon; with "gpu" runs one forward+rasterize of a tiny scene.
const path = require('path');
const addon = require(path.join(__dirname, 'webdgs_napi.node'));
if (addon.abiVersion() !== 1) { console.error('bad ABI version'); process.exit(1); }
const names = Object.keys(addon);
if (names.length < 57) { console.error('missing exports', names); process.exit(1); }
if (process.argv[2] === 'gpu') {
  const dev = addon.deviceCreate(0);
  const n = 64, W = 64, H = 48;
  // 12 fp16 per Gaussian: a grid of small opaque splats at z = 4 in front of an identity camera
  const g = new Uint16Array(n * 12), sh = new Uint16Array(n * 48);
  const f16 = (x) => { const f = new Float32Array([x]), u = new Uint32Array(f.buffer)[0]; const s = (u >> 16) & 0x8000, e = ((u >> 23) & 0xff) - 112, m = (u >> 13) & 0x3ff; return e <= 0 ? s : (s | (e << 10) | m); };
  for (let i = 0; i < n; i++) { const o = i * 12; g[o] = f16(((i % 8) - 3.5) * 0.25); g[o + 1] = f16((Math.floor(i / 8) - 3.5) * 0.2); g[o + 2] = f16(4); g[o + 3] = f16(2);
    g[o + 4] = f16(1); g[o + 8] = g[o + 9] = g[o + 10] = f16(-3); sh[i * 48] = f16(1.5); }
  const cam = new Float32Array(68); [0, 5, 10, 15, 16, 21, 26, 31].forEach((k) => { cam[k] = 1; });
  const fy = 60, zn = 0.01, zf = 100; cam[32] = 2 * fy / W; cam[37] = -2 * fy / H; cam[42] = zf / (zf - zn); cam[43] = 1; cam[46] = -zf * zn / (zf - zn);
  cam[64] = W; cam[65] = H; cam[66] = fy; cam[67] = fy;
  const gb = addon.bufferCreate(dev, g.byteLength), sb = addon.bufferCreate(dev, sh.byteLength), cb = addon.bufferCreate(dev, 272);
  addon.copyToDevice(dev, gb.ptr, g); addon.copyToDevice(dev, sb.ptr, sh); addon.copyToDevice(dev, cb.ptr, cam);
  const fwd = addon.tiledForwardCreate(dev, { numPoints: n, shDeg: 0, viewportWidth: W, viewportHeight: H });
  const rast = addon.tiledRasterizerCreate(dev, fwd);
  addon.tiledForwardEncode(fwd, gb.ptr, sb.ptr, cb.ptr, 0);
  addon.tiledRasterizerEncode(rast, W, H);
  addon.deviceSynchronize(dev);
  const img = new Uint8Array(addon.copyToHost(dev, addon.tiledRasterizerGet(rast, 0), W * H * 4));
  let lit = 0; for (let i = 0; i < W * H; i++) if (img[i * 4] > 0) lit++;
  const stats = new Uint32Array(addon.copyToHost(dev, addon.tiledForwardGetResources(fwd).statsBuffer, 16));
  console.log(`napi gpu smoke: E=${stats[0]} visible=${stats[1]} lit pixels=${lit}`);
  if (stats[1] !== n || lit === 0) process.exit(1);
  // ---- one training step through the addon: backward + Adam, eager then as a recorded command buffer, awaited as a Promise
  const target = addon.bufferCreate(dev, W * H * 4);
  addon.tiledRasterizerBlit(rast, target.ptr, W, H);                       // blitToTexture: a copy of the render is the "ground truth"
  const half = addon.bufferCreate(dev, (W / 2) * (H / 2) * 4);
  addon.tiledRasterizerBlit(rast, half.ptr, W / 2, H / 2);                 // linear-sampler blit to another size
  const bwd = addon.tiledBackwardCreate(dev, { numPoints: n, shDeg: 0, viewportWidth: W, viewportHeight: H });
  const opt = addon.optimizerCreate(dev, n, gb.ptr, sb.ptr);
  const encodeStep = () => {
    addon.tiledForwardEncode(fwd, gb.ptr, sb.ptr, cb.ptr, 0);
    addon.tiledRasterizerEncode(rast, W, H);
    const r = addon.tiledForwardGetResources(fwd);
    addon.tiledBackwardEncode(bwd, addon.tiledRasterizerGet(rast, 0), target.ptr, {
      splatBuffer: r.splatBuffer, tileOffsetsBuffer: addon.tiledRasterizerGet(rast, 3), tileIndicesBuffer: r.tileIndicesBuffer, cameraBuffer: cb.ptr,
      alphaTexture: addon.tiledRasterizerGet(rast, 1), nContribTexture: addon.tiledRasterizerGet(rast, 2) }, gb.ptr);
    addon.optimizerStep(opt, gb.ptr, sb.ptr, addon.tiledBackwardGet(bwd, 0), r.tileCountsBuffer);
  };
  encodeStep();                                                            // eager (allocates on first use)
  addon.deviceSynchronize(dev);
  addon.encoderBegin(dev); encodeStep(); const cmd = addon.encoderFinish(dev);   // recorded: GPUCommandBuffer
  addon.optimizerAdvanceIteration(opt, 0);
  addon.queueSubmit(dev, cmd); addon.queueSubmit(dev, cmd);
  addon.optimizerAdvanceIteration(opt, 1);
  const hp = addon.optimizerHyperparameters(opt, { lr_pos: 0.001 });
  const st = addon.tiledForwardCheck(fwd);
  const dens = addon.densifyCreate(dev, { numViews: 1, cloneThreshold: 1, splitThreshold: 0.5, pruneThreshold: 0.01, maxNewPointsPerStep: 16 });
  const prep = addon.densifyEncodePrepare(dens, n, gb.ptr, addon.tiledBackwardGet(bwd, 1));
  const total = addon.densifyReadTotal(dens);
  // the same through the staged encoders (encodeDecision / encodePrefixSum / encodeCapToMax / encodePrefixSum / encodeTotalOut)
  const dens2 = addon.densifyCreate(dev, { numViews: 1, cloneThreshold: 1, splitThreshold: 0.5, pruneThreshold: 0.01, maxNewPointsPerStep: 16 });
  const maxOut = addon.densifyStage(dens2, 4, n, 0, 0).maxOutPoints;
  addon.densifyStage(dens2, 0, n, gb.ptr, addon.tiledBackwardGet(bwd, 1));
  addon.densifyStage(dens2, 1, n, 0, 0); addon.densifyStage(dens2, 2, n, maxOut, 0); addon.densifyStage(dens2, 1, n, 0, 0); addon.densifyStage(dens2, 3, n, 0, 0);
  if (addon.densifyReadTotal(dens2) !== total || maxOut !== prep.maxOutPoints) { console.error('staged densify differs from encodePrepare'); process.exit(1); }
  addon.densifyDestroy(dens2);
  const state = addon.optimizerState(opt, 0);
  addon.queueOnSubmittedWorkDone(dev).then(() => {
    console.log(`napi gpu train: iteration=${addon.optimizerGetIteration(opt)} visible=${st.visibleCount} lr_pos=${hp.lr_pos.toFixed(4)} densify total=${total}/${prep.maxOutPoints} state=${typeof state.optPosBuffer}`);
    if (addon.optimizerGetIteration(opt) !== 3 || total === 0 || total > prep.maxOutPoints) process.exit(1);
    addon.commandBufferDestroy(cmd); addon.densifyDestroy(dens); addon.optimizerDestroy(opt); addon.tiledBackwardDestroy(bwd);
    addon.tiledRasterizerDestroy(rast); addon.tiledForwardDestroy(fwd);
    [gb, sb, cb, target, half].forEach((b) => addon.bufferDestroy(b.handle)); addon.deviceDestroy(dev);
    console.log('napi smoke ok:', names.length, 'exports');
  }).catch((e) => { console.error(e); process.exit(1); });
} else {
  console.log('napi smoke ok:', names.length, 'exports');
}
