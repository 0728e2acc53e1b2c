// node bindings/napi/densify_timing.js [config] -- where a densify event's host time goes in the JS Trainer (dev tool): wraps the event's methods with timers
'use strict';
const path = require('path');
const hip = require(path.join(__dirname, '..', 'ts', 'webdgs_hip.js'));
const synth = require(path.join(__dirname, '..', 'ts', 'synth.js'));
const { Trainer } = require(path.join(__dirname, '..', 'ts', 'trainer.js'));
const now = () => { const t = process.hrtime(); return (t[0] + t[1] * 1e-9) * 1e3; };
async function main() {
  const cfg = synth.CONFIGS[process.argv[2] || 'c3'];
  const dev = new hip.HipDevice(0);
  const scene = synth.makeGaussians(cfg), target = synth.makeTargetScene(scene.gaussians, scene.sh);
  const cams = synth.circleCameras(cfg, 8);
  const upload = (words) => { const b = dev.createBuffer({ size: words.byteLength }); dev.queue.writeBuffer(b, 0, words); return b; };
  const cloudOf = (s) => ({ type: 'full', num_points: cfg.num_points, sh_deg: cfg.sh_deg, gaussian_3d_buffer: upload(s.gaussians), sh_buffer: upload(s.sh) });
  const tpc = cloudOf(target), tcam = dev.createBuffer({ size: 272 });
  const tfw = new hip.TiledForwardPass(dev, tpc, tcam, { viewportWidth: cfg.width, viewportHeight: cfg.height, renderMode: 'gaussian' });
  const trs = new hip.TiledRasterizer({ device: dev, forwardPass: tfw, format: 'rgba8unorm' });
  const cameras = [], images = [];
  for (const cam of cams) {
    dev.queue.writeBuffer(tcam, 0, cam); tfw.encode(null); trs.encode(null, cfg.width, cfg.height);
    const img = dev.createBuffer({ size: 4 * cfg.width * cfg.height });
    dev.createCommandEncoder().copyBufferToBuffer(trs.getOutputTextureView(), 0, img, 0, 4 * cfg.width * cfg.height); dev.synchronize();
    images.push({ texture: img, width: cfg.width, height: cfg.height }); cameras.push({ camera: cam, width: cfg.width, height: cfg.height });
  }
  const t = new Trainer(dev, undefined, { pipelineDepth: 2 });
  t.setDensifyPruneConfig({ schedule: { enabled: true, warmupIterations: 20, interval: 20, stopIterations: 1000 } });
  t.setPointCloud(cloudOf(scene)); t.setDataset(cameras, images); t.setMaxIterations(1e9); t.start();
  const acc = {};
  const wrap = (obj, name, label) => { const f = obj[name]; obj[name] = function () { const t0 = now(); const r = f.apply(this, arguments);
    const done = () => { dev.synchronize(); acc[label] = (acc[label] || []); acc[label].push(now() - t0); };
    if (r && typeof r.then === 'function') return r.then((v) => { done(); return v; }); done(); return r; }; };
  wrap(t, 'runDensifyPruneMultiView', 'runDensifyPruneMultiView'); wrap(t, 'applyPointCloudSwap', 'applyPointCloudSwap');
  wrap(t, 'ensureMetricsPipelines', 'ensureMetricsPipelines');
  wrap(t, 'ensurePipelines', 'ensurePipelines'); wrap(t, 'invalidateCommandBuffers', 'invalidateCommandBuffers'); wrap(t, 'syncOptimizerState', 'syncOptimizerState');
  wrap(t, 'flushPointCloud', 'flushPointCloud'); wrap(t, 'drain', 'drain');
  for (const cls of ['TiledForwardPass', 'TiledBackwardPass']) wrap(hip[cls].prototype, 'setPointCloud', cls + '.setPointCloud');
  wrap(hip.Optimizer.prototype, 'destroy', 'Optimizer.destroy'); wrap(hip.Optimizer.prototype, 'setDeferredSH', 'Optimizer.setDeferredSH');
  wrap(hip.Optimizer.prototype, 'getStateBuffers', 'Optimizer.getStateBuffers'); wrap(hip.HipBuffer.prototype, 'destroy', 'HipBuffer.destroy');
  wrap(hip, 'allocatePointCloudLike', 'allocatePointCloudLike'); wrap(hip, 'allocateOptimizerStateBuffers', 'allocateOptimizerStateBuffers');
  wrap(hip.DensifyPrunePass.prototype, 'encodeScatter', 'encodeScatter'); wrap(hip.DensifyPrunePass.prototype, 'encodePrepare', 'encodePrepare');
  wrap(hip.DensifyPrunePass.prototype, 'readTotal', 'readTotal'); wrap(hip.DensifyPrunePass.prototype, 'ensureSize', 'ensureSize');
  {  // (no device synchronize in this one: the bare call)
    const f = hip.HipCommandBuffer.prototype.destroy;
    hip.HipCommandBuffer.prototype.destroy = function () { const t0 = now(); f.apply(this, arguments); (acc['HipCommandBuffer.destroy (bare)'] = acc['HipCommandBuffer.destroy (bare)'] || []).push(now() - t0); };
  }
  const stepTimes = [];
  for (let i = 0; i < 70; i++) { const t0 = now(); await t.step(); stepTimes.push([t.getIteration(), now() - t0]); }
  t.drain(); dev.synchronize();
  for (const k of Object.keys(acc)) console.log(k, 'n=' + acc[k].length, 'sum(ms)=' + acc[k].reduce((a, b) => a + b, 0).toFixed(2), 'last few:', acc[k].slice(-6).map((x) => x.toFixed(2)).join(' '));
  console.log('steps around the events (iteration: ms):', stepTimes.filter((s) => s[1] > 1.5).map((s) => `${s[0]}: ${s[1].toFixed(2)}`).join('  '));
  console.log('steps 36..60 (ms):', stepTimes.filter((s) => s[0] >= 36 && s[0] <= 60).map((s) => s[1].toFixed(2)).join(' '));
  console.log('total ms of steps 21..60:', stepTimes.filter((s) => s[0] >= 21 && s[0] <= 60).reduce((a, s) => a + s[1], 0).toFixed(2));
}
main().catch((e) => { console.error(e); process.exit(1); });
