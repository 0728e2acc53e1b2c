// node bindings/napi/lifecycle_run.js -- ten JS Trainers in a row on one device (set up, trained across a densify rebuild, destroyed): prints the free device memory
// after every cycle (tests/test_gpu_lifecycle.py: it must settle).
'use strict';
const path = require('path');
const hip = require(path.join(__dirname, '..', 'ts', 'webdgs_hip.js'));
const synth = require(path.join(__dirname, '..', 'ts', 'synth.js'));
const { Trainer } = require(path.join(__dirname, '..', 'ts', 'trainer.js'));

async function main() {
  const cfg = { config_id: 2, num_points: 20000, width: 320, height: 240, sh_deg: 1, fy: 550.0, s0: 0.006, name: 'lifecycle' };
  const dev = new hip.HipDevice(0);
  const scene = synth.makeGaussians(cfg), target = synth.makeTargetScene(scene.gaussians, scene.sh);
  const cams = synth.circleCameras(cfg, 4);
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
  trs.destroy(); tfw.destroy(); tpc.gaussian_3d_buffer.destroy(); tpc.sh_buffer.destroy(); tcam.destroy();
  const dens = { schedule: { enabled: true, warmupIterations: 10, interval: 10, stopIterations: 100 }, metricViews: 3, cloneThresholdCount: 5, splitScaleThreshold: 0.03,
    pruneOpacity: 0.2, maxNewPointsPerStep: 500 };
  const free = [], densified = [];
  for (let cycle = 0; cycle < 10; cycle++) {
    const t = new Trainer(dev, undefined, { pipelineDepth: 2, viewsPerStep: cycle % 2 ? 3 : 1 });
    t.setDensifyPruneConfig(dens);
    t.setPointCloud(cloudOf(scene));
    t.setDataset(cameras, images);
    t.start();
    for (let i = 0; i < 25; i++) await t.step();
    t.drain(); dev.synchronize();
    densified.push(t.getLastDensifyPruneIteration() === 20 && t.getPointCount() !== cfg.num_points);
    const cloud = t.pointCloud;
    t.destroy();
    cloud.gaussian_3d_buffer.destroy(); cloud.sh_buffer.destroy();
    dev.synchronize();
    free.push(Math.round(dev.memoryInfo().free / 1048576));
  }
  for (const im of images) im.texture.destroy();
  const cachedBefore = dev.memoryInfo().cached;
  console.log(JSON.stringify({ cycles: free.length, densified, free_mib: free, cached_mib: Math.round(cachedBefore / 1048576) }));
  dev.destroy();
}
main().catch((e) => { console.error(e); process.exit(1); });
