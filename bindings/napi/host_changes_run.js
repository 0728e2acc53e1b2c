// node bindings/napi/host_changes_run.js <dir> -- a live JS Trainer follows what its host changes under it: loss weights, learning rates, a dataset of
// another image size, stop / start, a densify schedule switched on -- the sequence of tests/test_gpu_js_host.py, which runs the same one through the
// Python host and compares cloud and optimizer state by sha256.
'use strict';
const fs = require('fs');
const path = require('path');
const crypto = require('crypto');
const hip = require(path.join(__dirname, '..', 'ts', 'webdgs_hip.js'));
const { Trainer } = require(path.join(__dirname, '..', 'ts', 'trainer.js'));

async function main() {
  const dir = process.argv[2];
  const meta = JSON.parse(fs.readFileSync(path.join(dir, 'meta.json'), 'utf8'));
  const u8 = (f) => new Uint8Array(fs.readFileSync(path.join(dir, f)));
  const dev = new hip.HipDevice(0);
  const upload = (bytes) => { const b = dev.createBuffer({ size: bytes.byteLength }); dev.queue.writeBuffer(b, 0, bytes); return b; };
  const dataset = (tag, w, h) => {
    const cams = new Float32Array(u8(`cameras_${tag}.bin`).buffer), img = u8(`images_${tag}.bin`), size = w * h * 4, cameras = [], images = [];
    for (let v = 0; v < meta.views; v++) {
      cameras.push({ camera: cams.slice(v * 68, v * 68 + 68), width: w, height: h });
      images.push({ texture: upload(img.subarray(v * size, (v + 1) * size)), width: w, height: h });
    }
    return { cameras, images };
  };
  const a = dataset('a', meta.a[0], meta.a[1]), b = dataset('b', meta.b[0], meta.b[1]);
  const draws = meta.draws.slice();
  let drawn = 0;
  const random = () => { if (drawn >= draws.length) throw new Error('more view draws than the schedule holds'); return (draws[drawn++] + 0.5) / meta.views; };
  const t = new Trainer(dev, undefined, { random, pipelineDepth: meta.pipeline_depth });
  t.setDensifyPruneConfig({ schedule: { enabled: false } });
  t.setPointCloud({ type: 'full', num_points: meta.num_points, sh_deg: meta.sh_deg, gaussian_3d_buffer: upload(u8('gaussians.bin')), sh_buffer: upload(u8('sh.bin')) });
  t.setDataset(a.cameras, a.images);
  t.start();
  const steps = async (n) => { for (let i = 0; i < n; i++) await t.step(); };
  const log = [];
  await steps(4);
  t.setTrainingConfig({ lambda_l1: 0.6, lambda_dssim: 0.4 });
  await steps(3);
  const hp = t.getOptimizerHyperparameters();
  t.setOptimizerHyperparameters({ lr_pos: hp.lr_pos * 2, lr_color: hp.lr_color * 0.5 });
  await steps(3);
  t.setDataset(b.cameras, b.images);
  await steps(3);
  log.push(t.getIteration());
  t.stop(); t.start();
  log.push(t.getIteration());
  await steps(2);
  t.setDensifyPruneConfig(meta.densify);
  await steps(6);
  t.drain(); dev.synchronize();
  const n = t.getPointCount();
  const sha = (buf, bytes) => crypto.createHash('sha256').update(Buffer.from(dev.readBuffer(buf, bytes))).digest('hex');
  const st = t.optimizer.getStateBuffers(), rowBytes = { optPosBuffer: 48, optRotBuffer: 48, optScaleBuffer: 48, optOpacityBuffer: 12, paramSH: 192, stateSH: 384 };
  const hashes = { gaussians: sha(t.pointCloud.gaussian_3d_buffer, n * 24), sh: sha(t.pointCloud.sh_buffer, n * 96) };
  for (const k of Object.keys(rowBytes)) hashes[`state_${k}`] = sha(st[k], n * rowBytes[k]);
  // the debug helper of trainer.ts:194 -- a swap to a zero-filled cloud of another size, applied by the host at a step boundary
  const before = { iteration: t.getIteration(), last_densify: t.getLastDensifyPruneIteration() };
  t.requestResizeTo(3000);
  const req = t.consumePointCloudSwapRequest();
  t.applyPointCloudSwap(req);
  await steps(2);
  t.drain(); dev.synchronize();
  if (drawn !== draws.length) throw new Error(`drew ${drawn} views, schedule has ${draws.length}`);
  const resized = { num_points: t.getPointCount(), iteration: t.getIteration(), gaussians: sha(t.pointCloud.gaussian_3d_buffer, 3000 * 24), sh: sha(t.pointCloud.sh_buffer, 3000 * 96),
    stateSH: sha(t.optimizer.getStateBuffers().stateSH, 3000 * 384) };
  console.log(JSON.stringify({ resized, num_points: n, iteration: before.iteration, iterations_seen: log, last_densify: before.last_densify, hashes,
    training_config: t.getTrainingConfig(), lr_pos: t.getOptimizerHyperparameters().lr_pos }));
  const last = t.pointCloud; t.destroy(); last.gaussian_3d_buffer.destroy(); last.sh_buffer.destroy();
  for (const d of [a, b]) for (const im of d.images) im.texture.destroy();
  dev.destroy();
}
main().catch((e) => { console.error(e); process.exit(1); });
