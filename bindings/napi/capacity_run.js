// node bindings/napi/capacity_run.js <depth> [views per step] -- a cloud of few, large splats outruns the tile-entry lists the library sized for it: the JS Trainer
// (maxTileEntries left at 0) doubles them, warns and goes on training (tests/test_gpu_js_host.py compares the outcome with the Python host's).
'use strict';
const path = require('path');
const crypto = require('crypto');
const hip = require(path.join(__dirname, '..', 'ts', 'webdgs_hip.js'));
const synth = require(path.join(__dirname, '..', 'ts', 'synth.js'));
const { Trainer } = require(path.join(__dirname, '..', 'ts', 'trainer.js'));

async function main() {
  const depth = Number(process.argv[2] || 1), vpr = Number(process.argv[3] || 1);
  const cfg = { config_id: 2, num_points: 6000, width: 512, height: 384, sh_deg: 1, fy: 550.0, s0: 0.2, name: 'few-large-splats' };
  const dev = new hip.HipDevice(0);
  const made = synth.makeGaussians(cfg), g = made.gaussians, sh = made.sh;
  const cams = synth.circleCameras(cfg, 2);
  const upload = (typed) => { const b = dev.createBuffer({ size: typed.byteLength }); dev.queue.writeBuffer(b, 0, typed); return b; };
  const pc = { type: 'full', num_points: cfg.num_points, sh_deg: cfg.sh_deg, gaussian_3d_buffer: upload(g), sh_buffer: upload(sh) };
  const black = new Uint8Array(cfg.width * cfg.height * 4);
  const images = cams.map(() => ({ texture: upload(black), width: cfg.width, height: cfg.height }));
  // a Viewer on the same kind of cloud: its first frame outruns the lists the library sized for it; readFrame() rebuilds the passes and renders again
  const { Viewer } = require(path.join(__dirname, '..', 'ts', 'viewer.js'));
  const vpc = { type: 'full', num_points: cfg.num_points, sh_deg: cfg.sh_deg, gaussian_3d_buffer: upload(g), sh_buffer: upload(sh) };
  const viewer = new Viewer(dev, null, { width: cfg.width, height: cfg.height });
  viewer.setPointCloud(vpc);
  viewer.setRenderMode('gaussian');
  dev.queue.writeBuffer(viewer.camera.uniform_buffer, 0, cams[0]);
  viewer.render(null);
  const frame = viewer.readFrame();
  const viewerOut = { frame: crypto.createHash('sha256').update(Buffer.from(frame.buffer, frame.byteOffset, frame.byteLength)).digest('hex'),
    cap: viewer.getForwardPass().getResources().maxTileEntries };
  viewer.forwardPass.destroy(); viewer.rasterizer.destroy(); vpc.gaussian_3d_buffer.destroy(); vpc.sh_buffer.destroy();
  try { dev.synchronize(); } catch (_e) { /* (nothing of the viewer's is left to report) */ }
  const warnings = [];
  const warn = console.warn; console.warn = (m) => warnings.push(String(m));
  const t = new Trainer(dev, undefined, { pipelineDepth: depth, viewsPerStep: vpr, lanes: vpr > 1 ? 2 : 0 });
  t.setDensifyPruneConfig({ schedule: { enabled: false } });
  t.setPointCloud(pc);
  t.setDataset(cams.map((c) => ({ camera: c, width: cfg.width, height: cfg.height })), images);
  t.start();
  for (let i = 0; i < 8; i++) { const ids = []; for (let k = 0; k < vpr; k++) ids.push((i + k) % 2); await t.stepViews(ids); }
  t.drain(); dev.synchronize();
  console.warn = warn;
  const cap = t.forwardPass.getResources().maxTileEntries;
  const needed = t.forwardPass.check().totalTileEntries;
  const n = t.getPointCount();
  const sha = (buf, bytes) => crypto.createHash('sha256').update(Buffer.from(dev.readBuffer(buf, bytes))).digest('hex');
  console.log(JSON.stringify({ viewer: viewerOut, grown: warnings.filter((w) => w.indexOf('tile-entry lists grown') >= 0).length, cap, needed, iteration: t.getIteration(),
    gaussians: sha(t.pointCloud.gaussian_3d_buffer, n * 24), sh: sha(t.pointCloud.sh_buffer, n * 96) }));
  const last = t.pointCloud; t.destroy(); last.gaussian_3d_buffer.destroy(); last.sh_buffer.destroy();
  for (const im of images) im.texture.destroy();
  dev.destroy();
}
main().catch((e) => { console.error(e); process.exit(1); });
