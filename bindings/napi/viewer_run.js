// node bindings/napi/viewer_run.js <dir> -- the TypeScript-side loaders, image ingest and Viewer on the GPU (tests/test_gpu_js_host.py): a .ply loaded
// by bindings/ts/loaders.js is rendered by bindings/ts/viewer.js in both render modes and on a resized canvas; then a Trainer trains the SAME
// cloud for a few steps on PNG ground truth and camera JSON loaded here, and the viewer -- which shares the PointCloud with the trainer as in
// the reference (main.ts:389, 524) and is never told about deferred SH writes -- renders it again.  Every frame goes back as raw rgba8.
'use strict';
const fs = require('fs');
const path = require('path');
const ts = (m) => require(path.join(__dirname, '..', 'ts', m));
const hip = ts('webdgs_hip.js'), loaders = ts('loaders.js'), images = ts('images.js');
const { Viewer } = ts('viewer.js'), { Trainer } = ts('trainer.js');

const dir = process.argv[2];
const meta = JSON.parse(fs.readFileSync(path.join(dir, 'meta.json'), 'utf8'));

async function main() {
  const dev = new hip.HipDevice(0);
  // meta.cloud_file: a .ply or COLMAP's points3D.bin; meta.camera_files: a camera JSON, or COLMAP's images.bin + cameras.bin (merged by loadCamera)
  const pc = loaders.loadPointCloud(fs.readFileSync(path.join(dir, meta.cloud_file || 'scene.ply')), dev);
  const cams = loaders.loadCamera((meta.camera_files || ['cams.json']).map((f) => ({ name: f, data: fs.readFileSync(path.join(dir, f)) })));
  const canvas = { width: meta.width, height: meta.height };
  const viewer = new Viewer(dev, null, canvas, 'rgba8unorm');
  viewer.setPointCloud(pc);
  viewer.camera.set_preset(cams[meta.view_camera]);
  const frame = (name) => { viewer.render(dev.createCommandEncoder()); fs.writeFileSync(path.join(dir, name), Buffer.from(viewer.readFrame().buffer)); };
  frame('out_frame_points.rgba');                     // the reference's viewer starts in 'pointcloud' mode (viewer.ts:51-55)
  viewer.setRenderMode('gaussian');
  frame('out_frame_gaussian.rgba');
  viewer.setPointSize(5); viewer.setRenderMode('pointcloud');
  frame('out_frame_points5.rgba');
  viewer.resize(meta.resized[0], meta.resized[1]);
  viewer.setRenderMode('gaussian');
  frame('out_frame_resized.rgba');
  viewer.resize(meta.width, meta.height);
  viewer.render(dev.createCommandEncoder());          // (a resized canvas gets a fresh swap-chain image)
  viewer.savePNG(path.join(dir, 'out_frame.png'));

  // ---- train the cloud the viewer shows; dataset in the loaders' own shapes (CameraData + LoadedImage)
  const loaded = images.loadImages(fs.readdirSync(path.join(dir, 'gt')).map((f) => path.join(dir, 'gt', f)), dev);
  const draws = meta.draws.slice();
  let drawn = 0;
  const t = new Trainer(dev, undefined, { random: () => (draws[drawn++] + 0.5) / cams.length });
  t.setDensifyPruneConfig({ schedule: { enabled: false } });
  t.setPointCloud(pc);
  t.setDataset(cams, loaded);
  t.start();
  for (let i = 0; i < meta.steps; i++) await t.step();
  frame('out_frame_trained.rgba');                    // no flushPointCloud(): the viewer's pass follows pointCloud.dcWords by itself
  const staleRows = new Uint32Array(hip.addon.copyToHost(dev.handle, pc.sh_buffer.ptr, pc.num_points * 96));
  const rows = new Uint32Array(dev.readBuffer(pc.sh_buffer, pc.num_points * 96));
  let stale = false; for (let i = 0; i < rows.length && !stale; i++) stale = rows[i] !== staleRows[i];
  fs.writeFileSync(path.join(dir, 'out_gaussians.bin'), Buffer.from(dev.readBuffer(pc.gaussian_3d_buffer, pc.num_points * 24)));
  fs.writeFileSync(path.join(dir, 'out_sh.bin'), Buffer.from(rows.buffer));
  // a state handle kept across steps is current when it is read (ADVICE r3): fetched BEFORE two more steps, read after them
  const kept = t.optimizer.getStateBuffers();
  for (let i = 0; i < 2; i++) await t.step();
  fs.writeFileSync(path.join(dir, 'out_state_pos_kept.bin'), Buffer.from(dev.readBuffer(kept.optPosBuffer, pc.num_points * 48)));
  t.destroy();                                        // the optimizer goes, the cloud stays: its rows are left current
  frame('out_frame_after_trainer.rgba');
  fs.writeFileSync(path.join(dir, 'out.json'), JSON.stringify({ type: pc.type, num_points: pc.num_points, sh_deg: pc.sh_deg, images: loaded.map((l) => [l.name, l.width, l.height]),
    cameras: cams.length, stale_rows_seen: stale, dc_words_after_trainer: pc.dcWords === null, png_bytes: fs.statSync(path.join(dir, 'out_frame.png')).size }));
  viewer.destroy();
  for (const l of loaded) l.texture.destroy();
  pc.gaussian_3d_buffer.destroy(); pc.sh_buffer.destroy();
  dev.destroy();
  console.log('VIEWER_RUN_OK');
}
main().catch((e) => { console.error(e); process.exit(1); });
