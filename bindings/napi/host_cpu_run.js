// node bindings/napi/host_cpu_run.js <dir> -- the TypeScript-side host's loaders, camera, image ingest and scene generator run WITHOUT a GPU
// (none of these modules loads the addon) on the files tests/test_js_host_cpu.py wrote into <dir>; every result goes back as raw bytes /
// JSON for a bit-for-bit comparison with the Python host (webdgs_amd/loaders.py, images.py, synth.py).
'use strict';
const fs = require('fs');
const path = require('path');
const ts = (m) => require(path.join(__dirname, '..', 'ts', m));
const loaders = ts('loaders.js'), images = ts('images.js'), synth = ts('synth.js'), camera = ts('camera.js'), parallel = null;

const dir = process.argv[2];
const meta = JSON.parse(fs.readFileSync(path.join(dir, 'meta.json'), 'utf8'));
const rd = (f) => fs.readFileSync(path.join(dir, f));
const wr = (f, typed) => fs.writeFileSync(path.join(dir, f), Buffer.from(typed.buffer, typed.byteOffset, typed.byteLength));
const out = { clouds: {}, errors: {} };

for (const name of meta.clouds) {   // PLY (full / normal) and COLMAP points3D.bin through loadPointCloud
  const pc = loaders.loadPointCloud(rd(name));
  out.clouds[name] = { type: pc.type, num_points: pc.num_points, sh_deg: pc.sh_deg };
  wr(`out_${name}.g`, pc.gaussians); wr(`out_${name}.sh`, pc.sh);
}
for (const name of Object.keys(meta.bad_clouds || {})) {
  try { loaders.loadPointCloud(rd(name)); out.errors[name] = null; } catch (e) { out.errors[name] = e.message; }
}
// plyreader.ts surface: decodeHeader -> [count, types, DataView]; readRawVertex walks the payload
{
  const [count, types, view] = loaders.decodeHeader(rd(meta.header_ply));
  const first = loaders.readRawVertex(0, view, types);
  out.header = { count, names: Object.keys(types), types, first_offset: first[0], first_vertex: first[1], nsh: [0, 1, 2, 3].map(loaders.nShCoeffs) };
  try { loaders.nShCoeffs(0.7320508075688772); } catch (e) { out.header.nsh_error = e.message; }
}
// export -> load round trip on the host
{
  const pc = loaders.loadPointCloud(rd(meta.clouds[0]));
  const again = loaders.loadPointCloud(loaders.exportPly(pc.gaussians, pc.sh, pc.sh_deg));
  out.export_round_trip = Buffer.from(again.gaussians.buffer).equals(Buffer.from(pc.gaussians.buffer)) && Buffer.from(again.sh.buffer).equals(Buffer.from(pc.sh.buffer));
  fs.writeFileSync(path.join(dir, 'out_export.ply'), loaders.exportPly(pc.gaussians, pc.sh, pc.sh_deg));
}
// cameras: JSON, images.bin + cameras.bin merged, each alone; the 68-float block of every camera on two canvas sizes
{
  const plain = (c) => Object.assign({}, c, { position: c.position ? Array.from(c.position) : undefined, rotation: c.rotation ? Array.from(c.rotation) : undefined });
  const json = loaders.loadCamera([{ name: 'cams.json', data: rd('cams.json') }]);
  const merged = loaders.loadCamera([{ name: 'sparse/0/images.bin', data: rd('images.bin') }, { name: 'sparse/0/cameras.bin', data: rd('cameras.bin') }]);
  const onlyImages = loaders.loadCamera([{ name: 'images.bin', data: rd('images.bin') }]);
  const onlyCameras = loaders.loadCamera({ name: 'cameras.bin', data: rd('cameras.bin') });
  out.cameras = { json: json.map(plain), merged: merged.map(plain), only_images: onlyImages.length, only_cameras: onlyCameras.map(plain) };
  const blocks = [];
  for (const c of json.concat(merged)) { blocks.push(loaders.cameraUniforms(c)); blocks.push(loaders.cameraUniforms(c, 333, 201)); }
  blocks.push(loaders.cameraUniforms({}, 200, 100));   // no pose, no intrinsics: Camera defaults
  wr('out_blocks.f32', Float32Array.from([].concat.apply([], blocks.map((b) => Array.from(b)))));
  // the Camera class over a stub device: set_preset -> on_update_canvas -> update_buffer writes the same block
  const written = [];
  const dev = { createBuffer: (d) => ({ size: d.size, destroy() {} }), queue: { writeBuffer: (_b, _o, data) => written.push(Float32Array.from(data)) } };
  const cam = new camera.Camera({ width: 333, height: 201 }, dev);
  cam.set_preset(json[0]);
  out.camera_class = { writes: written.length, equals_uniforms: Buffer.from(written[written.length - 1].buffer).equals(Buffer.from(loaders.cameraUniforms(json[0], 333, 201).buffer)),
    default_focal: written[0][67], look: Array.from(cam.look) };
  out.presets = camera.load_camera_presets(rd('cams.json')).map((p) => ({ position: Array.from(p.position), rotation: Array.from(p.rotation) }));
  try { loaders.loadCamera([{ name: 'x.bin', data: Buffer.from([1, 2, 3]) }]); } catch (e) { out.errors.camera = e.message; }
}
// images: filter, order, decode, drop
{
  const files = fs.readdirSync(path.join(dir, 'images')).map((f) => path.join(dir, 'images', f));
  const loaded = images.loadImages(files, null);
  out.images = loaded.map((im) => ({ name: im.name, width: im.width, height: im.height }));
  loaded.forEach((im, i) => wr(`out_image_${i}.rgba`, im.bitmap.data));
  const again = images.decodePNG(images.encodePNG(loaded[0].bitmap.data, loaded[0].width, loaded[0].height));
  out.png_round_trip = Buffer.from(again.data.buffer).equals(Buffer.from(loaded[0].bitmap.data.buffer));
}
// the synthetic scene generator
for (const s of meta.synth) {
  const cfg = Object.assign({}, synth.CONFIGS[s.config], { num_points: s.points });
  const sc = synth.makeGaussians(cfg), tg = synth.makeTargetScene(sc.gaussians, sc.sh);
  wr(`out_synth_${s.config}.g`, sc.gaussians); wr(`out_synth_${s.config}.sh`, sc.sh); wr(`out_synth_${s.config}.tg`, tg.gaussians); wr(`out_synth_${s.config}.tsh`, tg.sh);
  wr(`out_synth_${s.config}.cams`, Float32Array.from([].concat.apply([], synth.circleCameras(cfg, s.cameras).concat([synth.identityCamera(cfg)]).map((b) => Array.from(b)))));
}
// the slice arithmetic of the data-parallel step
{
  const p = ts('parallel.js');
  out.slices = meta.slices.map(([n, w]) => ({ n, w, slice: p.slicePoints(n, w), owned: Array.from({ length: w }, (_x, r) => p.ownedRange(n, w, r)) }));
  out.shard = p.shardViews([5, 6, 7, 8, 9], 1, 2);
}
fs.writeFileSync(path.join(dir, 'out.json'), JSON.stringify(out));
console.log('HOST_CPU_RUN_OK');
