// node bindings/napi/trainer_run.js <dir> -- drives bindings/ts/trainer.js (the reference-shaped Trainer over the N-API addon) on a
// dataset written by tests/test_gpu_napi.py, with the view draws fixed by the test, and dumps the trained cloud and optimizer state
// for a byte-for-byte comparison with the Python host's run of the same schedule.  meta.json selects the step's form: `views_per_step`
// (a batched step), `lanes`, `pipeline_depth`, `comm` ('capi': the sliced exchange through the library's RCCL communicator in a world
// of one), `keep_gradients`.  Also exercises get_prefix_scanner / get_dynamic_sorter and the pinned asynchronous read-back.
'use strict';
const fs = require('fs');
const path = require('path');
const hip = require(path.join(__dirname, '..', 'ts', 'webdgs_hip.js'));
const parallel = require(path.join(__dirname, '..', 'ts', 'parallel.js'));
const { Trainer } = require(path.join(__dirname, '..', 'ts', 'trainer.js'));

const dir = process.argv[2];
const meta = JSON.parse(fs.readFileSync(path.join(dir, 'meta.json'), 'utf8'));
const u8 = (name) => { const b = fs.readFileSync(path.join(dir, name)); return new Uint8Array(b.buffer, b.byteOffset, b.byteLength); };

async function main() {
  const dev = new hip.HipDevice(0);
  const upload = (bytes) => { const b = dev.createBuffer({ size: bytes.byteLength }); dev.queue.writeBuffer(b, 0, bytes); return b; };
  const pc = { type: 'full', num_points: meta.num_points, sh_deg: meta.sh_deg, gaussian_3d_buffer: upload(u8('gaussians.bin')), sh_buffer: upload(u8('sh.bin')) };
  const camBytes = u8('cameras.bin'), imgBytes = u8('images.bin');
  const cams = new Float32Array(camBytes.buffer.slice(camBytes.byteOffset, camBytes.byteOffset + camBytes.byteLength));
  const cameras = [], images = [];
  const imgSize = meta.width * meta.height * 4;
  for (let v = 0; v < meta.views; v++) {
    cameras.push({ camera: cams.slice(v * 68, v * 68 + 68), width: meta.width, height: meta.height });
    images.push({ texture: upload(imgBytes.subarray(v * imgSize, (v + 1) * imgSize)), width: meta.width, height: meta.height });
  }
  // the view draws, in order: views_per_step per step, plus the metric views of a step that densifies
  const draws = meta.draws.slice();
  let drawn = 0;
  const random = () => { if (drawn >= draws.length) throw new Error('more view draws than the schedule holds'); return (draws[drawn++] + 0.5) / meta.views; };
  const exchange = meta.comm === 'capi' ? new parallel.CapiExchange(dev, hip.Communicator.uniqueId(), 1, 0) : new parallel.Exchange();
  const t = new Trainer(dev, undefined, { random, viewsPerStep: meta.views_per_step || 1, lanes: meta.lanes || 0, pipelineDepth: meta.pipeline_depth || 1,
    keepGradients: !!meta.keep_gradients, exchange });
  t.setDensifyPruneConfig(meta.densify);
  if (meta.long_lists) t.longLists = meta.long_lists;   // long tile lists (csrc/longlist.h) with the test's threshold and scratch sizes
  t.setPointCloud(pc);
  t.setDataset(cameras, images);
  t.start();
  const sizes = [t.getPointCount()];
  let staleRowsSeen = false;
  for (let i = 0; i < meta.steps; i++) {
    await t.step();
    sizes.push(t.getPointCount());
  }
  t.drain();
  if (drawn !== draws.length) throw new Error(`drew ${drawn} views, schedule has ${draws.length}`);
  dev.synchronize();
  const n = t.getPointCount();
  const longLists = { settings: t.longLists, stats: t.forwardPass.longListStats() };
  // deferred SH writes: a raw device copy of the rows (no hook) is stale, a host read through the buffer is current -- no flushPointCloud() call here
  if (t.deferredSH) {
    const raw = new Uint32Array(hip.addon.copyToHost(dev.handle, t.pointCloud.sh_buffer.ptr, n * 96));
    const viaHook = new Uint32Array(dev.readBuffer(t.pointCloud.sh_buffer, n * 96));
    for (let i = 0; i < raw.length && !staleRowsSeen; i++) if (raw[i] !== viaHook[i]) staleRowsSeen = true;
  }
  // results: raw files, or (meta.hash_only: full-size runs) their sha256 in out_meta.json
  const hashes = {};
  const emit = (name, buffer, bytes) => {
    const data = Buffer.from(dev.readBuffer(buffer, bytes));
    if (meta.hash_only) hashes[name] = require('crypto').createHash('sha256').update(data).digest('hex');
    else fs.writeFileSync(path.join(dir, `out_${name}.bin`), data);
  };
  emit('gaussians', t.pointCloud.gaussian_3d_buffer, n * 24);
  emit('sh', t.pointCloud.sh_buffer, n * 96);
  const st = t.optimizer.getStateBuffers();
  const rowBytes = { optPosBuffer: 48, optRotBuffer: 48, optScaleBuffer: 48, optOpacityBuffer: 12, paramSH: 192, stateSH: 384 };
  for (const k of Object.keys(rowBytes)) emit(`state_${k}`, st[k], n * rowBytes[k]);
  if (meta.keep_gradients) emit('gradients', t.backwardPass.getGradientsBuffer(), n * 32);
  if (meta.skip_probes) {   // (full-size runs: the scanner / sorter / overflow probes below belong to the small cases)
    fs.writeFileSync(path.join(dir, 'out_meta.json'), JSON.stringify({ hashes, num_points: n, iteration: t.getIteration(), optimizer_iteration: t.optimizer.getIteration(), sizes,
      last_densify: t.getLastDensifyPruneIteration(), next_densify: t.getNextDensifyPruneIteration(), iters_per_s: t.getItersPerSec(), recorded_views: t.commandBuffers.size,
      stale_rows_seen: staleRowsSeen, exchange: exchange.name, long_lists: longLists }));
    const lastCloud = t.pointCloud;
    t.destroy(); exchange.destroy();
    lastCloud.gaussian_3d_buffer.destroy(); lastCloud.sh_buffer.destroy();
    for (const im of images) im.texture.destroy();
    dev.destroy();
    console.log('TRAINER_RUN_OK');
    return;
  }

  // ---- get_prefix_scanner / get_dynamic_sorter (prefix.ts:140, sort_dynamic.ts:252) and the pinned asynchronous read-back
  const count = 5000;
  const scanner = hip.get_prefix_scanner(count, dev);
  const vals = new Uint32Array(count); for (let i = 0; i < count; i++) vals[i] = (i * 2654435761) >>> 27;
  dev.queue.writeBuffer(scanner.input_buffer, 0, vals);
  scanner.set_count(count); scanner.scan(null);
  const scanned = new Uint32Array(dev.readBuffer(scanner.output_buffer, 4 * count));
  let run = 0, scanOk = true; for (let i = 0; i < count; i++) { if (scanned[i] !== run) scanOk = false; run = (run + vals[i]) >>> 0; }
  const stats = dev.createBuffer({ size: 16 }); dev.queue.writeBuffer(stats, 0, new Uint32Array([count, 0, 0, 0]));
  const sorter = hip.get_dynamic_sorter(count, dev, stats);
  const keys = new Uint32Array(count); for (let i = 0; i < count; i++) keys[i] = Math.imul(i ^ 0x9e37, 2246822519) >>> 8;
  const idx = new Uint32Array(count); for (let i = 0; i < count; i++) idx[i] = i;
  dev.queue.writeBuffer(sorter.ping_pong[0].sort_depths_buffer, 0, keys); dev.queue.writeBuffer(sorter.ping_pong[0].sort_indices_buffer, 0, idx);
  sorter.sort(null, 24);
  const out = sorter.ping_pong[sorter.final_out_index];
  const pinned = dev.createPinnedArrayBuffer(4 * count);
  const keyBuf = dev.createBuffer({ size: 4 * count });
  const sk = new Uint32Array(dev.readBuffer(out.sort_depths_buffer, 4 * count)), sv = new Uint32Array(dev.readBuffer(out.sort_indices_buffer, 4 * count));
  dev.queue.writeBuffer(keyBuf, 0, sk);
  const back = new Uint32Array(await dev.readBufferAsync(keyBuf, 0, pinned, 4 * count));
  let sortOk = true; for (let i = 0; i < count; i++) { if (sk[i] !== keys[sv[i]] || back[i] !== sk[i] || (i && (sk[i - 1] > sk[i] || (sk[i - 1] === sk[i] && sv[i - 1] > sv[i])))) sortOk = false; }
  sorter.destroy(); scanner.destroy(); stats.destroy(); keyBuf.destroy();

  // ---- a step whose tile-entry list overflows maxTileEntries is reported (an Error with code WDGS_E_CAPACITY out of step(): the deferred check of
  // onSubmittedWorkDone / the ticket wait) and changes nothing; the device stays usable.  Same form of the step as the run above.
  const overflow = { code: null, steps_until_error: 0, untouched: false, usable_after: false };
  {
    const fresh = () => ({ type: 'full', num_points: meta.num_points, sh_deg: meta.sh_deg, gaussian_3d_buffer: upload(u8('gaussians.bin')), sh_buffer: upload(u8('sh.bin')) });
    const make = (maxTileEntries) => {
      const tr = new Trainer(dev, undefined, { random: () => 0.1, viewsPerStep: meta.views_per_step || 1, lanes: meta.lanes || 0, pipelineDepth: meta.pipeline_depth || 1, maxTileEntries });
      tr.setDensifyPruneConfig({ schedule: { enabled: false } });
      tr.setPointCloud(fresh()); tr.setDataset(cameras, images); tr.start();
      return tr;
    };
    const small = make(4096);   // the scene needs several times that
    const before = Buffer.from(dev.readBuffer(small.pointCloud.gaussian_3d_buffer, meta.num_points * 24));
    try {
      for (let i = 0; i < 3; i++) { await small.step(); overflow.steps_until_error++; }
    } catch (e) { overflow.code = e.code || String(e); }
    try { dev.synchronize(); } catch (_e) { /* the step that was still in flight overflowed too */ }
    overflow.untouched = before.equals(Buffer.from(dev.readBuffer(small.pointCloud.gaussian_3d_buffer, meta.num_points * 24)));
    const sp = small.pointCloud; small.destroy(); sp.gaussian_3d_buffer.destroy(); sp.sh_buffer.destroy();
    const roomy = make(0);
    await roomy.step(); await roomy.step(); await roomy.step(); roomy.drain(); dev.synchronize();
    overflow.usable_after = roomy.getIteration() === 3 && !before.equals(Buffer.from(dev.readBuffer(roomy.pointCloud.gaussian_3d_buffer, meta.num_points * 24)));
    const rp = roomy.pointCloud; roomy.destroy(); rp.gaussian_3d_buffer.destroy(); rp.sh_buffer.destroy();
  }

  fs.writeFileSync(path.join(dir, 'out_meta.json'), JSON.stringify({ overflow, hashes, num_points: n, iteration: t.getIteration(), optimizer_iteration: t.optimizer.getIteration(), sizes,
    last_densify: t.getLastDensifyPruneIteration(), next_densify: t.getNextDensifyPruneIteration(), iters_per_s: t.getItersPerSec(), scan_ok: scanOk, sort_ok: sortOk,
    recorded_views: t.commandBuffers.size, recorded_keys: Array.from(t.commandBuffers.keys()), lanes: t.lanes, op_sets: t.opSets, stale_rows_seen: staleRowsSeen,
    exchange: exchange.name, long_lists: longLists }));
  const last = t.pointCloud;
  t.destroy();
  exchange.destroy();
  last.gaussian_3d_buffer.destroy(); last.sh_buffer.destroy();
  for (const im of images) im.texture.destroy();
  dev.destroy();
  console.log('TRAINER_RUN_OK');
}
main().catch((e) => { console.error(e); process.exit(1); });
