'use strict';
/*
 * parallel.js (+ parallel.d.ts) -- view-sharded data parallelism for the TypeScript-side host (SURVEY.md section 8(e)); the reference has
 * none (batch 1, one GPUDevice: src/trainer.ts:573).  Counterpart of webdgs_amd/parallel.py without torch: one node process per GPU,
 * every rank a full replica of the point cloud, the exchange driven through the library's own communicator (wdgs_comm_*: RCCL
 * queued on the device's stream, no host synchronisation inside a step):
 *
 *     reduce-scatter (sum) of the step's fp32 gradient block + visibility counts   -> rank r holds the sums for ITS slice
 *     Adam + re-pack on the owned slice only                                       -> 1 / world of the optimizer pass per rank
 *     all-gather of the re-packed 32-byte rows                                     -> every replica's point cloud is current again
 *
 * The 128-byte RCCL id travels from rank 0 to the others through a file (`rendezvous`): a launcher -- bench.js --gpus N -- gives every
 * rank the same path in WDGS_RENDEZVOUS; the rank, world size and device ordinal come from RANK / WORLD_SIZE / LOCAL_RANK, the names
 * torch.distributed.run uses.
 */
const fs = require('fs');
const hip = require('./webdgs_hip.js');

const GRAD_FLOATS = 14;   // pos3, opacity, rot4, log-sigma3, rgb3 (GaussianGradient component order)

/** Gaussians per rank: ceil(N / world) rounded up to 64; rank r owns [r * slice, min((r + 1) * slice, N)). */
function slicePoints(numPoints, world) {
  const per = Math.floor((Math.max(numPoints, 1) + world - 1) / world);
  return Math.floor((per + 63) / 64) * 64;
}
/** { first, count } of the slice `rank` owns. */
function ownedRange(numPoints, world, rank) {
  const sl = slicePoints(numPoints, world);
  const first = Math.min(rank * sl, numPoints);
  return { first, count: Math.max(0, Math.min((rank + 1) * sl, numPoints) - first) };
}
/** The views of one global batch that `rank` processes: a strided split (views rank, rank + world, ...). */
function shardViews(viewIds, rank, world) { return viewIds.filter((_v, i) => i % world === rank); }

/** What the Trainer needs from a communicator; this base class is the world of one (every call an identity). */
class Exchange {
  constructor() { this.worldSize = 1; this.rank = 0; this.name = 'none'; this.force = false; }
  exchangeGradients(_grad, _visible, _flag, _slicePoints) {}
  allgatherRows(_rows, _slicePoints) {}
  broadcast(_ptr, _bytes, _root) {}
  allreduceCounts(_counts, _count) {}
  destroy() {}
}

/** The library's communicator as the transport.  `force` (the default here: the communicator exists) issues the collectives even in a world
 *  of one, where they are identities -- which exercises RCCL and the sliced step on a one-GPU box. */
class CapiExchange extends Exchange {
  constructor(device, uniqueId, worldSize, rank) {
    super();
    this.comm = new hip.Communicator(device, uniqueId, worldSize, rank);
    this.worldSize = worldSize; this.rank = rank; this.force = true;
    this.name = 'wdgs_comm (RCCL on the device stream)';
  }
  exchangeGradients(grad, visible, flag, slicePts) { this.comm.exchangeGradients(grad, visible, flag, slicePts); }
  allgatherRows(rows, slicePts) { this.comm.allgatherRows(rows, slicePts); }
  broadcast(ptr, bytes, root) { if (bytes > 0) this.comm.broadcast(ptr, bytes, root); }
  allreduceCounts(counts, count) { if (count > 0) this.comm.allreduceCounts(counts, count); }
  destroy() { this.comm.destroy(); }
}

function sleepMs(ms) { Atomics.wait(new Int32Array(new SharedArrayBuffer(4)), 0, 0, ms); }

/** Rank 0 creates the id and publishes it (write to a temporary name, then rename: a reader never sees half a file); the others wait for it. */
function rendezvousId(file, rank, timeoutMs) {
  if (rank === 0) {
    const id = hip.Communicator.uniqueId();
    fs.writeFileSync(file + '.tmp', Buffer.from(id));
    fs.renameSync(file + '.tmp', file);
    return id;
  }
  const deadline = Date.now() + (timeoutMs || 120000);
  while (Date.now() < deadline) {
    if (fs.existsSync(file)) {
      const b = fs.readFileSync(file);
      if (b.length === 128) return b.buffer.slice(b.byteOffset, b.byteOffset + 128);
    }
    sleepMs(20);
  }
  throw new Error(`rank ${rank}: no RCCL id at ${file} after ${timeoutMs || 120000} ms (is rank 0 running?)`);
}

/** { rank, world, localRank } from RANK / WORLD_SIZE / LOCAL_RANK (1 process = 1 GPU). */
function envRanks() {
  const world = parseInt(process.env.WORLD_SIZE || '1', 10), rank = parseInt(process.env.RANK || '0', 10);
  return { rank, world, localRank: parseInt(process.env.LOCAL_RANK || String(rank), 10) };
}

/** The exchange of this process: none in a world of one (unless WDGS_COMM=capi asks for the communicator anyway), else CapiExchange over the
 *  id found at WDGS_RENDEZVOUS. */
function defaultExchange(device) {
  const { rank, world } = envRanks();
  if (world <= 1 && process.env.WDGS_COMM !== 'capi') return new Exchange();
  if (world <= 1) return new CapiExchange(device, hip.Communicator.uniqueId(), 1, 0);
  const file = process.env.WDGS_RENDEZVOUS;
  if (!file) throw new Error('WORLD_SIZE > 1 needs WDGS_RENDEZVOUS (a path every rank can read: rank 0 publishes the RCCL id there)');
  return new CapiExchange(device, rendezvousId(file, rank), world, rank);
}

module.exports = { GRAD_FLOATS, slicePoints, ownedRange, shardViews, Exchange, CapiExchange, rendezvousId, envRanks, defaultExchange };
