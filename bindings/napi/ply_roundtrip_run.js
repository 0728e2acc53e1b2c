// node bindings/napi/ply_roundtrip_run.js <file.ply> -- loadPointCloud + exportPly of a (large) .ply on the TypeScript-side host, no GPU: prints one JSON
// line with the cloud's header fields and the sha256 of the packed Gaussians, the packed SH and the re-exported file (tests/test_gpu_fullsize.py
// compares them with the Python host's at c5's 5 M Gaussians / 1.2 GB).
'use strict';
const fs = require('fs');
const path = require('path');
const crypto = require('crypto');
const loaders = require(path.join(__dirname, '..', 'ts', 'loaders.js'));
const sha = (typed) => crypto.createHash('sha256').update(Buffer.from(typed.buffer, typed.byteOffset, typed.byteLength)).digest('hex');
const t0 = Date.now();
const pc = loaders.loadPointCloud(fs.readFileSync(process.argv[2]));
const t1 = Date.now();
const again = loaders.exportPly(pc.gaussians, pc.sh, pc.sh_deg);
const t2 = Date.now();
console.log(JSON.stringify({ type: pc.type, num_points: pc.num_points, sh_deg: pc.sh_deg, gaussians: sha(pc.gaussians), sh: sha(pc.sh), exported: sha(again),
  exported_bytes: again.byteLength, load_ms: t1 - t0, export_ms: t2 - t1 }));
