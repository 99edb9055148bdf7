// Runs one sort through the JS shim's createSortWorker protocol (test driver; same input file format as
// oracle/wasm_ref.js).  usage: node sort_via_js.js <in.bin> <out.bin> [shared|cull]
'use strict';
const fs = require('fs');
const gs = require('./gsplat.js');
const [inPath, outPath, sharedArg] = process.argv.slice(2);
const shared = sharedArg === 'shared', cull = sharedArg === 'cull';
const buf = fs.readFileSync(inPath);
const ab = buf.buffer.slice(buf.byteOffset, buf.byteOffset + buf.byteLength);
const [n, renderCount, sortCount, range, useInt, dynamic, usePre] = new Uint32Array(ab, 0, 8);
let off = 32;
const take = (bytes) => { const b = ab.slice(off, off + bytes); off += bytes; return b; };
const indexes = new Uint32Array(take(4 * renderCount)), centers = take(16 * n), mvp = new Float32Array(take(64));
const sceneIdx = dynamic ? take(4 * n) : null, transforms = dynamic ? new Float32Array(take(32 * 64)) : null;
const pre = usePre ? take(4 * n) : null;
const precision = Math.round(Math.log2(range));
const worker = gs.createSortWorker(n, shared, true, !!useInt, !!dynamic, precision);
if (cull) worker.setFrustumCull(true);
worker.onmessage = (e) => {
  if (e.data.sortSetupPhase1Complete) {
    worker.postMessage({ centers: centers, sceneIndexes: sceneIdx, range: { from: 0, to: n - 1, count: n } });
    const sort = { modelViewProj: Array.from(mvp), splatRenderCount: renderCount, splatSortCount: sortCount, usePrecomputedDistances: !!usePre };
    if (shared) {
      new Uint32Array(e.data.indexesToSortBuffer, e.data.indexesToSortOffset, renderCount).set(indexes);
      if (transforms) new Float32Array(e.data.transformsBuffer, e.data.transformsOffset, 32 * 16).set(transforms);
      if (pre) new Uint8Array(e.data.precomputedDistancesBuffer).set(new Uint8Array(pre));
    } else {
      sort.indexesToSort = indexes; sort.transforms = transforms;
      if (pre) sort.precomputedDistances = useInt ? new Int32Array(pre) : new Float32Array(pre);
    }
    worker.postMessage({ sort });
  } else if (e.data.sortDone) {
    const out = shared ? new Uint32Array(worker.sortedIndexesBuffer, 0, e.data.splatRenderCount) : e.data.sortedIndexes;
    fs.writeFileSync(outPath, Buffer.from(out.buffer, out.byteOffset, out.byteLength));
    console.log(JSON.stringify({ sortTime: e.data.sortTime, status: e.data.status, splatRenderCount: e.data.splatRenderCount }));
    worker.terminate();
  }
};
