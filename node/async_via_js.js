// Test driver: the sort seam is asynchronous like the reference's Web Worker.  Posts two sorts back to back (different
// cameras) plus a second {centers} message while the first sort is in flight; checks that postMessage returns before any
// sortDone, that replies arrive in posting order and that a draw-side call (a second worker's sort) issued meanwhile works.
// usage: node async_via_js.js <in.bin> <outA.bin> <outB.bin>      (input format of oracle/wasm_ref.js; static integer)
'use strict';
const fs = require('fs');
const gs = require('./gsplat.js');
const [inPath, outA, outB] = process.argv.slice(2);
const buf = fs.readFileSync(inPath);
const ab = buf.buffer.slice(buf.byteOffset, buf.byteOffset + buf.byteLength);
const [n, renderCount, sortCount] = new Uint32Array(ab, 0, 8);
let off = 32;
const take = (bytes) => { const b = ab.slice(off, off + bytes); off += bytes; return b; };
const indexes = new Uint32Array(take(4 * renderCount)), centers = take(16 * n), mvpA = new Float32Array(take(64));
const mvpB = Float32Array.from(mvpA); mvpB[2] = -mvpA[2]; mvpB[6] = mvpA[10]; mvpB[10] = mvpA[6];   // another view direction
const worker = gs.createSortWorker(n, false, true, true, false, 16);
const events = [];
let repliesAtReturn = -1, replies = 0;
worker.onmessage = (e) => {
  if (e.data.sortSetupPhase1Complete) {
    worker.postMessage({ centers: centers, sceneIndexes: null, range: { from: 0, to: n - 1, count: n } });
    const sort = (m) => ({ sort: { modelViewProj: Array.from(m), splatRenderCount: renderCount, splatSortCount: sortCount,
                                   usePrecomputedDistances: false, indexesToSort: indexes, transforms: null } });
    worker.postMessage(sort(mvpA));
    worker.postMessage({ centers: centers, sceneIndexes: null, range: { from: 0, to: n - 1, count: n } });   // queued behind sort A
    worker.postMessage(sort(mvpB));
    repliesAtReturn = replies;                      // both postMessage calls have returned: no reply may have fired yet
    events.push('posted');
  } else if (e.data.sortDone) {
    replies++;
    events.push('sortDone' + replies);
    fs.writeFileSync(replies === 1 ? outA : outB, Buffer.from(e.data.sortedIndexes.buffer, e.data.sortedIndexes.byteOffset, e.data.sortedIndexes.byteLength));
    if (replies === 2) {
      worker.terminate();
      console.log(JSON.stringify({ repliesAtReturn, events, mvpB: Array.from(mvpB) }));
    }
  }
};
