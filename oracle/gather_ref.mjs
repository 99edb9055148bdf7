// oracle/gather_ref.mjs — TEST INFRASTRUCTURE.  Runs the REFERENCE's own cull: the text of Viewer.gatherSceneNodesForSort
// (/root/reference/src/Viewer.js:1969-2077) cut out of the source and evaluated with a fake `this`, over a tree built by the
// reference's own worker body and node classes (src/splattree/SplatTree.js:4-271, also cut out as text: the module wants a
// browser Worker).  THREE is oracle/three_min.mjs (r160 restatement).  Records, per camera, splatRenderCount and the index
// list the Viewer would hand to the sort worker.
// usage: node gather_ref.mjs <SplatTree.js> <Viewer.js> <Constants.js> <in.bin> <cameras.json> <out.json>
//   in.bin: uint32 count, maxDepth, maxCentersPerNode, pad, then float32[4*count] (x, y, z, index)
import fs from 'fs';
import crypto from 'crypto';
import * as THREE from './three_min.mjs';
const [treePath, viewerPath, constantsPath, inPath, camsPath, outPath] = process.argv.slice(2);

const cut = (src, startToken, open = '{', close = '}') => {      // text from startToken to the brace that closes its block
  const start = src.indexOf(startToken);
  if (start < 0) throw new Error(startToken + ' not found');
  let i = src.indexOf(open, start), depth = 0;
  for (; i < src.length; i++) {
    if (src[i] === open) depth++;
    else if (src[i] === close) { depth--; if (depth === 0) return src.slice(start, i + 1); }
  }
  throw new Error('unbalanced ' + startToken);
};

const run = async () => {
  const { Constants } = await import(constantsPath);
  const treeSrc = fs.readFileSync(treePath, 'utf8');
  const classes = cut(treeSrc, 'class SplatTreeNode') + '\n' + cut(treeSrc, 'class SplatSubTree') + '\n' +
                  cut(treeSrc, 'function createSplatTreeWorker(self)');
  const lib = new Function('THREE', 'var processSplatTreeNode;\n' + classes + '\nreturn { SplatSubTree, createSplatTreeWorker };')(THREE);
  const fake = { posted: null, postMessage(m) { this.posted = m; }, onmessage: null };
  lib.createSplatTreeWorker(fake);
  const buf = fs.readFileSync(inPath);
  const hdr = new Uint32Array(buf.buffer, buf.byteOffset, 4);
  const centers = new Float32Array(buf.buffer.slice(buf.byteOffset + 16, buf.byteOffset + 16 + 16 * hdr[0]));
  fake.onmessage({ data: { process: { centers: [centers], maxDepth: hdr[1], maxCentersPerNode: hdr[2] } } });
  const subTrees = fake.posted.subTrees.map((t) => lib.SplatSubTree.convertWorkerSubTree(t, null));

  const viewerSrc = fs.readFileSync(viewerPath, 'utf8');
  const field = cut(viewerSrc, 'gatherSceneNodesForSort = function()');                  // "name = function() {...}"
  const gather = new Function('THREE', 'Constants', 'return (' + field.slice(field.indexOf('function')) + ')();')(THREE, Constants);

  const cams = JSON.parse(fs.readFileSync(camsPath, 'utf8'));
  const out = [];
  for (const c of cams) {
    const viewer = {
      getRenderDimensions(v) { v.x = c.width; v.y = c.height; },
      camera: { fov: c.fov, matrixWorld: new THREE.Matrix4().fromArray(c.matrixWorld) },
      splatMesh: { getSplatTree() { return { subTrees }; }, dynamicMode: false, matrixWorld: new THREE.Matrix4().fromArray(c.meshWorld) },
      sortWorkerIndexesToSort: new Uint32Array(hdr[0]),
    };
    const r = gather.call(viewer, !!c.gatherAll);
    const list = viewer.sortWorkerIndexesToSort.slice(0, r.splatRenderCount);
    const mv = new THREE.Matrix4().copy(viewer.camera.matrixWorld).invert().multiply(viewer.splatMesh.matrixWorld);
    const f64hex = (v) => { const b = Buffer.alloc(8); b.writeDoubleLE(v); return b.toString('hex'); };
    out.push({ splatRenderCount: r.splatRenderCount, shouldSortAll: r.shouldSortAll,
               sha256: crypto.createHash('sha256').update(Buffer.from(list.buffer, list.byteOffset, list.byteLength)).digest('hex'),
               indexes: list.length <= 20000 ? Array.from(list) : null, modelView: mv.elements.map(f64hex) });
  }
  fs.writeFileSync(outPath, JSON.stringify({ leaves: subTrees[0].nodesWithIndexes.length, cameras: out }));
  console.log(JSON.stringify({ ok: true, leaves: subTrees[0].nodesWithIndexes.length, counts: out.map((o) => o.splatRenderCount) }));
};
run().catch((e) => { console.error(String(e && e.stack || e)); process.exit(1); });
