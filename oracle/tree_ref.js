// oracle/tree_ref.js — TEST INFRASTRUCTURE.  Runs the REFERENCE's own octree builder: the body of
// createSplatTreeWorker() in /root/reference/src/splattree/SplatTree.js:82-271 is three-free, so it is cut out of
// the source TEXT (the module itself imports 'three', which is not installed) and evaluated with a fake `self`.
// usage: node tree_ref.js <SplatTree.js> <in.bin> <out.json>
//   in.bin: uint32 count, uint32 maxDepth, uint32 maxCentersPerNode, uint32 pad, then float32[4*count] (x,y,z,index)
'use strict';
const fs = require('fs');
const [srcPath, inPath, outPath] = process.argv.slice(2);
const src = fs.readFileSync(srcPath, 'utf8');
const start = src.indexOf('function createSplatTreeWorker(self)');
if (start < 0) throw new Error('createSplatTreeWorker not found');
let i = src.indexOf('{', start), depth = 0, end = -1;
for (; i < src.length; i++) {
  if (src[i] === '{') depth++;
  else if (src[i] === '}') { depth--; if (depth === 0) { end = i + 1; break; } }
}
const fnText = src.slice(start, end);
const fake = { posted: null, postMessage(m) { this.posted = m; }, onmessage: null };
// the worker body assigns `processSplatTreeNode = function...` without declaring it: give it a binding
const make = new Function('self', 'var processSplatTreeNode;\n' + fnText + '\ncreateSplatTreeWorker(self);');
make(fake);
const buf = fs.readFileSync(inPath);
const hdr = new Uint32Array(buf.buffer, buf.byteOffset, 4);
const centers = new Float32Array(buf.buffer.slice(buf.byteOffset + 16, buf.byteOffset + 16 + 16 * hdr[0]));
fake.onmessage({ data: { process: { centers: [centers], maxDepth: hdr[1], maxCentersPerNode: hdr[2] } } });
const sub = fake.posted.subTrees[0];
// SplatSubTree.convertWorkerSubTree (SplatTree.js:55-79): leaves in DFS order, only those holding indexes
const leaves = [];
let leafCount = 0;
(function visit(node) {
  if (node.children.length === 0) {
    leafCount++;
    if (node.data && node.data.indexes && node.data.indexes.length > 0) {
      leaves.push({ min: node.min, max: node.max, center: node.center, depth: node.depth, indexes: node.data.indexes });
    }
  }
  for (const c of node.children) visit(c);
})(sub.rootNode);
// doubles are written as hex of their IEEE bits so the comparison is exact
const f64hex = (v) => { const b = Buffer.alloc(8); b.writeDoubleLE(v); return b.toString('hex'); };
fs.writeFileSync(outPath, JSON.stringify({
  sceneMin: sub.sceneMin, sceneMax: sub.sceneMax, leafCount: leafCount,
  leaves: leaves.map((l) => ({ min: l.min.map(f64hex), max: l.max.map(f64hex), center: l.center.map(f64hex),
                               depth: l.depth, indexes: l.indexes })) }));
console.log(JSON.stringify({ leaves: leaves.length, allLeaves: leafCount }));
