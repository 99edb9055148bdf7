// oracle/make_seam_bundle.mjs — TEST INFRASTRUCTURE (run by oracle/Makefile where /root/reference exists).
// Prepares what tests/test_node_seam.py needs of the reference so that it can also run on a GPU box, where /root/reference
// does not exist: into oracle/_ref/seam/ (git-ignored build output, like the reference's WASM sorter next to it; nothing of
// the reference enters this repository's history)
//   src/...            the import closure of the reference's loader modules (SplatBuffer, the INRIA-v1 PLY parser), unmodified
//   viewer_cut.json    the TEXT of the Viewer methods that call the two seams, cut out of src/Viewer.js by name:
//                      addSplatBuffersToMesh (:1189-1228), setupSortWorker (:1235-1300), runSplatSort (:1833-1964),
//                      gatherSceneNodesForSort (:1969-2077), updateSplatMesh (:651-677) and the lines of addSplatBuffers
//                      that queue the `centers` message and start the worker (:1108-1123)
// usage: node make_seam_bundle.mjs <reference root> <out dir>
import fs from 'fs';
import path from 'path';
const [refRoot, outDir] = process.argv.slice(2);
const srcRoot = path.join(refRoot, 'src');

const cut = (src, startToken) => {      // text from startToken to the brace that closes its block
  const start = src.indexOf(startToken);
  if (start < 0) throw new Error(startToken + ' not found');
  let i = src.indexOf('{', start), depth = 0;
  for (; i < src.length; i++) {
    if (src[i] === '{') depth++;
    else if (src[i] === '}') { depth--; if (depth === 0) return src.slice(start, i + 1); }
  }
  throw new Error('unbalanced ' + startToken);
};
const between = (src, a, b) => {
  const i = src.indexOf(a), j = src.indexOf(b, i);
  if (i < 0 || j < 0) throw new Error('anchor not found: ' + a + ' ... ' + b);
  return src.slice(i, j);
};

const copied = new Set();
const copyClosure = (file) => {
  const abs = path.resolve(file);
  if (copied.has(abs)) return;
  copied.add(abs);
  const text = fs.readFileSync(abs, 'utf8');
  const rel = path.relative(srcRoot, abs);
  const dst = path.join(outDir, 'src', rel);
  fs.mkdirSync(path.dirname(dst), { recursive: true });
  fs.writeFileSync(dst, text);
  const re = /(?:import|export)\s[^'"]*?from\s*['"]([^'"]+)['"]/g;
  let m;
  while ((m = re.exec(text))) {
    if (m[1].startsWith('.')) copyClosure(path.join(path.dirname(abs), m[1]));
  }
};

fs.mkdirSync(outDir, { recursive: true });
for (const entry of ['loaders/SplatBuffer.js', 'loaders/ply/INRIAV1PlyParser.js', 'Constants.js', 'LogLevel.js']) copyClosure(path.join(srcRoot, entry));

const viewer = fs.readFileSync(path.join(srcRoot, 'Viewer.js'), 'utf8');
const out = {
  addSplatBuffersToMesh: cut(viewer, 'addSplatBuffersToMesh = function()'),
  setupSortWorker: cut(viewer, 'setupSortWorker(splatMesh)'),
  runSplatSort: cut(viewer, 'runSplatSort = function()'),
  gatherSceneNodesForSort: cut(viewer, 'gatherSceneNodesForSort = function()'),
  updateSplatMesh: cut(viewer, 'updateSplatMesh = function()'),
  // addSplatBuffers (:1108-1123): build, then queue the centres for the worker and set it up
  queueCentersAndSetupWorker: between(viewer, 'const buildResults = this.addSplatBuffersToMesh(splatBuffers, splatBufferOptions, finalBuild,',
                                      'sortWorkerSetupPromise.then(() => {'),
  source: { file: 'src/Viewer.js', sha256: null },
};
fs.writeFileSync(path.join(outDir, 'viewer_cut.json'), JSON.stringify(out, null, 1));
fs.writeFileSync(path.join(outDir, 'package.json'), JSON.stringify({ type: 'module' }));   // the reference's .js files are ES modules (its package.json says so)
console.log(JSON.stringify({ ok: true, modules: copied.size, cuts: Object.keys(out).length - 1 }));
