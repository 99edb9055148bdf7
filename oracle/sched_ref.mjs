// oracle/sched_ref.mjs — TEST INFRASTRUCTURE.  Runs the REFERENCE's own sort trigger: the text of Viewer.runSplatSort
// (/root/reference/src/Viewer.js:1833-1964) cut out of the source and evaluated with a fake `this` (camera, splat mesh, a
// scripted gatherSceneNodesForSort, a sort worker that records what is posted).  THREE is oracle/three_min.mjs (r160
// restatement).  Per scripted step it records whether a sort was posted and the message's splatSortCount / splatRenderCount /
// modelViewProj.   usage: node sched_ref.mjs <Viewer.js> <script.json> <out.json>
import fs from 'fs';
import * as THREE from './three_min.mjs';
const [viewerPath, scriptPath, outPath] = process.argv.slice(2);

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

const run = async () => {
  const viewerSrc = fs.readFileSync(viewerPath, 'utf8');
  const field = cut(viewerSrc, 'runSplatSort = function()');                  // "name = function() {...}" + "()"
  const f64hex = (v) => { const b = Buffer.alloc(8); b.writeDoubleLE(v); return b.toString('hex'); };
  const out = [];
  for (const sc of JSON.parse(fs.readFileSync(scriptPath, 'utf8'))) {
    const runSplatSort = new Function('THREE', 'return (' + field.slice(field.indexOf('function')) + ')();')(THREE);   // fresh closure state
    const posted = [];
    const viewer = {
      initialized: true, sortRunning: false, gpuAcceleratedSort: false, sharedMemoryForWorkers: true, preSortMessages: [],
      perspectiveCamera: null,
      camera: { quaternion: new THREE.Quaternion(), position: new THREE.Vector3(), matrixWorld: new THREE.Matrix4(), projectionMatrix: new THREE.Matrix4() },
      splatMesh: { getSplatCount() { return sc.splatCount; }, dynamicMode: !!sc.dynamicMode, matrixWorld: new THREE.Matrix4().fromArray(sc.meshWorld),
                   fillTransformsArray() {} },
      next: null,
      gatherSceneNodesForSort() { return { splatRenderCount: this.next.splatRenderCount, shouldSortAll: !!this.next.shouldSortAll }; },
      sortWorker: { postMessage(m) { posted.push(m); } },
    };
    const steps = [];
    for (const st of sc.steps) {
      if (st.sortDone) { viewer.sortRunning = false; steps.push({ sortDone: true }); continue; }
      viewer.camera.matrixWorld.fromArray(st.matrixWorld);
      viewer.camera.projectionMatrix.fromArray(st.projection);
      viewer.camera.matrixWorld.decompose(viewer.camera.position, viewer.camera.quaternion, new THREE.Vector3());
      viewer.next = st;
      posted.length = 0;
      const ret = await runSplatSort.call(viewer, !!st.force, !!st.forceSortAll);
      await new Promise((r) => setImmediate(r));                              // the .then() of the (resolved) distance promise
      const m = posted.length ? posted[posted.length - 1].sort : null;
      steps.push({ returned: ret, posted: posted.length, splatSortCount: m ? m.splatSortCount : null, splatRenderCount: m ? m.splatRenderCount : null,
                   modelViewProj: m ? Array.from(m.modelViewProj).map(f64hex) : null, sortRunning: viewer.sortRunning });
    }
    out.push({ name: sc.name, steps });
  }
  fs.writeFileSync(outPath, JSON.stringify(out));
  console.log(JSON.stringify({ ok: true, scripts: out.length, posted: out.map((o) => o.steps.filter((s) => s.posted).length) }));
};
run().catch((e) => { console.error(e); process.exit(1); });
