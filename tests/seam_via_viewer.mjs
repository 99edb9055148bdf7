// tests/seam_via_viewer.mjs — drives the engine's two drop-in modules (node/SplatMesh.mjs, node/SortWorker.mjs) with the
// REFERENCE's own caller code: the text of Viewer.addSplatBuffersToMesh / setupSortWorker / runSplatSort /
// gatherSceneNodesForSort / updateSplatMesh (cut from /root/reference/src/Viewer.js into oracle/_ref/seam/viewer_cut.json by
// oracle/make_seam_bundle.mjs) is evaluated with a minimal Viewer object as `this`, and the splat data comes from the
// reference's own INRIA-v1 PLY parser -> SplatBuffer (imported from the same bundle).  Nothing here re-implements a caller.
//   load .ply -> SplatBuffer -> addSplatBuffersToMesh([buffer], [options], finalBuild)  (-> SplatMesh.build, octree when final)
//   -> queue the `centers` message, setupSortWorker -> runSplatSort (gatherSceneNodesForSort inside) -> sortDone ->
//   updateRenderIndexes -> updateSplatMesh (uniforms) -> renderer.render(splatMesh, camera)
// Outputs (outDir): frame.u8 (RGBA8, row 0 = bottom), sorted.u32 (the list handed to updateRenderIndexes), meta.json.
// usage: node --experimental-loader ../oracle/three_loader.mjs seam_via_viewer.mjs <bundle> <in.ply> <outDir> <config.json>
import fs from 'fs';
import path from 'path';
import { pathToFileURL, fileURLToPath } from 'url';
import * as THREE from 'three';
const [bundleDir, plyPath, outDir, cfgPath] = process.argv.slice(2);
const here = path.dirname(fileURLToPath(import.meta.url));
const imp = (p) => import(pathToFileURL(p).href);

const run = async () => {
  const cfg = JSON.parse(fs.readFileSync(cfgPath, 'utf8'));
  const { INRIAV1PlyParser } = await imp(path.join(bundleDir, 'src/loaders/ply/INRIAV1PlyParser.js'));
  const { Constants } = await imp(path.join(bundleDir, 'src/Constants.js'));
  const { LogLevel } = await imp(path.join(bundleDir, 'src/LogLevel.js'));
  const { SplatMesh } = await imp(path.join(here, '../node/SplatMesh.mjs'));
  const { createSortWorker } = await imp(path.join(here, '../node/SortWorker.mjs'));
  const cuts = JSON.parse(fs.readFileSync(path.join(bundleDir, 'viewer_cut.json'), 'utf8'));

  const free = { THREE, Constants, LogLevel, createSortWorker, MIN_SPLAT_COUNT_TO_SHOW_SPLAT_TREE_LOADING_SPINNER: 100000 };
  const names = Object.keys(free), values = names.map((k) => free[k]);
  const field = (text) => new Function(...names, 'return (' + text.slice(text.indexOf('function')) + ')();')(...values);   // `x = function() {...}()`
  const method = (text) => new Function(...names, 'return (function ' + text + ');')(...values);                             // `x(args) {...}`

  const buf = fs.readFileSync(plyPath);
  const ply = buf.buffer.slice(buf.byteOffset, buf.byteOffset + buf.byteLength);
  const splatBuffer = INRIAV1PlyParser.parseToUncompressedSplatBuffer(ply, cfg.shDegree);

  const camera = { fov: cfg.fov, isOrthographicCamera: false, zoom: 1, position: new THREE.Vector3(), quaternion: new THREE.Quaternion(),
                   matrixWorld: new THREE.Matrix4().fromArray(cfg.matrixWorld), projectionMatrix: new THREE.Matrix4().fromArray(cfg.projection) };
  camera.matrixWorldInverse = new THREE.Matrix4().copy(camera.matrixWorld).invert();
  camera.matrixWorld.decompose(camera.position, camera.quaternion, new THREE.Vector3());

  const viewer = {
    // the options the Viewer constructor would hold (src/Viewer.js:60-250)
    sharedMemoryForWorkers: !!cfg.sharedMemoryForWorkers, enableSIMDInSort: true, integerBasedSort: true, splatSortDistanceMapPrecision: 16,
    gpuAcceleratedSort: false, logLevel: LogLevel.None, devicePixelRatio: 1, focalAdjustment: 1.0, sceneRevealMode: 2, freeIntermediateSplatData: false,
    initialized: true, sortRunning: false, preSortMessages: [], sortWorker: null, runAfterNextSort: [], splatRenderCount: 0, splatSortCount: 0,
    camera, perspectiveCamera: null, renderer: null, loadingSpinner: { addTask() { return 1; }, removeTask() {}, setMinimized() {} },
    splatMesh: new SplatMesh(0, false, false, !!cfg.halfPrecisionCovariancesOnGPU, 1, false, true, !!cfg.antialiased, 1024, LogLevel.None,
                             cfg.shDegree, 1.0, 0.3),
    isDisposingOrDisposed() { return false; },
    getRenderDimensions(out) { out.x = cfg.width; out.y = cfg.height; },
    adjustForWebXRStereo() {}, forceRenderNextFrame() {}, disposeSortWorker() {},
  };
  viewer.addSplatBuffersToMesh = field(cuts.addSplatBuffersToMesh);
  viewer.setupSortWorker = method(cuts.setupSortWorker);
  viewer.runSplatSort = field(cuts.runSplatSort);
  viewer.gatherSceneNodesForSort = field(cuts.gatherSceneNodesForSort);
  viewer.updateSplatMesh = field(cuts.updateSplatMesh);
  const queueAndSetup = new Function(...names, 'return (function(splatBuffers, splatBufferOptions, finalBuild, showLoadingUIForSplatTreeBuild, ' +
    'replaceExisting, preserveVisibleRegion) { ' + cuts.queueCentersAndSetupWorker + '; return sortWorkerSetupPromise; });')(...values);
  // renderer.render(splatMesh, camera) (src/Viewer.js:1616): three calls every object's onBeforeRender
  viewer.renderer = { render(object, cam) { return object.onBeforeRender(this, null, cam); } };

  const treeReady = new Promise((resolve) => (cfg.finalBuild ? viewer.splatMesh.onSplatTreeReady(resolve) : resolve()));
  await queueAndSetup.call(viewer, [splatBuffer], [cfg.sceneOptions || {}], !!cfg.finalBuild, false, false, true);
  await treeReady;
  let handed = null;
  const realUpdate = viewer.splatMesh.updateRenderIndexes.bind(viewer.splatMesh);
  viewer.splatMesh.updateRenderIndexes = (indexes, count) => { handed = { indexes: Uint32Array.from(indexes.subarray(0, count)), count }; realUpdate(indexes, count); };
  // what the reference's code hands to the two seams: recorded so that the ctypes mirror can be driven with the same numbers
  const posted = [];
  const realPost = viewer.sortWorker.postMessage.bind(viewer.sortWorker);
  viewer.sortWorker.postMessage = (m) => { if (m.sort) posted.push(Array.from(m.sort.modelViewProj)); realPost(m); };
  let sorts = 0;
  for (;;) {                                               // the partial-sort queue: run until a sort of the whole list has landed
    await viewer.runSplatSort.call(viewer, sorts === 0, !!cfg.forceSortAll);
    await new Promise((r) => setImmediate(r));
    if (!viewer.sortPromise) break;
    await viewer.sortPromise;
    sorts++;
    if (viewer.splatSortCount >= viewer.splatRenderCount || sorts > 8) break;
  }
  viewer.updateSplatMesh.call(viewer);
  const frame = viewer.renderer.render(viewer.splatMesh, camera);
  fs.writeFileSync(path.join(outDir, 'frame.u8'), Buffer.from(frame.data.buffer, frame.data.byteOffset, frame.data.byteLength));
  fs.writeFileSync(path.join(outDir, 'sorted.u32'), Buffer.from(handed.indexes.buffer));
  const tree = viewer.splatMesh.getSplatTree();
  fs.writeFileSync(path.join(outDir, 'meta.json'), JSON.stringify({
    splatCount: viewer.splatMesh.getSplatCount(), maxSplatCount: viewer.splatMesh.getMaxSplatCount(), sorts,
    splatRenderCount: viewer.splatRenderCount, splatSortCount: viewer.splatSortCount, renderCountHanded: handed.count,
    width: frame.width, height: frame.height, leaves: tree ? tree.subTrees[0].nodesWithIndexes.length : 0,
    visible: frame.stats.visibleSplats, lastSortTime: viewer.lastSortTime,
    modelViewProj: posted[posted.length - 1], baseModelView: new THREE.Matrix4().copy(camera.matrixWorld).invert().elements,
    view: Array.from(viewer.splatMesh.core.cam.view), proj: Array.from(viewer.splatMesh.core.cam.proj),
    camPos: Array.from(viewer.splatMesh.core.cam.camPos), focal: Array.from(viewer.splatMesh.core.cam.focal) }));
  if (viewer.sortWorker) viewer.sortWorker.terminate();
  await viewer.splatMesh.dispose();
  console.log(JSON.stringify({ ok: true, sorts, splatRenderCount: viewer.splatRenderCount }));
};
run().catch((e) => { console.error(e); process.exit(1); });
