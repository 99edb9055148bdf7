// gsplat.js — JavaScript face of the MI355X engine: the reference's two seams, unchanged for their callers.
//
//   createSortWorker(splatCount, useSharedMemory, enableSIMDInSort, integerBasedSort, dynamicMode, precision)
//       same signature and message protocol as /root/reference/src/worker/SortWorker.js:83-256; the object it
//       returns has postMessage / onmessage / terminate / maxSplatCount exactly as Viewer.setupSortWorker and
//       runSplatSort use them (/root/reference/src/Viewer.js:1235-1320, 1921-1951).
//   SplatMeshHIP
//       the render seam of /root/reference/src/splatmesh/SplatMesh.js: build -> updateRenderIndexes (:1228-1235)
//       -> updateUniforms (:1248-1280) -> render (the draw of src/Viewer.js:1616), returning RGBA8 pixels.
//
// ES2019 / CommonJS on purpose: the container's Node is 12.22.  All compute happens in libgsplat_hip.so.
'use strict';
const path = require('path');
const addon = require(path.join(__dirname, 'gsplat_addon.node'));

const Constants = { DefaultSplatSortDistanceMapPrecision: 16, BytesPerInt: 4, BytesPerFloat: 4, MaxScenes: 32 };
const GS_SORT_INTEGER = 1, GS_SORT_DYNAMIC = 2, GS_MESH_COV_HALF = 1, GS_MESH_SH_U8 = 2, GS_DEST_DEPTH_UNORM24 = 1;
const GS_CAM_ANTIALIASED = 1, GS_CAM_POINT_CLOUD = 2, GS_CAM_ORTHOGRAPHIC = 4, GS_CAM_FADE_IN = 8, GS_CAM_SCENE_EFFECTS = 16, GS_CAM_DYNAMIC = 32;

let sharedContext = null;
function getContext(device) {
  if (!sharedContext) sharedContext = { handle: addon.contextCreate(device || 0), device: device || 0 };
  return sharedContext;
}

class HipSortWorker {
  constructor(splatCount, useSharedMemory, integerBasedSort, dynamicMode, precision, device) {
    this.maxSplatCount = splatCount;          // written by Viewer.js:1285
    this.onmessage = null;
    this.useSharedMemory = !!useSharedMemory;
    this.integerBasedSort = !!integerBasedSort;
    this.dynamicMode = !!dynamicMode;
    this.uploadedSplatCount = 0;
    this._busy = false;
    this._queue = [];
    this._terminated = false;
    this.synchronous = false;
    this.ctx = getContext(device);
    const flags = (this.integerBasedSort ? GS_SORT_INTEGER : 0) | (this.dynamicMode ? GS_SORT_DYNAMIC : 0);
    this.handle = addon.sorterCreate(this.ctx.handle, splatCount, flags, precision);
    // the reference hands views of its WASM memory back to the Viewer (SortWorker.js:180-191, Viewer.js:1270-1278)
    const AB = this.useSharedMemory && typeof SharedArrayBuffer !== 'undefined' ? SharedArrayBuffer : ArrayBuffer;
    this.indexesToSortBuffer = new AB(splatCount * 4);
    this.sortedIndexesBuffer = new AB(splatCount * 4);
    this.precomputedDistancesBuffer = new AB(splatCount * 4);
    this.transformsBuffer = new AB(Constants.MaxScenes * 64);
    const ready = { sortSetupPhase1Complete: true };
    if (this.useSharedMemory) {
      Object.assign(ready, { indexesToSortBuffer: this.indexesToSortBuffer, indexesToSortOffset: 0,
        sortedIndexesBuffer: this.sortedIndexesBuffer, sortedIndexesOffset: 0,
        precomputedDistancesBuffer: this.precomputedDistancesBuffer, precomputedDistancesOffset: 0,
        transformsBuffer: this.transformsBuffer, transformsOffset: 0 });
    }
    setImmediate(() => this._emit(ready));
  }

  _emit(data) { if (this.onmessage) this.onmessage({ data }); }

  // The reference's worker handles one message at a time, in posting order, on its own thread: postMessage returns at once
  // and sortDone arrives later (SortWorker.js:62-80, Viewer.js:1243-1264).  Same here: a sort runs on a libuv pool thread
  // (addon.sorterSortAsync); messages posted while it is in flight wait in a queue.
  postMessage(msg) {
    if (this._busy) { this._queue.push(msg); return; }
    this._handle(msg);
  }

  _drain() {
    this._busy = false;
    if (this._terminated) { this._destroy(); return; }
    while (!this._busy && this._queue.length) this._handle(this._queue.shift());
  }

  _handle(msg) {
    if (msg.centers) {                                                    // SortWorker.js:84-98
      const count = msg.range.count;
      const centers = this.integerBasedSort ? new Int32Array(msg.centers) : new Float32Array(msg.centers);
      const scene = this.dynamicMode ? new Uint32Array(msg.sceneIndexes) : null;
      addon.sorterUploadCenters(this.handle, msg.range.from, count, centers, scene);
      this.uploadedSplatCount = Math.max(this.uploadedSplatCount, msg.range.from + count);
    } else if (msg.sort) {                                                // SortWorker.js:99-115, 31-81
      const s = msg.sort;
      const renderCount = Math.min(s.splatRenderCount || 0, this.uploadedSplatCount);
      const sortCount = Math.min(s.splatSortCount || 0, this.uploadedSplatCount);
      const mvp = new Float32Array(s.modelViewProj);                      // fp64 -> fp32 like SortWorker.js:54
      let indexes, transforms, pre = null;
      if (this.useSharedMemory) {
        indexes = new Uint32Array(this.indexesToSortBuffer, 0, renderCount);
        transforms = new Float32Array(this.transformsBuffer);
        if (s.usePrecomputedDistances) pre = this.integerBasedSort ? new Int32Array(this.precomputedDistancesBuffer) : new Float32Array(this.precomputedDistancesBuffer);
      } else {
        indexes = s.indexesToSort ? new Uint32Array(s.indexesToSort.buffer || s.indexesToSort, s.indexesToSort.byteOffset || 0, renderCount) : null;
        transforms = s.transforms ? new Float32Array(Constants.MaxScenes * 16) : null;
        if (transforms) transforms.set(s.transforms);
        if (s.usePrecomputedDistances) pre = s.precomputedDistances;
      }
      if (pre && pre.length < this.uploadedSplatCount) throw new RangeError('precomputedDistances shorter than the uploaded splat count');
      if (this.dynamicMode && !transforms) transforms = new Float32Array(Constants.MaxScenes * 16);
      const out = this.useSharedMemory ? new Uint32Array(this.sortedIndexesBuffer, 0, renderCount) : new Uint32Array(renderCount);
      const culled = this.frustumCull;
      const done = (r) => {
        // under setFrustumCull the list holds only the kept splats: the Viewer draws `splatRenderCount` of them
        // (Viewer.js:1251-1262 passes e.data.splatRenderCount to updateRenderIndexes)
        const drawCount = culled ? r.resultCount : renderCount;
        const reply = { sortDone: true, splatSortCount: Math.min(sortCount, drawCount), splatRenderCount: drawCount, sortTime: r.sortTime, status: r.status };
        if (!this.useSharedMemory) reply.sortedIndexes = culled ? out.subarray(0, drawCount) : out;
        return reply;
      };
      if (this.synchronous) {                                             // test hook: the round-1 behaviour
        const reply = done(addon.sorterSort(this.handle, mvp, indexes, sortCount, renderCount, pre, this.dynamicMode ? transforms : null, out));
        setImmediate(() => this._emit(reply));
        return;
      }
      this._busy = true;
      addon.sorterSortAsync(this.handle, mvp, indexes, sortCount, renderCount, pre, this.dynamicMode ? transforms : null, out, (err, r) => {
        if (err) { this._drain(); throw err; }
        this._emit(done(r));
        this._drain();
      });
    } else if (msg.init) {
      // the reference ships its WASM bytes through an init message (SortWorker.js:116-199); nothing to do here
    }
  }

  // HIP-engine extras (no counterpart in the reference).  bindMesh: results stay on the device as positions in the mesh's
  // storage order; setVisibilityCull: full sorts keep only what mesh.project() of this frame's camera (and strip) draws
  bindMesh(mesh) { addon.sorterBindMesh(this.handle, mesh ? mesh.handle : null); }
  setVisibilityCull(enable) {
    addon.sorterSetVisibilityCull(this.handle, enable ? 1 : 0);
    this.frustumCull = this.frustumCull || !!enable;                       // replies carry the kept count either way
  }
  // fuse a per-splat frustum cull into full sorts
  setFrustumCull(enable) {
    addon.sorterSetFrustumCull(this.handle, enable ? 1 : 0);
    this.frustumCull = !!enable;
  }

  terminate() {                                                           // Viewer.js:1311
    this._terminated = true;
    if (!this._busy) this._destroy();                                     // else: once the sort in flight has landed
  }

  _destroy() {
    if (this.handle) { addon.sorterDestroy(this.handle); this.handle = null; }
  }
}

function createSortWorker(splatCount, useSharedMemory, enableSIMDInSort, integerBasedSort, dynamicMode,
                          splatSortDistanceMapPrecision = Constants.DefaultSplatSortDistanceMapPrecision, device = 0) {
  return new HipSortWorker(splatCount, useSharedMemory, integerBasedSort, dynamicMode, splatSortDistanceMapPrecision, device);
}

// THREE.DataUtils.toHalfFloat (three r160): clamp, then truncate through the base/shift tables
const halfTables = (() => {
  const base = new Uint32Array(512), shift = new Uint32Array(512);
  for (let i = 0; i < 256; i++) {
    const e = i - 127;
    if (e < -27) { base[i] = 0; base[i | 0x100] = 0x8000; shift[i] = 24; shift[i | 0x100] = 24; }
    else if (e < -14) { base[i] = 0x0400 >> (-e - 14); base[i | 0x100] = (0x0400 >> (-e - 14)) | 0x8000; shift[i] = -e - 1; shift[i | 0x100] = -e - 1; }
    else if (e <= 15) { base[i] = (e + 15) << 10; base[i | 0x100] = ((e + 15) << 10) | 0x8000; shift[i] = 13; shift[i | 0x100] = 13; }
    else if (e < 128) { base[i] = 0x7c00; base[i | 0x100] = 0xfc00; shift[i] = 24; shift[i | 0x100] = 24; }
    else { base[i] = 0x7c00; base[i | 0x100] = 0xfc00; shift[i] = 13; shift[i | 0x100] = 13; }
  }
  return { base, shift, f: new Float32Array(1), u: null };
})();
halfTables.u = new Uint32Array(halfTables.f.buffer);
function toHalfFloat(val) {
  halfTables.f[0] = Math.min(Math.max(val, -65504), 65504);
  const f = halfTables.u[0], e = (f >> 23) & 0x1ff;
  return halfTables.base[e] + ((f & 0x007fffff) >> halfTables.shift[e]);
}

class SplatMeshHIP {
  constructor(maxSplatCount, options = {}) {
    this.ctx = getContext(options.device);
    this.maxSplatCount = maxSplatCount;
    this.shDegree = options.sphericalHarmonicsDegree || 0;
    this.halfPrecisionCovariancesOnGPU = !!options.halfPrecisionCovariancesOnGPU;
    this.antialiased = !!options.antialiased;
    this.kernel2DSize = options.kernel2DSize === undefined ? 0.3 : options.kernel2DSize;
    this.maxScreenSpaceSplatSize = options.maxScreenSpaceSplatSize || 1024;
    this.splatScale = 1.0;
    this.pointCloudModeEnabled = false;
    this.dynamicMode = !!options.dynamicMode;                               // per-scene transforms (SplatMesh dynamicMode)
    this.enableOptionalEffects = !!options.enableOptionalEffects;           // per-scene opacity / visibility
    this.sphericalHarmonics8Bit = !!options.sphericalHarmonics8Bit;         // .ksplat compression level 2 SH
    this.orthographicMode = false;
    this.fadeIn = false;
    this.handle = addon.meshCreate(this.ctx.handle, maxSplatCount, this.shDegree,
      (this.halfPrecisionCovariancesOnGPU ? GS_MESH_COV_HALF : 0) | (this.sphericalHarmonics8Bit ? GS_MESH_SH_U8 : 0));
    this.splatCount = 0;
    this.renderCount = 0;
    this.indexes = null;
    this.sortWorker = null;
    this.cam = { view: new Float32Array(16), proj: new Float32Array(16), camPos: new Float32Array(3), focal: new Float32Array(2),
      width: 0, height: 0, splatScale: 1, kernel2d: this.kernel2DSize, maxSplatPx: this.maxScreenSpaceSplatSize, invFocalAdj: 1,
      shDegree: this.shDegree, flags: 0, tileRowBegin: 0, tileRowEnd: 0, orthoZoom: 1, fadeStartRadius: 0,
      sceneCenter: new Float32Array(3), viewMatrix: new Float32Array(16) };
  }
  // fillSplatDataArrays output (SplatMesh.js:1853-1902): centers F32[3n], covariances F32[6n], colors U8[4n], sh Uint16 half bits
  build(centers, covariances, colors, sphericalHarmonics, start = 0) {
    const n = centers.length / 3;
    let cov16 = null;
    if (this.halfPrecisionCovariancesOnGPU) {
      cov16 = new Uint16Array(covariances.length);
      for (let i = 0; i < covariances.length; i++) cov16[i] = toHalfFloat(covariances[i]);
    }
    const sh16 = this.shDegree && !this.sphericalHarmonics8Bit ? sphericalHarmonics : null;
    addon.meshUpload(this.handle, start, n, centers, cov16 ? null : covariances, cov16, colors, sh16);
    if (this.shDegree && this.sphericalHarmonics8Bit) addon.meshUploadShU8(this.handle, start, n, sphericalHarmonics);   // Uint8Array
    this.splatCount = Math.max(this.splatCount, start + n);
  }
  // sceneIndexes texture (SplatMesh.js:881-897) and the per-scene uniforms of updateUniforms (:1263-1276):
  // {sceneCount, transforms F32(16n), invCamPos F32(4n) = inverse(transform) * cameraPosition, opacity F32(n), visible U32(n),
  //  sh8Min F32(n), sh8Max F32(n)}
  setSceneIndexes(sceneIndexes, start = 0) { addon.meshUploadSceneIndexes(this.handle, start, sceneIndexes.length, sceneIndexes); }
  setScenes(params) { addon.meshSetScenes(this.handle, params); this.hasScenes = true; }
  // the fade-in of SplatMesh.updateVisibleRegionFadeDistance: visibleRegionFadeStartRadius + sceneCenter; null switches it off
  setFadeIn(sceneCenter, fadeStartRadius) {
    this.fadeIn = sceneCenter !== null && sceneCenter !== undefined;
    if (this.fadeIn) { this.cam.sceneCenter.set(sceneCenter); this.cam.fadeStartRadius = fadeStartRadius; }
  }
  getSplatCount() { return this.splatCount; }
  updateRenderIndexes(globalIndexes, renderSplatCount) { this.indexes = globalIndexes; this.sortWorker = null; this.renderCount = renderSplatCount; }
  useSortWorkerResult(worker, renderSplatCount) { this.sortWorker = worker; this.indexes = null; this.renderCount = renderSplatCount; }
  updateUniforms(renderDimensions, cameraFocalLengthX, cameraFocalLengthY, orthographicMode, orthographicZoom, inverseFocalAdjustment) {
    this.orthographicMode = !!orthographicMode;
    this.cam.orthoZoom = orthographicZoom === undefined ? 1 : orthographicZoom;
    this.cam.width = renderDimensions.x; this.cam.height = renderDimensions.y;
    this.cam.focal[0] = cameraFocalLengthX; this.cam.focal[1] = cameraFocalLengthY;
    this.cam.invFocalAdj = inverseFocalAdjustment;
  }
  // three's built-in uniforms; viewMatrixElements (camera.matrixWorldInverse) is only needed in dynamic mode
  setCameraMatrices(modelViewElements, projectionElements, cameraPosition, viewMatrixElements) {
    this.cam.view.set(modelViewElements); this.cam.proj.set(projectionElements); this.cam.camPos.set(cameraPosition);
    this.cam.viewMatrix.set(viewMatrixElements || modelViewElements);
  }
  setSplatScale(s = 1) { this.splatScale = s; }
  setPointCloudModeEnabled(e) { this.pointCloudModeEnabled = !!e; }
  // HIP-engine extra: whether very deep bins may be composited by many waves at once (same pixels either way)
  setDeepPass(enabled) { addon.meshSetDeepPass(this.handle, enabled ? 1 : 0); }
  // HIP-engine extra: composite as the browser's RGBA8 render target does - back to front, every channel rounded to 8 bits after
  // every splat (SplatMaterial3D.js:65-75 as a ROP executes it) - instead of in fp32 rounded once.  The mode walks the splats in
  // front of each quadrant's saturation depth (~4x the fp32 blend); full = true walks every list to its end (~70x: verification)
  setRop8(enabled, full = false) { addon.meshSetDrawMode(this.handle, enabled ? (full ? 2 : 1) : 0); }
  // The destination of the following draws: `depthTest: true, depthWrite: false` against `depth` (Float32Array W*H, window depth in
  // [0, 1], row 0 = bottom: what the host's opaque geometry left) and NormalBlending over `rgba` (Uint8Array 4*W*H) - the reference's
  // material state (SplatMaterial3D.js:72-73) when the splat mesh shares a scene with other objects (DropInViewer.js:34-42,
  // Viewer.js:1610-1616).  depthBits 24: compare as a 24-bit depth buffer.  Both null: a cleared target, no test.
  setDestination(depth, rgba, width, height, depthBits = 32) {
    addon.meshSetDestination(this.handle, depth || null, rgba || null, width >>> 0, height >>> 0, depthBits === 24 ? GS_DEST_DEPTH_UNORM24 : 0);
  }
  _camera() {
    const c = this.cam;
    c.splatScale = this.splatScale;
    c.flags = (this.antialiased ? GS_CAM_ANTIALIASED : 0) | (this.pointCloudModeEnabled ? GS_CAM_POINT_CLOUD : 0) |
      (this.orthographicMode ? GS_CAM_ORTHOGRAPHIC : 0) | (this.fadeIn ? GS_CAM_FADE_IN : 0) |
      (this.enableOptionalEffects ? GS_CAM_SCENE_EFFECTS : 0) | (this.dynamicMode ? GS_CAM_DYNAMIC : 0);
    return c;
  }
  // HIP-engine extras for a multi-GPU draw (no counterpart in the reference: one WebGL context).  One process per GPU:
  //   const group = new StripGroup(idBytesFromRank0, worldSize, rank)      (rank 0: StripGroup.uniqueId())
  //   worker.setVisibilityCull(true); mesh.useSortWorkerResult(worker, n); worker.bindMesh(mesh)
  //   per frame: mesh.project(rows) -> worker.postMessage({sort}) ... sortDone -> mesh.renderStrip(group, rowBegin, rowEnd, 0, out)
  // rows = [beginTileRow, endTileRow) of this rank; rowBegin / rowEnd = Uint32Array of every rank's PIXEL rows.
  project(tileRows) {
    const c = this._camera();
    c.tileRowBegin = tileRows ? tileRows[0] : 0; c.tileRowEnd = tileRows ? tileRows[1] : 0;
    addon.meshProject(this.handle, c);
  }
  renderStrip(group, rowBegin, rowEnd, root, out) {
    const c = this._camera();
    c.tileRowBegin = 0; c.tileRowEnd = 0;                                  // the library derives the strip from the row tables
    return addon.groupRenderGather(group.handle, this.handle, c, this.indexes, this.sortWorker ? this.sortWorker.handle : null,
                                   this.renderCount, rowBegin, rowEnd, root, out || null);
  }
  render(out) {
    const c = this._camera();
    c.tileRowBegin = 0; c.tileRowEnd = 0;
    const pixels = out || new Uint8Array(c.width * c.height * 4);
    const stats = addon.meshRender(this.handle, c, this.indexes, this.sortWorker ? this.sortWorker.handle : null, this.renderCount, pixels);
    return { pixels, stats };
  }
  dispose() { if (this.handle) { addon.meshDestroy(this.handle); this.handle = null; } }
}

// The strip gather of a multi-GPU draw (gs_group_*: RCCL behind the C ABI).  StripGroup.uniqueId() on one rank, the 128 bytes
// to every rank over any side channel (a pipe, a file, IPC), then `new StripGroup(id, worldSize, rank)` on each.
class StripGroup {
  static uniqueId() { return addon.groupUniqueId(); }
  constructor(id, worldSize, rank, device) {
    this.worldSize = worldSize; this.rank = rank;
    this.handle = addon.groupCreate(getContext(device).handle, worldSize > 1 ? id : null, worldSize, rank);
  }
  // the strip transfer of frame k runs beside the draw of frame k + 1 (renderStrip alternates its buffers itself); every rank
  // sets the same; wait() = all transfers issued so far have completed
  setOverlap(enabled) { addon.groupSetOverlap(this.handle, enabled ? 1 : 0); }
  wait() { addon.groupWait(this.handle); }
  dispose() { if (this.handle) { addon.groupDestroy(this.handle); this.handle = null; } }
}

module.exports = { createSortWorker, SplatMeshHIP, StripGroup, toHalfFloat, Constants, addon };
