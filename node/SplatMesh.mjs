// SplatMesh.mjs — drop-in for the reference's SplatMesh (/root/reference/src/splatmesh/SplatMesh.js) over the MI355X engine.
//
// Same constructor arguments, same `build(splatBuffers, sceneOptions, keepSceneTransforms, finalBuild, ...)`, same
// getIntegerCenters / getSplatCount / getMaxSplatCount / updateRenderIndexes / updateUniforms / fillTransformsArray /
// getSplatTree / setSplatScale ..., so that the Viewer's own code - addSplatBuffersToMesh (src/Viewer.js:1189-1228),
// setupSortWorker (:1235-1300), runSplatSort (:1833-1964), gatherSceneNodesForSort (:1969-2077), updateSplatMesh
// (:651-677) and the draw `renderer.render(splatMesh, camera)` (:1616) - runs against it UNCHANGED
// (tests/test_node_seam.py executes exactly that text).  Splat data reaches the device through the SplatBuffers' own
// fillSplat*Array methods, as SplatMesh.fillSplatDataArrays (:1853-1902) does; nothing here parses files.
//
// Integration = two imports (INTEGRATION.md):
//     import { SplatMesh } from '<this repo>/node/SplatMesh.mjs';          // was './splatmesh/SplatMesh.js'
//     import { createSortWorker } from '<this repo>/node/SortWorker.mjs';  // was './worker/SortWorker.js'
// plus the Viewer option `gpuAcceleratedSort: false` (the WebGL transform-feedback distance pass, SplatMesh.js:1701-1814, is
// subsumed by the device sort's own keying).
//
// `three` is the same peer dependency the reference imports (in this repo's tests it resolves to oracle/three_min.mjs).
// The frame lands in `this.frame` = {data: Uint8Array RGBA8 (row 0 = bottom, like gl.readPixels), width, height}; a host
// renderer shows it as a DataTexture on a full-screen quad (onBeforeRender is three's per-object hook, called by
// renderer.render).
import * as THREE from 'three';
import { createRequire } from 'module';
const require = createRequire(import.meta.url);
const { SplatMeshHIP: HipMeshCore, addon } = require('./gsplat.js');

export const SplatRenderMode = { ThreeD: 0, TwoD: 1 };
export const SceneRevealMode = { Default: 0, Gradual: 1, Instant: 2 };

// SplatScene (src/splatmesh/SplatScene.js) without the Object3D base: the transform is compose(position, quaternion, scale)
class HipSplatScene {
  constructor(splatBuffer, position = new THREE.Vector3(), quaternion = new THREE.Quaternion(), scale = new THREE.Vector3(1, 1, 1),
              minimumAlpha = 1, opacity = 1.0, visible = true) {
    this.splatBuffer = splatBuffer;
    this.position = new THREE.Vector3().copy(position);
    this.quaternion = new THREE.Quaternion().copy(quaternion);
    this.scale = new THREE.Vector3().copy(scale);
    this.transform = new THREE.Matrix4();
    this.matrix = new THREE.Matrix4();
    this.minimumAlpha = minimumAlpha;
    this.opacity = opacity;
    this.visible = visible;
  }
  copyTransformData(other) {
    this.position.copy(other.position); this.quaternion.copy(other.quaternion); this.scale.copy(other.scale);
    this.transform.copy(other.transform);
  }
  updateTransform() {                                       // :28-36 (a scene has no parent here: matrixWorld == matrix)
    this.matrix.compose(this.position, this.quaternion, this.scale);
    this.transform.copy(this.matrix);
  }
}

export class SplatMesh {
  // Same positional arguments as the reference's constructor (src/splatmesh/SplatMesh.js:34-40); the Viewer's text reads the
  // public fields by these names.
  constructor(...args) {
    const names = ['splatRenderMode', 'dynamicMode', 'enableOptionalEffects', 'halfPrecisionCovariancesOnGPU', 'devicePixelRatio',
                   'enableDistancesComputationOnGPU', 'integerBasedDistancesComputation', 'antialiased', 'maxScreenSpaceSplatSize',
                   'logLevel', 'sphericalHarmonicsDegree', 'sceneFadeInRateMultiplier', 'kernel2DSize'];
    const defaults = [SplatRenderMode.ThreeD, false, false, false, 1, true, false, false, 1024, 0, 0, 1.0, 0.3];
    names.forEach((name, k) => { this[name] = args[k] === undefined ? defaults[k] : args[k]; });
    if (this.splatRenderMode !== SplatRenderMode.ThreeD) throw new Error('SplatMesh (HIP): only SplatRenderMode.ThreeD is implemented');
    this.enableDistancesComputationOnGPU = false;           // the device sort keys the splats itself (see the header)
    Object.assign(this, {
      renderer: undefined, scenes: [], sceneOptions: undefined, minSphericalHarmonicsDegree: 0,
      splatTree: null, baseSplatTree: null, onSplatTreeReadyCallback: null, splatDataTextures: {},
      globalSplatIndexToLocalSplatIndexMap: [], globalSplatIndexToSceneIndexMap: [],
      lastBuildScenes: [], lastBuildSplatCount: 0, lastBuildMaxSplatCount: 0, lastBuildSceneCount: 0,
      firstRenderTime: -1, finalBuild: false, splatScale: 1.0, pointCloudModeEnabled: false,
      disposed: false, visible: false, frustumCulled: false,
      matrixWorld: new THREE.Matrix4(),                     // Object3D.matrixWorld (src/Viewer.js:1891 multiplies by it)
      core: null,                                           // the device-side mesh (gs_mesh_*)
      frame: null, shCompressionLevel: 1 });
  }

  // ---- statics, as the reference (:173-228, :1311-1341) --------------------------------------------------------------
  static buildScenes(parentObject, splatBuffers, sceneOptions) {
    const vec = (v, fallback) => new THREE.Vector3().fromArray(v || fallback);
    return splatBuffers.map((buffer, k) => {
      const o = sceneOptions[k] || {};
      return SplatMesh.createScene(buffer, vec(o.position, [0, 0, 0]), new THREE.Quaternion().fromArray(o.rotation || [0, 0, 0, 1]),
                                   vec(o.scale, [1, 1, 1]), o.splatAlphaRemovalThreshold || 1, o.opacity, o.visible);
    });
  }
  static createScene(splatBuffer, position, rotation, scale, minimumAlpha, opacity = 1.0, visible = true) {
    return new HipSplatScene(splatBuffer, position, rotation, scale, minimumAlpha, opacity, visible);
  }
  // global splat index -> (index inside its buffer, scene): every buffer owns getMaxSplatCount() consecutive global indexes
  static buildSplatIndexMaps(splatBuffers) {
    const localSplatIndexMap = [], sceneIndexMap = [];
    splatBuffers.forEach((buffer, sceneIndex) => {
      for (let local = 0, n = buffer.getMaxSplatCount(); local < n; local++) { localSplatIndexMap.push(local); sceneIndexMap.push(sceneIndex); }
    });
    return { localSplatIndexMap, sceneIndexMap };
  }
  static _sum(items, pick) { return items.reduce((n, item) => n + pick(item), 0); }
  static getTotalSplatCountForScenes(scenes) { return SplatMesh._sum(scenes, (sc) => (sc && sc.splatBuffer ? sc.splatBuffer.getSplatCount() : 0)); }
  static getTotalSplatCountForSplatBuffers(splatBuffers) { return SplatMesh._sum(splatBuffers, (b) => b.getSplatCount()); }
  static getTotalMaxSplatCountForScenes(scenes) { return SplatMesh._sum(scenes, (sc) => (sc && sc.splatBuffer ? sc.splatBuffer.getMaxSplatCount() : 0)); }
  static getTotalMaxSplatCountForSplatBuffers(splatBuffers) { return SplatMesh._sum(splatBuffers, (b) => b.getMaxSplatCount()); }

  // ---- build (:306-405) ----------------------------------------------------------------------------------------------
  build(splatBuffers, sceneOptions, keepSceneTransforms = true, finalBuild = false, onSplatTreeIndexesUpload, onSplatTreeConstruction,
        preserveVisibleRegion = true) {
    Object.assign(this, { sceneOptions, finalBuild });
    const incoming = SplatMesh.buildScenes(this, splatBuffers, sceneOptions);
    if (keepSceneTransforms) {
      const shared = Math.min(this.scenes.length, incoming.length);
      for (let k = 0; k < shared; k++) incoming[k].copyTransformData(this.getScene(k));
    }
    this.scenes = incoming;
    this.minSphericalHarmonicsDegree = splatBuffers.reduce((d, buffer) => Math.min(d, buffer.getMinSphericalHarmonicsDegree()),
                                                           Math.min(3, this.sphericalHarmonicsDegree));
    // An "update build" appends splats to the one buffer the previous build already uploaded (progressive loading); anything
    // else starts the device mesh afresh (the reference's isUpdateBuild rule, :336-352).
    const capacity = SplatMesh.getTotalMaxSplatCountForSplatBuffers(splatBuffers);
    const sameBuffers = splatBuffers.length === this.lastBuildScenes.length &&
                        splatBuffers.every((buffer, k) => buffer === this.lastBuildScenes[k].splatBuffer);
    const appendOnly = sameBuffers && this.scenes.length === 1 && this.lastBuildSceneCount === 1 && this.lastBuildMaxSplatCount === capacity;
    if (!appendOnly) {
      Object.assign(this, { lastBuildScenes: [], lastBuildSplatCount: 0, lastBuildMaxSplatCount: 0 });
      this.disposeMeshData();
      const maps = SplatMesh.buildSplatIndexMaps(splatBuffers);
      this.globalSplatIndexToLocalSplatIndexMap = maps.localSplatIndexMap;
      this.globalSplatIndexToSceneIndexMap = maps.sceneIndexMap;
    }
    this.updateTransforms();                                // the scenes' matrices exist before the first data fill
    const uploadedUpTo = this.getSplatCount(true);
    const update = this.refreshGPUDataFromSplatBuffers(appendOnly);
    Object.assign(this, { lastBuildScenes: this.scenes.slice(), lastBuildSplatCount: uploadedUpTo,
                          lastBuildMaxSplatCount: this.getMaxSplatCount(), lastBuildSceneCount: this.scenes.length });
    if (finalBuild && this.scenes.length) {
      const minAlphas = sceneOptions.map((o) => o.splatAlphaRemovalThreshold || 1);
      this.buildSplatTree(minAlphas, onSplatTreeIndexesUpload, onSplatTreeConstruction).then(() => {
        const callback = this.onSplatTreeReadyCallback;
        this.onSplatTreeReadyCallback = null;
        if (callback) callback(this.splatTree);
      });
    }
    this.visible = this.scenes.length > 0;
    return update;
  }

  // setupDataTextures' decisions (:637-690, 1060-1090) without the textures: covariance as fp16 when asked for, SH as fp16
  // (compression level <= 1) or uint8 (level 2: kept 8-bit end to end, like the reference's sphericalHarmonics8BitMode)
  _compressionLevels() { return this.scenes.map((sc) => sc.splatBuffer.compressionLevel); }
  getMaximumSplatBufferCompressionLevel() { const l = this._compressionLevels(); return l.length ? Math.max(...l) : undefined; }
  getMinimumSplatBufferCompressionLevel() { const l = this._compressionLevels(); return l.length ? Math.min(...l) : undefined; }
  getTargetCovarianceCompressionLevel() { return this.halfPrecisionCovariancesOnGPU ? 1 : 0; }
  getTargetSphericalHarmonicsCompressionLevel() { return Math.max(1, this.getMaximumSplatBufferCompressionLevel()); }

  // what the sort worker needs about the freshly uploaded range (the reply shape of :588-609)
  refreshGPUDataFromSplatBuffers(sinceLastBuildOnly) {
    const last = this.getSplatCount(true) - 1, first = sinceLastBuildOnly ? this.lastBuildSplatCount : 0;
    this.refreshDataTexturesFromSplatBuffers(sinceLastBuildOnly);
    return Object.assign({ from: first, to: last, count: last - first + 1 }, this.getDataForDistancesComputation(first, last));
  }

  // :621-635 + :900-1058: the splat buffers' own fill methods produce the arrays; they go to device planes instead of
  // padded data textures
  refreshDataTexturesFromSplatBuffers(sinceLastBuildOnly) {
    const maxSplatCount = this.getMaxSplatCount();
    const fromSplat = sinceLastBuildOnly ? this.lastBuildSplatCount : 0, toSplat = this.getSplatCount(true) - 1;
    if (!sinceLastBuildOnly || !this.core) {
      if (this.core) this.core.dispose();
      this.shCompressionLevel = this.getTargetSphericalHarmonicsCompressionLevel();
      this.core = new HipMeshCore(maxSplatCount, {
        sphericalHarmonicsDegree: this.minSphericalHarmonicsDegree, halfPrecisionCovariancesOnGPU: this.halfPrecisionCovariancesOnGPU,
        antialiased: this.antialiased, kernel2DSize: this.kernel2DSize, maxScreenSpaceSplatSize: this.maxScreenSpaceSplatSize,
        dynamicMode: this.dynamicMode, enableOptionalEffects: this.enableOptionalEffects,
        sphericalHarmonics8Bit: this.minSphericalHarmonicsDegree > 0 && this.shCompressionLevel === 2 });
      this.core.setSplatScale(this.splatScale);
      this.core.setPointCloudModeEnabled(this.pointCloudModeEnabled);
      if (this._destination) this.setDestination(this._destination);      // a rebuilt mesh keeps the host's destination
      if (this._rop8) this.core.setRop8(true, this._rop8 === 2);          // ... and its draw mode
      this.splatDataTextures = { baseData: {}, maxSplatCount,
        covariances: { compressionLevel: this.getTargetCovarianceCompressionLevel(), size: new THREE.Vector2(maxSplatCount, 1) },
        centerColors: { size: new THREE.Vector2(maxSplatCount, 1) } };       // sizes: Viewer.js:1289-1296 only logs them
    }
    const count = toSplat - fromSplat + 1;
    if (count <= 0) return;
    const covLevel = this.getTargetCovarianceCompressionLevel();
    const covariances = covLevel === 1 ? new Uint16Array(count * 6) : new Float32Array(count * 6);
    const centers = new Float32Array(count * 3);
    const colors = new Uint8Array(count * 4);
    const shComponents = [0, 9, 24][this.minSphericalHarmonicsDegree];
    let sh = null;
    if (shComponents) sh = this.shCompressionLevel === 2 ? new Uint8Array(count * shComponents) : new Uint16Array(count * shComponents);
    // updateBaseDataFromSplatBuffers (:900-913): source range [fromSplat, toSplat], written from 0 of these range arrays
    this.fillSplatDataArrays(covariances, null, null, centers, colors, sh, undefined, covLevel, 0, this.shCompressionLevel,
                             sinceLastBuildOnly ? fromSplat : undefined, sinceLastBuildOnly ? toSplat : undefined, 0);
    addon.meshUpload(this.core.handle, fromSplat, count, centers, covLevel === 1 ? null : covariances, covLevel === 1 ? covariances : null,
                     colors, sh && this.shCompressionLevel !== 2 ? sh : null);
    if (sh && this.shCompressionLevel === 2) addon.meshUploadShU8(this.core.handle, fromSplat, count, sh);
    this.core.splatCount = Math.max(this.core.splatCount, fromSplat + count);
    if (this.scenes.length > 1 || this.dynamicMode || this.enableOptionalEffects) {
      const sceneIndexes = new Uint32Array(count);
      for (let c = 0; c < count; c++) sceneIndexes[c] = this.globalSplatIndexToSceneIndexMap[fromSplat + c];
      this.core.setSceneIndexes(sceneIndexes, fromSplat);
    }
    this._scenesDirty = true;
  }

  getDataForDistancesComputation(start, end) {              // :572-581
    const centers = this.integerBasedDistancesComputation ? this.getIntegerCenters(start, end, true) : this.getFloatCenters(start, end, true);
    return { centers, sceneIndexes: this.getSceneIndexes(start, end) };
  }

  // :1853-1902, argument for argument
  fillSplatDataArrays(covariances, scales, rotations, centers, colors, sphericalHarmonics, applySceneTransform, covarianceCompressionLevel = 0,
                      scaleRotationCompressionLevel = 0, sphericalHarmonicsCompressionLevel = 1, srcStart, srcEnd, destStart = 0, sceneIndex) {
    const noScaleOverride = new THREE.Vector3();
    noScaleOverride.x = noScaleOverride.y = noScaleOverride.z = undefined;       // "no override" for fillSplatScaleRotationArray
    const matrix = new THREE.Matrix4();
    const oneScene = Number.isInteger(sceneIndex) && sceneIndex >= 0 && sceneIndex <= this.scenes.length;
    const firstScene = oneScene ? sceneIndex : 0, lastScene = oneScene ? sceneIndex : this.scenes.length - 1;
    if (applySceneTransform === undefined || applySceneTransform === null) applySceneTransform = !this.dynamicMode;
    if ((scales || rotations) && !(scales && rotations)) throw new Error('SplatMesh::fillSplatDataArrays() -> "scales" and "rotations" must both be valid.');
    let dest = destStart;
    for (let k = firstScene; k <= lastScene; k++) {
      const scene = this.getScene(k), buffer = scene.splatBuffer;
      let transform;                                                         // undefined = none (what the fill methods expect)
      if (applySceneTransform) { this.getSceneTransform(k, matrix); transform = matrix; }
      if (covariances) buffer.fillSplatCovarianceArray(covariances, transform, srcStart, srcEnd, dest, covarianceCompressionLevel);
      if (scales) buffer.fillSplatScaleRotationArray(scales, rotations, transform, srcStart, srcEnd, dest, scaleRotationCompressionLevel, noScaleOverride);
      if (centers) buffer.fillSplatCenterArray(centers, transform, srcStart, srcEnd, dest);
      if (colors) buffer.fillSplatColorArray(colors, scene.minimumAlpha, srcStart, srcEnd, dest);
      if (sphericalHarmonics) {
        buffer.fillSphericalHarmonicsArray(sphericalHarmonics, this.minSphericalHarmonicsDegree, transform, srcStart, srcEnd, dest,
                                           sphericalHarmonicsCompressionLevel);
      }
      dest += buffer.getSplatCount();
    }
  }

  // centres of splats [start, end] as the sort worker wants them (:1912-1948): x1000 rounded to int32, or float; 3 or 4 per splat
  _centers(start, end) {
    const xyz = new Float32Array((end - start + 1) * 3);
    this.fillSplatDataArrays(null, null, null, xyz, null, null, undefined, undefined, undefined, undefined, start);
    return xyz;
  }
  getIntegerCenters(start, end, padFour = false) {
    const xyz = this._centers(start, end), stride = padFour ? 4 : 3, out = new Int32Array((xyz.length / 3) * stride);
    for (let k = 0, o = 0; k < xyz.length; k += 3, o += stride) {
      out[o] = Math.round(xyz[k] * 1000.0); out[o + 1] = Math.round(xyz[k + 1] * 1000.0); out[o + 2] = Math.round(xyz[k + 2] * 1000.0);
      if (padFour) out[o + 3] = 1000;
    }
    return out;
  }
  getFloatCenters(start, end, padFour = false) {
    const xyz = this._centers(start, end);
    if (!padFour) return xyz;
    const out = new Float32Array((xyz.length / 3) * 4);
    for (let k = 0, o = 0; k < xyz.length; k += 3, o += 4) { out[o] = xyz[k]; out[o + 1] = xyz[k + 1]; out[o + 2] = xyz[k + 2]; out[o + 3] = 1.0; }
    return out;
  }
  getSceneIndexes(start, end) {                             // (:1667-1677; written at [start, end] of the array, as there)
    const out = new Uint32Array(end - start + 1);
    for (let g = start; g <= end; g++) out[g] = this.globalSplatIndexToSceneIndexMap[g];
    return out;
  }

  // ---- counts, scenes, transforms ------------------------------------------------------------------------------------
  getSplatCount(includeSinceLastBuild = false) {
    return includeSinceLastBuild ? SplatMesh.getTotalSplatCountForScenes(this.scenes) : this.lastBuildSplatCount;
  }
  getMaxSplatCount() { return SplatMesh.getTotalMaxSplatCountForScenes(this.scenes); }
  getScene(sceneIndex) {
    const scene = this.scenes[sceneIndex];
    if (!(sceneIndex >= 0 && sceneIndex < this.scenes.length)) throw new Error('SplatMesh::getScene() -> Invalid scene index.');
    return scene;
  }
  getSceneCount() { return this.scenes.length; }
  getSceneTransform(sceneIndex, outTransform) {             // :2019-2028
    const sc = this.getScene(sceneIndex);
    sc.updateTransform(this.dynamicMode);
    outTransform.copy(sc.transform);
  }
  getSplatBufferForSplat(globalIndex) { return this.getScene(this.globalSplatIndexToSceneIndexMap[globalIndex]).splatBuffer; }
  getSceneIndexForSplat(globalIndex) { return this.globalSplatIndexToSceneIndexMap[globalIndex]; }
  getSplatLocalIndex(globalIndex) { return this.globalSplatIndexToLocalSplatIndexMap[globalIndex]; }
  updateTransforms() { this.scenes.forEach((sc) => sc.updateTransform(this.dynamicMode)); this._scenesDirty = true; }
  fillTransformsArray(array) {                              // :1683-1699: 16 floats per scene; the slots of absent scenes become NaN, as there
    const flat = new Array(array.length);
    this.scenes.forEach((sc, k) => sc.transform.elements.forEach((value, j) => { flat[16 * k + j] = value; }));
    array.set(flat);
  }
  getSplatDataTextures() { return this.splatDataTextures; }
  setRenderer(renderer) { this.renderer = renderer; }
  freeIntermediateSplatData() {}                            // nothing is kept on the host
  updateVisibleRegionFadeDistance() {}                      // SceneRevealMode.Instant semantics: fadeInComplete = 1 (:1201-1226)
  computeDistancesOnGPU() { return Promise.resolve(true); } // subsumed by the device sort (gpuAcceleratedSort must be false)

  // ---- per-sort / per-frame ------------------------------------------------------------------------------------------
  updateRenderIndexes(globalIndexes, renderSplatCount) {    // :1228-1235
    if (renderSplatCount > 0 && this.firstRenderTime === -1) this.firstRenderTime = Date.now();
    if (this.core) this.core.updateRenderIndexes(globalIndexes, renderSplatCount);
  }
  updateUniforms(renderDimensions, cameraFocalLengthX, cameraFocalLengthY, orthographicMode, orthographicZoom, inverseFocalAdjustment) {   // :1248-1280
    if (this.getSplatCount() <= 0 || !this.core) return;
    this.core.updateUniforms({ x: renderDimensions.x * this.devicePixelRatio, y: renderDimensions.y * this.devicePixelRatio },
                             cameraFocalLengthX, cameraFocalLengthY, orthographicMode, orthographicZoom, inverseFocalAdjustment);
    if (this.dynamicMode || this.enableOptionalEffects) this._scenesDirty = true;
  }
  setSplatScale(splatScale = 1) { this.splatScale = splatScale; if (this.core) this.core.setSplatScale(splatScale); }
  getSplatScale() { return this.splatScale; }
  setPointCloudModeEnabled(enabled) { this.pointCloudModeEnabled = enabled; if (this.core) this.core.setPointCloudModeEnabled(enabled); }
  getPointCloudModeEnabled() { return this.pointCloudModeEnabled; }

  _uploadScenes(cameraPosition) {                           // the per-scene uniforms of updateUniforms (:1263-1276)
    const n = this.scenes.length;
    if (!(n > 1 || this.dynamicMode || this.enableOptionalEffects || (this.core.sphericalHarmonics8Bit && this.minSphericalHarmonicsDegree > 0))) return;
    const p = { sceneCount: n, transforms: new Float32Array(16 * n), invCamPos: new Float32Array(4 * n), opacity: new Float32Array(n),
                visible: new Uint32Array(n), sh8Min: new Float32Array(n), sh8Max: new Float32Array(n) };
    const inv = new THREE.Matrix4(), v = new THREE.Vector3();
    for (let i = 0; i < n; i++) {
      const scene = this.getScene(i);
      p.transforms.set(this.dynamicMode ? scene.transform.elements : new THREE.Matrix4().elements, 16 * i);
      inv.copy(scene.transform).invert();
      v.copy(cameraPosition).applyMatrix4(inv);
      p.invCamPos.set([v.x, v.y, v.z, 1], 4 * i);
      p.opacity[i] = Math.min(Math.max(scene.opacity, 0.0), 1.0);
      p.visible[i] = scene.visible ? 1 : 0;
      p.sh8Min[i] = scene.splatBuffer.minSphericalHarmonicsCoeff;
      p.sh8Max[i] = scene.splatBuffer.maxSphericalHarmonicsCoeff;
    }
    this.core.setScenes(p);
  }

  // The draw: renderer.render(splatMesh, camera) (src/Viewer.js:1616) reaches every object through onBeforeRender.
  // modelViewMatrix = camera.matrixWorldInverse * this.matrixWorld, projectionMatrix, cameraPosition: three's built-ins.
  onBeforeRender(renderer, scene, camera) { return this.renderFrame(camera); }

  // Drop-in mode (src/DropInViewer.js:34-42: the splat mesh is one object of the HOST's scene) and the Viewer's own threeScene
  // (src/Viewer.js:1610-1616: `renderer.render(this.threeScene, ...)` first, then the splat mesh with autoClear off).  The
  // reference's material is `depthTest: true, depthWrite: false` (SplatMaterial3D.js:72-73): splats behind what the other objects
  // drew are hidden, the rest blends over their colour.  A host hands that state over once per frame, after it has drawn its
  // opaque objects into a render target with a DepthTexture:
  //     renderer.readRenderTargetPixels(rt, 0, 0, w, h, colour)            -> Uint8Array RGBA8, row 0 = bottom
  //     depth: a FloatType DepthTexture copied out through a full-screen pass (or gl.readPixels(DEPTH_COMPONENT, FLOAT))
  //     splatMesh.setDestination({depth, colour, width: w, height: h, depthBits: 24})
  // and then draws `this.frame` (now the COMPLETE frame: splats over the scene) instead of blending it itself.  Pass null (or
  // nothing) to go back to splats alone over a cleared target.
  setDestination(dest) {
    this._destination = dest || null;
    if (!this.core) return;
    if (!dest) this.core.setDestination(null, null, 0, 0);
    else this.core.setDestination(dest.depth || null, dest.colour || null, dest.width, dest.height, dest.depthBits || 32);
  }
  // HIP-engine extra: draw as the browser's RGBA8 render target does (every channel rounded to 8 bits after every splat, back to
  // front: SplatMaterial3D.js:65-75 as a ROP executes it) instead of in fp32 rounded once: ~0.45 ms per 1080p frame of the garden
  // stand-in instead of 0.25; full = true walks every list to its end (~4 ms: verification).
  setRop8(enabled, full = false) {
    this._rop8 = enabled ? (full ? 2 : 1) : 0;
    if (this.core) this.core.setRop8(!!enabled, !!full);
  }
  renderFrame(camera, out) {
    if (!this.core || this.getSplatCount() <= 0) return null;
    const view = camera.matrixWorldInverse ? camera.matrixWorldInverse : new THREE.Matrix4().copy(camera.matrixWorld).invert();
    const modelView = new THREE.Matrix4().multiplyMatrices(view, this.matrixWorld);
    const position = new THREE.Vector3().setFromMatrixPosition(camera.matrixWorld);
    this.core.setCameraMatrices(modelView.elements, camera.projectionMatrix.elements, [position.x, position.y, position.z], view.elements);
    if (this._scenesDirty) { this._uploadScenes(position); this._scenesDirty = false; }
    const r = this.core.render(out);
    this.frame = { data: r.pixels, width: this.core.cam.width, height: this.core.cam.height, stats: r.stats };
    return this.frame;
  }

  // ---- splat tree (:231-280): built by the engine (on the device), exposed in the reference's shape ---------------------
  buildSplatTree(minAlphas = [], onSplatTreeIndexesUpload, onSplatTreeConstruction) {
    this.disposeSplatTree();
    return new Promise((resolve) => {
      const total = this.getSplatCount(true), splatCount = total;
      const centers = new Float32Array(total * 3), colors = new Uint8Array(total * 4);
      // getSplatCenter / getSplatColor of every splat (:239-244): scene transforms applied as for a static mesh, alpha filter
      this.fillSplatDataArrays(null, null, null, centers, null, null, this.dynamicMode ? false : true);
      const keep = new Uint8Array(splatCount);
      let dest = 0;
      for (const sc of this.scenes) {
        const s = this.scenes.indexOf(sc), buffer = sc.splatBuffer, n = buffer.getSplatCount();
        buffer.fillSplatColorArray(colors, 0, undefined, undefined, dest);
        const minAlpha = minAlphas[s] || 1;
        for (let i = 0; i < n; i++) keep[dest + i] = colors[4 * (dest + i) + 3] >= minAlpha ? 1 : 0;
        dest += n;
      }
      if (onSplatTreeIndexesUpload) onSplatTreeIndexesUpload(false);
      const handle = addon.treeCreate(this.core.ctx.handle, centers, keep, splatCount, 0, 8, 1000);      // maxDepth 8, 1000 per node (:236)
      if (onSplatTreeIndexesUpload) onSplatTreeIndexesUpload(true);
      if (onSplatTreeConstruction) onSplatTreeConstruction(false);
      this.splatTree = this.baseSplatTree = new HipSplatTree(handle, this);
      if (onSplatTreeConstruction) onSplatTreeConstruction(true);
      resolve();
    });
  }
  getSplatTree() { return this.splatTree; }
  onSplatTreeReady(callback) { this.onSplatTreeReadyCallback = callback; }
  disposeSplatTree() {
    if (this.baseSplatTree) this.baseSplatTree.dispose();
    this.splatTree = this.baseSplatTree = null;
  }
  disposeMeshData() { if (this.core) { this.core.dispose(); this.core = null; } }
  dispose() { this.disposeSplatTree(); this.disposeMeshData(); this.disposed = true; return Promise.resolve(); }
}

// SplatTree in the shape Viewer.gatherSceneNodesForSort walks (src/Viewer.js:1998-2059, src/splattree/SplatTree.js:275-318):
// subTrees[s].nodesWithIndexes[k] = {min, max, center: THREE.Vector3, data: {indexes}}.  The leaves come from gs_tree_read
// (built on the device, bit-identical to the reference's worker); `gather(...)` is the engine's own device-side gather for
// callers that skip the JS walk.
class HipSplatTree {
  constructor(handle, splatMesh) {
    this.handle = handle;
    this.splatMesh = splatMesh;
    this.maxDepth = 8;
    this.maxCentersPerNode = 1000;
    const t = addon.treeRead(handle), info = addon.treeInfo(handle);
    const nodes = [];
    for (let k = 0; k < t.depths.length; k++) {
      nodes.push({ min: new THREE.Vector3(t.bounds[6 * k], t.bounds[6 * k + 1], t.bounds[6 * k + 2]),
                   max: new THREE.Vector3(t.bounds[6 * k + 3], t.bounds[6 * k + 4], t.bounds[6 * k + 5]),
                   center: new THREE.Vector3(t.centers[3 * k], t.centers[3 * k + 1], t.centers[3 * k + 2]), depth: t.depths[k],
                   data: { indexes: t.indexes.subarray(t.offsets[k], t.offsets[k + 1]) }, children: [], id: k });
    }
    this.subTrees = [{ nodesWithIndexes: nodes, maxDepth: this.maxDepth, maxCentersPerNode: this.maxCentersPerNode }];
    this.leaves = info.allLeaves;
  }
  countLeaves() { return this.leaves; }
  visitLeaves(visitFunc) { for (const node of this.subTrees[0].nodesWithIndexes) visitFunc(node); }
  dispose() { if (this.handle) { addon.treeDestroy(this.handle); this.handle = null; } }
}

export { SplatMesh as SplatMeshHIP, HipSplatTree };
