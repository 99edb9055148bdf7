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
  constructor(splatRenderMode = SplatRenderMode.ThreeD, dynamicMode = false, enableOptionalEffects = false,
              halfPrecisionCovariancesOnGPU = false, devicePixelRatio = 1, enableDistancesComputationOnGPU = true,
              integerBasedDistancesComputation = false, antialiased = false, maxScreenSpaceSplatSize = 1024, logLevel = 0,
              sphericalHarmonicsDegree = 0, sceneFadeInRateMultiplier = 1.0, kernel2DSize = 0.3) {
    if (splatRenderMode !== SplatRenderMode.ThreeD) throw new Error('SplatMesh (HIP): only SplatRenderMode.ThreeD is implemented');
    this.renderer = undefined;
    this.splatRenderMode = splatRenderMode;
    this.dynamicMode = dynamicMode;
    this.enableOptionalEffects = enableOptionalEffects;
    this.halfPrecisionCovariancesOnGPU = halfPrecisionCovariancesOnGPU;
    this.devicePixelRatio = devicePixelRatio;
    this.enableDistancesComputationOnGPU = false;           // the device sort keys the splats itself (see the header)
    this.integerBasedDistancesComputation = integerBasedDistancesComputation;
    this.antialiased = antialiased;
    this.kernel2DSize = kernel2DSize;
    this.maxScreenSpaceSplatSize = maxScreenSpaceSplatSize;
    this.logLevel = logLevel;
    this.sphericalHarmonicsDegree = sphericalHarmonicsDegree;
    this.minSphericalHarmonicsDegree = 0;
    this.sceneFadeInRateMultiplier = sceneFadeInRateMultiplier;
    this.scenes = [];
    this.sceneOptions = undefined;
    this.splatTree = null;
    this.baseSplatTree = null;
    this.splatDataTextures = {};
    this.globalSplatIndexToLocalSplatIndexMap = [];
    this.globalSplatIndexToSceneIndexMap = [];
    this.lastBuildSplatCount = 0;
    this.lastBuildScenes = [];
    this.lastBuildMaxSplatCount = 0;
    this.lastBuildSceneCount = 0;
    this.firstRenderTime = -1;
    this.finalBuild = false;
    this.splatScale = 1.0;
    this.pointCloudModeEnabled = false;
    this.disposed = false;
    this.visible = false;
    this.frustumCulled = false;
    this.matrixWorld = new THREE.Matrix4();                 // Object3D.matrixWorld (src/Viewer.js:1891 multiplies by it)
    this.core = null;                                       // the device-side mesh (gs_mesh_*)
    this.frame = null;
    this.onSplatTreeReadyCallback = null;
    this.shCompressionLevel = 1;
  }

  // ---- statics, as the reference (:173-228, :1311-1341) --------------------------------------------------------------
  static buildScenes(parentObject, splatBuffers, sceneOptions) {
    const scenes = [];
    scenes.length = splatBuffers.length;
    for (let i = 0; i < splatBuffers.length; i++) {
      const options = sceneOptions[i] || {};
      const position = new THREE.Vector3().fromArray(options['position'] || [0, 0, 0]);
      const rotation = new THREE.Quaternion().fromArray(options['rotation'] || [0, 0, 0, 1]);
      const scale = new THREE.Vector3().fromArray(options['scale'] || [1, 1, 1]);
      scenes[i] = SplatMesh.createScene(splatBuffers[i], position, rotation, scale, options.splatAlphaRemovalThreshold || 1,
                                        options.opacity, options.visible);
    }
    return scenes;
  }
  static createScene(splatBuffer, position, rotation, scale, minimumAlpha, opacity = 1.0, visible = true) {
    return new HipSplatScene(splatBuffer, position, rotation, scale, minimumAlpha, opacity, visible);
  }
  static buildSplatIndexMaps(splatBuffers) {
    const localSplatIndexMap = [], sceneIndexMap = [];
    let total = 0;
    for (let s = 0; s < splatBuffers.length; s++) {
      const maxSplatCount = splatBuffers[s].getMaxSplatCount();
      for (let i = 0; i < maxSplatCount; i++) { localSplatIndexMap[total] = i; sceneIndexMap[total] = s; total++; }
    }
    return { localSplatIndexMap, sceneIndexMap };
  }
  static getTotalSplatCountForScenes(scenes) {
    let n = 0;
    for (const scene of scenes) if (scene && scene.splatBuffer) n += scene.splatBuffer.getSplatCount();
    return n;
  }
  static getTotalSplatCountForSplatBuffers(splatBuffers) { let n = 0; for (const b of splatBuffers) n += b.getSplatCount(); return n; }
  static getTotalMaxSplatCountForScenes(scenes) {
    let n = 0;
    for (const scene of scenes) if (scene && scene.splatBuffer) n += scene.splatBuffer.getMaxSplatCount();
    return n;
  }
  static getTotalMaxSplatCountForSplatBuffers(splatBuffers) { let n = 0; for (const b of splatBuffers) n += b.getMaxSplatCount(); return n; }

  // ---- build (:306-405) ----------------------------------------------------------------------------------------------
  build(splatBuffers, sceneOptions, keepSceneTransforms = true, finalBuild = false, onSplatTreeIndexesUpload, onSplatTreeConstruction,
        preserveVisibleRegion = true) {
    this.sceneOptions = sceneOptions;
    this.finalBuild = finalBuild;
    const maxSplatCount = SplatMesh.getTotalMaxSplatCountForSplatBuffers(splatBuffers);
    const newScenes = SplatMesh.buildScenes(this, splatBuffers, sceneOptions);
    if (keepSceneTransforms) {
      for (let i = 0; i < this.scenes.length && i < newScenes.length; i++) newScenes[i].copyTransformData(this.getScene(i));
    }
    this.scenes = newScenes;
    let minDegree = 3;
    for (const splatBuffer of splatBuffers) minDegree = Math.min(minDegree, splatBuffer.getMinSphericalHarmonicsDegree());
    this.minSphericalHarmonicsDegree = Math.min(minDegree, this.sphericalHarmonicsDegree);

    let splatBuffersChanged = splatBuffers.length !== this.lastBuildScenes.length;
    for (let i = 0; !splatBuffersChanged && i < splatBuffers.length; i++) {
      if (splatBuffers[i] !== this.lastBuildScenes[i].splatBuffer) splatBuffersChanged = true;
    }
    let isUpdateBuild = true;
    if (this.scenes.length !== 1 || this.lastBuildSceneCount !== this.scenes.length || this.lastBuildMaxSplatCount !== maxSplatCount ||
        splatBuffersChanged) isUpdateBuild = false;
    if (!isUpdateBuild) {
      this.lastBuildScenes = [];
      this.lastBuildSplatCount = 0;
      this.lastBuildMaxSplatCount = 0;
      this.disposeMeshData();
      const indexMaps = SplatMesh.buildSplatIndexMaps(splatBuffers);
      this.globalSplatIndexToLocalSplatIndexMap = indexMaps.localSplatIndexMap;
      this.globalSplatIndexToSceneIndexMap = indexMaps.sceneIndexMap;
    }
    this.updateTransforms();                                // the scenes' matrices exist before the first data fill
    const splatBufferSplatCount = this.getSplatCount(true);
    const dataUpdateResults = this.refreshGPUDataFromSplatBuffers(isUpdateBuild);
    for (let i = 0; i < this.scenes.length; i++) this.lastBuildScenes[i] = this.scenes[i];
    this.lastBuildSplatCount = splatBufferSplatCount;
    this.lastBuildMaxSplatCount = this.getMaxSplatCount();
    this.lastBuildSceneCount = this.scenes.length;
    if (finalBuild && this.scenes.length > 0) {
      this.buildSplatTree(sceneOptions.map((options) => options.splatAlphaRemovalThreshold || 1), onSplatTreeIndexesUpload,
                          onSplatTreeConstruction).then(() => {
        if (this.onSplatTreeReadyCallback) this.onSplatTreeReadyCallback(this.splatTree);
        this.onSplatTreeReadyCallback = null;
      });
    }
    this.visible = (this.scenes.length > 0);
    return dataUpdateResults;
  }

  // setupDataTextures' decisions (:637-690, 1060-1090) without the textures: covariance as fp16 when asked for, SH as fp16
  // (compression level <= 1) or uint8 (level 2: kept 8-bit end to end, like the reference's sphericalHarmonics8BitMode)
  getMaximumSplatBufferCompressionLevel() {
    let level;
    for (let i = 0; i < this.scenes.length; i++) {
      const l = this.getScene(i).splatBuffer.compressionLevel;
      if (i === 0 || l > level) level = l;
    }
    return level;
  }
  getMinimumSplatBufferCompressionLevel() {
    let level;
    for (let i = 0; i < this.scenes.length; i++) {
      const l = this.getScene(i).splatBuffer.compressionLevel;
      if (i === 0 || l < level) level = l;
    }
    return level;
  }
  getTargetCovarianceCompressionLevel() { return this.halfPrecisionCovariancesOnGPU ? 1 : 0; }
  getTargetSphericalHarmonicsCompressionLevel() { return Math.max(1, this.getMaximumSplatBufferCompressionLevel()); }

  refreshGPUDataFromSplatBuffers(sinceLastBuildOnly) {      // :588-609
    const splatCount = this.getSplatCount(true);
    this.refreshDataTexturesFromSplatBuffers(sinceLastBuildOnly);
    const updateStart = sinceLastBuildOnly ? this.lastBuildSplatCount : 0;
    const { centers, sceneIndexes } = this.getDataForDistancesComputation(updateStart, splatCount - 1);
    return { 'from': updateStart, 'to': splatCount - 1, 'count': splatCount - updateStart, 'centers': centers, 'sceneIndexes': sceneIndexes };
  }

  // :621-635 + :900-1058: the splat buffers' own fill methods produce the arrays; they go to device planes instead of
  // padded data textures
  refreshDataTexturesFromSplatBuffers(sinceLastBuildOnly) {
    const splatCount = this.getSplatCount(true);
    const maxSplatCount = this.getMaxSplatCount();
    const fromSplat = sinceLastBuildOnly ? this.lastBuildSplatCount : 0;
    const toSplat = splatCount - 1;
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
    const scaleOverride = new THREE.Vector3();
    scaleOverride.x = undefined; scaleOverride.y = undefined; scaleOverride.z = undefined;
    const tempTransform = new THREE.Matrix4();
    let startSceneIndex = 0, endSceneIndex = this.scenes.length - 1;
    if (sceneIndex !== undefined && sceneIndex !== null && sceneIndex >= 0 && sceneIndex <= this.scenes.length) {
      startSceneIndex = sceneIndex; endSceneIndex = sceneIndex;
    }
    for (let i = startSceneIndex; i <= endSceneIndex; i++) {
      if (applySceneTransform === undefined || applySceneTransform === null) applySceneTransform = this.dynamicMode ? false : true;
      const scene = this.getScene(i);
      const splatBuffer = scene.splatBuffer;
      let sceneTransform;
      if (applySceneTransform) { this.getSceneTransform(i, tempTransform); sceneTransform = tempTransform; }
      if (covariances) splatBuffer.fillSplatCovarianceArray(covariances, sceneTransform, srcStart, srcEnd, destStart, covarianceCompressionLevel);
      if (scales || rotations) {
        if (!scales || !rotations) throw new Error('SplatMesh::fillSplatDataArrays() -> "scales" and "rotations" must both be valid.');
        splatBuffer.fillSplatScaleRotationArray(scales, rotations, sceneTransform, srcStart, srcEnd, destStart, scaleRotationCompressionLevel, scaleOverride);
      }
      if (centers) splatBuffer.fillSplatCenterArray(centers, sceneTransform, srcStart, srcEnd, destStart);
      if (colors) splatBuffer.fillSplatColorArray(colors, scene.minimumAlpha, srcStart, srcEnd, destStart);
      if (sphericalHarmonics) {
        splatBuffer.fillSphericalHarmonicsArray(sphericalHarmonics, this.minSphericalHarmonicsDegree, sceneTransform, srcStart, srcEnd, destStart,
                                                sphericalHarmonicsCompressionLevel);
      }
      destStart += splatBuffer.getSplatCount();
    }
  }

  getIntegerCenters(start, end, padFour = false) {          // :1912-1926
    const splatCount = end - start + 1;
    const floatCenters = new Float32Array(splatCount * 3);
    this.fillSplatDataArrays(null, null, null, floatCenters, null, null, undefined, undefined, undefined, undefined, start);
    const componentCount = padFour ? 4 : 3;
    const intCenters = new Int32Array(splatCount * componentCount);
    for (let i = 0; i < splatCount; i++) {
      for (let t = 0; t < 3; t++) intCenters[i * componentCount + t] = Math.round(floatCenters[i * 3 + t] * 1000.0);
      if (padFour) intCenters[i * componentCount + 3] = 1000;
    }
    return intCenters;
  }
  getFloatCenters(start, end, padFour = false) {            // :1935-1948
    const splatCount = end - start + 1;
    const floatCenters = new Float32Array(splatCount * 3);
    this.fillSplatDataArrays(null, null, null, floatCenters, null, null, undefined, undefined, undefined, undefined, start);
    if (!padFour) return floatCenters;
    const padded = new Float32Array(splatCount * 4);
    for (let i = 0; i < splatCount; i++) {
      for (let t = 0; t < 3; t++) padded[i * 4 + t] = floatCenters[i * 3 + t];
      padded[i * 4 + 3] = 1.0;
    }
    return padded;
  }
  getSceneIndexes(start, end) {                             // :1667-1677
    const sceneIndexes = new Uint32Array(end - start + 1);
    for (let i = start; i <= end; i++) sceneIndexes[i] = this.globalSplatIndexToSceneIndexMap[i];
    return sceneIndexes;
  }

  // ---- counts, scenes, transforms ------------------------------------------------------------------------------------
  getSplatCount(includeSinceLastBuild = false) {
    return includeSinceLastBuild ? SplatMesh.getTotalSplatCountForScenes(this.scenes) : this.lastBuildSplatCount;
  }
  getMaxSplatCount() { return SplatMesh.getTotalMaxSplatCountForScenes(this.scenes); }
  getScene(sceneIndex) {
    if (sceneIndex < 0 || sceneIndex >= this.scenes.length) throw new Error('SplatMesh::getScene() -> Invalid scene index.');
    return this.scenes[sceneIndex];
  }
  getSceneCount() { return this.scenes.length; }
  getSceneTransform(sceneIndex, outTransform) {             // :2019-2028
    const scene = this.getScene(sceneIndex);
    scene.updateTransform(this.dynamicMode);
    outTransform.copy(scene.transform);
  }
  getSplatBufferForSplat(globalIndex) { return this.getScene(this.globalSplatIndexToSceneIndexMap[globalIndex]).splatBuffer; }
  getSceneIndexForSplat(globalIndex) { return this.globalSplatIndexToSceneIndexMap[globalIndex]; }
  getSplatLocalIndex(globalIndex) { return this.globalSplatIndexToLocalSplatIndexMap[globalIndex]; }
  updateTransforms() { for (let i = 0; i < this.scenes.length; i++) this.getScene(i).updateTransform(this.dynamicMode); this._scenesDirty = true; }
  fillTransformsArray(array) {                              // :1683-1699
    const temp = [];
    temp.length = array.length;
    for (let i = 0; i < this.scenes.length; i++) {
      const e = this.getScene(i).transform.elements;
      for (let j = 0; j < 16; j++) temp[i * 16 + j] = e[j];
    }
    array.set(temp);
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
    return new Promise((resolve) => {
      this.disposeSplatTree();
      const splatCount = this.getSplatCount(true);
      const centers = new Float32Array(splatCount * 3), colors = new Uint8Array(splatCount * 4);
      // getSplatCenter / getSplatColor of every splat (:239-244): scene transforms applied as for a static mesh, alpha filter
      this.fillSplatDataArrays(null, null, null, centers, null, null, this.dynamicMode ? false : true);
      const keep = new Uint8Array(splatCount);
      let dest = 0;
      for (let s = 0; s < this.scenes.length; s++) {
        const buffer = this.getScene(s).splatBuffer, n = buffer.getSplatCount();
        buffer.fillSplatColorArray(colors, 0, undefined, undefined, dest);
        const minAlpha = minAlphas[s] || 1;
        for (let i = 0; i < n; i++) keep[dest + i] = colors[4 * (dest + i) + 3] >= minAlpha ? 1 : 0;
        dest += n;
      }
      if (onSplatTreeIndexesUpload) onSplatTreeIndexesUpload(false);
      const handle = addon.treeCreate(this.core.ctx.handle, centers, keep, splatCount, 0, 8, 1000);      // maxDepth 8, 1000 per node (:236)
      if (onSplatTreeIndexesUpload) onSplatTreeIndexesUpload(true);
      if (onSplatTreeConstruction) onSplatTreeConstruction(false);
      this.baseSplatTree = new HipSplatTree(handle, this);
      this.splatTree = this.baseSplatTree;
      if (onSplatTreeConstruction) onSplatTreeConstruction(true);
      resolve();
    });
  }
  getSplatTree() { return this.splatTree; }
  onSplatTreeReady(callback) { this.onSplatTreeReadyCallback = callback; }
  disposeSplatTree() {
    if (this.baseSplatTree) this.baseSplatTree.dispose();
    this.splatTree = null;
    this.baseSplatTree = null;
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
