// oracle/assets_ref.mjs — TEST INFRASTRUCTURE.  Runs the REFERENCE's own asset code, imported in place from
// /root/reference/src with 'three' resolved to oracle/three_min.mjs (oracle/three_loader.mjs):
//   INRIAV1PlyParser.parseToUncompressedSplatBuffer   (src/loaders/ply/INRIAV1PlyParser.js:114-232)  .ply -> level-0 buffer
//   INRIAV1PlyParser.parseToUncompressedSplatArray + SplatBuffer.generateFromUncompressedSplatArrays
//                                                     (src/loaders/SplatBuffer.js:1177-1326)  -> .ksplat bytes, levels 0/1/2
//   SplatBuffer.fillSplatCenterArray / fillSplatScaleRotationArray / fillSplatCovarianceArray / fillSplatColorArray /
//   fillSphericalHarmonicsArray                       (src/loaders/SplatBuffer.js:307-734)    -> the arrays the seams consume
// and dumps every buffer and array into <outdir> for oracle/make_golden_assets.py.
// usage: node --experimental-loader ./three_loader.mjs assets_ref.mjs <reference/src> <in.ply> <outdir> <shDegree> <minAlpha>
import fs from 'fs';
import path from 'path';
const [srcRoot, plyPath, outDir, degArg, alphaArg] = process.argv.slice(2);
const shDegree = parseInt(degArg, 10), minimumAlpha = parseInt(alphaArg, 10);
const run = async () => {
  const { SplatBuffer } = await import(path.join(srcRoot, 'loaders/SplatBuffer.js'));
  const { INRIAV1PlyParser } = await import(path.join(srcRoot, 'loaders/ply/INRIAV1PlyParser.js'));
  const THREE = await import('three');
  const buf = fs.readFileSync(plyPath);
  const ply = buf.buffer.slice(buf.byteOffset, buf.byteOffset + buf.byteLength);
  const manifest = { shDegree, minimumAlpha, buffers: {} };
  const dump = (name, typed) => fs.writeFileSync(path.join(outDir, name), Buffer.from(typed.buffer, typed.byteOffset, typed.byteLength));

  const fills = (tag, sb) => {
    const n = sb.getSplatCount(), deg = Math.min(shDegree, sb.getMinSphericalHarmonicsDegree());
    const ncoef = deg === 0 ? 0 : (deg === 1 ? 9 : 24);
    const level = sb.compressionLevel, shLevel = Math.max(1, level);          // SplatMesh.js:1064-1066
    const centers = new Float32Array(3 * n), scales = new Float32Array(3 * n), rotations = new Float32Array(4 * n);
    const cov32 = new Float32Array(6 * n), cov16 = new Uint16Array(6 * n), colors = new Uint8Array(4 * n);
    sb.fillSplatCenterArray(centers, undefined, undefined, undefined, 0);
    sb.fillSplatScaleRotationArray(scales, rotations, undefined, undefined, undefined, 0, 0, undefined);
    sb.fillSplatCovarianceArray(cov32, undefined, undefined, undefined, 0, 0);
    sb.fillSplatCovarianceArray(cov16, undefined, undefined, undefined, 0, 1);
    sb.fillSplatColorArray(colors, minimumAlpha, undefined, undefined, 0);
    dump(`${tag}_centers.f32`, centers); dump(`${tag}_scales.f32`, scales); dump(`${tag}_rotations.f32`, rotations);
    dump(`${tag}_cov.f32`, cov32); dump(`${tag}_cov.u16`, cov16); dump(`${tag}_rgba.u8`, colors);
    let shDiffIdentity = null;
    if (ncoef) {
      const sh = shLevel === 2 ? new Uint8Array(ncoef * n) : new Uint16Array(ncoef * n);
      sb.fillSphericalHarmonicsArray(sh, deg, undefined, undefined, undefined, 0, shLevel);
      dump(`${tag}_sh.${shLevel === 2 ? 'u8' : 'u16'}`, sh);
      // static mode hands an identity scene transform to the same call (SplatMesh.js:1873-1898): how many values differ?
      const shT = shLevel === 2 ? new Uint8Array(ncoef * n) : new Uint16Array(ncoef * n);
      sb.fillSphericalHarmonicsArray(shT, deg, new THREE.Matrix4(), undefined, undefined, 0, shLevel);
      shDiffIdentity = 0;
      for (let i = 0; i < sh.length; i++) if (sh[i] !== shT[i]) shDiffIdentity++;
    }
    manifest.buffers[tag] = { splatCount: n, shDegree: deg, compressionLevel: level, shLevel, ncoef,
                              minSh: sb.minSphericalHarmonicsCoeff, maxSh: sb.maxSphericalHarmonicsCoeff,
                              sceneCenter: sb.sceneCenter.toArray(), shValuesChangedByIdentityTransform: shDiffIdentity };
    dump(`${tag}.ksplat`, new Uint8Array(sb.bufferData));
  };

  // 1. .ply -> file-order level-0 buffer (the progressive direct-to-buffer path)
  fills('ply', INRIAV1PlyParser.parseToUncompressedSplatBuffer(ply, shDegree));
  // 2. .ply -> rows -> the reference's own .ksplat writer at every compression level
  const rows = INRIAV1PlyParser.parseToUncompressedSplatArray(ply, shDegree);
  const flat = new Float64Array(rows.splatCount * rows.splats[0].length);
  rows.splats.forEach((s, i) => flat.set(s, i * s.length));
  dump('rows.f64', flat);
  manifest.rowLength = rows.splats[0].length;
  for (const level of [0, 1, 2]) {
    fills(`gen${level}`, SplatBuffer.generateFromUncompressedSplatArrays([rows], minimumAlpha, level, new THREE.Vector3()));
  }
  fs.writeFileSync(path.join(outDir, 'manifest.json'), JSON.stringify(manifest));
  console.log(JSON.stringify({ ok: true, splats: rows.splatCount }));
};
run().catch((e) => { console.error(String(e && e.stack || e)); process.exit(1); });
