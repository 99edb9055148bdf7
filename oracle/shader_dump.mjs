// oracle/shader_dump.mjs — TEST INFRASTRUCTURE.  Asks the REFERENCE's own material builder for its GLSL: imports
// /root/reference/src/splatmesh/SplatMaterial3D.js in place ('three' -> oracle/three_min.mjs) and calls
// SplatMaterial3D.build(dynamicMode, enableOptionalEffects, antialiased, maxScreenSpaceSplatSize, splatScale,
// pointCloudModeEnabled, maxSphericalHarmonicsDegree, kernel2DSize) for every permutation asked for; the vertex / fragment
// shader strings go to <outdir>/<name>.vert / .frag (scratch files of oracle/make_golden_raster.py, never committed).
// usage: node --experimental-loader ./three_loader.mjs shader_dump.mjs <reference/src> <outdir> <perms.json>
import fs from 'fs';
import path from 'path';
const [srcRoot, outDir, permsPath] = process.argv.slice(2);
const run = async () => {
  const { SplatMaterial3D } = await import(path.join(srcRoot, 'splatmesh/SplatMaterial3D.js'));
  const perms = JSON.parse(fs.readFileSync(permsPath, 'utf8'));
  for (const p of perms) {
    const m = SplatMaterial3D.build(!!p.dynamicMode, !!p.enableOptionalEffects, !!p.antialiased, p.maxScreenSpaceSplatSize,
                                    p.splatScale, !!p.pointCloudModeEnabled, p.maxSphericalHarmonicsDegree, p.kernel2DSize);
    fs.writeFileSync(path.join(outDir, p.name + '.vert'), m.vertexShader);
    fs.writeFileSync(path.join(outDir, p.name + '.frag'), m.fragmentShader);
    fs.writeFileSync(path.join(outDir, p.name + '.state.json'), JSON.stringify({ transparent: m.transparent, blending: m.blending,
      depthTest: m.depthTest, depthWrite: m.depthWrite, alphaTest: m.alphaTest, uniforms: Object.keys(m.uniforms) }));
  }
  console.log(JSON.stringify({ ok: true, count: perms.length }));
};
run().catch((e) => { console.error(String(e && e.stack || e)); process.exit(1); });
