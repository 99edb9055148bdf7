// Draws one frame through the JS shim's SplatMeshHIP (test driver).  usage: node render_via_js.js <in.bin> <out.bin>
// in.bin: uint32 header {n, shDegree, width, height, flags(1 ortho | 2 fade | 4 effects), sceneCount, 0, 0}, then
// centers F32[3n], cov F32[6n], rgba U8[4n], sh U16[ncoef*n], order U32[n], sceneIdx U32[n], modelView F32[16], proj F32[16],
// camPos F32[3], focal F32[2], orthoZoom F32[1], sceneCenter F32[3], fadeStart F32[1], opacity F32[sceneCount], visible U32[sceneCount]
// and, with flag 8 (a destination: drop-in mode's depth test and colour), depth F32[width*height], colour U8[4*width*height];
// flag 16: draw in GS_DRAW_ROP8 (the reference's RGBA8 target, rounded after every splat), flag 32: GS_DRAW_ROP8_FULL
'use strict';
const fs = require('fs');
const gs = require('./gsplat.js');
const [inPath, outPath] = process.argv.slice(2);
const buf = fs.readFileSync(inPath);
const ab = buf.buffer.slice(buf.byteOffset, buf.byteOffset + buf.byteLength);
const [n, shDegree, width, height, flags, sceneCount] = new Uint32Array(ab, 0, 8);
let off = 32;
const take = (Type, count) => { const a = new Type(ab.slice(off, off + count * Type.BYTES_PER_ELEMENT)); off += count * Type.BYTES_PER_ELEMENT; return a; };
const ncoef = [0, 9, 24][shDegree];
const centers = take(Float32Array, 3 * n), cov = take(Float32Array, 6 * n), rgba = take(Uint8Array, 4 * n);
const sh = take(Uint16Array, ncoef * n), order = take(Uint32Array, n), sceneIdx = take(Uint32Array, n);
const modelView = take(Float32Array, 16), proj = take(Float32Array, 16), camPos = take(Float32Array, 3), focal = take(Float32Array, 2);
const orthoZoom = take(Float32Array, 1)[0], sceneCenter = take(Float32Array, 3), fadeStart = take(Float32Array, 1)[0];
const opacity = take(Float32Array, sceneCount), visible = take(Uint32Array, sceneCount);
const dstDepth = (flags & 8) ? take(Float32Array, width * height) : null, dstColour = (flags & 8) ? take(Uint8Array, 4 * width * height) : null;
const mesh = new gs.SplatMeshHIP(n, { sphericalHarmonicsDegree: shDegree, enableOptionalEffects: !!(flags & 4) });
mesh.build(centers, cov, rgba, ncoef ? sh : null);
if (sceneCount > 1) mesh.setSceneIndexes(sceneIdx);
if (flags & 4) mesh.setScenes({ sceneCount, opacity, visible });
if (flags & 2) mesh.setFadeIn(sceneCenter, fadeStart);
mesh.updateRenderIndexes(order, n);
mesh.updateUniforms({ x: width, y: height }, focal[0], focal[1], !!(flags & 1), orthoZoom, 1.0);
mesh.setCameraMatrices(modelView, proj, camPos);
if (flags & 8) mesh.setDestination(dstDepth, dstColour, width, height, 24);
if (flags & 48) mesh.setRop8(true, !!(flags & 32));
const { pixels, stats } = mesh.render();
fs.writeFileSync(outPath, Buffer.from(pixels.buffer, pixels.byteOffset, pixels.byteLength));
// the multi-GPU entry points with a group of one: the same frame through gs_group_render_gather
const group = new gs.StripGroup(null, 1, 0);
const viaGroup = new Uint8Array(width * height * 4);
mesh.renderStrip(group, new Uint32Array([0]), new Uint32Array([height]), 0, viaGroup);
stats.stripIdentical = Buffer.compare(Buffer.from(viaGroup.buffer), Buffer.from(pixels.buffer, pixels.byteOffset, pixels.byteLength)) === 0;
group.dispose();
console.log(JSON.stringify(stats));
mesh.dispose();
