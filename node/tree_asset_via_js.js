// Test driver for the cull / asset bindings of the addon.
//   node tree_asset_via_js.js tree <centers.f32> <maxDepth> <maxCentersPerNode> [gather <modelView.f64> <w> <h> <out.u32>]
//   node tree_asset_via_js.js asset <file> <format 1|2> <maxShDegree> <out.bin>
'use strict';
const fs = require('fs');
const gs = require('./gsplat.js');
const a = gs.addon;
const [mode, ...rest] = process.argv.slice(2);
const f32 = (p) => { const b = fs.readFileSync(p); return new Float32Array(b.buffer.slice(b.byteOffset, b.byteOffset + b.byteLength)); };
if (mode === 'tree') {
  const centers = f32(rest[0]);
  const n = centers.length / 3;
  const gather = rest[3] === 'gather';
  const ctx = gather ? a.contextCreate(0) : null;
  const tree = a.treeCreate(ctx, centers, null, n, 0, parseInt(rest[1]), parseInt(rest[2]));
  const info = a.treeInfo(tree);
  if (gather) {
    const mvb = fs.readFileSync(rest[4]);
    const mv = new Float64Array(mvb.buffer.slice(mvb.byteOffset, mvb.byteOffset + 128));
    const out = new Uint32Array(info.splats);
    info.renderCount = a.treeGather(tree, mv, 50.0, parseFloat(rest[5]), parseFloat(rest[6]), 0, null, out);
    fs.writeFileSync(rest[7], Buffer.from(out.buffer, 0, 4 * info.renderCount));
  }
  a.treeDestroy(tree);
  if (ctx) a.contextDestroy(ctx);
  console.log(JSON.stringify(info));
} else if (mode === 'asset') {
  const bytes = fs.readFileSync(rest[0]);
  const r = a.assetLoad(new Uint8Array(bytes), parseInt(rest[1]), parseInt(rest[2]), 1, 0);
  const parts = [r.centers, r.cov, r.rgba].concat(r.sh ? [r.sh] : []).map((t) => Buffer.from(t.buffer, t.byteOffset, t.byteLength));
  fs.writeFileSync(rest[3], Buffer.concat(parts));
  console.log(JSON.stringify({ splatCount: r.splatCount, shDegree: r.shDegree, compressionLevel: r.compressionLevel, shLevel: r.shLevel }));
}
