// oracle/wasm_ref.js — TEST INFRASTRUCTURE.  Runs the REFERENCE's own prebuilt WebAssembly sorter
// (/root/reference/src/worker/sorter_no_simd_non_shared.wasm) under Node, laying memory out as
// src/worker/SortWorker.js:125-178 does, so oracle/make_golden.py can record known answers.
// usage: node wasm_ref.js <sorter.wasm> <in.bin> <out.bin> [repeat]
'use strict';
const fs = require('fs');
const [wasmPath, inPath, outPath, repeatArg] = process.argv.slice(2);
const repeat = parseInt(repeatArg || '1', 10);
const buf = fs.readFileSync(inPath);
const hdr = new Uint32Array(buf.buffer, buf.byteOffset, 8);
const [n, renderCount, sortCount, range, useInt, dynamic, usePre] = hdr;
let off = 32;
const take = (bytes) => { const b = buf.slice(off, off + bytes); off += bytes; return b; };
const indexes = take(4 * renderCount), centers = take(16 * n), mvp = take(64);
const sceneIdx = dynamic ? take(4 * n) : null, transforms = dynamic ? take(32 * 64) : null;
const pre = usePre ? take(4 * n) : null;

const page = 65536;
const sizes = { idx: 4 * n, centers: 16 * n, mvp: 64, pre: 4 * n, mapped: 4 * n, freq: 8 * range, sorted: 4 * n,
                scene: dynamic ? 4 * n : 0, tr: dynamic ? 32 * 64 : 0 };
let total = 32 * page; for (const k in sizes) total += sizes[k];
const pages = Math.floor(total / page) + 1;
const memory = new WebAssembly.Memory({ initial: pages, maximum: pages });
const o = {}; let cur = 0;
for (const k of ['idx', 'centers', 'mvp', 'pre', 'mapped', 'freq', 'sorted', 'scene', 'tr']) { o[k] = cur; cur += sizes[k]; }
const mem8 = new Uint8Array(memory.buffer);
mem8.set(indexes, o.idx); mem8.set(centers, o.centers); mem8.set(mvp, o.mvp);
if (dynamic) { mem8.set(sceneIdx, o.scene); mem8.set(transforms, o.tr); }
if (usePre) mem8.set(pre, o.pre);

const imports = { env: { memory: memory, __memory_base: 0, __table_base: 0,
  __indirect_function_table: new WebAssembly.Table({ initial: 0, element: 'anyfunc' }),
  __stack_pointer: new WebAssembly.Global({ value: 'i32', mutable: true }, cur + 16 * page) } };
WebAssembly.instantiate(fs.readFileSync(wasmPath), imports).then(({ instance }) => {
  const freq = new Uint32Array(memory.buffer, o.freq, range);
  let best = Infinity, sum = 0;
  for (let r = 0; r < repeat; r++) {
    freq.fill(0);                                              // SortWorker.js:53-55
    const t0 = process.hrtime.bigint();
    instance.exports.sortIndexes(o.idx, o.centers, o.pre, o.mapped, o.freq, o.mvp, o.sorted, o.scene, o.tr,
                                 range, sortCount, renderCount, n, usePre, useInt, dynamic);
    const dt = Number(process.hrtime.bigint() - t0) / 1e6;
    if (dt < best) best = dt;
    if (r > 0 || repeat === 1) sum += dt;                      // the first run warms the instance
  }
  fs.writeFileSync(outPath, Buffer.from(memory.buffer, o.sorted, 4 * renderCount));
  console.log(JSON.stringify({ ms: best, ms_mean: sum / Math.max(repeat - 1, 1), repeat: repeat, node: process.version, n: n, renderCount: renderCount, sortCount: sortCount }));
}).catch((e) => { console.error(String(e)); process.exit(1); });
