// bench_c1.js — BASELINE.json configs[0] (bonsai, integer sort only): the same {centers} / {sort} messages go to the
// reference's prebuilt WASM sorter (instantiated and laid out as /root/reference/src/worker/SortWorker.js:125-178 does,
// timed as :53-60) and to createSortWorker of gsplat.js (the HIP engine through the N-API addon); the two sortedIndexes
// arrays must be identical.  Driven by `bench.py --config C1`.
// usage: node bench_c1.js <centers.bin int32x4xn> <mvp.bin f64x16> <n> <steps> <warmup> <sorter.wasm | ->
'use strict';
const fs = require('fs');
const gs = require('./gsplat.js');
const [centersPath, mvpPath, nArg, stepsArg, warmupArg, wasmPath] = process.argv.slice(2);
const n = parseInt(nArg, 10), steps = parseInt(stepsArg, 10), warmup = parseInt(warmupArg, 10);
const cbuf = fs.readFileSync(centersPath);
const centers = cbuf.buffer.slice(cbuf.byteOffset, cbuf.byteOffset + cbuf.byteLength);
const mbuf = fs.readFileSync(mvpPath);
const mvp64 = new Float64Array(mbuf.buffer.slice(mbuf.byteOffset, mbuf.byteOffset + 128));
const identity = new Uint32Array(n);
for (let i = 0; i < n; i++) identity[i] = i;                         // Viewer.js:2061-2073: no tree -> identity list
const range = 1 << 16;
const ms = (t0) => Number(process.hrtime.bigint() - t0) / 1e6;

async function runWasm() {
  if (!wasmPath || wasmPath === '-' || !fs.existsSync(wasmPath)) return null;
  const page = 65536;
  const sizes = { idx: 4 * n, centers: 16 * n, mvp: 64, pre: 4 * n, mapped: 4 * n, freq: 8 * range, sorted: 4 * n };
  let total = 32 * page; for (const k in sizes) total += sizes[k];
  const pages = Math.floor(total / page) + 1;
  const memory = new WebAssembly.Memory({ initial: pages, maximum: pages });
  const o = {}; let cur = 0;
  for (const k of ['idx', 'centers', 'mvp', 'pre', 'mapped', 'freq', 'sorted']) { o[k] = cur; cur += sizes[k]; }
  new Uint8Array(memory.buffer).set(new Uint8Array(centers), o.centers);
  new Uint32Array(memory.buffer, o.idx, n).set(identity);
  new Float32Array(memory.buffer, o.mvp, 16).set(mvp64);               // fp64 -> fp32, SortWorker.js:54
  const imports = { env: { memory: memory, __memory_base: 0, __table_base: 0,
    __indirect_function_table: new WebAssembly.Table({ initial: 0, element: 'anyfunc' }),
    __stack_pointer: new WebAssembly.Global({ value: 'i32', mutable: true }, cur + 16 * page) } };
  const { instance } = await WebAssembly.instantiate(fs.readFileSync(wasmPath), imports);
  const freq = new Uint32Array(memory.buffer, o.freq, range);
  const reps = Math.max(3, Math.min(steps, 30));
  let sum = 0;
  for (let r = 0; r < reps + 1; r++) {
    const t0 = process.hrtime.bigint();
    freq.fill(0);                                                      // SortWorker.js:53-55 (inside its sortTime)
    instance.exports.sortIndexes(o.idx, o.centers, o.pre, o.mapped, o.freq, o.mvp, o.sorted, 0, 0, range, n, n, n, false, true, false);
    if (r > 0) sum += ms(t0);                                          // first run warms the instance
  }
  return { ms: sum / reps, reps, sorted: new Uint32Array(memory.buffer.slice(o.sorted, o.sorted + 4 * n)) };
}

function runHip() {
  return new Promise((resolve) => {
    const worker = gs.createSortWorker(n, false, false, true, false, 16);
    const sortMsg = () => ({ sort: { modelViewProj: Array.from(mvp64), splatRenderCount: n, splatSortCount: n,
                                     usePrecomputedDistances: false, indexesToSort: identity, transforms: null } });
    let done = 0, t0 = null, wall = 0, dev = 0, last = null;
    worker.onmessage = (e) => {
      if (e.data.sortSetupPhase1Complete) {
        worker.postMessage({ centers: centers, sceneIndexes: null, range: { from: 0, to: n - 1, count: n } });
        t0 = process.hrtime.bigint();
        worker.postMessage(sortMsg());
      } else if (e.data.sortDone) {
        if (done >= warmup) { wall += ms(t0); dev += e.data.sortTime; }
        last = e.data.sortedIndexes;
        done++;
        if (done < warmup + steps) { t0 = process.hrtime.bigint(); worker.postMessage(sortMsg()); }
        else { worker.terminate(); resolve({ ms: wall / steps, device_ms: dev / steps, sorted: last }); }
      }
    };
  });
}

(async () => {
  const hip = await runHip();
  const wasm = await runWasm();
  let identical = null;
  if (wasm) {
    identical = wasm.sorted.length === hip.sorted.length;
    for (let i = 0; identical && i < n; i++) identical = wasm.sorted[i] === hip.sorted[i];
  }
  console.log(JSON.stringify({ hip_ms: hip.ms, hip_device_ms: hip.device_ms, wasm_ms: wasm ? wasm.ms : null,
                               wasm_reps: wasm ? wasm.reps : 0, identical }));
})().catch((e) => { console.error(String(e && e.stack || e)); process.exit(1); });
