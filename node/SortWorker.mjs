// SortWorker.mjs — the sort seam as an ES module: `createSortWorker` with the reference's signature and message protocol
// (/root/reference/src/worker/SortWorker.js:202-256), over the MI355X sorter (node/gsplat.js -> N-API -> gs_sorter_*).
//     import { createSortWorker } from '<this repo>/node/SortWorker.mjs';      // was './worker/SortWorker.js'
import { createRequire } from 'module';
const require = createRequire(import.meta.url);
const { createSortWorker, Constants } = require('./gsplat.js');
export { createSortWorker, Constants };
