"""Seed-fuzz of the SORT SEAM's modes: random case dictionaries in the vocabulary of tests/kat_cases.py (splat count, render /
sort counts, distance-map precision 10..24, integer / float centres, static / dynamic with per-scene transforms, precomputed
distances, permuted index lists, grids of equal keys, int32 wrap-around, far offsets) built by kat_cases.make_case on seeds the
goldens never saw, sorted through the worker protocol and compared - sorted list, and keys / buckets / min / max where the
worker exposes them - with the C oracle (sort_oracle.c, pinned to the reference's own sorter).  Both ranking paths alternate
(LDS atomics and, through GSPLAT_NO_LDS_ATOMIC_RANK=1 in the environment, ballots: run the tool twice).
The oracle is the checker here, as in tests/.

usage: python tests/tools/soak_sort.py [iterations=200] [first_seed=100] [max_splats=300000]
       python tests/tools/soak_sort.py sizes 4095,4096,12289,...     the listed splat counts exactly (tile / chunk / table boundaries of
                                                               radix.hpp), each as int-16, float-20 and a permuted partial sort """
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import kat_cases
import oracle
from gaussiansplats3d_amd import Context, create_sort_worker

SIZES = None
if len(sys.argv) > 2 and sys.argv[1] == "sizes":
    SIZES = [int(v) for v in sys.argv[2].split(",")]
    sys.argv = [sys.argv[0], str(3 * len(SIZES)), "777", "2"]
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 100
max_n = int(sys.argv[3]) if len(sys.argv) > 3 else 300000
ctxs = [Context(0), Context(0, single_stream=True)]
failures = 0
t_start = time.perf_counter()
for it in range(iters):
    seed = seed0 + it
    rng = np.random.default_rng(seed)
    n = int(np.exp(rng.uniform(np.log(2), np.log(max_n))))
    render = int(rng.integers(max(1, n // 2), n + 1)) if rng.integers(0, 3) == 0 else n
    sort = int(rng.integers(0, render + 1)) if rng.integers(0, 3) == 0 else render
    mode = "int" if rng.integers(0, 3) else "float"
    # the Viewer's own clamp of splatSortDistanceMapPrecision (src/Viewer.js:208-210): 10..20 with integer centres, 10..24 with float
    precision = int(rng.integers(10, 21 if mode == "int" else 25)) if rng.integers(0, 2) else 16
    case = dict(name=f"soak{seed}", n=n, render=render, sort=sort, precision=precision, mode=mode)
    kind = int(rng.integers(0, 8))
    if kind == 0: case["dynamic"] = True
    if kind == 1: case["precomputed"] = True
    if kind == 2: case["grid"] = True
    if kind == 3 and mode == "int": case["huge"] = True
    if kind == 4: case["offset"] = float(rng.uniform(-800.0, 800.0))
    if rng.integers(0, 2) or render < n: case["permute"] = True
    if SIZES:
        n = SIZES[it // 3]
        case = [dict(name=f"size{n}i", n=n, render=n, sort=n, precision=16, mode="int"),
                dict(name=f"size{n}f", n=n, render=n, sort=n, precision=20, mode="float"),
                dict(name=f"size{n}p", n=n, render=n - n // 7, sort=n - n // 3, precision=16, mode="int", permute=True)][it % 3]
        render, sort = case["render"], case["sort"]
    label = " ".join(f"{k}={v}" for k, v in case.items() if k != "name")
    try:
        args = kat_cases.make_case(case, seed=seed)
        kw = {k: args[k] for k in ("sort_count", "render_count", "precision", "use_int", "dynamic")}
        expect, keys, buckets, (lo, hi), st = oracle.sort_indexes(args["indexes"], args["centers4"], args["mvp"], precomputed=args["precomputed"],
                                                                  scene_indexes=args["scene_indexes"], transforms=args["transforms"],
                                                                  return_intermediates=True, **kw)
        ctx = ctxs[it & 1]
        w = create_sort_worker(ctx, n, True, True, args["use_int"], args["dynamic"], args["precision"])
        w.post_message({"centers": args["centers4"], "sceneIndexes": args["scene_indexes"], "range": {"from": 0, "to": n - 1, "count": n}})
        reply = w.post_message({"sort": {"modelViewProj": args["mvp"], "splatRenderCount": args["render_count"],
                                         "splatSortCount": args["sort_count"], "usePrecomputedDistances": args["precomputed"] is not None,
                                         "indexesToSort": args["indexes"], "transforms": args["transforms"],
                                         "precomputedDistances": args["precomputed"]}})
        # (status 1 = a bucket was clamped: fp32 rounding can map the farthest splat to bucket == range at 24 bits; the oracle says when)
        assert reply["sortDone"] and reply["status"] == st, f"status {reply.get('status')} (oracle: {st})"
        assert np.array_equal(reply["sortedIndexes"], expect), "sorted list differs from the oracle's"
        s0 = render - sort
        if sort:
            assert np.array_equal(w.debug_read(0, render)[s0:], keys[s0:]), "keys differ"
            assert np.array_equal(w.debug_read(1, render)[s0:], buckets[s0:]), "buckets differ"
            assert (reply["stats"].key_min, reply["stats"].key_max) == (lo, hi), "min / max differ"
        w.terminate()
        print(f"ok   seed {seed}: {label}", flush=True)
    except Exception as e:
        failures += 1
        print(f"FAIL seed {seed}: {label}: {type(e).__name__}: {str(e)[:300]}", flush=True)
for c in ctxs:
    c.close()
print(f"soak_sort: {iters} iterations from seed {seed0}, {failures} failures, {time.perf_counter() - t_start:.0f} s"
      + (" (ballot ranking)" if os.environ.get("GSPLAT_NO_LDS_ATOMIC_RANK") else ""))
sys.exit(1 if failures else 0)
