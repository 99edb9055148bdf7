"""Seed-fuzz of the HOST octree builder (gs_tree_create without a context: no GPU) against the Python restatement of the
reference's createSplatTreeWorker (oracle/tree_oracle.py, pinned to the reference through recorded goldens): random clustered
scenes with centres quantised onto split planes, random maxDepth / maxCentersPerNode, an alpha filter and a first index.  Leaves
compared field by field (bounds and centres as exact doubles, depth, index lists, order).  Runs under AddressSanitizer too
(tests/tools/asan_assets.sh builds the instrumented library).

usage: python tests/tools/soak_tree_host.py [iterations=200] [first_seed=500] [max_splats=30000] """
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np

from oracle import tree_oracle
from gaussiansplats3d_amd import SplatTree

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 500
max_n = int(sys.argv[3]) if len(sys.argv) > 3 else 30000


def digest(leaves):
    h = hashlib.sha256()
    for lf in leaves:
        h.update(json.dumps([lf["min"], lf["max"], lf["center"], lf["depth"]]).encode())
        h.update(np.asarray(lf["indexes"], dtype=np.uint32).tobytes())
    return h.hexdigest()


failures = 0
t_start = time.perf_counter()
for it in range(iters):
    seed = seed0 + it
    rng = np.random.default_rng(seed)
    n = int(np.exp(rng.uniform(np.log(1), np.log(max_n))))
    k = int(rng.integers(1, 40))
    cl = rng.uniform(-5, 5, size=(k, 3))
    c = (cl[rng.integers(0, k, n)] + rng.normal(size=(n, 3)) * float(np.exp(rng.uniform(np.log(1e-3), np.log(1.0))))).astype(np.float32)
    if rng.integers(0, 3) == 0:
        step = float(rng.choice([0.25, 0.5, 1.0]))
        sel = rng.integers(0, 3, n) == 0
        c[sel] = (np.round(c[sel] / step) * step).astype(np.float32)
    max_depth = int(rng.integers(1, 10))
    if rng.integers(0, 8) == 0:
        c[:] = c[0]                                                  # every centre identical: the degenerate box (8^depth leaves:
        max_depth = min(max_depth, 3)                                # kept shallow, the pure-Python oracle recurses through them all)
    max_centers = int(rng.choice([2, 10, 50, 300, 1000]))
    use_alpha = bool(rng.integers(0, 2))
    alphas = rng.integers(0, 256, n).astype(np.uint8) if use_alpha else None
    min_alpha = int(rng.choice([1, 40, 200])) if use_alpha else 1
    first = int(rng.choice([0, 0, 1234]))
    label = f"seed {seed}: n={n} clusters={k} depth={max_depth} per_node={max_centers} alpha>={min_alpha if use_alpha else '-'} first={first}"
    try:
        while n / max_centers > 2000:
            max_centers *= 5                                         # (keeps the pure-Python oracle's recursion within seconds)
        tree = SplatTree(None, max_depth, max_centers).process_splat_mesh(c, alphas=alphas, min_alpha=min_alpha, first_index=first)
        leaves, all_leaves = tree_oracle.build_tree(c, (alphas >= min_alpha) if use_alpha else None, max_depth, max_centers, first_index=first)
        info = tree.info()
        assert (info.leaves, info.all_leaves) == (len(leaves), all_leaves), f"leaf counts {info.leaves}/{info.all_leaves} != oracle {len(leaves)}/{all_leaves}"
        bounds, centers, depths, offsets, indexes = tree.leaves()
        got = [dict(min=bounds[i, :3].tolist(), max=bounds[i, 3:].tolist(), center=centers[i].tolist(), depth=int(depths[i]),
                    indexes=indexes[offsets[i]:offsets[i + 1]].tolist()) for i in range(len(depths))]
        assert digest(got) == digest(leaves), "leaves differ"
        tree.dispose()
        print(f"ok   {label} | leaves {len(leaves)}", flush=True)
    except Exception as e:
        failures += 1
        print(f"FAIL {label}: {type(e).__name__}: {str(e)[:300]}", flush=True)
print(f"soak_tree_host: {iters} iterations from seed {seed0}, {failures} failures, {time.perf_counter() - t_start:.0f} s")
sys.exit(1 if failures else 0)
