"""Seed-fuzz of the OCTREE rows (build, cull / gather, gather -> sort -> draw): random clustered scenes (some centres quantised
onto split planes), random maxDepth / maxCentersPerNode, a random orbit pose, viewport and field of view.  Checked per iteration:
  * device-built tree == host-built tree (every leaf field) and == the Python restatement of the reference's worker
    (oracle/tree_oracle.py, itself pinned to the reference's createSplatTreeWorker through recorded goldens);
  * device gather == the oracle's gather (order included), with and without gatherAllNodes;
  * gather (kept on the device, ASYNCHRONOUS: the count never reaches the host, the copy is fused into the sort's key kernel)
    -> sort == the sort oracle on the oracle's list; with the per-splat frustum cull on top == oracle.culled_sort;
  * a partial sort of the gathered list (splatSortCount < splatRenderCount) on the synchronous path.
The oracles are the checkers here, as in tests/.

usage: python tests/tools/soak_tree.py [iterations=60] [first_seed=300] [max_splats=120000] """
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import oracle
from oracle import tree_oracle
from gaussiansplats3d_amd import Context, SplatTree, camera, create_sort_worker, util

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 300
max_n = int(sys.argv[3]) if len(sys.argv) > 3 else 120000
ctx = Context(0)
failures = 0
t_start = time.perf_counter()
for it in range(iters):
    seed = seed0 + it
    rng = np.random.default_rng(seed)
    n = int(np.exp(rng.uniform(np.log(50), np.log(max_n))))
    k = int(rng.integers(1, 60))
    cl = rng.uniform(-5, 5, size=(k, 3))
    c = (cl[rng.integers(0, k, n)] + rng.normal(size=(n, 3)) * float(np.exp(rng.uniform(np.log(0.02), np.log(1.0))))).astype(np.float32)
    if rng.integers(0, 3) == 0:
        step = float(rng.choice([0.25, 0.5, 1.0]))
        sel = rng.integers(0, 4, n) == 0
        c[sel] = (np.round(c[sel] / step) * step).astype(np.float32)          # centres exactly on coarse planes
    max_depth = int(rng.integers(2, 10))
    max_centers = int(rng.choice([20, 100, 300, 1000]))
    W, H = int(rng.integers(200, 2000)), int(rng.integers(150, 1200))
    cam = camera.orbit_cameras(str(rng.choice(["garden", "truck", "bonsai"])), W, H, 24)[int(rng.integers(0, 24))]
    label = f"seed {seed}: n={n} clusters={k} depth={max_depth} per_node={max_centers} {W}x{H}"
    try:
        tree = SplatTree(ctx, max_depth, max_centers).process_splat_mesh(c)
        os.environ["GSPLAT_TREE_HOST_BUILD"] = "1"
        host = SplatTree(ctx, max_depth, max_centers).process_splat_mesh(c)
        del os.environ["GSPLAT_TREE_HOST_BUILD"]
        a, b = tree.info(), host.info()
        assert (a.leaves, a.all_leaves, a.nodes, a.splats) == (b.leaves, b.all_leaves, b.nodes, b.splats), "device / host tree sizes differ"
        for x, y in zip(tree.leaves(), host.leaves()):
            assert np.array_equal(x, y), "device-built and host-built leaves differ"
        host.dispose()
        leaves, _ = tree_oracle.build_tree(c, None, max_depth, max_centers)
        ci = util.integer_centers(c)
        w = create_sort_worker(ctx, n)
        w.post_message({"centers": ci, "range": {"from": 0, "to": n - 1, "count": n}})
        kept = []
        for gather_all in (False, True):
            expect = tree_oracle.gather(leaves, cam.view, 50.0, W, H, gather_all)
            got = tree.gather_scene_nodes_for_sort(cam, gather_all_nodes=gather_all)
            assert got["splatRenderCount"] == len(expect), f"gather count {got['splatRenderCount']} != oracle {len(expect)} (gatherAll={gather_all})"
            assert np.array_equal(got["indexesToSort"], expect), f"gathered list differs (gatherAll={gather_all})"
            kept.append(len(expect))
        idx = tree_oracle.gather(leaves, cam.view, 50.0, W, H, False)
        if len(idx):
            # asynchronous gather + fused copy / key kernel, then with the per-splat frustum cull on top
            for fc in (False, True):
                w.set_frustum_cull(fc)
                r = tree.gather_scene_nodes_for_sort(cam, sort_worker=w, to_host=False, asynchronous=True)
                w.sort_gathered(cam.sort_mvp(), keep_on_device=True)
                st, _ = w.last_stats()
                got_list = w.debug_read(2, int(st.result_count))
                if fc:
                    kept_sorted, keep = oracle.culled_sort(idx, ci, cam.sort_mvp())
                    assert st.result_count == int(keep.sum()) and np.array_equal(got_list, kept_sorted), "asynchronous gather + frustum-culled sort differs"
                else:
                    assert st.result_count == len(idx) and np.array_equal(got_list, oracle.sort_indexes(idx, ci, cam.sort_mvp())), \
                        "asynchronous gather + sort differs"
            w.set_frustum_cull(False)
            # synchronous gather, partial sort
            r = tree.gather_scene_nodes_for_sort(cam, sort_worker=w, to_host=False)
            R = r["splatRenderCount"]
            sc = int(rng.integers(0, R + 1))
            reply = w.sort_gathered(cam.sort_mvp(), sc)
            assert np.array_equal(reply["sortedIndexes"], oracle.sort_indexes(idx, ci, cam.sort_mvp(), sort_count=sc, render_count=R)), "partial sort of the gathered list differs"
        w.terminate(); tree.dispose()
        print(f"ok   {label} | leaves {a.leaves} kept {kept[0]} of {kept[1]}", flush=True)
    except Exception as e:
        os.environ.pop("GSPLAT_TREE_HOST_BUILD", None)
        failures += 1
        print(f"FAIL {label}: {type(e).__name__}: {str(e)[:300]}", flush=True)
ctx.close()
print(f"soak_tree: {iters} iterations from seed {seed0}, {failures} failures, {time.perf_counter() - t_start:.0f} s")
sys.exit(1 if failures else 0)
