"""Stateful seed-fuzz: ONE mesh, ONE sorter and ONE octree kept alive over a random sequence of events, every frame checked
against oracles that are rebuilt from scratch from the current scene - what goes wrong in a long-lived viewer is state: storage
slots and block boxes after partial re-uploads, the per-mesh list-bin size, grown entry buffers, the previous draw's blend
statistics (bin order, deep pass), the two sets of vertex-stage outputs, culled sorts leaving their masks behind.
Events (one per step, random): a new camera on the orbit / a new viewport; a re-upload of a random index range with MOVED centres,
new covariances and colours (the sorter's centres follow); toggling the fused frustum cull; a visibility-culled frame
(vertex stage first); an octree-culled frame (asynchronous gather); a frame drawn as strips; a frame without statistics
(asynchronous, into a device buffer) followed by a synchronous one; the entry buffers shrunk so that the next draw overflows
and regrows.  Checked at every step: the sorted list against the sort oracle (or the culled / gathered variant), the frame
against the raster oracle within the stated tolerance.
The oracles are the checkers here, as in tests/.

usage: python tests/tools/soak_stateful.py [sequences=12] [first_seed=7000] [steps=30] [max_splats=50000] """
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch          # (before the engine's library: one HIP runtime per process, torch's, as in tests/conftest.py)

import helpers
import oracle
from gaussiansplats3d_amd import Context, SplatMesh, SplatTree, camera, create_sort_worker, util

if os.environ.get("SOAK_LIB"):          # an OLDER build of the library (entry points added since are dropped): does the soak find its bugs?
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from ab_libs import use_library
    use_library(os.environ["SOAK_LIB"])
seqs = int(sys.argv[1]) if len(sys.argv) > 1 else 12
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 7000
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
max_n = int(sys.argv[4]) if len(sys.argv) > 4 else 50000
failures = 0
frames_checked = 0
t_start = time.perf_counter()
for s in range(seqs):
    seed = seed0 + s
    rng = np.random.default_rng(seed)
    n = int(rng.integers(3000, max_n))
    sh_degree = int(rng.integers(0, 3))
    scene = helpers.small_scene(n, sh_degree, seed, scale=float(np.exp(rng.uniform(np.log(0.02), np.log(0.12)))))
    ctx = Context(0, single_stream=bool(s & 1))
    mesh = SplatMesh(ctx, n, sh_degree).build(scene.centers, scene.cov, scene.rgba, scene.sh if sh_degree else None)
    w = create_sort_worker(ctx, n)
    w.post_message({"centers": util.integer_centers(scene.centers), "range": {"from": 0, "to": n - 1, "count": n}})
    mesh.use_sorter_result(w, n)
    W, H = 320, 200
    cam = camera.orbit_cameras("garden", W, H, 24)[0]
    tree, tree_version = None, -1
    version = 0
    log = []
    try:
        for step in range(steps):
            ev = ["camera", "viewport", "reupload", "frustum", "visibility", "octree", "strips", "async", "overflow"][int(rng.integers(0, 9))]
            log.append(ev)
            ci = util.integer_centers(scene.centers)
            if ev == "camera":
                cam = camera.orbit_cameras("garden", W, H, 24)[int(rng.integers(0, 24))]
            elif ev == "viewport":
                W, H = int(rng.integers(80, 480)), int(rng.integers(64, 300))
                cam = camera.orbit_cameras("garden", W, H, 24)[int(rng.integers(0, 24))]
            elif ev == "reupload":
                a = int(rng.integers(0, n - 1)); b = int(rng.integers(a + 1, min(n, a + 1 + n // 2) + 1))
                fresh = helpers.small_scene(b - a, sh_degree, int(rng.integers(0, 1 << 30)), scale=float(np.exp(rng.uniform(np.log(0.02), np.log(0.12)))))
                scene.centers[a:b] = fresh.centers; scene.cov[a:b] = fresh.cov; scene.rgba[a:b] = fresh.rgba
                if sh_degree: scene.sh[a:b] = fresh.sh
                mesh.build(scene.centers[a:b], scene.cov[a:b], scene.rgba[a:b], scene.sh[a:b] if sh_degree else None, start=a)
                ci = util.integer_centers(scene.centers)
                w.post_message({"centers": ci[a:b], "range": {"from": a, "to": b - 1, "count": b - a}})
                version += 1
            mesh.set_camera(cam)
            mvp = cam.sort_mvp()
            full_order = oracle.sort_indexes(np.arange(n, dtype=np.uint32), ci, mvp)
            expect_list = full_order
            c, cov, rgba, sh = helpers.oracle_inputs(scene)
            ocam = oracle.make_camera(cam.model_view(), cam.projection, cam.position, W, H, sh_degree=sh_degree, sh_stored=sh_degree)
            if ev == "frustum":
                w.set_frustum_cull(True)
                w.sort_on_device(mvp, n)
                expect_list, keep = oracle.culled_sort(np.arange(n, dtype=np.uint32), ci, mvp)
                st, _ = w.last_stats()
                assert st.result_count == len(expect_list) and np.array_equal(w.debug_read(2, len(expect_list)), expect_list), "frustum-culled list differs"
                mesh.use_sorter_result(w, n)
                got = mesh.render()[0]
                w.set_frustum_cull(False)
            elif ev == "visibility":
                w.set_visibility_cull(True)
                mesh.project(None)
                w.sort_on_device(mvp, n)
                mesh.use_sorter_result(w, n)
                got = mesh.render()[0]
                w.set_visibility_cull(False)
            elif ev == "octree":
                if tree_version != version:
                    if tree is not None: tree.dispose()
                    tree = SplatTree(ctx, 8, int(rng.choice([50, 200, 1000]))).process_splat_mesh(scene.centers)
                    tree_version = version
                r = tree.gather_scene_nodes_for_sort(cam, sort_worker=w, to_host=False, asynchronous=True)
                w.sort_gathered(mvp, keep_on_device=True)
                st, _ = w.last_stats()
                R = int(st.result_count)
                got_list = w.debug_read(2, R)
                # the gathered set against the synchronous gather of the same tree (itself fuzzed against the oracle by soak_tree.py)
                ref = tree.gather_scene_nodes_for_sort(cam)
                assert R == ref["splatRenderCount"], "asynchronous gather kept another number of splats"
                expect_list = oracle.sort_indexes(ref["indexesToSort"], ci, mvp)
                assert np.array_equal(got_list, expect_list), "octree-culled list differs"
                mesh.use_sorter_result(w, r["splatRenderCount"])
                got = mesh.render()[0]
            else:
                w.sort_on_device(mvp, n)
                assert np.array_equal(w.debug_read(2, n), full_order), "sorted list differs from the oracle's"
                mesh.use_sorter_result(w, n)
                if ev == "strips":
                    rows = (H + 15) // 16
                    cuts = sorted(set([0, rows] + [int(v) for v in rng.integers(0, rows + 1, size=3)]))
                    got = np.concatenate([mesh.render(tile_rows=(a, b))[0] for a, b in zip(cuts[:-1], cuts[1:])], axis=0)
                elif ev == "async":
                    buf = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda:0")
                    torch.cuda.synchronize()
                    for _ in range(3):
                        w.sort_on_device(mvp, n)
                        mesh.render(out_device_ptr=buf.data_ptr(), want_stats=False, to_host=False)
                    ctx.synchronize()
                    got = buf.cpu().numpy()
                    again = mesh.render()[0]
                    assert np.array_equal(got, again), "asynchronous and synchronous frames differ"
                elif ev == "overflow":
                    mesh.debug_set_entry_capacity(1024)
                    got, st = mesh.render()
                else:
                    got = mesh.render()[0]
            fb, q, amb, frags = oracle.render(ocam, c, cov, rgba, sh, expect_list)
            helpers.compare_frames(got, fb, amb, f"step {step} ({ev})")
            frames_checked += 1
        print(f"ok   seed {seed}: n={n} sh{sh_degree} {'one stream' if s & 1 else 'streams'} | " + " ".join(log), flush=True)
    except Exception as e:
        failures += 1
        print(f"FAIL seed {seed}: n={n} sh{sh_degree} after [{' '.join(log)}]: {type(e).__name__}: {str(e)[:400]}", flush=True)
    if tree is not None: tree.dispose()
    w.terminate(); mesh.dispose(); ctx.close()
print(f"soak_stateful: {seqs} sequences x {steps} steps from seed {seed0}, {frames_checked} frames checked, {failures} failures, {time.perf_counter() - t_start:.0f} s")
sys.exit(1 if failures else 0)
