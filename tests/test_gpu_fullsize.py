"""-m gpu: INVARIANTS of the engine at BASELINE.json's full sizes - NOT parity.  Every test here compares the engine with
itself: storage-order invariance, bound-sorter vs host-index equivalence, strips == full frame, idempotence, statistics
consistency.  They would pass for an engine that is consistently wrong; parity against the oracle at these sizes is
tests/test_gpu_crops.py (oracle-rendered windows of the same frames) and, bit-exact for the sort, tests/test_gpu_sort.py."""
import numpy as np
import pytest

from gaussiansplats3d_amd import Context, SplatMesh, camera, create_sort_worker, scenes, util
from gaussiansplats3d_amd import _lib as L

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def c3():
    scene = scenes.make_config_scene("C3")                      # 5.8 M splats, SH-2: the headline workload
    cam = camera.demo_camera("garden", 1920, 1080)
    return scene, cam


def _mesh(ctx, scene, keep_order=False):
    m = SplatMesh(ctx, scene.count, scene.sh_degree, scene.cov_half)
    if keep_order:                                              # re-create with GS_MESH_KEEP_ORDER
        m.dispose()
        import ctypes as C
        m.handle = C.c_void_p()
        L.check(m.lib.gs_mesh_create(ctx.handle, scene.count, scene.sh_degree, L.GS_MESH_KEEP_ORDER, C.byref(m.handle)))
    m.build(scene.centers, scene.cov, scene.rgba, scene.sh if scene.sh_degree else None)
    return m


def test_c3_full_size_invariants(ctx, c3):
    scene, cam = c3
    n = scene.count
    w = create_sort_worker(ctx, n)
    w.post_message({"centers": util.integer_centers(scene.centers), "range": {"from": 0, "to": n - 1, "count": n}})
    order = w.post_message({"sort": {"modelViewProj": cam.sort_mvp(), "splatRenderCount": n, "splatSortCount": n}})["sortedIndexes"]
    assert np.array_equal(np.bincount(order, minlength=n), np.ones(n, np.int64))      # a permutation

    morton = _mesh(ctx, scene)
    morton.set_camera(cam)
    morton.update_render_indexes(order, n)                      # host indexes -> translated through perm on the device
    a, sa = morton.render()
    assert sa.visible_splats > 100000 and sa.tile_entries >= sa.visible_splats and sa.tiles16 > sa.tile_entries

    # 1. storage order is invisible: upload order kept on the device gives the same pixels
    plain = _mesh(ctx, scene, keep_order=True)
    plain.set_camera(cam)
    plain.update_render_indexes(order, n)
    b, sb = plain.render()
    np.testing.assert_array_equal(a, b)
    assert (sa.visible_splats, sa.tile_entries, sa.tiles16) == (sb.visible_splats, sb.tile_entries, sb.tiles16)
    plain.dispose()

    # 2. a sorter bound to the mesh (device-resident, Morton-space payload) == the host index path
    w.sort_on_device(cam.sort_mvp(), n)                         # unbound result: caller's indexes
    morton.use_sorter_result(w, n)                              # binds for the NEXT sort; this draw still translates
    c, _ = morton.render()
    np.testing.assert_array_equal(a, c)
    w.sort_on_device(cam.sort_mvp(), n)                         # bound result: positions in the mesh's order
    d, _ = morton.render()
    np.testing.assert_array_equal(a, d)
    np.testing.assert_array_equal(w.debug_read(2, n), order)    # host-visible result is still in the caller's numbering

    # 3. idempotence and strips (multi-GPU unit) at full size
    e, _ = morton.render()
    np.testing.assert_array_equal(d, e)
    parts = [morton.render(tile_rows=r)[0] for r in ((0, 17), (17, 18), (18, 45), (45, 68))]
    np.testing.assert_array_equal(np.concatenate(parts, axis=0), a)

    # 3b. the per-splat frustum cull fused into the sort: same pixels, and the list is the reference's minus the dropped
    import oracle
    w.set_frustum_cull(True)
    culled = w.post_message({"sort": {"modelViewProj": cam.sort_mvp(), "splatRenderCount": n, "splatSortCount": n}})
    expect, keep = oracle.culled_sort(np.arange(n, dtype=np.uint32), util.integer_centers(scene.centers), cam.sort_mvp())
    assert culled["stats"].result_count == len(expect) < n
    np.testing.assert_array_equal(culled["sortedIndexes"], expect)
    w.sort_on_device(cam.sort_mvp(), n)
    f, _ = morton.render()
    np.testing.assert_array_equal(a, f)
    w.set_frustum_cull(False)
    w.sort_on_device(cam.sort_mvp(), n)

    # 4. statistics agree with the per-splat vertex-stage outputs (of a full-frame draw: the last draw above was a strip)
    _, sa = morton.render()
    recs, rects, vis = morton.debug_records()
    assert int(vis.sum()) == sa.visible_splats
    x0, y0, x1, y1 = rects[vis, 0] & 0xFFFF, rects[vis, 0] >> 16, rects[vis, 1] & 0xFFFF, rects[vis, 1] >> 16
    assert int(((x1 - x0 + 1).astype(np.int64) * (y1 - y0 + 1)).sum()) == sa.tiles16
    sh = int(np.log2(sa.list_bin_px // 16))                    # 16-px tiles per list-bin edge, as a shift
    bins = ((x1 >> sh) - (x0 >> sh) + 1).astype(np.int64) * ((y1 >> sh) - (y0 >> sh) + 1)
    assert int(bins.sum()) == sa.tile_entries
    assert int(morton.bin_entry_counts().sum()) == sa.tile_entries
    w.terminate()
    morton.dispose()


def test_c4_sixteen_million_sort_properties(ctx):
    """configs[3]: 16 M uniform-random Gaussians — the radix-sort stress case: bit-exact vs the pinned C oracle."""
    import oracle
    n = 16_000_000
    rng = np.random.default_rng(2026)
    c = rng.uniform(-10, 10, (n, 3)).astype(np.float32)
    ci = util.integer_centers(c)
    cam = camera.demo_camera("synthetic16m", 3840, 2160)
    mvp = cam.sort_mvp()
    w = create_sort_worker(ctx, n)
    w.post_message({"centers": ci, "range": {"from": 0, "to": n - 1, "count": n}})
    got = w.post_message({"sort": {"modelViewProj": mvp, "splatRenderCount": n, "splatSortCount": n}})["sortedIndexes"]
    exp, keys, buckets, _, st = oracle.sort_indexes(np.arange(n, dtype=np.uint32), ci, mvp, return_intermediates=True)
    bs = buckets[got].astype(np.int64)
    assert (np.diff(bs) <= 0).all()
    np.testing.assert_array_equal(got, exp)
    w.terminate()
