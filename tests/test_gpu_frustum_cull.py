"""-m gpu: the per-splat frustum cull fused into the sort (gs_sorter_set_frustum_cull) through the C ABI.

The contract (include/gsplat_hip.h): the culled sort returns the reference's sorted list with the dropped splats removed —
keys, range and buckets still span every list position — so (1) the keep bits equal the fp32 numpy restatement bit for
bit, (2) the list equals filter(reference sort), (3) the kept set covers everything the vertex stage draws, and (4) the
frame is bit-identical to the one drawn from the full list."""
import numpy as np
import pytest

import helpers
import oracle
import tree_cases
from gaussiansplats3d_amd import Context, SplatMesh, SplatTree, camera, create_sort_worker, util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = Context(0)
    yield c
    c.close()


def _worker(ctx, centers, integer=True, precision=16):
    n = centers.shape[0]
    w = create_sort_worker(ctx, n, integer_based_sort=integer, splat_sort_distance_map_precision=precision)
    c4 = util.integer_centers(centers) if integer else np.c_[centers, np.ones(n)].astype(np.float32)
    w.post_message({"centers": c4, "range": {"from": 0, "to": n - 1, "count": n}})
    return w, c4


def _wide_scene(n, sh_degree, seed):
    """small_scene plus a shell all around the camera, so the cull has plenty to drop on every side."""
    scene = helpers.small_scene(n, sh_degree, seed)
    rng = np.random.default_rng(seed + 1)
    cam_pos = np.array(camera.DEMO_POSES["garden"][1])
    shell = rng.normal(size=(n // 2, 3))
    shell = cam_pos + shell / np.linalg.norm(shell, axis=1, keepdims=True) * rng.uniform(0.05, 12.0, size=(n // 2, 1))
    scene.centers[: n // 2] = shell.astype(np.float32)
    return scene


@pytest.mark.parametrize("pose,integer,precision,n", [("garden", True, 16, 50000), ("truck", True, 16, 4097),
                                                      ("garden", False, 16, 30000), ("bonsai", True, 20, 20000),
                                                      ("garden", True, 10, 63), ("garden", True, 16, 1)])
def test_culled_sort_is_the_reference_sort_with_dropped_splats_removed(ctx, pose, integer, precision, n):
    scene = _wide_scene(max(n, 2), 0, seed=400 + n % 97)
    centers = scene.centers[:n]
    cam = camera.demo_camera(pose, 1920, 1080)
    w, c4 = _worker(ctx, centers, integer, precision)
    w.set_frustum_cull(True)
    idx = np.arange(n, dtype=np.uint32)
    reply = w.post_message({"sort": {"modelViewProj": cam.sort_mvp(), "splatRenderCount": n, "splatSortCount": n}})
    expect, keep = oracle.culled_sort(idx, c4, cam.sort_mvp(), precision=precision, use_int=integer)
    np.testing.assert_array_equal(w.keep_bits(n), keep)
    assert reply["stats"].result_count == keep.sum() == len(expect)
    np.testing.assert_array_equal(reply["sortedIndexes"], expect)
    if n >= 4097:
        assert 0 < keep.sum() < n, "the case should drop something but not everything"
    # switching the cull off again restores the plain contract
    w.set_frustum_cull(False)
    reply = w.post_message({"sort": {"modelViewProj": cam.sort_mvp(), "splatRenderCount": n, "splatSortCount": n}})
    np.testing.assert_array_equal(reply["sortedIndexes"], oracle.sort_indexes(idx, c4, cam.sort_mvp(), precision=precision,
                                                                               use_int=integer))
    assert reply["stats"].result_count == n
    w.terminate()


def test_everything_dropped_and_nothing_dropped(ctx):
    cam = camera.demo_camera("garden", 640, 360)
    pos = np.array(camera.DEMO_POSES["garden"][1]); look = np.array(camera.DEMO_POSES["garden"][2])
    fwd = (look - pos) / np.linalg.norm(look - pos)
    rng = np.random.default_rng(3)
    behind = (pos - fwd * rng.uniform(1, 5, size=(500, 1)) + rng.normal(size=(500, 3)) * 0.1).astype(np.float32)
    front = (pos + fwd * rng.uniform(2, 5, size=(500, 1)) + rng.normal(size=(500, 3)) * 0.05).astype(np.float32)
    for centers, kept in ((behind, 0), (front, 500)):
        w, c4 = _worker(ctx, centers)
        w.set_frustum_cull(True)
        reply = w.post_message({"sort": {"modelViewProj": cam.sort_mvp(), "splatRenderCount": 500, "splatSortCount": 500}})
        assert reply["stats"].result_count == kept and len(reply["sortedIndexes"]) == kept
        mesh = SplatMesh(ctx, 500, 0, False)
        rgba = np.full((500, 4), 200, np.uint8)
        cov = np.tile(np.array([1e-3, 0, 0, 1e-3, 0, 1e-3], np.float32), (500, 1))
        mesh.build(centers, cov, rgba, None)
        mesh.set_camera(cam)
        w.sort_on_device(cam.sort_mvp(), 500)
        mesh.use_sorter_result(w, 500)
        img, stats = mesh.render()
        assert (stats.visible_splats > 0) == (kept > 0)
        assert (img.any()) == (kept > 0)
        w.terminate()
        mesh.dispose()


@pytest.mark.parametrize("sh_degree,w_h,ortho", [(0, (640, 360), False), (2, (1000, 600), False), (1, (333, 211), False),
                                                 (0, (640, 360), True)])
def test_frame_from_the_culled_list_is_bit_identical(ctx, sh_degree, w_h, ortho):
    scene = _wide_scene(30000, sh_degree, seed=410 + sh_degree)
    width, height = w_h
    if ortho:
        pos, look, up = camera.DEMO_POSES["garden"][1], camera.DEMO_POSES["garden"][2], camera.DEMO_POSES["garden"][0]
        cam = camera.OrthographicCamera(width, height, pos, look, up, zoom=60.0)
    else:
        cam = camera.demo_camera("garden", width, height)
    mesh = SplatMesh(ctx, scene.count, sh_degree, False)
    mesh.build(scene.centers, scene.cov, scene.rgba, scene.sh if sh_degree else None)
    mesh.set_camera(cam)
    w, c4 = _worker(ctx, scene.centers)
    w.sort_on_device(cam.sort_mvp(), scene.count)
    mesh.use_sorter_result(w, scene.count)
    mesh.render()                                        # settles the per-mesh list-bin size (chosen from a measured draw)
    full, s_full = mesh.render()
    parts_full = [mesh.render(tile_rows=r)[0] for r in ((0, 5), (5, (height + 15) // 16))]
    _, _, drawn = mesh.debug_records()
    w.set_frustum_cull(True)
    w.sort_on_device(cam.sort_mvp(), scene.count)
    culled, s_cull = mesh.render()
    keep = w.keep_bits(scene.count)
    stats, _ = w.last_stats()
    assert stats.result_count == keep.sum() < scene.count
    assert not (drawn & ~keep).any(), "the cull dropped a splat the vertex stage draws"
    assert s_cull.visible_splats == s_full.visible_splats and s_cull.tile_entries == s_full.tile_entries
    np.testing.assert_array_equal(culled, full)
    parts = [mesh.render(tile_rows=r)[0] for r in ((0, 5), (5, (height + 15) // 16))]          # multi-GPU strips too
    np.testing.assert_array_equal(np.concatenate(parts, axis=0), np.concatenate(parts_full, axis=0))
    # and the host-visible list is in the caller's numbering although the mesh stores its splats in Morton order
    expect, _ = oracle.culled_sort(np.arange(scene.count, dtype=np.uint32), c4, cam.sort_mvp())
    np.testing.assert_array_equal(w.debug_read(2, len(expect)), expect)
    w.terminate()
    mesh.dispose()


def test_composes_with_the_octree_gather(ctx):
    """octree node cull -> device list -> per-splat cull + sort: the reference pipeline's list, filtered."""
    from oracle import tree_oracle
    case = tree_cases.make_case("clusters40k")
    c = case["centers"]
    n = c.shape[0]
    cam = camera.demo_camera("garden", 1280, 720)
    tree = SplatTree(ctx, 8, 300).process_splat_mesh(c)
    w, c4 = _worker(ctx, c)
    w.set_frustum_cull(True)
    leaves, _ = tree_oracle.build_tree(c, None, 8, 300)
    idx = tree_oracle.gather(leaves, cam.view, 50.0, 1280, 720)
    r = tree.gather_scene_nodes_for_sort(cam, sort_worker=w, to_host=False)
    assert r["splatRenderCount"] == len(idx)
    reply = w.sort_gathered(cam.sort_mvp())
    expect, keep = oracle.culled_sort(idx, c4, cam.sort_mvp())
    assert 0 < keep.sum() < len(idx)
    np.testing.assert_array_equal(w.keep_bits(len(idx)), keep)
    np.testing.assert_array_equal(reply["sortedIndexes"], expect)
    w.terminate()
    tree.dispose()


def test_unsupported_combinations_fail_loudly(ctx):
    scene = helpers.small_scene(1000, 0, seed=5)
    cam = camera.demo_camera("garden", 320, 180)
    w, c4 = _worker(ctx, scene.centers)
    w.set_frustum_cull(True)
    with pytest.raises(RuntimeError, match="full sort"):
        w.post_message({"sort": {"modelViewProj": cam.sort_mvp(), "splatRenderCount": 1000, "splatSortCount": 400}})
    with pytest.raises(RuntimeError, match="full sort"):
        w.post_message({"sort": {"modelViewProj": cam.sort_mvp(), "splatRenderCount": 1000, "splatSortCount": 1000,
                                 "usePrecomputedDistances": True, "precomputedDistances": np.zeros(1000, np.int32)}})
    w.terminate()
    d = create_sort_worker(ctx, 1000, dynamic_mode=True)
    with pytest.raises(RuntimeError, match="dynamic"):
        d.set_frustum_cull(True)
    d.terminate()
