"""-m gpu: the visibility cull (gs_mesh_project -> gs_sorter_set_visibility_cull sort -> gs_mesh_render) - how the tile-row
strips of a multi-GPU draw shard the sort - and the strip gather behind the C ABI (gs_group_*, a group of one here)."""
import numpy as np
import pytest
import torch

import helpers
import oracle
from gaussiansplats3d_amd import Context, SplatMesh, camera, create_sort_worker, util
from gaussiansplats3d_amd import dist as gdist

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = Context(0)
    yield c
    c.close()


def _setup(ctx, n=60000, sh=2, seed=21, w=640, h=360):
    scene = helpers.small_scene(n, sh, seed=seed)
    cam = camera.demo_camera("garden", w, h)
    worker = create_sort_worker(ctx, n)
    ci = util.integer_centers(scene.centers)
    worker.post_message({"centers": ci, "range": {"from": 0, "to": n - 1, "count": n}})
    mesh = SplatMesh(ctx, n, sh)
    mesh.build(scene.centers, scene.cov, scene.rgba, scene.sh)
    mesh.set_camera(cam)
    return scene, cam, ci, worker, mesh


def test_visibility_culled_sort_is_the_reference_list_restricted_to_what_the_frame_draws(ctx):
    scene, cam, ci, worker, mesh = _setup(ctx)
    n, mvp = scene.count, cam.sort_mvp()
    worker.sort_on_device(mvp, n)
    mesh.use_sorter_result(worker, n)                    # binds the sorter to the mesh
    worker.sort_on_device(mvp, n)
    full, _ = mesh.render()
    _, _, vis = mesh.debug_records()                     # vertex-stage survivors, original splat numbering
    assert 1000 < vis.sum() < n
    order = oracle.sort_indexes(np.arange(n, dtype=np.uint32), ci, mvp)

    worker.set_visibility_cull(True)
    with pytest.raises(Exception):                       # no projection of this frame yet
        worker.post_message({"sort": {"modelViewProj": mvp, "splatRenderCount": n, "splatSortCount": n}})
    mesh.project()
    reply = worker.post_message({"sort": {"modelViewProj": mvp, "splatRenderCount": n, "splatSortCount": n}})
    np.testing.assert_array_equal(reply["sortedIndexes"], order[vis[order]])
    assert reply["stats"].result_count == int(vis.sum())
    np.testing.assert_array_equal(worker.keep_bits(n), vis)
    got, st = mesh.render()                              # consumes the projection
    np.testing.assert_array_equal(got, full)
    assert st.visible_splats == int(vis.sum())

    # strips: every "rank" projects, sorts and draws only its strip; together they are the full frame
    rows = (cam.height + 15) // 16
    parts, kept = [], []
    for strip in ((0, 7), (7, 8), (8, 15), (15, rows)):
        mesh.project(strip)
        worker.sort_on_device(mvp, n)
        img, s = mesh.render(tile_rows=strip)
        parts.append(img)
        kept.append(worker.last_stats()[0].result_count)
        assert s.visible_splats == kept[-1]
    np.testing.assert_array_equal(np.concatenate(parts, axis=0), full)
    assert max(kept) < int(vis.sum())                    # a strip really sorts less than the frame
    worker.set_visibility_cull(False)
    worker.terminate()
    mesh.dispose()


def test_a_projection_is_consumed_once_and_only_by_its_own_camera(ctx):
    scene, cam, ci, worker, mesh = _setup(ctx, n=20000, seed=5, w=320, h=200)
    n = scene.count
    order = oracle.sort_indexes(np.arange(n, dtype=np.uint32), ci, cam.sort_mvp())
    mesh.update_render_indexes(order, n)
    want, _ = mesh.render()
    other = camera.orbit_cameras("garden", 320, 200, 8)[3]
    mesh.set_camera(other)
    mesh.project()                                       # projected for `other` ...
    mesh.set_camera(cam)
    got, _ = mesh.render()                               # ... drawn with `cam`: the stale projection must not be used
    np.testing.assert_array_equal(got, want)
    mesh.project()
    a, _ = mesh.render()
    b, _ = mesh.render()                                 # second draw projects again itself
    np.testing.assert_array_equal(a, want)
    np.testing.assert_array_equal(b, want)
    worker.terminate()
    mesh.dispose()


def test_group_of_one_gathers_its_strip_into_the_frame(ctx):
    """gs_group_* with world_size 1 (no RCCL involved): the root's own strip lands at its rows of the full frame."""
    group = gdist.StripGroup(ctx, 0, 1)
    h, w = 96, 64
    strips = [(2, 5)]                                    # tile rows -> pixel rows [32, 80)
    strip = torch.randint(0, 255, (48, w, 4), dtype=torch.uint8, device="cuda")
    full = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    group.gather_strips(strip.data_ptr(), full.data_ptr(), w, strips, h)
    ctx.synchronize()
    assert torch.equal(full[32:80], strip) and not full[:32].any() and not full[80:].any()
    group.close()


def test_overlapped_gathers_of_a_group_of_one():
    """gs_group_set_overlap: the transfer runs on the group's own stream beside what the context's stream does next, out of
    / into alternating buffers; the context's stream waits for the previous transfer before it goes on.  Twenty frames of
    changing content through two buffer sets, every one checked."""
    s = torch.cuda.Stream()
    c = Context(0, stream=s.cuda_stream, single_stream=True)
    group = gdist.StripGroup(c, 0, 1)
    group.set_overlap(True)
    h, w = 2048, 1024                                    # 6 MB strips: a transfer long enough to overlap something
    strips = [(2, 98)]                                   # tile rows -> pixel rows [32, 1568)
    rows = 1568 - 32
    bufs = [torch.zeros((rows, w, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
    fulls = [torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
    torch.cuda.synchronize()
    seen = []
    for k in range(20):
        with torch.cuda.stream(s):
            bufs[k & 1].fill_(10 + k)                    # "the draw" of frame k, on the context's stream
        group.gather_strips(bufs[k & 1].data_ptr(), fulls[k & 1].data_ptr(), w, strips, h)
        if k >= 1:
            # what call k - 1 gathered: complete once the context's stream has passed the wait that call k put in front of
            # everything enqueued after it, and not written again before call k + 1's transfer
            with torch.cuda.stream(s):
                seen.append((k - 1, fulls[(k - 1) & 1][40, 3, 1].clone(), fulls[(k - 1) & 1][1567, w - 1, 2].clone()))
    group.wait()
    torch.cuda.synchronize()
    for k, a, b in seen:
        assert int(a) == 10 + k and int(b) == 10 + k, (k, int(a), int(b))
    assert int(fulls[1][100, 5, 0]) == 29 and int(fulls[0][100, 5, 0]) == 28
    assert not fulls[1][:32].any() and not fulls[1][1568:].any()
    group.set_overlap(False)
    strip = torch.randint(0, 255, (rows, w, 4), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    group.gather_strips(strip.data_ptr(), fulls[0].data_ptr(), w, strips, h)
    c.synchronize()
    assert torch.equal(fulls[0][32:1568], strip)
    group.close()
    c.close()


def test_visibility_mask_lifecycle(ctx):
    """The mask a projection leaves is consumed (and re-zeroed) by the sort that reads it.  Whatever the order of calls - two
    projections in a row, a sort that covers fewer splats than were projected, a projection for another camera - the next
    visibility-culled sort must see exactly the survivors of ITS projection."""
    scene, cam, ci, worker, mesh = _setup(ctx, n=30000, seed=9, w=320, h=200)
    n = scene.count
    other = camera.orbit_cameras("garden", 320, 200, 8)[2]
    worker.sort_on_device(cam.sort_mvp(), n)
    mesh.use_sorter_result(worker, n)
    worker.set_visibility_cull(True)

    def survivors(c):
        mesh.set_camera(c)
        mesh.update_render_indexes(np.arange(n, dtype=np.uint32), n)
        mesh.render()
        vis = mesh.debug_records()[2]
        mesh.use_sorter_result(worker, n)
        return vis

    vis_cam, vis_other = survivors(cam), survivors(other)
    assert (vis_cam != vis_other).any() and vis_cam.sum() > 500 and vis_other.sum() > 500

    def culled_sort(c, count=n):
        mesh.set_camera(c)
        mesh.project()
        return worker.post_message({"sort": {"modelViewProj": c.sort_mvp(), "splatRenderCount": count, "splatSortCount": count}})

    def expect(c, vis, count=n):
        order = oracle.sort_indexes(np.arange(count, dtype=np.uint32), ci, c.sort_mvp())
        return order[vis[order]]

    np.testing.assert_array_equal(culled_sort(cam)["sortedIndexes"], expect(cam, vis_cam))
    np.testing.assert_array_equal(culled_sort(cam)["sortedIndexes"], expect(cam, vis_cam))          # mask re-zeroed by the sort
    mesh.set_camera(other)
    mesh.project()                                                                                    # never consumed ...
    np.testing.assert_array_equal(culled_sort(cam)["sortedIndexes"], expect(cam, vis_cam))          # ... and must not leak
    half = n // 2
    np.testing.assert_array_equal(culled_sort(other, half)["sortedIndexes"], expect(other, vis_other, half))   # bits >= half stay set
    np.testing.assert_array_equal(culled_sort(cam)["sortedIndexes"], expect(cam, vis_cam))
    np.testing.assert_array_equal(worker.keep_bits(n), vis_cam)
    worker.set_visibility_cull(False)
    worker.terminate()
    mesh.dispose()


def test_full_frame_projection_leaves_the_original_order_mask_to_the_sorter(ctx):
    """Round 6: a full-frame gs_mesh_project for a mesh whose bound sorter culls by visibility no longer sets one bit per survivor
    with an atomicOr at inv_perm[position] (25 us of a C3 frame); the sorter derives the mask from the vertex stage's storage-order
    mask through its position map (k_mask_derive_count).  The list is the reference's restricted to what the frame draws, the keep
    bits are the vertex stage's survivors, the frame is the full sort's - also for a list shorter than the mesh, and beside strips
    (which keep the atomics).  $GSPLAT_NO_LAZY_MASK=1 and both front ends: tests/tools/soak_paths.sh."""
    scene, cam, ci, worker, mesh = _setup(ctx, n=90000, sh=1, seed=33)
    n, mvp = scene.count, cam.sort_mvp()
    worker.sort_on_device(mvp, n)
    mesh.use_sorter_result(worker, n)
    worker.sort_on_device(mvp, n)
    full, _ = mesh.render()
    _, _, vis = mesh.debug_records()
    order = oracle.sort_indexes(np.arange(n, dtype=np.uint32), ci, mvp)
    worker.set_visibility_cull(True)

    def culled(strip=None):
        mesh.project(strip)
        reply = worker.post_message({"sort": {"modelViewProj": mvp, "splatRenderCount": n, "splatSortCount": n}})
        bits = worker.keep_bits(n).copy()
        img, st = mesh.render(tile_rows=strip)
        return reply["sortedIndexes"].copy(), bits, img, int(st.visible_splats)

    lazy = culled()
    np.testing.assert_array_equal(lazy[0], order[vis[order]])
    np.testing.assert_array_equal(lazy[1], vis)
    np.testing.assert_array_equal(lazy[2], full)
    strip = (4, 11)
    lazy_strip = culled(strip)
    # fewer splats sorted than the mesh holds: the mask covers the list's positions only
    m = n // 2
    order_m = oracle.sort_indexes(np.arange(m, dtype=np.uint32), ci, mvp)
    mesh.project()
    reply = worker.post_message({"sort": {"modelViewProj": mvp, "splatRenderCount": m, "splatSortCount": m}})
    np.testing.assert_array_equal(reply["sortedIndexes"], order_m[vis[order_m]])
    assert lazy_strip[3] < lazy[3]                           # a strip draws (and sorts) less than the frame
    worker.set_visibility_cull(False)
    worker.terminate()
    mesh.dispose()
