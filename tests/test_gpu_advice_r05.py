"""-m gpu: regression tests for the round-5 advisor findings, each named after what used to go wrong."""
import ctypes as C

import numpy as np
import pytest

import helpers
import oracle
from gaussiansplats3d_amd import Context, SplatMesh, SplatTree, camera, create_sort_worker, util
from gaussiansplats3d_amd import _lib as L
from gaussiansplats3d_amd._lib import GsError

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = Context(0, single_stream=True)
    yield c
    c.close()


def _mesh(ctx, n=3000, w=128, h=80, seed=7):
    scene = helpers.small_scene(n, 0, seed=seed)
    cam = camera.demo_camera("garden", w, h)
    mesh = SplatMesh(ctx, scene.count, 0).build(scene.centers, scene.cov, scene.rgba, None)
    mesh.set_camera(cam)
    mesh.update_render_indexes(np.arange(scene.count, dtype=np.uint32), scene.count)
    return scene, cam, mesh


def test_set_destination_refuses_arrays_that_disagree_or_are_not_images(ctx):
    """A depth array smaller than rgba (or a flat one) made gs_mesh_set_destination copy width * height * 4 bytes out of a shorter
    numpy buffer; device pointers without a size died on `size[0]`."""
    _, cam, mesh = _mesh(ctx)
    H, W = cam.height, cam.width
    depth = np.ones((H, W), np.float32)
    rgba = np.zeros((H, W, 4), np.uint8)
    with pytest.raises(ValueError):
        mesh.set_destination(depth=depth[: H // 2], rgba=rgba)             # sizes disagree
    with pytest.raises(ValueError):
        mesh.set_destination(depth=depth.reshape(-1))                       # not [H, W]
    with pytest.raises(ValueError):
        mesh.set_destination(rgba=rgba[:, :, :3])                           # not [H, W, 4]
    with pytest.raises(ValueError):
        mesh.set_destination(depth_device_ptr=0x1000)                       # device pointer without a size
    with pytest.raises(ValueError):
        mesh.set_destination(depth=depth, size=(W + 1, H))                  # size disagrees with the array
    with pytest.raises(ValueError):
        mesh.set_destination(depth=depth, depth_device_ptr=0x1000, size=(W, H))
    plain, _ = mesh.render()
    mesh.set_destination(depth=depth, rgba=rgba)                            # the well-formed call still works
    with_dest, _ = mesh.render()
    np.testing.assert_array_equal(with_dest, plain)                         # depth 1.0 passes everything, colour (0,0,0,0) adds nothing
    mesh.set_destination()
    mesh.dispose()


def test_a_refused_destination_leaves_the_mesh_without_one(ctx):
    """The header promises that a refused gs_mesh_set_destination leaves the mesh WITHOUT a destination; the C side used to commit
    the depth pointer before the later steps could fail."""
    _, cam, mesh = _mesh(ctx)
    plain, _ = mesh.render()
    depth = np.zeros((cam.height, cam.width), np.float32)                   # depth 0 in front of everything: an empty frame
    mesh.set_destination(depth=depth)
    hidden, _ = mesh.render()
    assert hidden.any() == False and plain.any()
    d = L.Destination()
    d.depth_host = depth.ctypes.data
    d.width, d.height = cam.width, cam.height
    d.flags = 0x80                                                          # unknown flag: refused after the old destination was dropped
    assert mesh.lib.gs_mesh_set_destination(mesh.handle, C.byref(d)) < 0
    again, _ = mesh.render()
    np.testing.assert_array_equal(again, plain)                             # no destination at all: neither the old nor half of the new
    mesh.dispose()


def test_rop8_verification_refuses_a_destination_the_last_draw_did_not_see(ctx):
    """gs_mesh_debug_rop8 combined the LAST draw's width / depth mode with the mesh's CURRENT destination pointers."""
    _, cam, mesh = _mesh(ctx)
    mesh.render()
    mesh.rop8_window(0, 0, 32, 32)
    mesh.set_destination(depth=np.full((cam.height, cam.width), 0.5, np.float32))
    with pytest.raises(GsError):
        mesh.rop8_window(0, 0, 32, 32)                                      # set since the draw
    mesh.render()
    mesh.rop8_window(0, 0, 32, 32)
    mesh.set_destination()
    with pytest.raises(GsError):
        mesh.rop8_window(0, 0, 32, 32)                                      # cleared since the draw
    mesh.render()
    mesh.rop8_window(0, 0, 32, 32)
    mesh.dispose()


def test_timed_draws_do_not_leak_into_the_kernel_clock(ctx):
    """Timed draws bracket the whole vertex stage, sampled untimed draws k_project alone; both used to be summed into one clock."""
    _, cam, mesh = _mesh(ctx, n=20000, w=320, h=200)
    mesh.render(to_host=False, want_stats=False)
    ctx.synchronize()
    mesh.kernel_time(0, reset=True)
    for _ in range(6):
        mesh.render(to_host=False, want_stats=True)                         # timed: whole-stage samples only
    k_ms, k_n = mesh.kernel_time(0, reset=False)
    s_ms, s_n = mesh.kernel_time(1, reset=True)
    assert k_n == 0 and k_ms == 0.0
    assert s_n == 6 and s_ms > 0.0
    for _ in range(16):
        mesh.render(to_host=False, want_stats=False)                        # untimed: every 8th launch is a kernel-only sample
    k_ms, k_n = mesh.kernel_time(0, reset=False)
    s_ms, s_n = mesh.kernel_time(1, reset=True)
    assert k_n >= 1 and k_ms > 0.0 and s_n == 0
    with pytest.raises(GsError):
        mesh.kernel_time(2)
    mesh.dispose()


def test_a_gathered_list_survives_a_full_frame_visibility_culled_sort():
    """The streaming front end of a visibility-culled sort writes pay_in and never touches idx_in, yet has_gathered was dropped on
    every visibility-culled sort: a later gs_sorter_sort_gathered failed for no reason."""
    ctx = Context(0)
    scene = helpers.small_scene(30000, 0, seed=41)
    n = scene.count
    ci = util.integer_centers(scene.centers)
    cam = camera.demo_camera("garden", 640, 360)
    mvp = cam.sort_mvp()
    tree = SplatTree(ctx, 8, 200).process_splat_mesh(scene.centers)
    w = create_sort_worker(ctx, n)
    w.post_message({"centers": ci, "range": {"from": 0, "to": n - 1, "count": n}})
    mesh = SplatMesh(ctx, n, 0).build(scene.centers, scene.cov, scene.rgba, None)
    mesh.set_camera(cam)
    g = tree.gather_scene_nodes_for_sort(cam, sort_worker=w, to_host=True)
    idx = g["indexesToSort"].copy()
    first = w.sort_gathered(mvp)["sortedIndexes"].copy()                    # the list now sits in idx_in, copied
    np.testing.assert_array_equal(first, oracle.sort_indexes(idx, ci, mvp))
    w.sort_on_device(mvp, n)
    mesh.use_sorter_result(w, n)
    w.set_visibility_cull(True)
    mesh.project()                                                          # a full frame: the streaming front end
    w.sort_on_device(mvp, n)
    mesh.render(to_host=False)
    w.set_visibility_cull(False)
    second = w.sort_gathered(mvp)["sortedIndexes"]
    np.testing.assert_array_equal(second, first)
    w.terminate()
    mesh.dispose()
    tree.dispose()
    ctx.close()
