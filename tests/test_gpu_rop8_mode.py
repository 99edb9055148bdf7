"""-m gpu: GS_DRAW_ROP8 - the reference's blend state as a GPU executes it (SplatMaterial3D.js:65-75: NormalBlending into an RGBA8
target, every channel rounded to 8 bits after EVERY splat, back to front) as a draw mode of the render seam, against the
ROP-emulating oracle (raster_oracle.c, rop8) and against the engine's own verification kernel (gs_mesh_debug_rop8)."""
import numpy as np
import pytest

import helpers
import oracle
from gaussiansplats3d_amd import Context, SplatMesh, camera, create_sort_worker, util
from test_gpu_depth import _occluder, _order

pytestmark = pytest.mark.gpu
EQUAL, MAXDIFF = 0.995, 1          # the verification kernel's own gate (tests/test_gpu_crops.py: ROP8_EQUAL / ROP8_MAX)


@pytest.fixture(scope="module")
def ctx():
    c = Context(0)
    yield c
    c.close()


def _ref8(fb):
    return np.floor(np.clip(fb, 0, 1) * 255.0 + 0.5).astype(np.int32)


@pytest.mark.parametrize("full", [False, True])
@pytest.mark.parametrize("sh_degree,cov_half,w,h,n", [(0, False, 256, 144, 4000), (2, False, 320, 200, 20000), (1, True, 200, 120, 3000)])
def test_rop8_draw_matches_the_rop_emulating_oracle_on_every_pixel(ctx, sh_degree, cov_half, w, h, n, full):
    """Both shapes of the mode: GS_DRAW_ROP8 (the splats in front of each quadrant's saturation depth) and GS_DRAW_ROP8_FULL (every list
    to its end)."""
    scene = helpers.small_scene(n, sh_degree, seed=500 + sh_degree, cov_half=cov_half)
    cam = camera.demo_camera("garden", w, h)
    order = _order(scene, cam)
    mesh = SplatMesh(ctx, scene.count, scene.sh_degree, scene.cov_half)
    mesh.build(scene.centers, scene.cov, scene.rgba, scene.sh if scene.sh_degree else None)
    mesh.set_camera(cam)
    mesh.update_render_indexes(order, scene.count)
    fp32, st32 = mesh.render()
    mesh.set_draw_mode(rop8=True, full=full)
    got, st = mesh.render()
    for _ in range(2):                             # the mode's own statistics now order the bins (same mode, same view): same pixels
        np.testing.assert_array_equal(mesh.render()[0], got)
    c, cov, rgba, sh = helpers.oracle_inputs(scene)
    ocam = oracle.make_camera(cam.model_view(), cam.projection, cam.position, w, h, scene.sh_degree, scene.sh_degree)
    (fb8, _), = oracle.render_windows(ocam, c, cov, rgba, sh, order, windows=[(0, 0, w, h)], rop8=True)[0]
    d = np.abs(got.astype(np.int32) - _ref8(fb8))
    assert got[..., 3].any()
    if not full:                                   # the bounded walk: colour to the gate, alpha within 2 steps (tile_blend.hip)
        assert d[..., 3].max() <= 2
        d = d[..., :3]
    assert d.max() <= MAXDIFF and (d == 0).mean() >= EQUAL, (int(d.max()), float((d == 0).mean()))
    # ... and the verification kernel, which walks the same lists with its own formulation of the fragment rule, agrees with the mode
    win = (w // 4, h // 4, 96, 64)
    r8 = mesh.rop8_window(*win).astype(np.int32)
    dd = np.abs(r8 - got[win[1]:win[1] + win[3], win[0]:win[0] + win[2]].astype(np.int32))
    assert dd.max() <= 1 and (dd == 0).mean() >= 0.995, (int(dd.max()), float((dd == 0).mean()))
    # the mode walks every list to its end: at least the pairs the fp32 draw walked before its pixels saturated
    assert st.splats_walked >= st32.splats_walked and st.visible_splats == st32.visible_splats
    # strips reproduce the frame bit for bit, and the fp32 mode comes back unchanged
    rows = (h + 15) // 16
    cut = rows // 2 + 1
    parts = [mesh.render(tile_rows=r)[0] for r in ((0, cut), (cut, rows))]
    np.testing.assert_array_equal(np.concatenate(parts, axis=0), got)
    mesh.set_draw_mode(rop8=False)
    again, _ = mesh.render()
    np.testing.assert_array_equal(again, fp32)
    with pytest.raises(Exception):
        from gaussiansplats3d_amd import _lib as L
        L.check(mesh.lib.gs_mesh_set_draw_mode(mesh.handle, 7))
    mesh.dispose()


def test_rop8_draw_over_a_destination_tests_depth_and_blends_over_its_colour(ctx):
    w, h = 320, 200
    scene = helpers.small_scene(5000, 2, seed=321)
    cam = camera.demo_camera("garden", w, h)
    ocam, s, depth, dst, zw, vis = _occluder(scene, cam, w, h, seed=5)
    order = _order(scene, cam)
    mesh = SplatMesh(ctx, scene.count, scene.sh_degree, scene.cov_half)
    mesh.build(scene.centers, scene.cov, scene.rgba, scene.sh)
    mesh.set_camera(cam)
    mesh.update_render_indexes(order, scene.count)
    for unorm24, full in ((False, False), (True, False), (False, True)):
        mesh.set_draw_mode(rop8=True, full=full)
        mesh.set_destination(depth=depth, rgba=dst, depth_unorm24=unorm24)
        got, _ = mesh.render()
        (fb8, _), = oracle.render_windows(ocam, *s, order, windows=[(0, 0, w, h)], rop8=True, depth=depth, depth_unorm24=unorm24, dst_rgba=dst)[0]
        d = np.abs(got.astype(np.int32) - _ref8(fb8))
        if not full:
            assert d[..., 3].max() <= 2
            d = d[..., :3]
        assert d.max() <= MAXDIFF and (d == 0).mean() >= EQUAL, (unorm24, full, int(d.max()), float((d == 0).mean()))
        corner = got[-h // 6:, -w // 5:]                            # in front of everything: the destination untouched
        np.testing.assert_array_equal(corner, dst[-h // 6:, -w // 5:])
    mesh.set_destination()
    mesh.dispose()


def test_rop8_draw_from_a_device_resident_sort_and_on_one_stream():
    """The mode behind the sorter seam (device-resident order, bound sorter) on a single-stream context: same pixels as from host
    indexes on the default context."""
    w, h = 400, 230
    scene = helpers.small_scene(30000, 1, seed=91)
    cam = camera.demo_camera("garden", w, h)
    frames = []
    for single in (True, False):
        c = Context(0, single_stream=single)
        mesh = SplatMesh(c, scene.count, scene.sh_degree)
        mesh.build(scene.centers, scene.cov, scene.rgba, scene.sh)
        mesh.set_camera(cam)
        mesh.set_draw_mode(rop8=True)
        if single:
            wk = create_sort_worker(c, scene.count)
            wk.post_message({"centers": util.integer_centers(scene.centers), "range": {"from": 0, "to": scene.count - 1, "count": scene.count}})
            wk.sort_on_device(cam.sort_mvp(), scene.count)
            mesh.use_sorter_result(wk, scene.count)
            for _ in range(2):
                wk.sort_on_device(cam.sort_mvp(), scene.count)
                img, _ = mesh.render()
            wk.terminate()
        else:
            mesh.update_render_indexes(_order(scene, cam), scene.count)
            img, _ = mesh.render()
        frames.append(img)
        mesh.dispose()
        c.close()
    np.testing.assert_array_equal(frames[0], frames[1])


def test_rop8_statistics_do_not_schedule_the_next_fp32_draw(ctx):
    """A GS_DRAW_ROP8 draw walks every list whole: its per-bin counters would order the next fp32 draw's bins - and send hundreds of
    them to the deep pass - from numbers that say nothing about an fp32 draw.  The draw after a mode switch is scheduled as a first
    draw is (no order, no deep bins), its pixels are the fp32 frame's, and the draws after it use their own statistics again."""
    scene = helpers.small_scene(120000, 0, seed=17, scale=0.03)
    cam = camera.demo_camera("garden", 640, 360)
    mesh = SplatMesh(ctx, scene.count, 0).build(scene.centers, scene.cov, scene.rgba, None)
    mesh.set_camera(cam)
    mesh.update_render_indexes(_order(scene, cam), scene.count)
    for _ in range(3):
        fp32, st = mesh.render()
    mesh.set_draw_mode(rop8=True, full=True)
    _, st8 = mesh.render()
    assert st8.splats_walked > st.splats_walked
    mesh.set_draw_mode(rop8=False)
    for _ in range(3):
        again, st2 = mesh.render()
        np.testing.assert_array_equal(again, fp32)
        assert mesh.deep_pass_info()["bins"].size == 0          # nothing of this small frame belongs in the deep pass
    assert st2.splats_walked == st.splats_walked
    mesh.dispose()
