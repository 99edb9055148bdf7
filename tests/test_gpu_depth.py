"""-m gpu: the render seam's DESTINATION (gs_mesh_set_destination) - the reference's `depthTest: true, depthWrite: false`
(/root/reference/src/splatmesh/SplatMaterial3D.js:72-73) against what other scene geometry drew first, and NormalBlending over its
colour (draw order src/Viewer.js:1610-1616; drop-in mode src/DropInViewer.js:34-42).  The engine against the raster oracle with the
same destination (oracle semantics pinned in tests/test_depth_oracle.py): strict 1/255; strips, both executors of the chunked
composite and both contexts produce the same bits; the RGBA8-ROP verification kernel sees the same destination."""
import numpy as np
import pytest

import helpers
import oracle
from gaussiansplats3d_amd import Context, GsError, SplatMesh, camera, create_sort_worker, util
from test_gpu_deep import Rig, _pile

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = Context(0)
    yield c
    c.close()


def _occluder(scene, cam, w, h, seed, tilt=0.02):
    """An opaque plane through the MIDDLE of the scene, as the host's own geometry would leave it in the depth buffer: the stored
    depth is the median splat's, tilted across the frame so that the cut runs THROUGH splats' footprints (the test is per pixel,
    not per splat), with a hole (depth 1: nothing drawn there) and a region in front of everything (depth 0)."""
    c, cov, rgba, sh = helpers.oracle_inputs(scene)
    ocam = oracle.make_camera(cam.model_view(), cam.projection, cam.position, w, h, scene.sh_degree, scene.sh_degree)
    p = oracle.project(ocam, c, cov, rgba, sh)
    zw = (p["ndcz"] * np.float32(0.5) + np.float32(0.5)).astype(np.float32)
    vis = p["visible"] == 1
    mid, spread = np.quantile(zw[vis], 0.2), zw[vis].std()
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    depth = (mid + tilt * spread * ((xx - w / 2) / w + (yy - h / 2) / h) * 8.0).astype(np.float32)
    depth[: h // 6, : w // 5] = 1.0
    depth[-h // 6:, -w // 5:] = 0.0
    rng = np.random.default_rng(seed)
    dst = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
    dst[..., 3] = np.where(depth < 1.0, 255, 0)                # opaque where the occluder is, untouched clear colour in the hole
    dst[depth >= 1.0, :3] = 0
    return ocam, (c, cov, rgba, sh), depth, dst, zw, vis


def _order(scene, cam):
    return oracle.sort_indexes(np.arange(scene.count, dtype=np.uint32), util.integer_centers(scene.centers), cam.sort_mvp())


@pytest.mark.parametrize("sh_degree,cov_half,w,h,unorm24,with_colour", [(0, False, 256, 144, False, True), (2, False, 320, 200, True, True),
                                                                        (1, True, 200, 120, False, False)])
def test_frame_with_an_occluder_through_the_scene_matches_the_oracle(ctx, sh_degree, cov_half, w, h, unorm24, with_colour):
    scene = helpers.small_scene(4000, sh_degree, seed=300 + sh_degree, cov_half=cov_half)
    cam = camera.demo_camera("garden", w, h)
    ocam, s, depth, dst, zw, vis = _occluder(scene, cam, w, h, seed=5)
    order = _order(scene, cam)
    mesh = SplatMesh(ctx, scene.count, scene.sh_degree, scene.cov_half)
    mesh.build(scene.centers, scene.cov, scene.rgba, scene.sh if scene.sh_degree else None)
    mesh.set_camera(cam)
    mesh.update_render_indexes(order, scene.count)
    plain, _ = mesh.render()
    mesh.set_destination(depth=depth, rgba=dst if with_colour else None, depth_unorm24=unorm24)
    got, stats = mesh.render()
    fb, q, amb, frags = oracle.render(ocam, *s, order, depth=depth, depth_unorm24=unorm24, dst_rgba=dst if with_colour else None)
    fb0, _, _, frags0 = oracle.render(ocam, *s, order)
    assert 0.15 * frags0 < frags < 0.85 * frags0                 # the occluder really hides about half of the fragments
    print(helpers.compare_frames(got, fb, amb, f"depth-tested sh{sh_degree} unorm24={unorm24} colour={with_colour}", strict=True))
    assert not np.array_equal(got, plain)
    # in front of everything: the destination untouched; the rop8 verification kernel sees the same destination
    corner = got[-h // 6:, -w // 5:]
    assert np.array_equal(corner, dst[-h // 6:, -w // 5:] if with_colour else np.zeros_like(corner))
    win = (w // 3, h // 3, 64, 48)
    r8 = mesh.rop8_window(*win)
    crops, _ = oracle.render_windows(ocam, *s, order, windows=[win], rop8=True, depth=depth, depth_unorm24=unorm24,
                                     dst_rgba=dst if with_colour else None)
    ofb = crops[0][0]
    ref8 = np.floor(np.clip(ofb, 0, 1) * 255.0 + 0.5).astype(np.int32)
    d8 = np.abs(r8.astype(np.int32) - ref8)
    assert d8.max() <= 1 and (d8 == 0).mean() >= 0.995, (int(d8.max()), float((d8 == 0).mean()))
    # clearing the destination restores the plain frame, bit for bit
    mesh.set_destination()
    again, _ = mesh.render()
    assert np.array_equal(again, plain)
    mesh.dispose()


def test_strips_contexts_and_device_buffers_reproduce_the_depth_tested_frame(ctx):
    import torch
    w, h = 400, 230
    scene = helpers.small_scene(6000, 2, seed=77)
    cam = camera.demo_camera("garden", w, h)
    _, _, depth, dst, _, _ = _occluder(scene, cam, w, h, seed=9)
    frames = {}
    for name, c in (("serial", Context(0, single_stream=True)), ("default", ctx)):
        mesh = SplatMesh(c, scene.count, 2)
        mesh.build(scene.centers, scene.cov, scene.rgba, scene.sh)
        mesh.set_camera(cam)
        wk = create_sort_worker(c, scene.count)
        wk.post_message({"centers": util.integer_centers(scene.centers), "range": {"from": 0, "to": scene.count - 1, "count": scene.count}})
        mesh.use_sorter_result(wk, scene.count)
        if name == "serial":
            mesh.set_destination(depth=depth, rgba=dst, depth_unorm24=True)
        else:                                                  # the same destination handed over as device buffers
            d_dev = torch.from_numpy(depth).cuda()
            c_dev = torch.from_numpy(dst).cuda()
            torch.cuda.synchronize()
            mesh.set_destination(depth_device_ptr=d_dev.data_ptr(), rgba_device_ptr=c_dev.data_ptr(), size=(w, h), depth_unorm24=True)
        wk.sort_on_device(cam.sort_mvp(), scene.count)
        full, _ = mesh.render()
        frames[name] = full
        rows = (h + 15) // 16
        for cuts in ([(0, 3), (3, 4), (4, 9), (9, rows)], [(0, rows // 2), (rows // 2, rows)]):
            strips = []
            for r in cuts:                                     # a rank's frame: vertex stage of the strip, culled sort, draw
                wk.set_visibility_cull(True)
                mesh.project(r)
                wk.sort_on_device(cam.sort_mvp(), scene.count)
                strips.append(mesh.render(tile_rows=r)[0])
            wk.set_visibility_cull(False)
            assert np.array_equal(np.concatenate(strips, axis=0), full), (name, cuts)
        # asynchronous frames in flight (two record sets on the default context) draw the same pixels
        wk.sort_on_device(cam.sort_mvp(), scene.count)
        for _ in range(6):
            mesh.render(want_stats=False, to_host=False)
        last, _ = mesh.render()
        assert np.array_equal(last, full)
        wk.terminate()
        mesh.dispose()
        if name == "serial":
            c.close()
    assert np.array_equal(frames["serial"], frames["default"])


def test_deep_pass_and_per_bin_kernel_agree_under_a_destination(ctx):
    """The chunked composite's two executors with a depth test in the chain: a pile thousands of splats deep, half of it behind
    the occluder."""
    W, H = 480, 270
    cam = camera.demo_camera("garden", W, H)
    scene = _pile(60000, 41)
    _, _, depth, dst, _, _ = _occluder(scene, cam, W, H, seed=11, tilt=0.3)
    rig = Rig(ctx, scene, cam)
    rig.mesh.set_destination(depth=depth, rgba=dst)
    rig.mesh.set_deep_pass(False)
    plain, st0 = rig.draw()
    assert rig.mesh.deep_pass_info()["chunks_closed_by_bins"] >= 2
    rig.mesh.set_deep_pass(True)
    frames = [rig.draw() for _ in range(3)]
    assert len(rig.mesh.deep_pass_info()["bins"]) >= 1
    for f, _ in frames:
        np.testing.assert_array_equal(f, plain)
    rig.close()


def test_destination_validation_and_an_empty_draw(ctx):
    scene = helpers.small_scene(500, 0, seed=2)
    cam = camera.demo_camera("garden", 128, 80)
    mesh = SplatMesh(ctx, scene.count, 0)
    mesh.build(scene.centers, scene.cov, scene.rgba, None)
    mesh.set_camera(cam)
    rng = np.random.default_rng(4)
    dst = rng.integers(0, 256, size=(80, 128, 4), dtype=np.uint8)
    mesh.set_destination(rgba=dst)
    mesh.update_render_indexes(np.zeros(0, dtype=np.uint32), 0)
    empty, _ = mesh.render()
    assert np.array_equal(empty, dst)                          # no splats: the destination colour comes back unchanged
    mesh.set_destination(depth=np.ones((64, 64), dtype=np.float32))
    mesh.update_render_indexes(np.arange(scene.count, dtype=np.uint32), scene.count)
    with pytest.raises(GsError):                               # not this camera's viewport
        mesh.render()
    mesh.set_destination()
    mesh.render()
    mesh.dispose()
