"""-m gpu: the HIP render seam against the fp32 raster oracle, through the C ABI."""
import numpy as np
import pytest

import helpers
import oracle
from gaussiansplats3d_amd import Context, SplatMesh, camera, create_sort_worker, scenes, util

pytestmark = pytest.mark.gpu
K_POWER = np.float32(2.4022448)


@pytest.fixture(scope="module")
def ctx():
    c = Context(0)
    yield c
    c.close()


def build_mesh(ctx, scene, **kw):
    mesh = SplatMesh(ctx, scene.count, scene.sh_degree, scene.cov_half, **kw)
    mesh.build(scene.centers, scene.cov, scene.rgba, scene.sh if scene.sh_degree else None)
    return mesh


def sorted_order(scene, cam):
    ci = util.integer_centers(scene.centers)
    return oracle.sort_indexes(np.arange(scene.count, dtype=np.uint32), ci, cam.sort_mvp())


def oracle_frame(scene, cam, order, sh_degree=None, **kw):
    c, cov, rgba, sh = helpers.oracle_inputs(scene)
    ocam = oracle.make_camera(cam.model_view(), cam.projection, cam.position, cam.width, cam.height,
                              sh_degree=scene.sh_degree if sh_degree is None else sh_degree,
                              sh_stored=scene.sh_degree, **kw)
    return ocam, oracle.render(ocam, c, cov, rgba, sh, order)


@pytest.mark.parametrize("sh_degree,cov_half,w,h", [(0, False, 256, 144), (1, False, 200, 120), (2, False, 256, 144),
                                                    (2, True, 320, 200), (0, True, 130, 70)])
def test_framebuffer_matches_oracle(ctx, sh_degree, cov_half, w, h):
    scene = helpers.small_scene(3000, sh_degree, seed=100 + sh_degree, cov_half=cov_half)
    cam = camera.demo_camera("garden", w, h)
    order = sorted_order(scene, cam)
    mesh = build_mesh(ctx, scene)
    mesh.set_camera(cam)
    mesh.update_render_indexes(order, scene.count)
    got, stats = mesh.render()
    _, (fb, q, amb, frags) = oracle_frame(scene, cam, order)
    assert frags > 1000 and stats.tile_entries > 0
    print(helpers.compare_frames(got, fb, amb, f"sh{sh_degree} half={cov_half} {w}x{h}"))
    mesh.dispose()


def test_vertex_stage_records_match_oracle(ctx):
    scene = helpers.small_scene(5000, 2, seed=7)
    cam = camera.demo_camera("garden", 640, 360)
    mesh = build_mesh(ctx, scene)
    mesh.set_camera(cam)
    mesh.update_render_indexes(np.arange(scene.count, dtype=np.uint32), scene.count)
    mesh.render()
    recs, rects, on_screen = mesh.debug_records()
    c, cov, rgba, sh = helpers.oracle_inputs(scene)
    ocam = oracle.make_camera(cam.model_view(), cam.projection, cam.position, cam.width, cam.height, 2, 2)
    o = oracle.project(ocam, c, cov, rgba, sh)
    f = recs.view(np.float32)
    assert ((rects[on_screen, 0] & 0xFFFF) <= (rects[on_screen, 1] & 0xFFFF)).all()
    vis = o["visible"] == 1
    # every splat we keep is visible for the oracle; the ones we drop are invisible or touch no pixel centre
    assert not (on_screen & ~vis).any()
    k = on_screen
    assert k.sum() > 1000
    np.testing.assert_allclose(f[k, 0], o["cx"][k], rtol=0, atol=2e-3)
    np.testing.assert_allclose(f[k, 1], o["cy"][k], rtol=0, atol=2e-3)
    n1 = o["b1x"] ** 2 + o["b1y"] ** 2
    n2 = o["b2x"] ** 2 + o["b2y"] ** 2
    for col, num, den in ((2, "b1x", n1), (3, "b1y", n1), (4, "b2x", n2), (5, "b2y", n2)):
        exp = K_POWER * o[num][k] / den[k]
        np.testing.assert_allclose(f[k, col], exp, rtol=2e-4, atol=1e-7)
    r16 = (recs[k, 6] & 0xFFFF) / 65535.0
    g16 = (recs[k, 6] >> 16) / 65535.0
    b16 = (recs[k, 7] & 0xFFFF) / 65535.0
    a16 = (recs[k, 7] >> 16) / 65535.0
    for got, name in ((r16, "r"), (g16, "g"), (b16, "b"), (a16, "a")):
        np.testing.assert_allclose(got, o[name][k], rtol=0, atol=1.0 / 65535.0 + 1e-6)
    mesh.dispose()


def test_sorter_result_stays_on_device_and_matches_host_indexes(ctx):
    scene = helpers.small_scene(4000, 0, seed=21)
    cam = camera.demo_camera("garden", 320, 180)
    mesh = build_mesh(ctx, scene)
    mesh.set_camera(cam)
    order = sorted_order(scene, cam)
    mesh.update_render_indexes(order, scene.count)
    a, _ = mesh.render()
    w = create_sort_worker(ctx, scene.count)
    ci = util.integer_centers(scene.centers)
    w.post_message({"centers": ci, "range": {"from": 0, "to": scene.count - 1, "count": scene.count}})
    w.sort_on_device(cam.sort_mvp(), scene.count)
    mesh.use_sorter_result(w, scene.count)
    b, _ = mesh.render()
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(w.debug_read(2, scene.count), order)
    w.terminate()
    mesh.dispose()


def test_tile_row_strips_tile_the_full_frame(ctx):
    """Multi-GPU sharding unit: rendering tile-row strips separately reproduces the full frame exactly."""
    scene = helpers.small_scene(3000, 1, seed=33)
    cam = camera.demo_camera("garden", 300, 170)           # 11 tile rows, last one partial
    mesh = build_mesh(ctx, scene)
    mesh.set_camera(cam)
    mesh.update_render_indexes(sorted_order(scene, cam), scene.count)
    full, _ = mesh.render()
    parts = [mesh.render(tile_rows=r)[0] for r in ((0, 3), (3, 4), (4, 9), (9, 11))]
    np.testing.assert_array_equal(np.concatenate(parts, axis=0), full)
    mesh.dispose()


def test_draw_order_is_honoured(ctx):
    """Two overlapping opaque-ish splats: swapping the index order swaps which one is on top."""
    cam = camera.demo_camera("garden", 64, 64)
    pos = np.array(camera.DEMO_POSES["garden"][1]); look = np.array(camera.DEMO_POSES["garden"][2])
    fwd = (look - pos) / np.linalg.norm(look - pos)
    centers = np.stack([pos + fwd * 2.0, pos + fwd * 2.5]).astype(np.float32)
    cov = np.tile(np.array([[0.05, 0, 0, 0.05, 0, 0.05]], np.float32), (2, 1))
    rgba = np.array([[255, 0, 0, 250], [0, 0, 255, 250]], np.uint8)
    from gaussiansplats3d_amd import scenes
    scene = scenes.SplatScene(centers, cov, rgba, np.zeros((2, 0), np.float16), 0)
    mesh = build_mesh(ctx, scene)
    mesh.set_camera(cam)
    c, covo, rg, _ = helpers.oracle_inputs(scene)
    for order in (np.array([1, 0], np.uint32), np.array([0, 1], np.uint32)):
        mesh.update_render_indexes(order, 2)
        got, _ = mesh.render()
        ocam = oracle.make_camera(cam.model_view(), cam.projection, cam.position, 64, 64)
        fb, q, amb, _ = oracle.render(ocam, c, covo, rg, None, order)
        helpers.compare_frames(got, fb, amb, "order")
        top = got[32, 32]
        assert (top[0] > top[2]) == (order[-1] == 0)        # last drawn = on top
    mesh.dispose()


def test_empty_and_offscreen(ctx):
    scene = helpers.small_scene(500, 0, seed=5)
    pos = np.array(camera.DEMO_POSES["garden"][1]); look = np.array(camera.DEMO_POSES["garden"][2])
    fwd = (look - pos) / np.linalg.norm(look - pos)
    behind = pos - fwd * np.linspace(0.5, 9.0, 500)[:, None]                # strictly behind the eye
    scene.centers[:] = (behind + np.cross(fwd, [0.3, 0.1, 0.2]) * np.sin(np.arange(500))[:, None]).astype(np.float32)
    cam = camera.demo_camera("garden", 128, 72)
    mesh = build_mesh(ctx, scene)
    mesh.set_camera(cam)
    mesh.update_render_indexes(np.arange(500, dtype=np.uint32), 500)
    got, stats = mesh.render()
    assert stats.tile_entries == 0 and not got.any()
    mesh.update_render_indexes(np.zeros(0, np.uint32), 0)   # renderSplatCount == 0
    got, stats = mesh.render()
    assert not got.any()
    mesh.dispose()


def test_entry_buffer_grows_on_overflow(ctx):
    """A few hundred near-camera splats cover the whole 1080p screen: with the entry buffers shrunk to 4096 entries the
    draw overflows and must regrow them and redraw."""
    rng = np.random.default_rng(9)
    n = 700
    cam = camera.demo_camera("garden", 1920, 1080)
    pos = np.array(camera.DEMO_POSES["garden"][1]); look = np.array(camera.DEMO_POSES["garden"][2])
    fwd = (look - pos) / np.linalg.norm(look - pos)
    from gaussiansplats3d_amd import scenes
    centers = (pos + fwd * rng.uniform(1.0, 2.0, (n, 1)) + rng.normal(size=(n, 3)) * 0.05).astype(np.float32)
    cov = np.tile(np.array([[0.5, 0, 0, 0.5, 0, 0.5]], np.float32), (n, 1))
    rgba = rng.integers(1, 255, (n, 4), dtype=np.uint8); rgba[:, 3] = 3
    scene = scenes.SplatScene(centers, cov, rgba, np.zeros((n, 0), np.float16), 0)
    mesh = build_mesh(ctx, scene)
    mesh.debug_set_entry_capacity(4096)
    mesh.set_camera(cam)
    order = sorted_order(scene, cam)
    mesh.update_render_indexes(order, n)
    got, stats = mesh.render()
    assert stats.overflowed == 1 and stats.tile_entries > 8 * n and stats.entry_capacity >= stats.tile_entries
    _, (fb, q, amb, _) = oracle_frame(scene, cam, order)
    print(helpers.compare_frames(got, fb, amb, "overflow"))
    mesh.dispose()


def _rot(axis, deg):
    a = np.radians(deg)
    c, s = np.cos(a), np.sin(a)
    x, y, z = np.asarray(axis, np.float64) / np.linalg.norm(axis)
    return np.array([[c + x * x * (1 - c), x * y * (1 - c) - z * s, x * z * (1 - c) + y * s],
                     [y * x * (1 - c) + z * s, c + y * y * (1 - c), y * z * (1 - c) - x * s],
                     [z * x * (1 - c) - y * s, z * y * (1 - c) + x * s, c + z * z * (1 - c)]])


def _cov6(scale, R=np.eye(3)):
    M = R @ np.diag(np.asarray(scale, np.float64))
    S = M @ M.T
    return [S[0, 0], S[0, 1], S[0, 2], S[1, 1], S[1, 2], S[2, 2]]


# (label, centre in view space of a camera at the origin looking down -z, cov6, expected to be drawn)
_TANX = np.tan(np.radians(25.0)) * 640 / 360
HAND_PICKED = [
    ("axis aligned", (0.0, 0.0, -5.0), _cov6((0.1, 0.2, 0.05)), True),
    ("rotated", (0.7, -0.4, -4.0), _cov6((0.3, 0.05, 0.1), _rot((1, 2, 3), 37.0)), True),
    ("needle seen end-on", (-0.5, 0.3, -3.0), _cov6((0.01, 0.01, 0.6)), True),
    ("behind the camera", (0.0, 0.0, 5.0), _cov6((0.1, 0.1, 0.1)), False),
    ("inside the 1.2x guard band", (1.19 * _TANX * 6.0, 0.4, -6.0), _cov6((0.5, 0.5, 0.5)), True),   # centre off screen, quad on
    ("outside the 1.2x guard band", (1.21 * _TANX * 6.0, 0.4, -6.0), _cov6((0.2, 0.2, 0.2)), False),
    ("outside the guard band in y", (0.0, -1.21 * np.tan(np.radians(25.0)) * 6.0, -6.0), _cov6((0.2, 0.2, 0.2)), False),
    ("beyond the far plane", (0.0, 0.0, -1500.0), _cov6((5.0, 5.0, 5.0)), False),
    ("nearer than the near plane", (0.0, 0.0, -0.05), _cov6((0.001, 0.001, 0.001)), False),
    # r = sqrt(max(0.1, .)) makes lambda2 = t - r negative for sub-pixel splats (a ~ d ~ 0.3): the reference drops them
    ("eigenvalue floor, lambda2 < 0", (0.2, 0.2, -40.0), _cov6((1e-4, 1e-4, 1e-4)), False),
    ("zero covariance", (-0.3, 0.1, -2.0), _cov6((0.0, 0.0, 0.0)), False),
    ("eigenvalue floor active, kept", (0.2, 0.2, -40.0), _cov6((0.0463, 0.0463, 0.0463)), True),
    ("1024 px clamp active", (0.1, 0.0, -0.4), _cov6((3.0, 0.002, 0.002), _rot((0, 0, 1), 20.0)), True),
    ("large and close, both axes clamped", (0.0, 0.05, -0.3), _cov6((2.0, 2.0, 0.01)), True),
    ("taller than wide at the centre", (0.0, 0.0, -3.0), _cov6((0.05, 0.3, 0.05)), True),
    ("far and faint", (3.0, 2.0, -600.0), _cov6((2.0, 1.0, 3.0), _rot((1, 0, 1), 60.0)), True),
]


def test_hand_picked_projection_cases(ctx):
    """vertex-stage known answers on hand-picked splats — accept / reject decisions must be
    identical and the kept records equal to the oracle's; then the composited frame of the same set."""
    W, H = 640, 360
    cam = camera.PerspectiveCamera(W, H, (0.0, 0.0, 0.0), (0.0, 0.0, -1.0), (0.0, 1.0, 0.0))
    n = len(HAND_PICKED)
    centers = np.array([c for _, c, _, _ in HAND_PICKED], dtype=np.float32)
    cov = np.array([v for _, _, v, _ in HAND_PICKED], dtype=np.float32)
    rng = np.random.default_rng(5)
    rgba = rng.integers(40, 256, size=(n, 4), dtype=np.uint8)
    mesh = SplatMesh(ctx, n, 0, False)
    mesh.build(centers, cov, rgba, None)
    mesh.set_camera(cam)
    order = np.argsort(centers[:, 2], kind="stable").astype(np.uint32)        # far -> near (view z ascending)
    mesh.update_render_indexes(order, n)
    got, stats = mesh.render()
    recs, rects, kept = mesh.debug_records()
    ocam = oracle.make_camera(cam.model_view(), cam.projection, cam.position, W, H, 0, 0)
    o = oracle.project(ocam, centers, cov, rgba, None)
    f = recs.view(np.float32)
    for i, (label, _, _, drawn) in enumerate(HAND_PICKED):
        assert bool(o["visible"][i]) == drawn, f"oracle: {label}"
        assert bool(kept[i]) == drawn, f"device: {label}"       # every drawn case of the table touches pixel centres
        if not drawn:
            continue
        assert abs(f[i, 0] - o["cx"][i]) <= 2e-3 and abs(f[i, 1] - o["cy"][i]) <= 2e-3, label
        n1 = o["b1x"][i] ** 2 + o["b1y"][i] ** 2
        n2 = o["b2x"][i] ** 2 + o["b2y"][i] ** 2
        exp = K_POWER * np.array([o["b1x"][i] / n1, o["b1y"][i] / n1, o["b2x"][i] / n2, o["b2y"][i] / n2])
        np.testing.assert_allclose(f[i, 2:6], exp, rtol=3e-4, atol=1e-6, err_msg=label)
    # the clamp and the floor really are active in the cases that claim it
    i = [l for l, *_ in HAND_PICKED].index("1024 px clamp active")
    assert abs(np.hypot(o["b1x"][i], o["b1y"][i]) - 1024.0) < 1e-2
    i = [l for l, *_ in HAND_PICKED].index("eigenvalue floor active, kept")
    l1 = (o["b1x"][i] ** 2 + o["b1y"][i] ** 2) / 8.0
    l2 = (o["b2x"][i] ** 2 + o["b2y"][i] ** 2) / 8.0
    assert abs((l1 - l2) - 2.0 * np.sqrt(0.1)) < 1e-4          # lambda1 - lambda2 = 2 r with r pinned at sqrt(0.1)
    fb, q, amb, frags = oracle.render(ocam, centers, cov, rgba, None, order)
    print(helpers.compare_frames(got, fb, amb, "hand-picked"))
    mesh.dispose()


def test_wide_entry_keys_match_the_16_bit_path(ctx, monkeypatch):
    """More than 65536 list bins switch the entry sort to 32-bit keys (a 65536 x 17408 px viewport at 128-px lists), so the
    path is forced through the GSPLAT_WIDE_ENTRY_KEYS test hook instead; it must produce the same pixels."""
    scene = helpers.small_scene(3000, 1, seed=61)
    cam = camera.demo_camera("garden", 1000, 600)
    order = sorted_order(scene, cam)
    mesh = build_mesh(ctx, scene)
    mesh.set_camera(cam)
    mesh.update_render_indexes(order, scene.count)
    narrow, s_narrow = mesh.render()
    mesh.dispose()
    monkeypatch.setenv("GSPLAT_WIDE_ENTRY_KEYS", "1")
    wide_ctx = Context(0)
    mesh = build_mesh(wide_ctx, scene)
    mesh.set_camera(cam)
    mesh.update_render_indexes(order, scene.count)
    wide, s_wide = mesh.render()
    assert narrow.any() and s_wide.tile_entries == s_narrow.tile_entries > 0
    np.testing.assert_array_equal(wide, narrow)
    mesh.dispose()
    wide_ctx.close()


def test_every_list_bin_size_gives_the_same_pixels(ctx, monkeypatch):
    """The list-bin size only changes how entries are grouped: 32-px lists (608 of them at 1000x600: a TWO-pass entry sort
    whose last pass publishes the ranges with atomics), 64, 128 and 256 px (one pass, ranges from the digit totals) must
    all produce the same frame, and that frame matches the oracle."""
    scene = helpers.small_scene(4000, 1, seed=71)
    cam = camera.demo_camera("garden", 1000, 600)
    order = sorted_order(scene, cam)
    frames = {}
    for shift in (1, 2, 3, 4):
        monkeypatch.setenv("GSPLAT_LIST_SHIFT", str(shift))
        mesh = build_mesh(ctx, scene)                      # the switch is read when the mesh is created
        mesh.set_camera(cam)
        mesh.update_render_indexes(order, scene.count)
        frames[shift], stats = mesh.render()
        assert stats.list_bin_px == 16 << shift
        parts = [mesh.render(tile_rows=r)[0] for r in ((0, 11), (11, 38))]
        np.testing.assert_array_equal(np.concatenate(parts, axis=0), frames[shift])
        mesh.dispose()
    for shift in (1, 2, 4):
        np.testing.assert_array_equal(frames[shift], frames[3])
    _, (fb, q, amb, frags) = oracle_frame(scene, cam, order)
    print(helpers.compare_frames(frames[3], fb, amb, "list bins"))


def test_asynchronous_draws_choose_their_list_bins_without_statistics(ctx, monkeypatch):
    """The size of the list bins follows what the last draws saw (16-px tiles per visible splat).  That used to happen only when
    somebody asked for statistics: a render loop that never does stayed on the first guess for ever - 128-px lists under a scene of
    tiny splats, every 32-px bin scanning 16 bins' worth of entries (the capture-like stand-in drew in 4.5 ms instead of 1.2).  Every
    draw now leaves {visible, tiles} in mapped host words and the next draws read them."""
    scene = helpers.small_scene(6000, 0, seed=23, scale=0.2)    # large splats: lists larger than the first guess (128 px) pay
    cam = camera.demo_camera("garden", 960, 540)
    order = sorted_order(scene, cam)

    def fresh():
        m = build_mesh(ctx, scene)
        m.set_camera(cam)
        m.update_render_indexes(order, scene.count)
        return m

    a = fresh()                                             # with statistics: the first guess, then what the rule settles on
    frame, first = a.render()
    for _ in range(3):
        _, settled = a.render()
    a.dispose()
    assert settled.list_bin_px != first.list_bin_px, (first.list_bin_px, settled.list_bin_px)  # (else this scene proves nothing)
    b = fresh()                                             # never asks: draws that return nothing to the host
    for _ in range(6):
        b.render(to_host=False, want_stats=False)
        ctx.synchronize()                                   # (a frame boundary: the words of that draw have arrived)
    got, st = b.render()
    assert st.list_bin_px == settled.list_bin_px
    np.testing.assert_array_equal(got, frame)
    b.dispose()
    monkeypatch.setenv("GSPLAT_NO_ASYNC_LIST_BINS", "1")    # the A/B switch: rounds 1-5
    c = fresh()
    for _ in range(6):
        c.render(to_host=False, want_stats=False)
        ctx.synchronize()
    assert c.last_stats().list_bin_px == first.list_bin_px
    c.dispose()


def test_smaller_list_bins_chosen_asynchronously_come_with_room_for_their_entries(ctx):
    """Smaller list bins mean more entries.  When the words a draw left say "smaller bins", the entry buffers grow to the bound those
    words imply (one entry per 16-px tile touched) BEFORE the first draw that uses the new size - not one truncated frame later."""
    from gaussiansplats3d_amd import _lib as L
    scene = helpers.small_scene(60000, 0, seed=29, scale=0.002)   # splats of a pixel or two: one tile each
    cam = camera.demo_camera("garden", 960, 540)
    order = sorted_order(scene, cam)

    def fresh():
        m = build_mesh(ctx, scene)
        m.set_camera(cam)
        m.update_render_indexes(order, scene.count)
        return m

    a = fresh()
    frame, first = a.render()
    for _ in range(3):
        _, settled = a.render()
    a.dispose()
    assert settled.list_bin_px < first.list_bin_px and settled.tile_entries > first.tile_entries, \
        (first.list_bin_px, settled.list_bin_px, int(first.tile_entries), int(settled.tile_entries))
    b = fresh()
    b.debug_set_entry_capacity(max(1024, (int(first.tile_entries) + int(settled.tile_entries)) // 2))   # fits the first guess only
    warnings = 0
    for _ in range(6):
        b.render(to_host=False, want_stats=False)
        warnings += b.last_status == L.GS_WARN_FRAME_TRUNCATED
        ctx.synchronize()
    got, st = b.render()
    assert warnings == 0 and st.list_bin_px == settled.list_bin_px and not st.overflowed
    np.testing.assert_array_equal(got, frame)
    b.dispose()


def test_a_slowly_moving_camera_keeps_the_bin_order_and_the_pixels(ctx, monkeypatch):
    """The blend takes its bins costliest-first from the previous frame while the picture moved with little parallax since (tile_bin.hip:
    a whole-picture shift is followed bin by bin), and in row-major order otherwise; the deep pass's members always come from the previous frame.  Scheduling only:
    every frame of a slow and of a fast camera path equals the frame of a fresh mesh, under the gate, with the gate shut
    ($GSPLAT_ORDER_MOTION=0: the same view only) and wide open (100)."""
    scene = helpers.small_scene(90000, 1, seed=31, scale=0.03)          # tiny splats: deep bins, long lists
    w, h = 640, 360
    slow = camera.orbit_cameras("garden", w, h, 1440)[:6]                # a quarter of a degree per frame
    fast = camera.orbit_cameras("garden", w, h, 30)[:4]                  # twelve degrees per frame
    up, pos, look = (np.asarray(v, dtype=np.float64) for v in camera.DEMO_POSES["garden"])
    axis, rel = up / np.linalg.norm(up), look - pos

    def turned(deg):                                                     # the camera turning about its own position
        a = np.radians(deg)
        r = rel * np.cos(a) + np.cross(axis, rel) * np.sin(a) + axis * np.dot(axis, rel) * (1.0 - np.cos(a))
        return camera.PerspectiveCamera(w, h, tuple(pos), tuple(pos + r), tuple(up))
    turning = [turned(d) for d in (0.0, 3.0, 6.0, 9.0, 4.0, -7.0)]        # (the statistics are read shifted by whole bins)
    path = slow + fast + slow[::-1] + turning
    orders = [sorted_order(scene, c) for c in path]
    want = []
    for c, o in zip(path, orders):
        m = build_mesh(ctx, scene)
        m.set_camera(c)
        m.update_render_indexes(o, scene.count)
        want.append(m.render()[0])
        m.dispose()
    assert any(f.any() for f in want)
    for limit in (None, "0", "100"):
        if limit is None:
            monkeypatch.delenv("GSPLAT_ORDER_MOTION", raising=False)
        else:
            monkeypatch.setenv("GSPLAT_ORDER_MOTION", limit)
        m = build_mesh(ctx, scene)                                       # (the limit is read per draw)
        for lap in range(2):
            for k, (c, o) in enumerate(zip(path, orders)):
                m.set_camera(c)
                m.update_render_indexes(o, scene.count)
                got, _ = m.render()
                np.testing.assert_array_equal(got, want[k], err_msg=f"limit {limit} lap {lap} frame {k}")
        m.dispose()


def test_asynchronous_draws_heal_an_overflowing_entry_buffer(ctx):
    """A draw that returns nothing to the host cannot re-run itself when its entry buffer overflows; the next draw notices
    (mapped host mirror, no synchronisation), grows the buffer and says so once.  Moving camera: every pose needs a
    different number of entries."""
    import torch
    from gaussiansplats3d_amd import _lib as L
    scene = helpers.small_scene(6000, 0, seed=77, scale=0.12)
    mesh = build_mesh(ctx, scene)
    cams = camera.orbit_cameras("garden", 320, 200, 12)
    orders = [sorted_order(scene, c) for c in cams]
    want = []
    for c, o in zip(cams, orders):                           # synchronous draws: the frames to expect
        mesh.set_camera(c)
        mesh.update_render_indexes(o, scene.count)
        want.append(mesh.render()[0])
    mesh.debug_set_entry_capacity(1024)                      # far too small for any pose
    out = torch.zeros((200, 320, 4), dtype=torch.uint8, device="cuda")
    warnings, frames = 0, []
    for rounds in range(3):
        for c, o in zip(cams, orders):
            mesh.set_camera(c)
            mesh.update_render_indexes(o, scene.count)
            mesh.render(out_device_ptr=out.data_ptr(), to_host=False, want_stats=False)      # asynchronous
            warnings += mesh.last_status == L.GS_WARN_FRAME_TRUNCATED
            ctx.synchronize()
            frames.append(out.cpu().numpy().copy())
    assert warnings >= 1, "the overflow was never reported"
    assert not np.array_equal(frames[0], want[0]), "the first draw cannot have fitted 1024 entries"
    for k in range(len(cams)):                               # the last orbit is complete: buffers have grown to the largest need
        np.testing.assert_array_equal(frames[2 * len(cams) + k], want[k])
    mesh.dispose()


def test_stage_times_only_when_asked_for(ctx):
    """Stage events are recorded for calls that return statistics, or for every call after
    gs_context_set_stage_timing; an untimed call reports 0 ms and the same counts / pixels."""
    scene = helpers.small_scene(4000, 1, seed=55)
    cam = camera.demo_camera("garden", 256, 144)
    mesh = build_mesh(ctx, scene)
    mesh.set_camera(cam)
    worker = create_sort_worker(ctx, scene.count)
    worker.post_message({"centers": util.integer_centers(scene.centers), "range": {"from": 0, "to": scene.count - 1, "count": scene.count}})
    worker.sort_on_device(cam.sort_mvp(), scene.count)
    mesh.use_sorter_result(worker, scene.count)
    img_timed, st = mesh.render()
    assert st.device_ms > 0 and st.blend_ms > 0 and st.project_ms > 0
    mesh.render(want_stats=False, to_host=False)
    ctx.synchronize()
    untimed = mesh.last_stats()
    assert untimed.device_ms == 0 and untimed.bin_ms == 0 and untimed.blend_ms == 0
    assert untimed.tile_entries == st.tile_entries and untimed.visible_splats == st.visible_splats
    ss, _ = worker.last_stats()
    assert ss.device_ms == 0 and ss.result_count == scene.count
    ctx.set_stage_timing(True)
    try:
        worker.sort_on_device(cam.sort_mvp(), scene.count)
        mesh.render(want_stats=False, to_host=False)
        ctx.synchronize()
        assert mesh.last_stats().device_ms > 0 and worker.last_stats()[0].device_ms > 0
    finally:
        ctx.set_stage_timing(False)
    img, _ = mesh.render(want_stats=False)
    assert np.array_equal(img, img_timed)
    worker.terminate()
    mesh.dispose()


def test_range_uploads_like_update_data_textures_from_base_data(ctx):
    """SplatMesh.updateDataTexturesFromBaseData(fromSplat, toSplat) takes any range any number of times
    (/root/reference/src/splatmesh/SplatMesh.js:900-1062): progressive loads append, edits replace a sub-range.  Whatever the
    sequence of gs_mesh_upload calls, the frame must equal the one of a mesh built once from the final arrays (splats seen
    before keep their storage slots, new ones get slots along the Morton curve of their own run)."""
    n = 6000
    scene = helpers.small_scene(n, 2, seed=88)
    cam = camera.demo_camera("garden", 320, 200)
    order = sorted_order(scene, cam)

    def frame_of(mesh):
        mesh.set_camera(cam)
        mesh.update_render_indexes(order, n)
        return mesh.render()[0]

    want = frame_of(build_mesh(ctx, scene))
    sl = lambda a, lo, hi: a[lo:hi]                                   # noqa: E731

    def upload(mesh, lo, hi, src=scene):
        mesh.build(sl(src.centers, lo, hi), sl(src.cov, lo, hi), sl(src.rgba, lo, hi), sl(src.sh, lo, hi), start=lo)

    # progressive: three appended runs, the last one overlapping the second
    prog = SplatMesh(ctx, n, 2)
    upload(prog, 0, 2500)
    upload(prog, 2500, 4000)
    upload(prog, 3500, n)                                             # [3500,4000) again (same data) + fresh [4000,n)
    np.testing.assert_array_equal(frame_of(prog), want)
    # an edit: scribble over a sub-range, then restore it - and a range that straddles two earlier runs
    rng = np.random.default_rng(5)
    bad = helpers.small_scene(n, 2, seed=89)
    upload(prog, 1000, 3000, src=bad)
    assert np.abs(frame_of(prog).astype(int) - want.astype(int)).max() > 20
    upload(prog, 1000, 3000)
    np.testing.assert_array_equal(frame_of(prog), want)
    # the edited mesh equals a mesh built once from the edited arrays
    lo, hi = 2200, 5100
    edited = helpers.small_scene(n, 2, seed=88)
    for name in ("centers", "cov", "rgba", "sh"):
        getattr(edited, name)[lo:hi] = getattr(bad, name)[lo:hi]
    upload(prog, lo, hi, src=bad)
    order_e = sorted_order(edited, cam)
    once = build_mesh(ctx, edited)
    for m in (prog, once):
        m.set_camera(cam)
        m.update_render_indexes(order_e, n)
    np.testing.assert_array_equal(prog.render()[0], once.render()[0])
    del rng
    prog.dispose()
    once.dispose()


def test_a_range_upload_that_leaves_a_hole_draws_like_zero_filled_textures(ctx):
    """gs_mesh_upload accepts any range, so the FIRST upload may be [a, b) with a > 0: splats [0, a) then are what the
    reference's zero-filled data textures hold (centre 0, covariance 0, alpha 0: SplatMesh.js:686-697) - defined planes, a
    bijective storage permutation, nothing drawn.  The frame must equal the one of a mesh whose hole was uploaded as
    zeros, with the full index list naming the never-uploaded splats as well."""
    n, a = 5000, 1800
    scene = helpers.small_scene(n, 2, seed=93)
    cam = camera.demo_camera("garden", 320, 200)
    zeroed = helpers.small_scene(n, 2, seed=93)
    for name in ("centers", "cov", "rgba", "sh"):
        getattr(zeroed, name)[:a] = 0
    order = sorted_order(zeroed, cam)
    want_mesh = build_mesh(ctx, zeroed)
    want_mesh.set_camera(cam)
    want_mesh.update_render_indexes(order, n)
    want = want_mesh.render()[0]
    holed = SplatMesh(ctx, n, 2)
    holed.build(scene.centers[a:], scene.cov[a:], scene.rgba[a:], scene.sh[a:], start=a)     # [0, a) never uploaded
    holed.set_camera(cam)
    holed.update_render_indexes(order, n)
    got, st = holed.render()
    np.testing.assert_array_equal(got, want)
    assert got[..., 3].any()
    # the hole filled later: the ordinary frame
    holed.build(scene.centers[:a], scene.cov[:a], scene.rgba[:a], scene.sh[:a], start=0)
    full_order = sorted_order(scene, cam)
    holed.update_render_indexes(full_order, n)
    once = build_mesh(ctx, scene)
    once.set_camera(cam)
    once.update_render_indexes(full_order, n)
    np.testing.assert_array_equal(holed.render()[0], once.render()[0])
    for m in (want_mesh, holed, once):
        m.dispose()


def test_out_of_range_indexes_are_harmless(ctx):
    """A stale or wrong index list must neither fault the GPU nor read foreign memory (WebGL's out-of-range texelFetch is
    harmless too): the mesh draws nothing for entries >= the uploaded splat count, the sorter clamps them."""
    scene = helpers.small_scene(3000, 1, seed=91)
    cam = camera.demo_camera("garden", 256, 144)
    order = sorted_order(scene, cam)
    mesh = build_mesh(ctx, scene)
    mesh.set_camera(cam)
    mesh.update_render_indexes(order, scene.count)
    mesh.render()
    mesh.update_render_indexes(order, scene.count + 1)                  # more than uploaded: refused
    with pytest.raises(Exception):
        mesh.render()
    mesh.update_render_indexes(order, scene.count)
    mesh2 = SplatMesh(ctx, scene.count, 1).build(scene.centers, scene.cov, scene.rgba, scene.sh)
    mesh2.set_camera(cam)
    stale = order.copy()
    stale[::7] = 0xFFFFFFF0                                             # a seventh of the list points nowhere
    mesh2.update_render_indexes(stale, scene.count)
    got, _ = mesh2.render()
    keep = np.ones(scene.count, bool)
    keep[::7] = False
    mesh.update_render_indexes(order[keep], int(keep.sum()))
    ref, _ = mesh.render()
    np.testing.assert_array_equal(got, ref)                             # = the frame of the valid entries alone
    # sorter: list entries beyond the uploaded centres are clamped to the last splat, never dereferenced
    worker = create_sort_worker(ctx, scene.count)
    worker.post_message({"centers": util.integer_centers(scene.centers), "range": {"from": 0, "to": scene.count - 1, "count": scene.count}})
    idx = np.arange(scene.count, dtype=np.uint32)
    idx[5] = 0xFFFFFFFF
    r = worker.post_message({"sort": {"modelViewProj": cam.sort_mvp(), "splatRenderCount": scene.count, "splatSortCount": scene.count,
                                      "indexesToSort": idx}})
    assert r["sortedIndexes"].shape[0] == scene.count
    with pytest.raises(Exception):                                      # a list shorter than the counts: refused on the host
        worker.post_message({"sort": {"modelViewProj": cam.sort_mvp(), "splatRenderCount": scene.count, "splatSortCount": scene.count,
                                      "indexesToSort": idx[:100]}})
    worker.terminate()
    for m in (mesh, mesh2):
        m.dispose()


def test_block_boxes_follow_a_re_upload_of_moved_splats(ctx, monkeypatch):
    """Splats uploaded before keep their storage slots (scattered over the Morton run of the first upload), so replacing the
    data of a sub-range must redo the block boxes of every block of that run, not of [from, from + count) - else a block keeps
    the box of what it held before and k_project's whole-block test drops splats that moved into view.
    The scene sits behind the camera, a sub-range is moved in front of it: the frame must equal the one drawn with the block test
    off, and it must not be empty."""
    n = 6000
    scene = scenes.scene_like(n, 1, 777)
    W, H = 320, 192
    cam = camera.PerspectiveCamera(W, H, (0.0, 0.0, 0.0), (0.0, 0.0, -1.0), (0.0, 1.0, 0.0))
    moved = scenes.scene_like(n, 1, 777)
    behind = scene.centers.copy()
    behind[:, 2] = np.abs(behind[:, 2]) + 12.0                          # every splat behind the camera (it looks down -z)
    scene.centers[:] = behind
    moved.centers[:] = behind
    lo, hi = 1000, 2500
    moved.centers[lo:hi, 2] = -(np.abs(moved.centers[lo:hi, 2]) - 12.0) - 3.0      # ... this range in front of it
    frames = []
    for off in (False, True):
        if off:
            monkeypatch.setenv("GSPLAT_NO_BLOCK_CULL", "1")
        m = build_mesh(ctx, scene)
        if off:
            monkeypatch.delenv("GSPLAT_NO_BLOCK_CULL")
        m.set_camera(cam)
        m.update_render_indexes(sorted_order(scene, cam), n)
        assert int(m.render()[1].visible_splats) == 0
        m.build(moved.centers[lo:hi], moved.cov[lo:hi], moved.rgba[lo:hi], moved.sh[lo:hi], start=lo)
        m.update_render_indexes(sorted_order(moved, cam), n)
        f, st = m.render()
        frames.append((f, int(st.visible_splats)))
        m.dispose()
    assert frames[1][1] > 100
    assert frames[0][1] == frames[1][1]
    np.testing.assert_array_equal(frames[0][0], frames[1][0])


def test_block_level_cull_changes_nothing(ctx, monkeypatch):
    """k_project drops whole 256-splat storage blocks whose box fails the frustum (or cannot reach the rank's strip) before it
    reads their centres.  Same frames, same visible counts and the same per-splat visibility masks as with the block test off
    (GSPLAT_NO_BLOCK_CULL=1), for the demo pose, orbit poses that look away from most of the cloud, a camera inside it, and
    strips of a multi-GPU draw; and the test must really fire (most of the scene is outside the frustum)."""
    n = 120000
    scene = scenes.scene_like(n, 1, 4242)
    W, H = 640, 360
    cams = [camera.demo_camera("garden", W, H)] + camera.orbit_cameras("garden", W, H, 6)[1:4] + \
           [camera.PerspectiveCamera(W, H, (0.1, 0.2, -0.3), (3.0, 1.0, 2.0), (0.0, -1.0, 0.0))]
    on = build_mesh(ctx, scene)
    monkeypatch.setenv("GSPLAT_NO_BLOCK_CULL", "1")
    off = build_mesh(ctx, scene)
    monkeypatch.delenv("GSPLAT_NO_BLOCK_CULL")
    rows = (H + 15) // 16
    for cam in cams:
        order = sorted_order(scene, cam)
        frames = []
        for m in (on, off):
            m.set_camera(cam)
            m.update_render_indexes(order, n)
            f, st = m.render()
            vis = m.debug_records(n)[2]
            strips = [m.render(tile_rows=r)[0] for r in ((0, 5), (5, 6), (6, rows))]
            frames.append((f, int(st.visible_splats), vis, strips))
        np.testing.assert_array_equal(frames[0][0], frames[1][0])
        assert frames[0][1] == frames[1][1] and 0 < frames[0][1] < n // 2
        np.testing.assert_array_equal(frames[0][2], frames[1][2])
        for a, b in zip(frames[0][3], frames[1][3]):
            np.testing.assert_array_equal(a, b)
        np.testing.assert_array_equal(np.concatenate(frames[0][3], axis=0), frames[0][0])
    on.dispose()
    off.dispose()


@pytest.mark.gpu
def test_fork_join_context_draws_the_same_frames_with_serial_frames():
    """GS_CTX_FORK_JOIN: the sort and the vertex stage of a frame run side by side on their own streams, frames stay serial.  The
    frames of a moving camera equal the single-stream context's bit for bit, sort results included; and a sort really waits for
    the previous draw (the draw after a re-sort with another camera never sees a half-written order: every frame is checked)."""
    from gaussiansplats3d_amd import Context, create_sort_worker
    scene = helpers.small_scene(40000, 2, seed=17)
    n = scene.count
    ci = util.integer_centers(scene.centers)
    cams = camera.orbit_cameras("garden", 320, 180, 8)
    frames = {}
    for mode in ("single", "fork"):
        c = Context(0, single_stream=(mode == "single"), fork_join=(mode == "fork"))
        w = create_sort_worker(c, n)
        w.post_message({"centers": ci, "range": {"from": 0, "to": n - 1, "count": n}})
        m = build_mesh(c, scene)
        m.use_sorter_result(w, n)
        out = []
        for cam in cams:
            m.set_camera(cam)
            w.sort_on_device(cam.sort_mvp(), n)
            out.append(m.render()[0].copy())
        frames[mode] = out
        w.terminate(); m.dispose(); c.close()
    for a, b in zip(frames["single"], frames["fork"]):
        assert a[..., 3].any()
        np.testing.assert_array_equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("strips", [False, True])
def test_frames_in_flight_on_two_record_sets_equal_the_serial_frames(strips):
    """A context with streams of its own keeps TWO sets of vertex-stage outputs (records, rects, masks): the vertex stage of frame
    k + 1 writes one while frame k is binned and blended from the other.  Eight frames of a moving camera are enqueued without a
    single synchronisation, each into a device buffer of its own, and compared with the single-stream context's frames; with
    `strips` every frame is a rank's strip (vertex stage first, visibility-culled sort: the per-index mask belongs to a set too)."""
    import torch
    from gaussiansplats3d_amd import Context, create_sort_worker
    scene = helpers.small_scene(200000, 2, seed=23)
    n = scene.count
    ci = util.integer_centers(scene.centers)
    W, H = 640, 368
    cams = camera.orbit_cameras("garden", W, H, 8)
    rows = (6, 15) if strips else None
    frames = {}
    for mode in ("single", "streams"):
        c = Context(0, single_stream=(mode == "single"))
        w = create_sort_worker(c, n)
        w.post_message({"centers": ci, "range": {"from": 0, "to": n - 1, "count": n}})
        m = build_mesh(c, scene)
        m.use_sorter_result(w, n)
        w.set_visibility_cull(strips)
        m.set_camera(cams[0])
        h = m.strip_shape(rows)[0]
        bufs = [torch.zeros((h, W, 4), dtype=torch.uint8, device="cuda:0") for _ in cams]
        torch.cuda.synchronize()
        for rep in range(2):                                   # the second round reuses both sets with frames still in flight
            for cam, buf in zip(cams, bufs):
                m.set_camera(cam)
                if strips:
                    m.project(rows)
                w.sort_on_device(cam.sort_mvp(), n)
                m.render(tile_rows=rows, out_device_ptr=buf.data_ptr(), want_stats=False, to_host=False)
        c.synchronize()
        frames[mode] = [b.cpu().numpy() for b in bufs]
        w.terminate(); m.dispose(); c.close()
    for a, b in zip(frames["single"], frames["streams"]):
        assert a[..., 3].any()
        np.testing.assert_array_equal(a, b)


def test_live_block_list_and_per_workgroup_block_tests_draw_the_same_frame(ctx, monkeypatch):
    """Round 5: k_block_test decides per storage block before k_project runs and k_project walks its list of live blocks;
    $GSPLAT_NO_BLOCK_LIST restores the round-4 shape (every workgroup tests its own block).  Records, masks and frames must not
    depend on it: full frame, strips, a pose that sees nothing, and a second draw (the two counters take turns)."""
    scene = helpers.small_scene(40000, 2, seed=23)
    cam = camera.demo_camera("garden", 640, 360)
    pos, look = np.array(camera.DEMO_POSES["garden"][1]), np.array(camera.DEMO_POSES["garden"][2])
    fwd = (look - pos) / np.linalg.norm(look - pos)
    away = camera.PerspectiveCamera(640, 360, tuple(pos - 30.0 * fwd), tuple(pos - 60.0 * fwd), camera.DEMO_POSES["garden"][0])
    order = sorted_order(scene, cam)
    outs = {}
    for mode in ("list", "per_workgroup"):
        # (the separate test is skipped for scenes that were mostly in view at their last measured draw: forced here)
        monkeypatch.setenv("GSPLAT_NO_BLOCK_LIST" if mode == "per_workgroup" else "GSPLAT_BLOCK_TEST_ALWAYS", "1")
        mesh = build_mesh(ctx, scene)
        monkeypatch.delenv("GSPLAT_NO_BLOCK_LIST", raising=False)
        monkeypatch.delenv("GSPLAT_BLOCK_TEST_ALWAYS", raising=False)
        mesh.update_render_indexes(order, scene.count)
        res = []
        for c, rows in ((cam, None), (cam, (3, 9)), (away, None), (cam, None), (cam, (0, 2))):
            mesh.set_camera(c)
            frame, st = mesh.render(tile_rows=rows)
            recs, rects, vis = mesh.debug_records()
            res.append((frame, int(st.visible_splats), int(st.tile_entries), vis, recs[vis], rects[vis]))
        outs[mode] = res
        mesh.dispose()
    assert outs["list"][0][1] > 1000 and outs["list"][2][1] == 0 and not outs["list"][2][0].any()
    for a, b in zip(outs["list"], outs["per_workgroup"]):
        assert a[1] == b[1] and a[2] == b[2]
        for x, y in zip((a[0], a[3], a[4], a[5]), (b[0], b[3], b[4], b[5])):
            np.testing.assert_array_equal(x, y)


def test_asynchronous_draws_learn_how_much_of_the_scene_is_in_view(ctx):
    """Where the vertex stage runs its block test depends on the share of the splats in view at the last full-frame draw; draws that
    never ask for statistics learn it from a mapped word k_bin_emit leaves (mesh.hip, mesh_read_view_share).  A scene seen whole ends
    up without a block test, the same scene seen from inside with the separate kernel - and the frames are the ones a
    statistics-reading mesh draws."""
    import torch
    w, h = 320, 200
    scene = helpers.small_scene(30000, 0, seed=91, scale=0.03)
    order = np.arange(scene.count, dtype=np.uint32)
    up, pos, look = (np.asarray(v, dtype=np.float64) for v in camera.DEMO_POSES["garden"])
    fwd = (look - pos) / np.linalg.norm(look - pos)
    out = torch.empty((h, w, 4), dtype=torch.uint8, device="cuda")
    for eye, expect in ((pos - fwd * 6.0, 2), (pos + fwd * 6.0, 1)):        # the whole slab in view (99.8 %) / from inside it (14 %)
        cam = camera.PerspectiveCamera(w, h, tuple(eye), tuple(eye + fwd), tuple(up))
        ref_mesh = SplatMesh(ctx, scene.count, 0).build(scene.centers, scene.cov, scene.rgba, None)
        ref_mesh.set_camera(cam)
        ref_mesh.update_render_indexes(order, scene.count)
        ref, st = ref_mesh.render()
        share = st.visible_splats / scene.count
        assert (share > 0.95) if expect == 2 else (share < 0.3), share
        mesh = SplatMesh(ctx, scene.count, 0).build(scene.centers, scene.cov, scene.rgba, None)
        mesh.set_camera(cam)
        mesh.update_render_indexes(order, scene.count)
        for _ in range(4):
            mesh.render(out_device_ptr=out.data_ptr(), want_stats=False, to_host=False)     # asynchronous: no statistics, device output
            torch.cuda.synchronize()
        info = mesh.view_share()
        assert info["projected"] == scene.count and info["visible"] == st.visible_splats, (info, st.visible_splats)
        assert info["block_test"] == expect, info
        assert np.array_equal(out.cpu().numpy(), ref)
        ref_mesh.dispose(); mesh.dispose()


def test_fused_binner_draws_the_same_frame_and_statistics(ctx, monkeypatch):
    """$GSPLAT_BIN_FUSED=1 (k_bin_fused: count + emit in one launch behind a scan across the running grid; built in round 6,
    slower than the two kernels, kept as an A/B switch) must produce the same entries: same frame, same counters - also when the
    entry buffer overflows, for a strip, and for a list whose length lives on the device (a visibility-culled sort)."""
    scene = helpers.small_scene(150000, 1, seed=77)
    cam = camera.demo_camera("garden", 640, 360)
    n = scene.count
    w = create_sort_worker(ctx, n)
    w.post_message({"centers": util.integer_centers(scene.centers), "range": {"from": 0, "to": n - 1, "count": n}})
    mesh = build_mesh(ctx, scene)
    mesh.set_camera(cam)
    mvp = cam.sort_mvp()
    w.sort_on_device(mvp, n)
    mesh.use_sorter_result(w, n)

    def frames():
        out = []
        w.set_visibility_cull(False)
        w.sort_on_device(mvp, n)
        img, st = mesh.render()
        out.append((img, int(st.visible_splats), int(st.tile_entries), int(st.tiles16)))
        img, st = mesh.render(tile_rows=(5, 13))
        out.append((img, int(st.visible_splats), int(st.tile_entries), int(st.tiles16)))
        w.set_visibility_cull(True)
        mesh.project()
        w.sort_on_device(mvp, n)
        img, st = mesh.render()
        out.append((img, int(st.visible_splats), int(st.tile_entries), int(st.tiles16)))
        w.set_visibility_cull(False)
        return out

    plain = frames()
    monkeypatch.setenv("GSPLAT_BIN_FUSED", "1")
    fused = frames()
    mesh.debug_set_entry_capacity(4096)                 # an overflowing draw is redone with a grown buffer
    w.sort_on_device(mvp, n)
    img, st = mesh.render()
    monkeypatch.delenv("GSPLAT_BIN_FUSED")
    for (a, *sa), (b, *sb) in zip(plain, fused):
        np.testing.assert_array_equal(a, b)
        assert sa == sb
    np.testing.assert_array_equal(img, plain[0][0])
    assert plain[0][1] > 1000 and plain[0][2] > plain[0][1]
    w.terminate()
    mesh.dispose()
