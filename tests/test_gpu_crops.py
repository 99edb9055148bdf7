"""-m gpu: raster parity AT THE BASELINE.json CONFIGURATIONS.  The fp32 oracle cannot rasterise 5.8 M splats over a whole
1080p ... 8K frame in test time, but it can over crops: every splat of the (pinned) sorted list is projected once by the
oracle itself and composited into a handful of 64x64-px windows (oracle.render_windows), which are compared with the same
pixels of the engine's full-size frame under the stated framebuffer tolerance (helpers.compare_frames).

Windows per config: centre, the four corners, the densest list bin of the draw, one straddling a list-bin boundary and
one straddling the cut between two multi-GPU strips (the frame is also drawn as strips and must equal the full draw).
Beyond the BASELINE.json configurations at their demo pose: the translucent C3T scene (the long-list path of the blend: 7x
the (splat, tile) pairs of the opaque stand-in), C3 at poses 15 / 30 / 45 of the 60-pose orbit, and the capture-like C3S
scene (surfels, bimodal opacity).
The same crops also measure the gap between the fp32 composite (the parity target) and the reference's real render
target, which rounds to RGBA8 after every splat (src/splatmesh/SplatMaterial3D.js:65-75): printed, written to
gpurun_out/crops_<cfg>.json when that directory exists, and reported in DESIGN.md — not gated."""
import json
import os

import numpy as np
import pytest

import helpers
import oracle
from gaussiansplats3d_amd import Context, SplatMesh, camera, create_sort_worker, scenes, util

pytestmark = pytest.mark.gpu
WIN = 64


@pytest.fixture(scope="module")
def ctx():
    c = Context(0)
    yield c
    c.close()


def mesh_frame_reaches_corners(counts):
    return bool(counts[0, 0] and counts[0, -1] and counts[-1, 0] and counts[-1, -1])


def _windows(W, H, mesh, cut_row, n_max):
    """[(label, x0, y0, w, h)] in GL window coordinates."""
    counts = mesh.bin_entry_counts()
    lb = int(mesh.last_stats().list_bin_px)
    by, bx = np.unravel_index(int(np.argmax(counts)), counts.shape)
    clampx = lambda x: int(min(max(x, 0), W - WIN))
    clampy = lambda y: int(min(max(y, 0), H - WIN))
    # a list-bin corner near the middle of the screen, and the strip cut (a 16-px tile row that is not a list-bin row)
    lx, ly = (W // 2 // lb) * lb, (H // 2 // lb) * lb
    wins = [("centre", clampx(W // 2 - WIN // 2), clampy(H // 2 - WIN // 2)),
            ("densest-list-bin", clampx(bx * lb + lb // 4), clampy(by * lb + lb // 4)),
            ("list-bin-boundary", clampx(lx - WIN // 2), clampy(ly - WIN // 2)),
            ("strip-cut", clampx(W // 3), clampy(cut_row * 16 - WIN // 2)),
            ("corner-bl", 0, 0), ("corner-tr", W - WIN, H - WIN), ("corner-br", W - WIN, 0), ("corner-tl", 0, H - WIN)]
    if not mesh_frame_reaches_corners(counts):          # C4's cube does not fill the frame: quarter points instead
        wins[4:] = [("quarter-bl", W // 4, H // 4), ("quarter-tr", 3 * W // 4 - WIN, 3 * H // 4 - WIN),
                    ("quarter-br", 3 * W // 4 - WIN, H // 4), ("quarter-tl", W // 4, 3 * H // 4 - WIN)]
    return [(name, x, y, WIN, WIN) for name, x, y in wins[:n_max]]


_scene_cache = {}


ROP8_MAX, ROP8_EQUAL = 1.0, 0.995     # measured r04: 99.85 % .. 100 % of the channel values equal, never more than 1 apart (profiles/r04*_crops_*.json)


def _scene(cfg_name):
    """The last generated scene is kept (C5 is C3's scene, the orbit poses share it: ~15 s of numpy each)."""
    key = "C3" if cfg_name == "C5" else cfg_name
    if key not in _scene_cache:
        _scene_cache.clear()
        _scene_cache[key] = scenes.make_config_scene(key)
    return _scene_cache[key]


def _crop_parity(ctx, cfg_name, n_windows, cam=None, tag=None):
    cfg = scenes.CONFIGS[cfg_name]
    W, H = cfg["width"], cfg["height"]
    scene = _scene(cfg_name)
    cam = cam or camera.demo_camera(cfg["pose"], W, H)
    tag = tag or cfg_name
    n = scene.count
    ci = util.integer_centers(scene.centers)
    mvp = cam.sort_mvp()

    worker = create_sort_worker(ctx, n)
    worker.post_message({"centers": ci, "range": {"from": 0, "to": n - 1, "count": n}})
    mesh = SplatMesh(ctx, n, scene.sh_degree, scene.cov_half)
    mesh.build(scene.centers, scene.cov, scene.rgba, scene.sh if scene.sh_degree else None)
    mesh.set_camera(cam)
    worker.sort_on_device(mvp, n)
    mesh.use_sorter_result(worker, n)
    mesh.render()                                   # may grow the entry buffer / settle the list-bin size
    worker.sort_on_device(mvp, n)
    frame, stats = mesh.render()
    rows = (H + 15) // 16
    cut = rows // 2 + 1                              # 16-px tile row of the strip cut: not a multiple of a 128-px list row
    strips = [mesh.render(tile_rows=r)[0] for r in ((0, cut), (cut, rows))]
    np.testing.assert_array_equal(np.concatenate(strips, axis=0), frame)
    mesh.render()                                   # statistics / list ranges of a full-frame draw again
    wins = _windows(W, H, mesh, cut, n_windows)

    # the reference's order from the pinned sort oracle; the engine's own device-resident order must be the same list
    order = oracle.sort_indexes(np.arange(n, dtype=np.uint32), ci, mvp)
    np.testing.assert_array_equal(worker.debug_read(2, n), order)

    c, cov, rgba, _ = helpers.oracle_inputs(scene)
    ocam = oracle.make_camera(cam.model_view(), cam.projection, cam.position, W, H, scene.sh_degree, scene.sh_degree)
    sh = scene.sh if scene.sh_degree else None
    boxes = [w[1:] for w in wins]
    crops, frags = oracle.render_windows(ocam, c, cov, rgba, sh, order, boxes)
    bounds = []                                      # per window: how far per-splat RGBA8 rounding can drift from the exact composite
    crops8, _ = oracle.render_windows(ocam, c, cov, rgba, sh, order, boxes, rop8=True, error_bounds=bounds)
    assert frags > 1000
    report = {"config": tag, "width": W, "height": H, "splats": n, "visible": int(stats.visible_splats),
              "list_bin_px": int(stats.list_bin_px), "oracle_fragments": frags, "windows": []}
    amb_pixels = 0
    for (name, x0, y0, w, h), (fb, amb), (fb8, _), ebound in zip(wins, crops, crops8, bounds):
        got = frame[y0:y0 + h, x0:x0 + w]
        msg = helpers.compare_frames(got, fb, amb, f"{tag} {name} @({x0},{y0})", strict=True)
        assert got[..., 3].any(), f"{tag} {name}: the window is empty, it checks nothing"
        ref = np.clip(fb, 0, 1) * 255.0
        ref8 = np.clip(fb8, 0, 1) * 255.0
        # a14 on the reference's own terms: the engine's verification path composites this window back to front with the RGBA8
        # rounding after every splat (gs_mesh_debug_rop8) - compared with the ROP-emulating oracle, which does the same from its
        # own vertex stage.  The two differ only where a rounding step sits within the vertex stages' tolerance of a half:
        # <= 0.15 % of the channel values, by one (r04).
        mine8 = mesh.rop8_window(x0, y0, w, h).astype(np.float32)
        d8 = np.abs(mine8 - np.round(ref8))
        rop8_equal, rop8_max = float((d8 == 0).mean()), float(d8.max())
        assert rop8_max <= ROP8_MAX and rop8_equal >= ROP8_EQUAL, \
            f"{tag} {name}: rop8 verification path vs the ROP-emulating oracle: {rop8_equal:.4f} equal, max {rop8_max:.0f} (limits {ROP8_EQUAL} / {ROP8_MAX})"
        gap = np.abs(ref - ref8)                     # fp32 composite vs per-splat RGBA8 rounding, both by the oracle
        ours = np.abs(got.astype(np.float32) - np.round(ref8))  # the engine's frame vs the ROP-emulating oracle (whole 1/255 steps)
        amb_pixels += int(amb.sum())
        report["windows"].append({"name": name, "x0": x0, "y0": y0, "parity": msg, "ambiguous_pixels": int(amb.sum()),
                                  "rop8_gap_max": round(float(gap.max()), 3), "rop8_gap_mean": round(float(gap.mean()), 4),
                                  "rop8_path_equal_frac": round(rop8_equal, 5), "rop8_path_max": rop8_max,
                                  "engine_vs_rop8_max": round(float(ours.max()), 3),
                                  "engine_vs_rop8_mean": round(float(ours.mean()), 4),
                                  "rop8_bound_max": round(float(ebound.max()) + ENGINE_SLACK, 3),
                                  "rop8_bound_headroom_min": round(float((ebound[..., None] + ENGINE_SLACK - ours).min()), 3)})
        print(msg, "| rop8 gap max %.2f mean %.3f (1/255 units) | rop8 path: %.4f equal, max %.0f" % (gap.max(), gap.mean(), rop8_equal, rop8_max))
        # The reference's real target is RGBA8 and rounds after every splat (SplatMaterial3D.js:65-75); the engine composites in
        # fp32 and rounds once.  The distance between the two is not an engine error, but it is GATED so that it cannot grow
        # unnoticed: worst 3-4 and mean 0.3-0.9 of 1/255 on every window of rounds 2-3 (profiles/r03z_crops_*.json).
        # Round 5: the limit is DERIVED per pixel instead of fitted (round 4's flat "max 4" sat exactly on C3T's and C3S's worst
        # pixel): every blend into RGBA8 rounds by <= 0.5 and later splats scale the accumulated error by (1 - alpha), so the
        # ROP-emulating oracle sits within e = sum_k 0.5 * T_k of the exact composite (raster_oracle.c keeps the recursion
        # e <- (1 - alpha) e + 0.5 per pixel), and the engine - the fp32 composite rounded once - within 0.5 + its own measured
        # distance to the fp32 oracle (<= 0.52 on every crop, gated above by strict=True).
        excess = ours - (ebound[..., None] + ENGINE_SLACK)
        assert excess.max() <= 0.0 and ours.mean() <= 1.0, \
            f"{tag} {name}: engine vs the RGBA8-ROP-emulating oracle: {excess.max():+.2f} beyond the per-pixel bound (max diff {ours.max():.2f}, mean {ours.mean():.3f}; mean limit 1.0 of 1/255)"
        assert ours.max() <= ROP8_RECORDED_MAX[tag] + 1.0, \
            f"{tag} {name}: engine vs the RGBA8-ROP-emulating oracle: worst pixel {ours.max():.1f} / 255, recorded {ROP8_RECORDED_MAX[tag]:.0f} (tripwire: recorded + 1)"
    # GS_DRAW_ROP8 (round 6): the same frame drawn in the reference's RGBA8-per-splat mode, on the windows the
    # ROP-emulating oracle rasterised: >= 99.5 % of the channel values equal, never more than 1 apart - the verification kernel's gate
    if tag in ROP8_MODE_CONFIGS:
        report["rop8_mode"] = {}
        for label, full in (("bounded", False), ("full", True)):
            mesh.set_draw_mode(rop8=True, full=full)
            worker.sort_on_device(mvp, n)
            frame8, st8 = mesh.render()
            mesh.set_draw_mode(rop8=False)
            eq, alpha_eq, worst = [], [], 0.0
            for (name, x0, y0, w, h), (fb8, _) in zip(wins, crops8):
                d = np.abs(frame8[y0:y0 + h, x0:x0 + w].astype(np.float32) - np.round(np.clip(fb8, 0, 1) * 255.0))
                if not full:
                    # the bounded walk: colour to the full walk's gate; the ALPHA channel may sit up to 2 steps off where it stalls
                    # below 255 (alpha = q8(a + (1 - a) alpha) stops moving once a (255 - alpha) < 0.5, at a value that depends on
                    # the whole list - the far splats the bounded walk leaves out included; tile_blend.hip)
                    assert d[..., 3].max() <= 2.0, f"{tag} {name}: GS_DRAW_ROP8 (bounded) alpha channel {d[..., 3].max():.0f} steps from the oracle"
                    alpha_eq.append(float((d[..., 3] == 0).mean()))
                    d = d[..., :3]
                eq.append(float((d == 0).mean()))
                worst = max(worst, float(d.max()))
                assert d.max() <= ROP8_MAX and (d == 0).mean() >= ROP8_EQUAL, \
                    f"{tag} {name}: GS_DRAW_ROP8 ({label}) vs the ROP-emulating oracle: {(d == 0).mean():.4f} equal, max {d.max():.0f}"
            report["rop8_mode"][label] = {"equal_frac_min": round(min(eq), 5), "equal_frac_mean": round(float(np.mean(eq)), 5), "max_diff": worst,
                                          "channels": "rgba" if full else "rgb (alpha: <= 2 steps)",
                                          "alpha_equal_frac_min": round(min(alpha_eq), 5) if alpha_eq else None,
                                          "splats_walked": int(st8.splats_walked), "blend_ms": round(float(st8.blend_ms), 4)}
            print(f"{tag} GS_DRAW_ROP8 {label}: equal >= {min(eq):.4f}, max {worst:.0f}, walked {int(st8.splats_walked)}, blend {float(st8.blend_ms):.3f} ms")
    report["ambiguous_pixels_total"] = amb_pixels
    report["entries_scanned"] = int(stats.entries_scanned)
    report["splats_walked"] = int(stats.splats_walked)
    report["tile_entries"] = int(stats.tile_entries)
    _write_report(f"crops_{tag}.json", report)
    worker.terminate()
    mesh.dispose()
    return report


ROP8_MODE_CONFIGS = ("C3", "C3T", "C2")   # configurations whose crops are also drawn in GS_DRAW_ROP8
ENGINE_SLACK = 1.05        # the engine's own final rounding (0.5) + its strict-gated distance to the fp32 oracle (<= 0.5) + fp32 noise
# Tripwire beside the derived bound: the bound reaches 18.5 / 255 on translucent content while the engine
# sits at <= 4, so a regression that doubled the distance to the ROP result would pass it.  The worst pixel of every
# configuration as recorded in round 5 (profiles/r05z_crops_*.json: `engine_vs_rop8_max`), plus one step.
ROP8_RECORDED_MAX = {"C2": 3.0, "C3": 3.0, "C3S": 4.0, "C3T": 4.0, "C4": 3.0, "C5": 3.0, "C3orbit15": 3.0, "C3orbit30": 3.0, "C3orbit45": 2.0}


def _write_report(name, report):
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, name), "w") as f:
            json.dump(report, f, indent=1)


def _whole_frame(ctx, cfg_name, W, H):
    """EVERY pixel of a frame of the configuration's whole scene (all of its splats, its SH degree and covariance format, the demo
    pose) at a resolution the fp32 oracle can rasterise in full, against the oracle - the device's own sorted order checked against
    the pinned sort oracle first."""
    scene = _scene(cfg_name)
    cam = camera.demo_camera(scenes.CONFIGS[cfg_name]["pose"], W, H)
    n = scene.count
    ci = util.integer_centers(scene.centers)
    worker = create_sort_worker(ctx, n)
    worker.post_message({"centers": ci, "range": {"from": 0, "to": n - 1, "count": n}})
    mesh = SplatMesh(ctx, n, scene.sh_degree, scene.cov_half)
    mesh.build(scene.centers, scene.cov, scene.rgba, scene.sh if scene.sh_degree else None)
    mesh.set_camera(cam)
    mvp = cam.sort_mvp()
    worker.sort_on_device(mvp, n)
    mesh.use_sorter_result(worker, n)
    mesh.render()                                   # may grow the entry buffer / settle the list-bin size
    worker.sort_on_device(mvp, n)
    frame, stats = mesh.render()
    order = oracle.sort_indexes(np.arange(n, dtype=np.uint32), ci, mvp)
    np.testing.assert_array_equal(worker.debug_read(2, n), order)
    c, cov, rgba, _ = helpers.oracle_inputs(scene)
    ocam = oracle.make_camera(cam.model_view(), cam.projection, cam.position, W, H, scene.sh_degree, scene.sh_degree)
    (fb, amb), = oracle.render_windows(ocam, c, cov, rgba, scene.sh if scene.sh_degree else None, order, [(0, 0, W, H)])[0]
    assert frame[..., 3].any(), f"{cfg_name}: the frame is empty, it checks nothing"
    msg = helpers.compare_frames(frame, fb, amb, f"{cfg_name} whole frame {W}x{H}")
    err = np.abs(frame.astype(np.float32) - np.clip(fb, 0, 1) * 255.0)
    _write_report(f"whole_frame_{cfg_name}_{W}x{H}.json",
                  {"config": f"{cfg_name} at {W}x{H}, every pixel", "splats": n, "visible": int(stats.visible_splats),
                   "list_bin_px": int(stats.list_bin_px), "splats_walked": int(stats.splats_walked), "parity": msg,
                   "worst_1_255": round(float(err.max()), 3), "pixels_beyond_1_255": int((err.max(axis=-1) > 1.5).sum()),
                   "pixels": W * H, "ambiguous_pixels": int(amb.sum())})
    print(msg)
    worker.terminate()
    mesh.dispose()


def test_c3_garden_1080p_crops_match_oracle(ctx):
    _crop_parity(ctx, "C3", 8)


def test_c3_orbit_poses_crops_match_oracle(ctx):
    """Off the demo pose: poses 15, 30 and 45 of the 60-pose orbit (a quarter, a half and three quarters of the way round
    the look-at point), 4 windows each."""
    cfg = scenes.CONFIGS["C3"]
    cams = camera.orbit_cameras(cfg["pose"], cfg["width"], cfg["height"], 60)
    reports = [_crop_parity(ctx, "C3", 4, cam=cams[k], tag=f"C3orbit{k}") for k in (15, 30, 45)]
    _write_report("crops_C3orbit.json", {"config": "C3orbit", "poses": [15, 30, 45], "reports": reports})


def test_c3_whole_frame_at_480x270_matches_oracle(ctx):
    """The crops above cover 8 windows of 64 x 64 px per configuration (1.6 % of a 1080p frame).  This closes "some tile nobody
    cropped" from the other side: all 5.8 M splats, SH-2, at 480 x 270 - where a pixel sees about sixteen times the splats of a
    1080p pixel."""
    _whole_frame(ctx, "C3", 480, 270)


def test_c5_garden_8k_crops_match_oracle(ctx):
    _crop_parity(ctx, "C5", 4)


def test_c2_truck_1080p_crops_match_oracle(ctx):
    _crop_parity(ctx, "C2", 8)


def test_c2_whole_frame_at_480x270_matches_oracle(ctx):
    """ truck stand-in: 2.5 M splats, SH-0."""
    _whole_frame(ctx, "C2", 480, 270)


def test_c4_sixteen_million_4k_crops_match_oracle(ctx):
    _crop_parity(ctx, "C4", 8)


def test_c4_whole_frame_at_480x270_matches_oracle(ctx):
    """ 16 M uniform random splats, fp16 covariances."""
    _whole_frame(ctx, "C4", 480, 270)


def test_c3t_translucent_1080p_crops_match_oracle(ctx):
    """The long-list path: pixels do not saturate early, the blend scans and walks its entry lists (9.3 M entries staged,
    3.6 M pairs walked per frame against 0.5 M for C3)."""
    rep = _crop_parity(ctx, "C3T", 8)
    assert rep["splats_walked"] > 2_000_000, rep["splats_walked"]


def test_c3t_whole_frame_at_480x270_matches_oracle(ctx):
    """ the translucent scene: long lists that do not saturate, chunks closed by the per-bin kernel."""
    _whole_frame(ctx, "C3T", 480, 270)


def test_c3s_capture_like_1080p_crops_match_oracle(ctx):
    """Surface-like stand-in: flat anisotropic splats on 2-D manifolds, 40 % of them nearly transparent, camera outside the
    object (scenes.capture_like)."""
    _crop_parity(ctx, "C3S", 8)


def test_c3s_whole_frame_at_480x270_matches_oracle(ctx):
    """ the capture-like scene: surfels, 32-px lists, the deep pass."""
    _whole_frame(ctx, "C3S", 480, 270)
