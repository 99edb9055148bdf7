"""Seed-fuzz parity soak: what the GPU tests check on fixed seeds, here on seeds nobody has looked at.  Per iteration a scene
(random size, SH degree, covariance format, splat scale), a camera on a random orbit pose and a random viewport; checked:
  * the depth sort, bit for bit, against the C oracle (sort_oracle.c, itself pinned to the reference's sorter), integer and
    float centres alternating, and the frustum-culled sort = the full list minus the culled splats;
  * the frame against the fp32 raster oracle within the stated tolerance (tests/helpers.compare_frames);
  * three strips of tile rows == the full frame, byte for byte; the same frame from a context with streams of its own;
  * a visibility-culled sort draws the same frame;
  * every other scene: a random destination (depth planes with per-pixel noise or holes, fp32 or 24-bit, with or without colour)
    against the oracle with the same destination (same tolerance), strips included;
  * every third scene: the same frame in GS_DRAW_ROP8 (the reference's RGBA8 target, rounded after every splat) against the
    ROP-emulating oracle (>= 99.5 % of the channel values equal, never more than 1 apart), over the destination when there is one.
The oracle is the checker here, as in tests/ (this tool is test infrastructure, not product).

usage: python tests/tools/soak.py [iterations=24] [first_seed=1000] [max_splats=60000]   -> one line per iteration, "soak: N iterations, 0 failures" """
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import helpers
import oracle
from gaussiansplats3d_amd import Context, SplatMesh, camera, create_sort_worker, util

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 24
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
max_n = int(sys.argv[3]) if len(sys.argv) > 3 else 60000
failures = 0
def rop8_check(mesh, ocam, inputs, order, W, H, cuts, depth=None, unorm24=False, dst=None, full=False):
    """GS_DRAW_ROP8 / GS_DRAW_ROP8_FULL (gs_mesh_set_draw_mode): the frame against the ROP-emulating oracle - >= 99.5 % of the channel
    values equal, never more than 1 apart (gs_mesh_debug_rop8's gate; the bounded walk: colour to that gate, alpha <= 2 steps) - and
    strips that tile it byte for byte."""
    mesh.set_draw_mode(rop8=True, full=full)
    try:
        got, _ = mesh.render()
        (fb8, _), = oracle.render_windows(ocam, *inputs, order, windows=[(0, 0, W, H)], rop8=True, depth=depth, depth_unorm24=unorm24, dst_rgba=dst)[0]
        d = np.abs(got.astype(np.int32) - np.floor(np.clip(fb8, 0, 1) * 255.0 + 0.5).astype(np.int32))
        if not full:
            assert d[..., 3].max() <= 2, f"ROP8 mode (bounded) alpha channel {d[..., 3].max()} steps from the oracle"
            d = d[..., :3]
        assert d.max() <= 1 and (d == 0).mean() >= 0.995, f"ROP8 mode ({'full' if full else 'bounded'}) vs the ROP-emulating oracle: {(d == 0).mean():.4f} equal, max {d.max()}"
        parts = [mesh.render(tile_rows=(a, b))[0] for a, b in zip(cuts[:-1], cuts[1:])]
        assert np.array_equal(np.concatenate(parts, axis=0), got), "strips do not tile the ROP8 frame"
    finally:
        mesh.set_draw_mode(rop8=False)
    return f"rop8{' full' if full else ''} {(d == 0).mean():.4f} equal max {int(d.max())}"


t_start = time.perf_counter()
ctx1 = Context(0, single_stream=True)
ctx2 = Context(0)
for it in range(iters):
    seed = seed0 + it
    rng = np.random.default_rng(seed)
    n = int(rng.integers(2000, max_n))
    sh_degree = int(rng.integers(0, 3))
    cov_half = bool(rng.integers(0, 2))
    scale = float(np.exp(rng.uniform(np.log(0.01), np.log(0.12))))
    W, H = int(rng.integers(64, 520)), int(rng.integers(48, 300))
    float_centres = bool(it & 1)
    scene = helpers.small_scene(n, sh_degree, seed, scale=scale, cov_half=cov_half)
    cam = camera.orbit_cameras("garden", W, H, 24)[int(rng.integers(0, 24))]
    label = f"seed {seed}: n={n} sh{sh_degree} {'f16' if cov_half else 'f32'}cov scale={scale:.3f} {W}x{H} {'float' if float_centres else 'int'} centres"
    try:
        centres = util.float_centers(scene.centers) if float_centres else util.integer_centers(scene.centers)
        expect = oracle.sort_indexes(np.arange(n, dtype=np.uint32), centres, cam.sort_mvp(), use_int=not float_centres)
        frames = {}
        for name, ctx in (("one stream", ctx1), ("streams", ctx2)):
            w = create_sort_worker(ctx, n, integer_based_sort=not float_centres)
            w.post_message({"centers": centres, "range": {"from": 0, "to": n - 1, "count": n}})
            mesh = SplatMesh(ctx, n, scene.sh_degree, scene.cov_half).build(scene.centers, scene.cov, scene.rgba,
                                                                           scene.sh if scene.sh_degree else None)
            mesh.set_camera(cam)
            w.sort_on_device(cam.sort_mvp(), n)
            got_order = w.debug_read(2, n)
            assert np.array_equal(got_order, expect), "sorted list differs from the oracle's"
            mesh.use_sorter_result(w, n)
            full, st = mesh.render()
            frames[name] = full
            if name == "one stream":
                c, cov, rgba, sh = helpers.oracle_inputs(scene)
                ocam = oracle.make_camera(cam.model_view(), cam.projection, cam.position, W, H, sh_degree=sh_degree, sh_stored=sh_degree)
                fb, q, amb, frags = oracle.render(ocam, c, cov, rgba, sh, expect)
                msg = helpers.compare_frames(full, fb, amb, "frame")
                rows = (H + 15) // 16
                cuts = sorted(set([0, rows] + [int(v) for v in rng.integers(0, rows + 1, size=2)]))
                parts = [mesh.render(tile_rows=(a, b))[0] for a, b in zip(cuts[:-1], cuts[1:])]
                assert np.array_equal(np.concatenate(parts, axis=0), full), "strips do not tile the full frame"
                # frustum-culled sort: the full list minus the culled splats, same frame
                if not float_centres:
                    w.set_frustum_cull(True)
                    w.sort_on_device(cam.sort_mvp(), n)
                    culled, _ = mesh.render()
                    assert np.array_equal(culled, full), "frame from the frustum-culled list differs"
                    w.set_frustum_cull(False)
                # visibility-culled sort (the per-rank path): vertex stage first, then the sort over what it kept
                w.set_visibility_cull(True)
                mesh.project(None)
                w.sort_on_device(cam.sort_mvp(), n)
                vis, _ = mesh.render()
                assert np.array_equal(vis, full), "frame from the visibility-culled list differs"
                w.set_visibility_cull(False)
                # a destination (depth the host's own geometry left + its colour): the engine against the oracle with the same one,
                # strips tile it; clearing it restores the plain frame
                if rng.integers(0, 2):
                    w.sort_on_device(cam.sort_mvp(), n)
                    proj = oracle.project(ocam, c, cov, rgba, sh)
                    zw = (proj["ndcz"] * np.float32(0.5) + np.float32(0.5)).astype(np.float32)
                    seen = zw[proj["visible"] == 1]
                    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
                    if seen.size:
                        mid, spread = float(np.quantile(seen, rng.uniform(0.05, 0.6))), float(seen.std())
                    else:
                        mid, spread = 0.5, 0.1
                    kind = int(rng.integers(0, 3))
                    depth = mid + spread * rng.uniform(-0.3, 0.3) * ((xx - W / 2) / W + (yy - H / 2) / H)
                    if kind == 1:
                        depth = depth + spread * 0.2 * rng.standard_normal((H, W))            # per-pixel noise: no two pixels cut alike
                    if kind == 2:
                        depth = np.where(rng.random((H, W)) < 0.5, depth, 1.0)                  # holes pixel by pixel
                    depth = np.clip(depth, 0.0, 1.0).astype(np.float32)
                    unorm24 = bool(rng.integers(0, 2))
                    dst = rng.integers(0, 256, size=(H, W, 4), dtype=np.uint8) if rng.integers(0, 2) else None
                    mesh.set_destination(depth=depth, rgba=dst, depth_unorm24=unorm24)
                    got_d, _ = mesh.render()
                    fb_d, _, amb_d, _ = oracle.render(ocam, c, cov, rgba, sh, expect, depth=depth, depth_unorm24=unorm24, dst_rgba=dst)
                    msg += " | " + helpers.compare_frames(got_d, fb_d, amb_d, f"destination kind {kind}{' unorm24' if unorm24 else ''}")
                    parts = [mesh.render(tile_rows=(a, b))[0] for a, b in zip(cuts[:-1], cuts[1:])]
                    assert np.array_equal(np.concatenate(parts, axis=0), got_d), "strips do not tile the depth-tested frame"
                    if it % 3 == 0:                            # ... and in the reference's RGBA8-per-splat mode over the same destination
                        msg += " | " + rop8_check(mesh, ocam, (c, cov, rgba, sh), expect, W, H, cuts, depth, unorm24, dst, full=bool(it % 2))
                    mesh.set_destination()
                    again, _ = mesh.render()
                    assert np.array_equal(again, full), "clearing the destination does not restore the plain frame"
                elif it % 3 == 0:
                    w.sort_on_device(cam.sort_mvp(), n)
                    msg += " | " + rop8_check(mesh, ocam, (c, cov, rgba, sh), expect, W, H, cuts, full=bool(it % 2))
                    again, _ = mesh.render()
                    assert np.array_equal(again, full), "leaving the ROP8 mode does not restore the plain frame"
            w.terminate(); mesh.dispose()
        assert np.array_equal(frames["one stream"], frames["streams"]), "the default context's frame differs from the one-stream context's"
        print(f"ok   {label} | entries {st.tile_entries} visible {st.visible_splats} | {msg}", flush=True)
    except Exception as e:                                     # keep going: a soak reports every failing seed
        failures += 1
        print(f"FAIL {label}: {type(e).__name__}: {str(e)[:300]}", flush=True)
ctx1.close(); ctx2.close()
print(f"soak: {iters} iterations from seed {seed0}, {failures} failures, {time.perf_counter() - t_start:.0f} s")
sys.exit(1 if failures else 0)
