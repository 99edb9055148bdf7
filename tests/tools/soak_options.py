"""Seed-fuzz of the SHADER PERMUTATIONS: per iteration a random scene and a random combination of the options
that reach the reference's shaders - antialiased, pointCloudMode, splatScale, kernel2DSize, maxScreenSpaceSplatSize,
focalAdjustment, the evaluated SH degree, half-precision covariances - on one of four pipelines: perspective, orthographic,
dynamicMode (three scenes with random rigid / scaled transforms), enableOptionalEffects (per-scene opacity / visibility) with or
without the distance fade-in.  Checked: the frame against the fp32 raster oracle configured the same way
(tests/helpers.compare_frames), and narrow strips of tile rows == the full frame byte for byte (the strip pre-test of the vertex
stage may only drop splats that cannot reach the strip, whatever the permutation).
The oracle is the checker here, as in tests/.

usage: python tests/tools/soak_options.py [iterations=120] [first_seed=4000] [max_splats=40000] """
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import helpers
import oracle
from gaussiansplats3d_amd import Context, SplatMesh, camera, util

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 120
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
max_n = int(sys.argv[3]) if len(sys.argv) > 3 else 40000
ctx = Context(0)
failures = 0
t_start = time.perf_counter()


def random_transform(rng, scale_ok):
    a = rng.normal(size=3); a /= np.linalg.norm(a)
    th = np.deg2rad(rng.uniform(-25.0, 25.0))
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    R = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)
    M = np.eye(4)
    M[:3, :3] = R * (float(rng.uniform(0.6, 1.7)) if scale_ok and rng.integers(0, 2) else 1.0)
    M[:3, 3] = rng.normal(size=3) * 0.25
    return M.T.reshape(16)


for it in range(iters):
    seed = seed0 + it
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1500, max_n))
    sh_degree = int(rng.integers(0, 3))
    cov_half = bool(rng.integers(0, 3) == 0)
    scale = float(np.exp(rng.uniform(np.log(0.015), np.log(0.15))))
    W, H = int(rng.integers(96, 420)), int(rng.integers(64, 260))
    pipeline = ["perspective", "orthographic", "dynamic", "effects"][int(rng.integers(0, 4))]
    opts = {}
    if rng.integers(0, 3) == 0: opts["antialiased"] = True
    if rng.integers(0, 5) == 0: opts["point_cloud_mode"] = True
    if rng.integers(0, 2): opts["splat_scale"] = float(rng.choice([0.5, 0.8, 1.5, 2.5]))
    if rng.integers(0, 3) == 0: opts["kernel_2d_size"] = float(rng.choice([0.0, 0.1, 0.6]))
    if rng.integers(0, 4) == 0: opts["max_screen_space_splat_size"] = float(rng.choice([12.0, 24.0, 200.0]))
    focal_adj = float(rng.choice([0.5, 1.0, 1.0, 2.0]))
    eval_deg = int(rng.integers(0, sh_degree + 1)) if rng.integers(0, 3) == 0 else None
    fade = pipeline == "effects" and bool(rng.integers(0, 2))
    label = (f"seed {seed}: {pipeline} n={n} sh{sh_degree}" + (f"->{eval_deg}" if eval_deg is not None else "") +
             f" {'f16' if cov_half else 'f32'}cov scale={scale:.3f} {W}x{H} focal_adj={focal_adj} fade={fade} " +
             ",".join(f"{k}={v}" for k, v in opts.items()))
    try:
        scene = helpers.small_scene(n, sh_degree, seed, scale=scale, cov_half=cov_half)
        up, pos, look = camera.DEMO_POSES["garden"]
        if pipeline == "orthographic":
            zoom = float(rng.uniform(15.0, 80.0))
            cam = camera.OrthographicCamera(W, H, pos, look, up, zoom=zoom)
        else:
            cam = camera.orbit_cameras("garden", W, H, 24)[int(rng.integers(0, 24))]
        sidx = (rng.integers(0, 3, n)).astype(np.uint32) if pipeline in ("dynamic", "effects") else None
        kw = dict(opts)
        if pipeline == "dynamic": kw["dynamic_mode"] = True
        if pipeline == "effects": kw["enable_optional_effects"] = True
        mesh = SplatMesh(ctx, n, sh_degree, cov_half, **kw)
        mesh.build(scene.centers, scene.cov, scene.rgba, scene.sh if sh_degree else None, **({"scene_indexes": sidx} if sidx is not None else {}))
        c, cov, rgba, sh = helpers.oracle_inputs(scene)
        ocam = oracle.make_camera(cam.model_view(), cam.projection, cam.position, W, H,
                                  sh_degree=sh_degree if eval_deg is None else eval_deg, sh_stored=sh_degree,
                                  splat_scale=opts.get("splat_scale", 1.0), kernel2d=opts.get("kernel_2d_size", 0.3),
                                  max_splat_px=opts.get("max_screen_space_splat_size", 1024.0), focal_adjustment=focal_adj,
                                  antialiased=opts.get("antialiased", False), point_cloud=opts.get("point_cloud_mode", False))
        if pipeline == "orthographic":
            ocam.orthographic, ocam.ortho_zoom = 1, zoom
            order = oracle.sort_indexes(np.arange(n, dtype=np.uint32), util.integer_centers(scene.centers), cam.sort_mvp())
        elif pipeline == "dynamic":
            transforms = [np.eye(4).reshape(16), random_transform(rng, True), random_transform(rng, False)]
            mesh.set_scenes(transforms=transforms, camera_position=cam.position)
            ocam = oracle.set_scenes(ocam, view_matrix=cam.view, transforms=transforms, camera_position=cam.position, dynamic=True)
            order = rng.permutation(n).astype(np.uint32)                 # any fixed draw order: both sides use it
        elif pipeline == "effects":
            opacity = [1.0, float(rng.uniform(0.02, 0.9)), float(rng.choice([0.005, 0.5]))]
            visible = (1, int(rng.integers(0, 2)), 1)
            mesh.set_scenes(opacity=opacity, visible=visible)
            ocam = oracle.set_scenes(ocam, opacity=opacity, visible=visible, effects=True)
            order = oracle.sort_indexes(np.arange(n, dtype=np.uint32), util.integer_centers(scene.centers), cam.sort_mvp())
            if fade:
                center = scene.centers.mean(axis=0)
                radius = float(np.quantile(np.linalg.norm(scene.centers - center, axis=1), rng.uniform(0.2, 0.8)))
                mesh.set_fade_in(center, radius)
                ocam.fade_in, ocam.fade_start = 1, radius
                ocam.scene_center[:] = center.astype(np.float32).tolist()
        else:
            order = oracle.sort_indexes(np.arange(n, dtype=np.uint32), util.integer_centers(scene.centers), cam.sort_mvp())
        mesh.set_camera(cam, focal_adjustment=focal_adj, spherical_harmonics_degree=eval_deg)
        mesh.update_render_indexes(order, n)
        got, st = mesh.render()
        fb, q, amb, frags = oracle.render(ocam, c, cov, rgba, sh, order, **({"scene_indexes": sidx} if sidx is not None else {}))
        msg = helpers.compare_frames(got, fb, amb, "frame")
        rows = (H + 15) // 16
        step = int(rng.integers(1, 4))
        parts = [mesh.render(tile_rows=(r, min(r + step, rows)))[0] for r in range(0, rows, step)]
        assert np.array_equal(np.concatenate(parts, axis=0), got), "strips do not tile the full frame"
        mesh.dispose()
        print(f"ok   {label} | visible {st.visible_splats} frags {frags} | {msg}", flush=True)
    except Exception as e:
        failures += 1
        print(f"FAIL {label}: {type(e).__name__}: {str(e)[:300]}", flush=True)
ctx.close()
print(f"soak_options: {iters} iterations from seed {seed0}, {failures} failures, {time.perf_counter() - t_start:.0f} s")
sys.exit(1 if failures else 0)
