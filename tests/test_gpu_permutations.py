"""-m gpu: the reference's shader permutations against the fp32 raster oracle: orthographic camera,
distance fade-in, per-scene opacity / visibility (enableOptionalEffects), per-scene transforms (dynamicMode), 8-bit SH."""
import numpy as np
import pytest

import helpers
import oracle
from gaussiansplats3d_amd import Context, SplatMesh, camera, util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = Context(0)
    yield c
    c.close()


def _order(scene, cam):
    return oracle.sort_indexes(np.arange(scene.count, dtype=np.uint32), util.integer_centers(scene.centers), cam.sort_mvp())


def _ocam(scene, cam, **kw):
    return oracle.make_camera(cam.model_view(), cam.projection, cam.position, cam.width, cam.height,
                              sh_degree=scene.sh_degree, sh_stored=scene.sh_degree, **kw)


def test_orthographic_camera(ctx):
    scene = helpers.small_scene(2500, 1, seed=41)
    up, pos, look = camera.DEMO_POSES["garden"]
    cam = camera.OrthographicCamera(320, 200, pos, look, up, zoom=40.0)      # 40 px per world unit
    order = _order(scene, cam)
    mesh = SplatMesh(ctx, scene.count, 1).build(scene.centers, scene.cov, scene.rgba, scene.sh)
    mesh.set_camera(cam)
    mesh.update_render_indexes(order, scene.count)
    got, stats = mesh.render()
    c, cov, rgba, sh = helpers.oracle_inputs(scene)
    ocam = _ocam(scene, cam)
    ocam.orthographic, ocam.ortho_zoom = 1, 40.0
    fb, q, amb, frags = oracle.render(ocam, c, cov, rgba, sh, order)
    assert frags > 1000
    print(helpers.compare_frames(got, fb, amb, "ortho"))
    # sanity: it is not the perspective image
    pcam = camera.demo_camera("garden", 320, 200)
    mesh.set_camera(pcam)
    persp, _ = mesh.render()
    assert np.abs(persp.astype(int) - got.astype(int)).max() > 30
    mesh.dispose()


def test_fade_in(ctx):
    scene = helpers.small_scene(2500, 0, seed=42)
    cam = camera.demo_camera("garden", 256, 144)
    order = _order(scene, cam)
    center = scene.centers.mean(axis=0)
    radius = float(np.median(np.linalg.norm(scene.centers - center, axis=1)))   # half of the splats inside the fade start
    mesh = SplatMesh(ctx, scene.count, 0).build(scene.centers, scene.cov, scene.rgba)
    mesh.set_fade_in(center, radius)
    mesh.set_camera(cam)
    mesh.update_render_indexes(order, scene.count)
    got, _ = mesh.render()
    c, cov, rgba, sh = helpers.oracle_inputs(scene)
    ocam = _ocam(scene, cam)
    ocam.fade_in, ocam.fade_start = 1, radius
    ocam.scene_center[:] = center.astype(np.float32).tolist()
    fb, q, amb, _ = oracle.render(ocam, c, cov, rgba, sh, order)
    print(helpers.compare_frames(got, fb, amb, "fade-in"))
    full = oracle.render(_ocam(scene, cam), c, cov, rgba, sh, order)[1]
    assert np.abs(full.astype(int) - q.astype(int)).max() > 20, "fade-in should change the image"
    mesh.dispose()


def _three_scenes(n, sh_degree, seed):
    scene = helpers.small_scene(n, sh_degree, seed=seed)
    scene_idx = (np.arange(n) % 3).astype(np.uint32)
    return scene, scene_idx


@pytest.mark.parametrize("visible", [(1, 1, 1), (1, 0, 1)])
def test_scene_opacity_and_visibility(ctx, visible):
    scene, sidx = _three_scenes(3000, 0, 43)
    cam = camera.demo_camera("garden", 256, 144)
    order = _order(scene, cam)
    opacity = [1.0, 0.4, 0.005]                            # the third scene is below the 0.01 cut-off
    mesh = SplatMesh(ctx, scene.count, 0, enable_optional_effects=True)
    mesh.build(scene.centers, scene.cov, scene.rgba, scene_indexes=sidx)
    mesh.set_scenes(opacity=opacity, visible=visible)
    mesh.set_camera(cam)
    mesh.update_render_indexes(order, scene.count)
    got, _ = mesh.render()
    c, cov, rgba, sh = helpers.oracle_inputs(scene)
    ocam = oracle.set_scenes(_ocam(scene, cam), opacity=opacity, visible=visible, effects=True)
    fb, q, amb, _ = oracle.render(ocam, c, cov, rgba, sh, order, scene_indexes=sidx)
    print(helpers.compare_frames(got, fb, amb, f"effects vis={visible}"))
    mesh.dispose()


def test_dynamic_mode_scene_transforms(ctx):
    """dynamicMode: modelView = viewMatrix * transforms[scene]; SH view direction from the camera position in the
    scene's own frame."""
    scene, sidx = _three_scenes(3000, 2, 44)
    cam = camera.demo_camera("garden", 256, 144)

    def rot_t(axis, deg, t):
        a = np.deg2rad(deg)
        R = np.eye(4)
        i, j = [(1, 2), (0, 2), (0, 1)][axis]
        R[i, i] = R[j, j] = np.cos(a); R[i, j] = -np.sin(a); R[j, i] = np.sin(a)
        R[:3, 3] = t
        return R.T.reshape(16)                              # column-major elements
    transforms = [np.eye(4).reshape(16), rot_t(1, 12.0, (0.3, -0.1, 0.2)), rot_t(2, -20.0, (-0.2, 0.25, 0.0))]
    order = np.arange(scene.count, dtype=np.uint32)[::-1].copy()     # any fixed draw order: both sides use it
    mesh = SplatMesh(ctx, scene.count, 2, dynamic_mode=True)
    mesh.build(scene.centers, scene.cov, scene.rgba, scene.sh, scene_indexes=sidx)
    mesh.set_scenes(transforms=transforms, camera_position=cam.position)
    mesh.set_camera(cam)
    mesh.update_render_indexes(order, scene.count)
    got, _ = mesh.render()
    c, cov, rgba, sh = helpers.oracle_inputs(scene)
    ocam = oracle.set_scenes(_ocam(scene, cam), view_matrix=cam.view, transforms=transforms, camera_position=cam.position,
                             dynamic=True)
    fb, q, amb, frags = oracle.render(ocam, c, cov, rgba, sh, order, scene_indexes=sidx)
    assert frags > 1000
    print(helpers.compare_frames(got, fb, amb, "dynamic"))
    static = oracle.render(_ocam(scene, cam), c, cov, rgba, sh, order)[1]
    assert np.abs(static.astype(int) - q.astype(int)).max() > 30, "the transforms should move splats"
    mesh.dispose()


@pytest.mark.parametrize("sh_degree", [1, 2])
def test_8bit_spherical_harmonics(ctx, sh_degree):
    scene = helpers.small_scene(2500, sh_degree, seed=45 + sh_degree)
    rng = np.random.default_rng(5)
    ncoef = 9 if sh_degree == 1 else 24
    sh8 = rng.integers(0, 256, size=(scene.count, ncoef), dtype=np.uint8)
    lo, hi = -1.5, 1.5                                      # Constants.SphericalHarmonics8BitCompressionRange defaults
    cam = camera.demo_camera("garden", 256, 144)
    order = _order(scene, cam)
    mesh = SplatMesh(ctx, scene.count, sh_degree, spherical_harmonics_8bit=True)
    mesh.build(scene.centers, scene.cov, scene.rgba, sh8)
    mesh.set_scenes(sh8_range=[(lo, hi)])
    mesh.set_camera(cam)
    mesh.update_render_indexes(order, scene.count)
    got, _ = mesh.render()
    c, cov, rgba, _ = helpers.oracle_inputs(scene)
    ocam = oracle.set_scenes(_ocam(scene, cam), sh8_range=[(lo, hi)])
    ocam.sh8 = 1
    fb, q, amb, _ = oracle.render(ocam, c, cov, rgba, sh8.astype(np.float32), order)
    print(helpers.compare_frames(got, fb, amb, f"sh8 degree {sh_degree}"))
    mesh.dispose()


@pytest.mark.parametrize("opts", [
    dict(antialiased=True),
    dict(point_cloud_mode=True),
    dict(splat_scale=0.5),
    dict(splat_scale=2.5),
    dict(kernel_2d_size=0.1),
    dict(max_screen_space_splat_size=24.0),          # the 1024-px clamp of SplatMaterial3D.js:195-196, made active
    dict(focal_adjustment=2.0),
    dict(evaluate_sh_degree=1),                      # sphericalHarmonicsDegree uniform below the stored degree
    dict(evaluate_sh_degree=0),
    dict(antialiased=True, splat_scale=1.5, focal_adjustment=0.5),
], ids=lambda o: ",".join(f"{k}={v}" for k, v in o.items()))
def test_viewer_options(ctx, opts):
    """Viewer / SplatMesh options that reach the shaders as uniforms or #defines: antialiased, pointCloudMode, splatScale,
    kernel2DSize, maxScreenSpaceSplatSize, focalAdjustment, sphericalHarmonicsDegree."""
    opts = dict(opts)
    scene = helpers.small_scene(2500, 2, seed=60)
    cam = camera.demo_camera("garden", 256, 144)
    order = _order(scene, cam)
    focal_adj = opts.pop("focal_adjustment", 1.0)
    eval_deg = opts.pop("evaluate_sh_degree", None)
    mesh = SplatMesh(ctx, scene.count, 2, **opts).build(scene.centers, scene.cov, scene.rgba, scene.sh)
    mesh.set_camera(cam, focal_adjustment=focal_adj, spherical_harmonics_degree=eval_deg)
    mesh.update_render_indexes(order, scene.count)
    got, _ = mesh.render()
    c, cov, rgba, sh = helpers.oracle_inputs(scene)
    ocam = oracle.make_camera(cam.model_view(), cam.projection, cam.position, cam.width, cam.height,
                              sh_degree=2 if eval_deg is None else eval_deg, sh_stored=2,
                              splat_scale=opts.get("splat_scale", 1.0), kernel2d=opts.get("kernel_2d_size", 0.3),
                              max_splat_px=opts.get("max_screen_space_splat_size", 1024.0), focal_adjustment=focal_adj,
                              antialiased=opts.get("antialiased", False), point_cloud=opts.get("point_cloud_mode", False))
    fb, q, amb, frags = oracle.render(ocam, c, cov, rgba, sh, order)
    assert frags > 200
    print(helpers.compare_frames(got, fb, amb, str(opts)))
    base = oracle.render(_ocam(scene, cam), c, cov, rgba, sh, order)[1]
    assert np.abs(base.astype(int) - q.astype(int)).max() > 8, "the option should change the image"
    mesh.dispose()


def _strips_equal_frame(mesh, cam, label, strip_rows=2):
    """Narrow strips of tile rows, drawn one by one, must tile the full frame bit for bit: the strip pre-test of the vertex
    stage (a bound of the splat's extent from its covariance's spectral-radius plane) may only drop splats that cannot reach
    the strip, whatever the shader permutation."""
    full, st = mesh.render()
    assert st.visible_splats > 100, label
    rows = (cam.height + 15) // 16
    parts = [mesh.render(tile_rows=(r, min(r + strip_rows, rows)))[0] for r in range(0, rows, strip_rows)]
    np.testing.assert_array_equal(np.concatenate(parts, axis=0), full, err_msg=label)


@pytest.mark.parametrize("opts", [
    dict(), dict(antialiased=True), dict(point_cloud_mode=True), dict(splat_scale=2.5), dict(kernel_2d_size=0.1, point_cloud_mode=True),
    dict(max_screen_space_splat_size=24.0), dict(focal_adjustment=2.0, splat_scale=1.7), dict(half_precision_covariances=True),
], ids=lambda o: ",".join(f"{k}={v}" for k, v in o.items()) or "base")
def test_strips_tile_the_frame_under_viewer_options(ctx, opts):
    opts = dict(opts)
    half = opts.pop("half_precision_covariances", False)
    focal_adj = opts.pop("focal_adjustment", 1.0)
    scene = helpers.small_scene(6000, 1, seed=71, scale=0.12, cov_half=half)     # large splats: many reach several strips
    cam = camera.demo_camera("garden", 320, 208)
    mesh = SplatMesh(ctx, scene.count, 1, scene.cov_half, **opts).build(scene.centers, scene.cov, scene.rgba, scene.sh)
    mesh.set_camera(cam, focal_adjustment=focal_adj)
    mesh.update_render_indexes(_order(scene, cam), scene.count)
    _strips_equal_frame(mesh, cam, str(opts))
    mesh.dispose()


def test_strips_tile_the_frame_orthographic_and_dynamic(ctx):
    up, pos, look = camera.DEMO_POSES["garden"]
    scene = helpers.small_scene(5000, 1, seed=72, scale=0.1)
    ocam = camera.OrthographicCamera(320, 208, pos, look, up, zoom=40.0)
    mesh = SplatMesh(ctx, scene.count, 1).build(scene.centers, scene.cov, scene.rgba, scene.sh)
    mesh.set_camera(ocam)
    mesh.update_render_indexes(_order(scene, ocam), scene.count)
    _strips_equal_frame(mesh, ocam, "orthographic")
    mesh.dispose()
    # dynamic mode: per-scene transforms (one of them scales by 1.6: the mesh-level row norms do not describe it, the pre-test
    # must fall back to the clamp) 
    scene, sidx = _three_scenes(4500, 1, 73)
    cam = camera.demo_camera("garden", 320, 208)
    big = np.diag([1.6, 1.6, 1.6, 1.0]); big[:3, 3] = (0.2, -0.1, 0.3)
    transforms = [np.eye(4).reshape(16), big.T.reshape(16), np.eye(4).reshape(16)]
    mesh = SplatMesh(ctx, scene.count, 1, dynamic_mode=True)
    mesh.build(scene.centers, scene.cov * 4.0, scene.rgba, scene.sh, scene_indexes=sidx)
    mesh.set_scenes(transforms=transforms, camera_position=cam.position)
    mesh.set_camera(cam)
    mesh.update_render_indexes(np.arange(scene.count, dtype=np.uint32), scene.count)
    _strips_equal_frame(mesh, cam, "dynamic")
    mesh.dispose()
