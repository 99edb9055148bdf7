"""Seeded inputs of the raster known-answer tests: shared by oracle/make_golden_raster.py (which pushes them through the
reference's own shader text) and tests/test_raster_ref.py (which pushes them through the C raster oracle and the HIP vertex
stage).  Every case = a small scene, a camera and the Viewer options of one shader permutation."""
import numpy as np

import helpers
from gaussiansplats3d_amd import camera
from gaussiansplats3d_amd.util import to_half_three


def shader_builds():
    """Arguments of SplatMaterial3D.build per compiled shader (dynamicMode, enableOptionalEffects, antialiased,
    maxScreenSpaceSplatSize, splatScale, pointCloudModeEnabled, maxSphericalHarmonicsDegree, kernel2DSize)."""
    base = dict(dynamicMode=False, enableOptionalEffects=False, antialiased=False, maxScreenSpaceSplatSize=1024, splatScale=1.0,
                pointCloudModeEnabled=False, kernel2DSize=0.3)
    return {
        "base0": dict(base, maxSphericalHarmonicsDegree=0),
        "base1": dict(base, maxSphericalHarmonicsDegree=1),
        "base2": dict(base, maxSphericalHarmonicsDegree=2),
        "aa2": dict(base, maxSphericalHarmonicsDegree=2, antialiased=True),
        "small1": dict(base, maxSphericalHarmonicsDegree=1, maxScreenSpaceSplatSize=48, kernel2DSize=0.1),
        "effects1": dict(base, maxSphericalHarmonicsDegree=1, enableOptionalEffects=True),
        "dynamic2": dict(base, maxSphericalHarmonicsDegree=2, dynamicMode=True),
    }


CASES = ["sh0", "sh1_half", "sh2", "sh2_eval1", "antialiased", "point_cloud", "orthographic", "clamp_scale_focal", "fade_in",
         "sh8", "effects", "dynamic"]


def _uniforms(cam, scene_sh, **kw):
    fa = kw.get("focal_adjustment", 1.0)
    fx, fy = cam.focal(fa)
    u = dict(model_view=cam.model_view(kw.get("mesh_world")), projection=cam.projection, view_matrix=cam.view,
             camera_position=cam.position, focal=(fx, fy), viewport=(cam.width, cam.height),
             ortho_zoom=float(getattr(cam, "zoom", 1.0)), inverse_focal_adjustment=1.0 / fa, splat_scale=kw.get("splat_scale", 1.0),
             orthographic=int(bool(getattr(cam, "is_orthographic", False))), point_cloud=int(kw.get("point_cloud", False)),
             sh_degree=kw.get("sh_degree", scene_sh), fade_in_complete=0 if kw.get("fade") else 1, scene_count=kw.get("scene_count", 1),
             scene_center=list(kw["fade"][0]) if kw.get("fade") else [0.0, 0.0, 0.0],
             fade_start_radius=kw["fade"][1] if kw.get("fade") else 0.0, transforms=kw.get("transforms", []),
             scene_opacity=kw.get("opacity", []), scene_visibility=kw.get("visible", []), sh8_range=kw.get("sh8_range", (-1.5, 1.5)))
    return u


def make_case(name):
    up, pos, look = camera.DEMO_POSES["garden"]
    cam = camera.demo_camera("garden", 512, 288)
    n = 2000
    case = dict(cov_half=False, sh8=False, scene_idx=None, kernel2d=0.3, max_splat_px=1024.0, antialiased=False)
    if name == "sh0":
        sc = helpers.small_scene(n, 0, seed=301)
        case.update(build="base0", uniforms=_uniforms(cam, 0))
    elif name == "sh1_half":
        sc = helpers.small_scene(n, 1, seed=302, cov_half=True)
        case.update(build="base1", cov_half=True, uniforms=_uniforms(cam, 1))
    elif name == "sh2":
        sc = helpers.small_scene(n, 2, seed=303)
        case.update(build="base2", uniforms=_uniforms(cam, 2))
    elif name == "sh2_eval1":                     # degree-2 data, sphericalHarmonicsDegree uniform = 1
        sc = helpers.small_scene(n, 2, seed=304)
        case.update(build="base2", uniforms=_uniforms(cam, 2, sh_degree=1))
    elif name == "antialiased":
        sc = helpers.small_scene(n, 2, seed=305, scale=0.01)
        case.update(build="aa2", antialiased=True, uniforms=_uniforms(cam, 2))
    elif name == "point_cloud":
        sc = helpers.small_scene(n, 0, seed=306)
        case.update(build="base0", uniforms=_uniforms(cam, 0, point_cloud=True))
    elif name == "orthographic":
        sc = helpers.small_scene(n, 1, seed=307)
        cam = camera.OrthographicCamera(512, 288, pos, look, up, zoom=40.0)
        case.update(build="base1", uniforms=_uniforms(cam, 1))
    elif name == "clamp_scale_focal":             # maxScreenSpaceSplatSize 48 (clamps), kernel2DSize 0.1, splatScale, focalAdjustment
        sc = helpers.small_scene(n, 1, seed=308, scale=0.2)
        case.update(build="small1", kernel2d=0.1, max_splat_px=48.0, uniforms=_uniforms(cam, 1, splat_scale=1.7, focal_adjustment=2.0))
    elif name == "fade_in":
        sc = helpers.small_scene(n, 0, seed=309)
        center = sc.centers.mean(axis=0)
        radius = float(np.median(np.linalg.norm(sc.centers - center, axis=1)))
        case.update(build="base0", uniforms=_uniforms(cam, 0, fade=(center.astype(np.float32).tolist(), radius)))
    elif name == "sh8":
        sc = helpers.small_scene(n, 2, seed=310)
        case.update(build="base2", sh8=True, uniforms=_uniforms(cam, 2, sh8_range=(-0.9, 1.1)))
    elif name == "effects":
        sc = helpers.small_scene(n, 1, seed=311)
        case.update(build="effects1", scene_idx=(np.arange(n) % 3).astype(np.uint32),
                    uniforms=_uniforms(cam, 1, scene_count=3, opacity=[1.0, 0.4, 0.005], visible=[1, 1, 1]))
    elif name == "dynamic":
        sc = helpers.small_scene(n, 2, seed=312)
        c, s = np.cos(0.3), np.sin(0.3)
        t1 = np.array([[c, 0, s, 0.4], [0, 1, 0, -0.2], [-s, 0, c, 0.1], [0, 0, 0, 1]]).T.reshape(16)
        t2 = np.array([[1.2, 0, 0, -0.5], [0, 1.2, 0, 0.3], [0, 0, 1.2, 0.0], [0, 0, 0, 1]]).T.reshape(16)
        case.update(build="dynamic2", scene_idx=(np.arange(n) % 3).astype(np.uint32),
                    uniforms=_uniforms(cam, 2, scene_count=3, transforms=[np.eye(4).T.reshape(16), t1, t2]))
    else:
        raise KeyError(name)
    cov = sc.cov
    case["cov16"] = to_half_three(cov) if case["cov_half"] else None
    if case["cov_half"]:
        cov = case["cov16"].view(np.float16).astype(np.float32)
    sh = sc.sh.astype(np.float32) if sc.sh_degree else np.zeros((n, 0), np.float32)      # fp16 storage, widened by the sampler
    sh_u8 = None
    if case["sh8"]:                               # compression level 2: uint8 storage, the sampler returns byte / 255
        lo, hi = case["uniforms"]["sh8_range"]
        sh_u8 = np.clip(np.floor((np.clip(sh, lo, hi) - lo) / (hi - lo) * 255.0), 0, 255).astype(np.uint8)
        sh = sh_u8.astype(np.float32) / np.float32(255.0)
    case.update(scene=sc, camera=cam, centers=sc.centers, cov=cov, rgba=sc.rgba, sh_stored=sc.sh_degree, sh_sampled=sh, sh_u8=sh_u8)
    return case


def fragment_samples():
    rng = np.random.default_rng(77)
    vp = np.concatenate([rng.uniform(-3.2, 3.2, (1500, 2)),
                         np.array([[np.sqrt(8.0), 0.0], [2.0, 2.0], [0.0, 0.0], [2.0000002, 2.0], [1.9999999, 2.0]])]).astype(np.float32)
    vc = rng.uniform(0, 1, (vp.shape[0], 4)).astype(np.float32)
    return vp, vc
