"""Raster parity PINNED to the reference's shader text.  tests/golden/raster_ref.npz holds what the reference's OWN GLSL
(the strings SplatMaterial3D.build() returns for 7 shader builds: SH degree 0 / 1 / 2, antialiased, a small
maxScreenSpaceSplatSize + kernel2DSize, enableOptionalEffects, dynamicMode) computes for seeded scenes when it is executed
statement by statement in IEEE fp32 (oracle/make_golden_raster.py: token rewrites + oracle/glsl_shim.hpp, data textures laid
out as SplatMesh.setupDataTextures does).  Compared here:
  CPU tier  oracle/raster_oracle.c's vertex stage (the checker every GPU raster test relies on) and its per-fragment rule;
  GPU tier  the HIP vertex stage (k_project) directly.
What stays a restatement: GL's fixed-function parts (quad rasterisation, the blend equation of three's NormalBlending)."""
import json
import os

import numpy as np
import pytest

import oracle
import raster_cases

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "raster_ref.npz"))
SQRT8 = np.float32(np.sqrt(np.float32(8.0)))


def _from_shader(res, viewport):
    """gl_Position / varyings of the four quad corners -> what was drawn and where (float64 arithmetic on fp32 outputs)."""
    pos = res[:, :, 0:4].astype(np.float64)
    reject = (res[:, 0, 2] == 2.0) & (res[:, 0, 3] == 1.0) & (res[:, 0, 0] == 0.0)        # gl_Position = (0,0,2,1)
    drawn = np.isfinite(res[:, :, 0:4]).all(axis=(1, 2)) & ~reject
    z = pos[:, 0, 2]
    drawn &= (z >= -1.0) & (z <= 1.0)                         # the quad sits at its centre's depth: GL clips it as a whole
    vp = np.asarray(viewport, np.float64)
    centre = (pos[:, :, 0:2].mean(axis=1) * 0.5 + 0.5) * vp
    b1 = (pos[:, 2, 0:2] - pos[:, 1, 0:2]) * 0.5 * vp * 0.5   # corners (-1,1) -> (1,1): 2 * basisVector1 in NDC
    b2 = (pos[:, 1, 0:2] - pos[:, 0, 0:2]) * 0.5 * vp * 0.5   # corners (-1,-1) -> (-1,1): 2 * basisVector2
    return drawn, centre, b1, b2, res[:, 0, 4:8].astype(np.float64)


def _oracle_camera(case):
    u, cam = case["uniforms"], case["camera"]
    ocam = oracle.make_camera(u["model_view"], u["projection"], u["camera_position"], cam.width, cam.height,
                              sh_degree=u["sh_degree"], sh_stored=case["sh_stored"], splat_scale=u["splat_scale"],
                              kernel2d=case["kernel2d"], max_splat_px=case["max_splat_px"],
                              focal_adjustment=1.0 / u["inverse_focal_adjustment"], antialiased=case["antialiased"],
                              point_cloud=bool(u["point_cloud"]))
    if u["orthographic"]:
        ocam.orthographic, ocam.ortho_zoom = 1, u["ortho_zoom"]
    if not u["fade_in_complete"]:
        ocam.fade_in, ocam.fade_start = 1, u["fade_start_radius"]
        ocam.scene_center[:] = u["scene_center"]
    if case["sh8"]:
        ocam.sh8 = 1
        oracle.set_scenes(ocam, sh8_range=[u["sh8_range"]] * max(u["scene_count"], 1), opacity=[1.0] * max(u["scene_count"], 1))
    if case["build"] == "effects1":
        oracle.set_scenes(ocam, opacity=u["scene_opacity"], visible=u["scene_visibility"], effects=True)
    if case["build"] == "dynamic2":
        oracle.set_scenes(ocam, view_matrix=u["view_matrix"], transforms=u["transforms"], camera_position=u["camera_position"],
                          dynamic=True)
    return ocam


@pytest.mark.parametrize("name", raster_cases.CASES)
def test_c_oracle_vertex_stage_matches_the_reference_shader(name):
    case = raster_cases.make_case(name)
    drawn, centre, b1, b2, colour = _from_shader(G["vs_" + name], case["uniforms"]["viewport"])
    sh = None
    if case["sh_stored"]:
        sh = case["sh_u8"].astype(np.float32) if case["sh8"] else case["sh_sampled"]
    o = oracle.project(_oracle_camera(case), case["centers"], case["cov"], case["rgba"], sh, scene_indexes=case["scene_idx"])
    vis = o["visible"] == 1
    assert drawn.sum() > 100
    np.testing.assert_array_equal(vis, drawn, err_msg="accept / reject decisions differ from the shader's")
    k = drawn
    np.testing.assert_allclose(o["cx"][k], centre[k, 0], rtol=0, atol=2e-3)
    np.testing.assert_allclose(o["cy"][k], centre[k, 1], rtol=0, atol=2e-3)
    for col, ref in (("b1x", b1[:, 0]), ("b1y", b1[:, 1]), ("b2x", b2[:, 0]), ("b2y", b2[:, 1])):
        # the shader adds the basis to an NDC centre of magnitude <= 1.2 before the harness subtracts it again: that costs
        # up to 1.2 * 2^-24 * viewport / 2 px of absolute precision
        np.testing.assert_allclose(o[col][k], ref[k], rtol=2e-4, atol=1e-4 * max(case["uniforms"]["viewport"]) / 100)
    tol = 2e-4 if name == "dynamic" else 3e-6                 # dynamic: GLSL inverse() in fp32 vs the host's fp64 inverse
    for ch, col in enumerate("rgba"):
        np.testing.assert_allclose(o[col][k], colour[k, ch], rtol=0, atol=tol)
    # vPosition of the four corners = corner * sqrt(8)
    corners = np.array([[-1, -1], [-1, 1], [1, 1], [1, -1]], np.float32) * SQRT8
    np.testing.assert_array_equal(G["vs_" + name][k][:, :, 8:10], np.broadcast_to(corners, (int(k.sum()), 4, 2)))


def test_fragment_rule_matches_the_reference_shader():
    """SplatMaterial3D.js:235-251 executed: A = dot(vPosition, vPosition); A > 8 discards; opacity = exp(-0.5 A) * vColor.a."""
    vp, vc = raster_cases.fragment_samples()
    col, disc = G["fs_color"], G["fs_discard"].astype(bool)
    A = vp[:, 0] * vp[:, 0] + vp[:, 1] * vp[:, 1]             # fp32, the operation order of dot()
    assert A.dtype == np.float32
    np.testing.assert_array_equal(disc, A > np.float32(8.0))
    keep = ~disc
    np.testing.assert_array_equal(col[keep, :3], vc[keep, :3])
    alpha = np.exp(np.float32(-0.5) * A[keep]).astype(np.float32) * vc[keep, 3]
    np.testing.assert_allclose(col[keep, 3], alpha, rtol=3e-7, atol=0)


def test_recorded_shader_builds_and_blend_state():
    meta = json.loads(bytes(G["meta"]).decode())
    assert set(meta) == set(raster_cases.shader_builds())
    for m in meta.values():                                   # SplatMaterial3D.js:65-75
        assert m["state"]["blending"] == 1 and m["state"]["transparent"] and m["state"]["depthTest"] and not m["state"]["depthWrite"]
        assert len(m["vert_sha256"]) == 64


# ------------------------------------------------------------------------------------------------ GPU tier
@pytest.fixture(scope="module")
def ctx():
    from gaussiansplats3d_amd import Context
    c = Context(0)
    yield c
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", raster_cases.CASES)
def test_hip_vertex_stage_matches_the_reference_shader(ctx, name):
    from gaussiansplats3d_amd import SplatMesh
    case = raster_cases.make_case(name)
    u, cam, sc = case["uniforms"], case["camera"], case["scene"]
    n = sc.count
    mesh = SplatMesh(ctx, n, sc.sh_degree, half_precision_covariances=case["cov_half"], antialiased=case["antialiased"],
                     kernel_2d_size=case["kernel2d"], max_screen_space_splat_size=case["max_splat_px"], splat_scale=u["splat_scale"],
                     point_cloud_mode=bool(u["point_cloud"]), spherical_harmonics_8bit=case["sh8"],
                     dynamic_mode=case["build"] == "dynamic2", enable_optional_effects=case["build"] == "effects1")
    mesh.build(sc.centers, sc.cov, sc.rgba, case["sh_u8"] if case["sh8"] else (sc.sh if sc.sh_degree else None),
               scene_indexes=case["scene_idx"])
    if case["sh8"] or case["scene_idx"] is not None:
        k = max(u["scene_count"], 1)
        mesh.set_scenes(transforms=u["transforms"] or None, camera_position=u["camera_position"],
                        opacity=u["scene_opacity"] or [1.0] * k, visible=u["scene_visibility"] or [1] * k,
                        sh8_range=[u["sh8_range"]] * k)
    if not u["fade_in_complete"]:
        mesh.set_fade_in(u["scene_center"], u["fade_start_radius"])
    mesh.set_camera(cam, focal_adjustment=1.0 / u["inverse_focal_adjustment"], spherical_harmonics_degree=u["sh_degree"])
    mesh.update_render_indexes(np.arange(n, dtype=np.uint32), n)
    mesh.render()
    recs, rects, on_screen = mesh.debug_records()
    drawn, centre, b1, b2, colour = _from_shader(G["vs_" + name], u["viewport"])
    # the engine additionally drops splats whose footprint reaches no pixel centre of the viewport: a subset of `drawn`
    assert not (on_screen & ~drawn).any(), "the engine draws a splat the reference's shader rejects"
    k = on_screen
    assert k.sum() > 80
    f = recs.view(np.float32)
    np.testing.assert_allclose(f[k, 0], centre[k, 0], rtol=0, atol=3e-3)
    np.testing.assert_allclose(f[k, 1], centre[k, 1], rtol=0, atol=3e-3)
    K = 2.4022448                                              # sqrt(4 log2 e): record = K * b / |b|^2
    for col, b in ((2, b1), (4, b2)):
        nrm = (b[k] ** 2).sum(axis=1)
        want = K * b[k] / nrm[:, None]
        err = np.abs(f[k, col:col + 2] - want).max(axis=1)     # relative to the vector's length (a component may be ~0)
        assert (err <= 5e-4 * np.sqrt((want ** 2).sum(axis=1)) + 1e-6).all(), float((err / np.sqrt((want ** 2).sum(axis=1))).max())
    tol = 2e-4 if name == "dynamic" else 2e-5                 # unorm16 storage of the record's colour
    got = np.stack([(recs[k, 6] & 0xFFFF), (recs[k, 6] >> 16), (recs[k, 7] & 0xFFFF), (recs[k, 7] >> 16)], axis=1) / 65535.0
    np.testing.assert_allclose(got, np.clip(colour[k], 0, 1), rtol=0, atol=tol)
    # every splat the shader draws and the engine drops must be one whose ellipse misses all pixel centres of the frame
    dropped = drawn & ~on_screen
    ext_x, ext_y = np.sqrt(b1[:, 0] ** 2 + b2[:, 0] ** 2), np.sqrt(b1[:, 1] ** 2 + b2[:, 1] ** 2)      # half extents of the ellipse's box
    W, H = u["viewport"]
    inside = (centre[:, 0] + ext_x > 0.5) & (centre[:, 0] - ext_x < W - 0.5) & (centre[:, 1] + ext_y > 0.5) & (centre[:, 1] - ext_y < H - 0.5)
    thin = np.minimum(ext_x, ext_y) < 0.5 + 1e-3              # its box can slip between two rows / columns of pixel centres
    assert (thin | ~inside)[dropped].all(), "a dropped splat must miss every pixel centre of the frame"
    mesh.dispose()
