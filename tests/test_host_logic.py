"""CPU tier: host-side mirrors (camera math, fp16 narrowing, strip partition, scene generators)."""
import numpy as np
import pytest

from gaussiansplats3d_amd import camera, dist as gdist, scenes, util


def test_perspective_matches_three_r160_formula():
    p = camera.make_perspective(50.0, 16 / 9, 0.1, 1000.0).reshape(4, 4).T
    f = 1.0 / np.tan(np.deg2rad(25.0))
    assert p[0, 0] == pytest.approx(f / (16 / 9), rel=1e-14)
    assert p[1, 1] == pytest.approx(f, rel=1e-14)
    assert p[2, 2] == pytest.approx(-(1000.1) / (999.9), rel=1e-14)
    assert p[2, 3] == pytest.approx(-2 * 1000.0 * 0.1 / 999.9, rel=1e-14)
    assert p[3, 2] == -1.0 and p[3, 3] == 0.0


def test_look_at_is_rigid_and_faces_target():
    up, pos, look = camera.DEMO_POSES["garden"]
    cam = camera.PerspectiveCamera(1920, 1080, pos, look, up)
    mw = cam.matrix_world.reshape(4, 4).T
    R = mw[:3, :3]
    np.testing.assert_allclose(R.T @ R, np.eye(3), atol=1e-14)
    assert np.linalg.det(R) == pytest.approx(1.0)
    v = cam.view.reshape(4, 4).T @ np.array([*look, 1.0])
    assert abs(v[0]) < 1e-12 and abs(v[1]) < 1e-12 and v[2] < 0          # target on the -Z axis
    fx, fy = cam.focal()
    assert fx == pytest.approx(fy)                                       # square pixels
    assert fy == pytest.approx(540.0 / np.tan(np.deg2rad(25.0)))


def test_sort_mvp_row3_is_clip_z():
    cam = camera.demo_camera("truck", 1920, 1080)
    mvp = cam.sort_mvp().reshape(4, 4).T
    p = np.array([0.3, -0.2, 0.5, 1.0])
    clip = cam.projection.reshape(4, 4).T @ (cam.view.reshape(4, 4).T @ p)
    assert (mvp @ p)[2] == pytest.approx(clip[2], rel=1e-12)


def test_to_half_three_truncates():
    x = np.array([1.0, 1.0 + 2 ** -11, 1.0 + 2 ** -10 - 2 ** -20, -3.999, 65504.0, 1e9, 6.1e-5, 6.0e-8, 1e-9], np.float32)
    h = util.to_half_three(x).view(np.float16).astype(np.float64)
    assert h[0] == 1.0 and h[1] == 1.0 and h[2] == 1.0                   # toward zero, never to nearest
    assert h[3] == -3.998046875
    assert h[4] == 65504.0 and h[5] == 65504.0                           # clamped, not inf
    assert abs(h[6]) <= 6.1e-5 and h[7] == 2.0 ** -24 and h[8] == 0.0
    mag = np.abs(x[:4].astype(np.float64))
    assert (np.abs(h[:4]) <= mag).all()


def test_integer_centers_match_oracle_rule():
    c = np.array([[0.0005, -0.0005, 1.2345], [-1.0005, 2.5, -2.5]], np.float32)
    got = util.integer_centers(c)
    assert got.dtype == np.int32 and (got[:, 3] == 1000).all()
    np.testing.assert_array_equal(got[:, :3], np.floor(c.astype(np.float64) * 1000 + 0.5).astype(np.int32))


@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
def test_balanced_strips_cover_rows_contiguously(world):
    rng = np.random.default_rng(world)
    cost = rng.integers(0, 100000, size=68).astype(np.float64)
    cost[20:30] *= 20                                                     # dense band of tile rows
    strips = gdist.balanced_row_strips(cost, world)
    assert len(strips) == world and strips[0][0] == 0 and strips[-1][1] == 68
    for (a, b), (c, d) in zip(strips[:-1], strips[1:]):
        assert b == c and a <= b
    loads = [cost[a:b].sum() for a, b in strips]
    assert max(loads) <= cost.sum() / world + cost.max() + 1


def test_strips_with_more_ranks_than_rows():
    strips = gdist.equal_row_strips(3, 8)
    assert strips[0][0] == 0 and strips[-1][1] == 3 and sum(b - a for a, b in strips) == 3


def test_scene_generators_are_deterministic_and_well_formed():
    a = scenes.make_config_scene("C3", 20000)
    b = scenes.make_config_scene("C3", 20000)
    np.testing.assert_array_equal(a.centers, b.centers)
    np.testing.assert_array_equal(a.sh.view(np.uint16), b.sh.view(np.uint16))
    assert a.sh.shape == (20000, 24) and a.sh.dtype == np.float16 and a.rgba.dtype == np.uint8
    assert (a.rgba[:, 3] >= 1).all()
    # covariances are symmetric positive definite: leading minors > 0
    c = a.cov.astype(np.float64)
    m2 = c[:, 0] * c[:, 3] - c[:, 1] ** 2
    det = (c[:, 0] * (c[:, 3] * c[:, 5] - c[:, 4] ** 2) - c[:, 1] * (c[:, 1] * c[:, 5] - c[:, 4] * c[:, 2])
           + c[:, 2] * (c[:, 1] * c[:, 4] - c[:, 3] * c[:, 2]))
    assert (c[:, 0] > 0).all() and (m2 > 0).all() and (det > 0).all()
    d = scenes.make_config_scene("C4", 5000)
    assert d.cov_half and d.sh_degree == 0
