"""CPU tier: oracle/three_min.mjs (the three.js r160 primitives the reference's code is executed with when goldens are
recorded) against independent numpy arithmetic.  Needs Node; the goldens themselves do not (they are committed)."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

from gaussiansplats3d_amd import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(shutil.which("node") is None, reason="node not installed")


@pytest.fixture(scope="module")
def result(tmp_path_factory):
    rng = np.random.default_rng(160)
    mats = []
    for _ in range(12):
        a = rng.normal(size=(4, 4)); b = rng.normal(size=(4, 4))
        a[3] = [0, 0, 0, 1] if rng.random() < 0.5 else a[3]
        mats.append(dict(a=a.T.reshape(16).tolist(), b=b.T.reshape(16).tolist(), v=rng.normal(size=3).tolist(),
                         q=rng.normal(size=4).tolist(), s=rng.uniform(0.2, 3.0, size=3).tolist()))
    floats = np.concatenate([rng.normal(size=300) * 10.0 ** rng.integers(-9, 6, 300), [0.0, -0.0, 65504.0, 65519.9, 1e9, -1e9,
                             6.1e-5, 5.9e-8, 2.0 ** -25, 1.0, -1.0, 0.333251953125]]).astype(np.float32)
    d = tmp_path_factory.mktemp("three")
    path = str(d / "in.json")
    json.dump(dict(mats=mats, floats=[float(x) for x in floats]), open(path, "w"))
    out = subprocess.check_output(["node", "--no-warnings", os.path.join(ROOT, "oracle", "three_min_check.mjs"), path], text=True)
    return mats, floats, json.loads(out)


def _m(a):
    return np.array(a, dtype=np.float64).reshape(4, 4).T          # column-major 16 -> math matrix


def test_matrix4_algebra(result):
    mats, _, r = result
    for k, c in enumerate(mats):
        A, B = _m(c["a"]), _m(c["b"])
        np.testing.assert_allclose(_m(r["invert"][k]), np.linalg.inv(A), rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(_m(r["multiply"][k]), A @ B, rtol=1e-13, atol=1e-14)
        np.testing.assert_allclose(_m(r["premultiply"][k]), B @ A, rtol=1e-13, atol=1e-14)
        np.testing.assert_allclose(r["determinant"][k], np.linalg.det(A), rtol=1e-10)
        v4 = A @ np.array([*c["v"], 1.0])
        np.testing.assert_allclose(r["applyMatrix4"][k], v4[:3] / v4[3], rtol=1e-12, atol=1e-14)
        A3, B3 = A[:3, :3], B[:3, :3]
        m3 = lambda e: np.array(e).reshape(3, 3).T                # noqa: E731
        np.testing.assert_allclose(m3(r["m3"][k]["mul"]), A3 @ B3, rtol=1e-13, atol=1e-14)
        np.testing.assert_allclose(m3(r["m3"][k]["pre"]), B3 @ A3, rtol=1e-13, atol=1e-14)
        np.testing.assert_array_equal(m3(r["m3"][k]["tr"]), A3.T)


def test_compose_decompose_and_normalisation(result):
    mats, _, r = result
    for k, c in enumerate(mats):
        q = np.array(c["q"]); q = q / np.linalg.norm(q)
        np.testing.assert_allclose(r["quatNormalize"][k], q, rtol=1e-14)
        x, y, z, w = q
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                      [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                      [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
        M = np.eye(4); M[:3, :3] = R * np.array(c["s"])[None, :]; M[:3, 3] = c["v"]
        np.testing.assert_allclose(_m(r["compose"][k]), M, rtol=1e-13, atol=1e-14)
        d = np.array(r["decompose"][k])
        np.testing.assert_allclose(d[:3], c["v"], rtol=1e-14)
        np.testing.assert_allclose(d[7:], c["s"], rtol=1e-12)
        assert min(np.abs(d[3:7] - q).max(), np.abs(d[3:7] + q).max()) < 1e-12       # q and -q are the same rotation
        v = np.array(c["v"])
        np.testing.assert_allclose(r["normalize"][k], v / np.linalg.norm(v), rtol=1e-14)
    assert r["zeroNormalize"] == [0, 0, 0] and r["zeroQuat"] == [0, 0, 0, 1]


def test_half_float_tables(result):
    _, floats, r = result
    # toHalfFloat truncates (it does not round): the host mirror used by the tests and by gsplat.js states the same rule
    expect = util.to_half_three(floats.reshape(-1, 1)).reshape(-1)
    np.testing.assert_array_equal(np.array(r["toHalf"], dtype=np.uint16), expect)
    got = np.array([float(v) for v in r["fromHalf"]], dtype=np.float64)
    ref = np.arange(65536, dtype=np.uint16).view(np.float16).astype(np.float64)
    both_nan = np.isnan(got) & np.isnan(ref)
    np.testing.assert_array_equal(got[~both_nan], ref[~both_nan])                     # fromHalfFloat is exact
    persp = _m(r["perspective"])
    assert persp[0, 0] == pytest.approx(0.5) and persp[3, 2] == -1 and persp[2, 3] == pytest.approx(-2 * 1000 * 0.1 / 999.9)
