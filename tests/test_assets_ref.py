"""CPU tier: the native asset readers (csrc/assets.hip) against goldens recorded by running the REFERENCE's own loader code
under Node (oracle/make_golden_assets.py -> oracle/assets_ref.mjs: src/loaders/ply/INRIAV1PlyParser.js and
src/loaders/SplatBuffer.js imported in place, 'three' resolved to the r160 restatement oracle/three_min.mjs).

Pinned here, bit for bit: PLY row -> splat (exp scale, sigmoid / SH-DC colour with floor + clamp, quaternion normalised
twice, f_rest re-ordering), the .ksplat files the reference's own writer produces at compression levels 0 / 1 / 2 (bucket
tables with full and partial buckets, uint16 bucket-relative centres, half scales / rotations, 8-bit SH) and what its
fill routines return for them: centres, scales, rotations, covariances (fp32 and THREE.DataUtils.toHalfFloat bits), RGBA
with the alpha threshold, SH as half bits / uint8."""
import json
import os

import numpy as np
import pytest

from gaussiansplats3d_amd import assets

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["sh2", "sh1", "sh0"]


def _load(case):
    g = np.load(os.path.join(GOLDEN, f"assets_ref_{case}.npz"))
    man = json.loads(bytes(g["manifest"]).decode())
    return g, man


def _check(a, g, tag, b):
    got = a.fill(minimum_alpha=1, half_precision_covariances=False, want_scale_rotation=True)
    got16 = a.fill(minimum_alpha=1, half_precision_covariances=True)
    assert a.info.splat_count == b["splatCount"] and a.info.sh_degree == b["shDegree"]
    assert a.info.compression_level == b["compressionLevel"] and a.info.sh_level == b["shLevel"]
    eq = lambda x, y, what: np.testing.assert_array_equal(x.view(np.uint32) if x.dtype == np.float32 else x,     # noqa: E731
                                                          y.view(np.uint32) if y.dtype == np.float32 else y, err_msg=f"{tag} {what}")
    eq(got["centers"], g[f"{tag}_centers"], "centres")
    eq(got["scales"], g[f"{tag}_scales"], "scales")
    eq(got["rotations"], g[f"{tag}_rotations"], "rotations")
    eq(got["rgba"], g[f"{tag}_rgba"], "rgba")
    eq(got["cov"], g[f"{tag}_cov32"], "covariances fp32")
    eq(got16["cov_f16"], g[f"{tag}_cov16"], "covariances half")
    if b["ncoef"]:
        sh = got["sh_u8"] if b["shLevel"] == 2 else got["sh_f16"]
        eq(sh, g[f"{tag}_sh"], "spherical harmonics")
        if b["shLevel"] == 2:
            assert abs(a.info.sh_min - b["minSh"]) < 1e-6 and abs(a.info.sh_max - b["maxSh"]) < 1e-6
    np.testing.assert_allclose(np.array(a.info.scene_center[:]), b["sceneCenter"], atol=0)


@pytest.mark.parametrize("case", CASES)
def test_ply_reader_matches_the_reference_parser(case):
    """INRIAV1PlyParser.parseToUncompressedSplatBuffer -> SplatBuffer fill routines, file order."""
    g, man = _load(case)
    a = assets.SplatAsset(bytes(g["ply_bytes"]), "ply", man["shDegree"])
    _check(a, g, "ply", man["buffers"]["ply"])
    a.close()


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("level", [0, 1, 2])
def test_ksplat_reader_matches_the_reference_on_files_the_reference_wrote(case, level):
    g, man = _load(case)
    tag = f"gen{level}"
    a = assets.SplatAsset(bytes(g[f"{tag}_ksplat"]), "ksplat", man["shDegree"])
    _check(a, g, tag, man["buffers"][tag])
    a.close()


def test_reference_rows_match_the_python_restatement():
    """The independent Python restatement (oracle/asset_oracle.py) agrees with the reference's parsed rows too."""
    g, man = _load("sh2")
    rows = g["rows"]
    assert rows.shape[0] == 420 and np.isfinite(rows[:, :3]).all()
    a = assets.SplatAsset(bytes(g["ply_bytes"]), "ply", 2)
    got = a.fill(minimum_alpha=0, want_scale_rotation=True)
    np.testing.assert_array_equal(got["centers"], rows[:, 0:3].astype(np.float32))
    np.testing.assert_array_equal(got["scales"], rows[:, 3:6].astype(np.float32))
    np.testing.assert_array_equal(got["rgba"], rows[:, 10:14].astype(np.uint8))
    a.close()
