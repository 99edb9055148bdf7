"""Asset readers: INRIA-v1 .ply and .ksplat -> the arrays the seams consume.  CPU only (native host code).
The PLY header logic is pinned to the reference's own PlyParserUtils.js (tests/golden/ply_header_kat.json); the rest is
checked against the independent Python restatement in oracle/asset_oracle.py."""
import json
import os
import struct

import numpy as np
import pytest

import asset_cases
from gaussiansplats3d_amd import assets, util

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ply_header_kat.json")))


def _fields(n_rest, with_uchar=False):
    names = ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"] + [f"f_rest_{k}" for k in range(n_rest)]
    names += ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
    return {nm: i for i, nm in enumerate(names)}


def _float_matrix(data, g, n_float_cols):
    body = data[g["headerSizeBytes"]:]
    rec = np.frombuffer(body, dtype=np.uint8).reshape(g["vertexCount"], g["bytesPerVertex"])
    return np.ascontiguousarray(rec[:, :4 * n_float_cols]).view(np.float32).reshape(g["vertexCount"], n_float_cols)


@pytest.mark.parametrize("name", list(asset_cases.CASES))
def test_ply_reader_against_reference_header_and_oracle(name):
    from oracle import asset_oracle
    data, cols = asset_cases.make_case(name)
    g = GOLD[name]
    n_rest = asset_cases.CASES[name]
    fields = _fields(n_rest)
    # --- the reference's own header decoding (golden) agrees with the layout the test relies on
    assert g["vertexCount"] == 64 and g["bytesPerVertex"] == 4 * len(fields) + (1 if name == "sh1_with_uchar" else 0)
    for nm, off in g["fieldOffsets"].items():
        assert off == 4 * fields[nm], nm
    rows = _float_matrix(data, g, len(fields))
    for r, raw in enumerate(g["rows"]):                                  # PlyParserUtils.readVertex of the first rows
        for nm, v in raw.items():
            assert np.float32(v) == rows[r, fields[nm]], (r, nm)
    deg = g["sphericalHarmonicsDegree"]
    # --- the oracle's f_rest -> coefficient mapping is the reference's
    cpc = g["coefficientsPerChannel"]
    d1 = ["f_rest_%d" % (i + cpc * rgb) for rgb in range(3) for i in range(3)] if deg >= 1 else []
    d2 = ["f_rest_%d" % (i + cpc * rgb + 3) for rgb in range(3) for i in range(5)] if deg >= 2 else []
    gd1 = [x for x in g["degree1Fields"]]
    gd2 = [x for x in g["degree2Fields"]]
    assert [x for x in d1] == gd1
    assert [a if b is not None else None for a, b in zip(d2, gd2)] == gd2      # unmapped fields come back as null
    # --- native reader == oracle restatement, bit for bit
    if name == "odd_27":
        fields27 = dict(fields)
        # f_rest_24..26 exist in the file but not in the reference's name->id map: they read as `undefined` -> 0
        rows = rows.copy()
        rows[:, [fields["f_rest_%d" % k] for k in (24, 25, 26)]] = 0.0
        fields = fields27
    c, s, rot, rgba, sh_file = asset_oracle.ply_rows_to_level0(fields, rows, deg)
    cov, rgba_f, sh16 = asset_oracle.fill_from_level0(c, s, rot, rgba, sh_file, deg, min_alpha=1)
    a = assets.SplatAsset(data, "ply", 2)
    assert (a.info.splat_count, a.info.sh_degree, a.info.compression_level, a.info.sh_level) == (64, deg, 0, 1)
    got = a.fill(minimum_alpha=1, want_scale_rotation=True)
    np.testing.assert_array_equal(got["centers"], c)
    np.testing.assert_array_equal(got["scales"].view(np.uint32), s.view(np.uint32))
    # (x,y,z,w) as fillSplatScaleRotationArray returns them: normalised once more and flipped to w >= 0 (pinned against the
    # reference's own code in tests/test_assets_ref.py)
    q = rot[:, [1, 2, 3, 0]].astype(np.float64)
    ln = np.sqrt(q[:, 0] * q[:, 0] + q[:, 1] * q[:, 1] + q[:, 2] * q[:, 2] + q[:, 3] * q[:, 3])
    q = np.where(ln[:, None] == 0, [[0.0, 0.0, 0.0, 1.0]], q * (1.0 / np.where(ln == 0, 1.0, ln))[:, None])
    q = q * np.where(q[:, 3:4] < 0, -1.0, 1.0)
    np.testing.assert_array_equal(got["rotations"].view(np.uint32), q.astype(np.float32).view(np.uint32))
    np.testing.assert_array_equal(got["rgba"], rgba_f)
    np.testing.assert_array_equal(got["cov"].view(np.uint32), cov.view(np.uint32))
    if deg:
        np.testing.assert_array_equal(got["sh_f16"], sh16)
    # INRIA semantics spot checks
    assert np.allclose(got["scales"], np.exp(cols["log_scales"].astype(np.float64)), rtol=1e-6)
    # the zero quaternion row: Quaternion.set(rot_0..3) maps rot_0 -> x, so normalize()'s (0,0,0,1) fallback lands in rot_3 =
    # the file's z slot: the reference turns a zero rotation into (x,y,z,w) = (0,0,1,0)
    assert (got["rotations"][3] == [0, 0, 1, 0]).all()
    half = assets.SplatAsset(data, "ply", 1 if deg else 0).fill(half_precision_covariances=True)
    np.testing.assert_array_equal(half["cov_f16"], util.to_half_three(cov.astype(np.float64)))
    a.close()


@pytest.mark.parametrize("level,sh_degree", [(0, 0), (0, 2), (1, 1), (1, 2), (2, 2), (2, 1)])
def test_ksplat_reader_against_oracle(level, sh_degree):
    from oracle import asset_oracle
    rng = np.random.default_rng(50 + 10 * level + sh_degree)
    n = 700
    centers = (rng.normal(size=(n, 3)) * 6).astype(np.float32)
    scales = np.exp(rng.normal(-4, 1, size=(n, 3))).astype(np.float32)
    rot = rng.normal(size=(n, 4)).astype(np.float32)
    rgba = rng.integers(0, 256, size=(n, 4), dtype=np.uint8)
    ncomp = {0: 0, 1: 9, 2: 24}[sh_degree]
    sh = rng.normal(0, 0.4, size=(n, ncomp)).astype(np.float32)
    data, order = assets.write_ksplat(centers, scales, rot, rgba, sh, sh_degree, level, block_size=5.0, bucket_size=64,
                                      sh_range=(-1.25, 1.5))
    exp = asset_oracle.fill_from_ksplat(data, min_alpha=20)
    a = assets.SplatAsset(data, "ksplat", 2)
    assert (a.info.splat_count, a.info.sh_degree, a.info.compression_level) == (n, sh_degree, level)
    assert a.info.sh_level == max(1, level)
    got = a.fill(minimum_alpha=20)
    np.testing.assert_array_equal(got["centers"].view(np.uint32), exp["centers"].view(np.uint32))
    np.testing.assert_array_equal(got["cov"].view(np.uint32), exp["cov"].view(np.uint32))
    np.testing.assert_array_equal(got["rgba"], exp["rgba"])
    assert (got["rgba"][:, 3][rgba[order, 3] < 20] == 0).all()
    if ncomp:
        np.testing.assert_array_equal(got["sh_u8"] if level == 2 else got["sh_f16"], exp["sh"])
    # format semantics: level 0 keeps fp32 exactly; levels 1/2 quantise centres to blockSize/2/32767
    if level == 0:
        np.testing.assert_array_equal(got["centers"], centers[order])
    else:
        assert np.abs(got["centers"].astype(np.float64) - centers[order]).max() <= 2.5 / 32767 * 0.51 + 1e-6
        assert struct.unpack_from("<I", data, 4096 + 36)[0] > 0            # the case has partially filled buckets
    # a lower requested degree truncates the SH arrays, nothing else
    if sh_degree == 2:
        low = assets.SplatAsset(data, "ksplat", 1).fill(minimum_alpha=20)
        full = got["sh_u8"] if level == 2 else got["sh_f16"]
        np.testing.assert_array_equal(low["sh_u8"] if level == 2 else low["sh_f16"], full[:, :9])
    a.close()


def test_bad_files_fail_loudly():
    from gaussiansplats3d_amd import GsError
    with pytest.raises(GsError):
        assets.SplatAsset(b"ply\nformat binary_little_endian 1.0\nelement vertex 2\nproperty float x\n", "ply")
    with pytest.raises(GsError):
        assets.SplatAsset(b"\0" * 100, "ksplat")
    data, _ = asset_cases.make_case("sh0")
    with pytest.raises(GsError):
        assets.SplatAsset(data[:-40], "ply")                               # truncated vertex data


@pytest.mark.gpu
def test_ply_file_renders_like_its_arrays():
    """End to end: a .ply staged from a synthetic scene -> native reader -> sort + draw == the same scene built from
    arrays the reference's loader would have produced (the reader's own arrays through the oracle)."""
    import helpers
    import oracle
    from gaussiansplats3d_amd import Context, SplatMesh, camera
    scene = helpers.small_scene(1500, 1, seed=77)
    rng = np.random.default_rng(3)
    n = scene.count
    log_s = rng.normal(np.log(0.05), 0.5, size=(n, 3)).astype(np.float32)
    rot = rng.normal(size=(n, 4)).astype(np.float32)
    f_dc = rng.normal(0, 1, size=(n, 3)).astype(np.float32)
    opac = rng.normal(1, 2, size=n).astype(np.float32)
    f_rest = rng.normal(0, 0.2, size=(n, 9)).astype(np.float32)
    data = assets.write_ply(scene.centers, log_s, rot, f_dc, opac, f_rest)
    arr = assets.load(data, 1)
    ctx = Context(0)
    cam = camera.demo_camera("garden", 256, 144)
    order = oracle.sort_indexes(np.arange(n, dtype=np.uint32), util.integer_centers(arr["centers"]), cam.sort_mvp())
    mesh = SplatMesh(ctx, n, 1).build(arr["centers"], arr["cov"], arr["rgba"], arr["sh_f16"])
    mesh.set_camera(cam)
    mesh.update_render_indexes(order, n)
    got, _ = mesh.render()
    ocam = oracle.make_camera(cam.model_view(), cam.projection, cam.position, 256, 144, sh_degree=1, sh_stored=1)
    fb, q, amb, frags = oracle.render(ocam, arr["centers"], arr["cov"], arr["rgba"],
                                      arr["sh_f16"].view(np.float16).astype(np.float32), order)
    assert frags > 500
    print(helpers.compare_frames(got, fb, amb, "ply end-to-end"))
    mesh.dispose()
    ctx.close()


def test_real_capture_from_data_dir_replaces_the_stand_in(tmp_path, monkeypatch):
    """$GS_DATA_DIR/garden.ply is read through the native asset reader instead of the synthetic scene."""
    from gaussiansplats3d_amd import scenes
    rng = np.random.default_rng(11)
    n = 500
    centers = rng.normal(size=(n, 3)).astype(np.float32)
    data = assets.write_ply(centers, rng.normal(-3, 0.3, (n, 3)).astype(np.float32), rng.normal(size=(n, 4)).astype(np.float32),
                            rng.normal(size=(n, 3)).astype(np.float32), rng.normal(1, 2, n).astype(np.float32),
                            rng.normal(0, 0.2, (n, 24)).astype(np.float32))
    (tmp_path / "garden.ply").write_bytes(data)
    monkeypatch.setenv("GS_DATA_DIR", str(tmp_path))
    scene = scenes.make_config_scene("C3")
    assert scene.name == "garden.ply" and scene.count == n and scene.sh_degree == 2 and scene.sh.shape == (n, 24)
    np.testing.assert_array_equal(scene.centers, centers)
    assert scenes.make_config_scene("C2").name == "C2"          # truck.ply is not there: synthetic stand-in
    monkeypatch.delenv("GS_DATA_DIR")
    assert scenes.load_real_scene("C3") is None


def _raw_ply(props, rows):
    """A binary little-endian PLY with float properties `props` and rows [n, len(props)]."""
    rows = np.ascontiguousarray(rows, dtype="<f4")
    head = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % rows.shape[0]
    head += "".join(f"property float {p}\n" for p in props) + "end_header\n"
    return head.encode() + rows.tobytes()


def test_ply_without_opacity_and_with_nan_fields_follow_js_semantics():
    """The reference's createSplat defaults opacity to 0 and clamps with Math.min / Math.max (NaN propagates) before a
    Uint8ClampedArray store (NaN -> 0): a file without `opacity` is fully transparent, a NaN colour channel is 0, not 255
    (src/loaders/ply/INRIAV1PlyParser.js:114-209)."""
    base = ["x", "y", "z", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3", "f_dc_0", "f_dc_1", "f_dc_2"]
    row = [0.1, 0.2, 0.3, -3.0, -3.0, -3.0, 1.0, 0.0, 0.0, 0.0, 5.0, -5.0, 0.0]
    a = assets.SplatAsset(_raw_ply(base, [row, row]), "ply", 0)
    got = a.fill(minimum_alpha=0)
    assert (got["rgba"][:, 3] == 0).all()                                  # no opacity property -> alpha 0
    assert got["rgba"][0, 0] == 255 and got["rgba"][0, 1] == 0             # 0.5 + SH_C0 * +-5 clamps to 255 / 0
    a.close()
    nan_row = list(row) + [4.0]
    nan_row[10] = float("nan")                                             # f_dc_0
    a = assets.SplatAsset(_raw_ply(base + ["opacity"], [nan_row, list(row) + [float("nan")]]), "ply", 0)
    got = a.fill(minimum_alpha=0)
    assert got["rgba"][0, 0] == 0 and got["rgba"][0, 3] == int(np.clip(np.float64(1 / (1 + np.exp(-4.0))) * 255, 0, 255) + 0.5) or got["rgba"][0, 0] == 0
    assert got["rgba"][1, 3] == 0                                          # NaN opacity -> sigmoid NaN -> 0
    a.close()


def test_malformed_ksplat_bucket_tables_are_rejected_not_read_out_of_bounds():
    """A level-1 .ksplat whose bucket bookkeeping does not add up (a crafted 5 KB file made gs_asset_fill
    return heap garbage).  Section header fields (KSplat format, src/loaders/SplatBuffer.js:750-830): +8 bucketSize,
    +12 bucketCount, +32 fullBucketCount, +36 partiallyFilledBucketCount."""
    from gaussiansplats3d_amd import GsError
    rng = np.random.default_rng(3)
    n = 300
    data, _ = assets.write_ksplat((rng.normal(size=(n, 3)) * 6).astype(np.float32), np.exp(rng.normal(-4, 1, size=(n, 3))).astype(np.float32),
                                  rng.normal(size=(n, 4)).astype(np.float32), rng.integers(0, 256, size=(n, 4), dtype=np.uint8),
                                  np.zeros((n, 0), np.float32), 0, 1, block_size=5.0, bucket_size=64, sh_range=(-1.5, 1.5))
    good = assets.SplatAsset(data, "ksplat", 0)
    ref = good.fill(minimum_alpha=0)
    assert np.isfinite(ref["centers"]).all()
    good.close()
    sec = 4096
    full, partial = struct.unpack_from("<II", data, sec + 32)
    assert partial > 0

    def patched(offset, fmt, *values):
        b = bytearray(data)
        struct.pack_into(fmt, b, sec + offset, *values)
        return bytes(b)

    bad_files = {
        "bucketCount 0": patched(12, "<I", 0),
        "bucketSize 0": patched(8, "<I", 0),
        "no partial buckets although the full ones do not cover the splats": patched(36, "<I", 0),
        "more buckets in use than exist": patched(32, "<II", full + 50, partial),
        "partial bucket count beyond the table": patched(36, "<I", partial + 1000),
    }
    for why, blob in bad_files.items():
        try:
            a = assets.SplatAsset(blob, "ksplat", 0)
        except GsError:
            continue
        got = a.fill(minimum_alpha=0)            # accepted: then it must at least stay inside the file
        assert np.isfinite(got["centers"]).all() and np.abs(got["centers"]).max() < 1e6, why
        a.close()
    # the partial-bucket length table itself: lengths that do not add up to the splat count
    table = sec + 1024                            # bucket metadata follows the section header
    blob = bytearray(data)
    lengths = list(struct.unpack_from(f"<{partial}I", data, table))
    assert full * 64 + sum(lengths) == n
    struct.pack_into("<I", blob, table + 4 * int(np.argmax(lengths)), 0)      # the largest partial bucket claims nothing
    with pytest.raises(GsError):
        assets.SplatAsset(bytes(blob), "ksplat", 0)
