"""Seed-fuzz of the ASSET READERS (host code: runs without a GPU).  Per iteration either
  * a random INRIA-v1 .ply (1..4000 splats; 0 / 9 / 24 / 45 / an odd number of f_rest fields; an optional trailing uchar property;
    zero quaternions, extreme opacities and scales sprinkled in), read at a random requested SH degree and minimum alpha, fp32 or
    fp16 covariances, compared bit for bit with oracle/asset_oracle.py (the Python restatement pinned to the reference's parser), or
  * a random .ksplat written by assets.write_ksplat (compression level 0 / 1 / 2, SH degree 0..2, random block and bucket sizes and
    SH range), compared bit for bit with asset_oracle.fill_from_ksplat;
and then a MUTATION pass over the same file: random bytes of the header (and, for .ksplat, of the section and bucket tables)
overwritten, the file truncated at a random point - the reader must either fail with GsError or return arrays of the size it
announced; it must never read outside the buffer (a crash of this process is the failure signal; run under `timeout`).
The oracle is the checker here, as in tests/.

usage: python tests/tools/soak_assets.py [iterations=300] [first_seed=100] """
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

from oracle import asset_oracle
from gaussiansplats3d_amd import GsError, assets, util

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 100
failures = mutations = rejected = 0
t_start = time.perf_counter()


def ply_fields(n_rest):
    names = ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"] + [f"f_rest_{k}" for k in range(n_rest)]
    names += ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
    return {nm: i for i, nm in enumerate(names)}


def mutate(rng, data, fmt, table_bytes):
    """a damaged copy: a few bytes of the first `table_bytes` overwritten and / or the tail cut off"""
    b = bytearray(data)
    kind = int(rng.integers(0, 3))
    if kind != 1:
        for _ in range(int(rng.integers(1, 6))):
            b[int(rng.integers(0, min(table_bytes, len(b))))] = int(rng.integers(0, 256))
    if kind != 0:
        b = b[:int(rng.integers(0, len(b)))]
    return bytes(b)


for it in range(iters):
    seed = seed0 + it
    rng = np.random.default_rng(seed)
    n = int(np.exp(rng.uniform(0.0, np.log(4000.0))))
    fmt = "ply" if rng.integers(0, 2) else "ksplat"
    try:
        if fmt == "ply":
            n_rest = int(rng.choice([0, 9, 24, 45, 27, 12]))
            cols = dict(centers=(rng.normal(size=(n, 3)) * 5).astype(np.float32), log_scales=rng.normal(-4, 1.5, size=(n, 3)).astype(np.float32),
                        rotations=rng.normal(size=(n, 4)).astype(np.float32), f_dc=rng.normal(0, 1.5, size=(n, 3)).astype(np.float32),
                        opacity=rng.normal(0, 4, size=n).astype(np.float32),
                        f_rest=rng.normal(0, 0.3, size=(n, n_rest)).astype(np.float32) if n_rest else None)
            cols["rotations"][rng.integers(0, n)] = 0.0
            cols["opacity"][rng.integers(0, n)] = float(rng.choice([-60.0, 60.0]))
            extra = rng.integers(0, 256, n).astype(np.uint8) if rng.integers(0, 4) == 0 else None
            data = assets.write_ply(cols["centers"], cols["log_scales"], cols["rotations"], cols["f_dc"], cols["opacity"], cols["f_rest"], extra)
            want = int(rng.integers(0, 3))
            min_alpha = int(rng.choice([0, 1, 1, 20, 200]))
            label = f"seed {seed}: ply n={n} f_rest={n_rest} uchar={extra is not None} want_sh={want} min_alpha={min_alpha}"
            a = assets.SplatAsset(data, "ply", want)
            deg = int(a.info.sh_degree)
            fields = ply_fields(n_rest)
            header = data.index(b"end_header\n") + len(b"end_header\n")
            stride = 4 * len(fields) + (1 if extra is not None else 0)
            rec = np.frombuffer(data[header:], dtype=np.uint8).reshape(n, stride)
            rows = np.ascontiguousarray(rec[:, :4 * len(fields)]).view(np.float32).reshape(n, len(fields)).copy()
            # the reference reads f_rest_0 .. f_rest_{k-1}, k = 45 / 24 / 9 / 0 = the largest of those the file holds
            # (INRIAV1PlyParser.decodeHeaderLines); a coefficient index beyond that names a field that was never read: undefined -> 0
            read = 45 if n_rest >= 45 else 24 if n_rest >= 24 else 9 if n_rest >= 9 else 0
            for k in range(read, n_rest):
                rows[:, fields[f"f_rest_{k}"]] = 0.0
            assert a.info.splat_count == n, "splat count"
            c, s, rot, rgba, sh_file = asset_oracle.ply_rows_to_level0(fields, rows, deg)
            cov, rgba_f, sh16 = asset_oracle.fill_from_level0(c, s, rot, rgba, sh_file, deg, min_alpha=min_alpha)
            got = a.fill(minimum_alpha=min_alpha)
            assert np.array_equal(got["centers"], c), "centres"
            assert np.array_equal(got["rgba"], rgba_f), "colours"
            assert np.array_equal(got["cov"].view(np.uint32), cov.view(np.uint32)), "covariances"
            if deg:
                assert np.array_equal(got["sh_f16"], sh16), "spherical harmonics"
            half = assets.SplatAsset(data, "ply", want)
            assert np.array_equal(half.fill(minimum_alpha=min_alpha, half_precision_covariances=True)["cov_f16"], util.to_half_three(cov.astype(np.float64))), "fp16 covariances"
            half.close(); a.close()
            table = header
        else:
            level = int(rng.integers(0, 3))
            sh_degree = int(rng.integers(0, 3))
            ncomp = {0: 0, 1: 9, 2: 24}[sh_degree]
            centers = (rng.normal(size=(n, 3)) * float(rng.uniform(0.5, 20.0))).astype(np.float32)
            scales = np.exp(rng.normal(-4, 1.5, size=(n, 3))).astype(np.float32)
            rot = rng.normal(size=(n, 4)).astype(np.float32)
            rgba = rng.integers(0, 256, size=(n, 4), dtype=np.uint8)
            sh = rng.normal(0, 0.5, size=(n, ncomp)).astype(np.float32)
            block, bucket = float(rng.choice([1.0, 5.0, 12.5])), int(rng.choice([16, 64, 256]))
            lo = float(rng.uniform(-2.0, -0.5)); hi = float(rng.uniform(0.5, 2.0))
            min_alpha = int(rng.choice([0, 1, 20, 200]))
            want = int(rng.integers(0, 3))
            label = f"seed {seed}: ksplat n={n} level={level} sh{sh_degree} block={block} bucket={bucket} range=({lo:.2f},{hi:.2f}) want_sh={want} min_alpha={min_alpha}"
            data, order = assets.write_ksplat(centers, scales, rot, rgba, sh, sh_degree, level, block_size=block, bucket_size=bucket, sh_range=(lo, hi))
            exp = asset_oracle.fill_from_ksplat(data, min_alpha=min_alpha, max_sh_degree=want)
            a = assets.SplatAsset(data, "ksplat", want)
            assert (a.info.splat_count, a.info.compression_level) == (n, level), "header"
            got = a.fill(minimum_alpha=min_alpha)
            assert np.array_equal(got["centers"].view(np.uint32), exp["centers"].view(np.uint32)), "centres"
            assert np.array_equal(got["cov"].view(np.uint32), exp["cov"].view(np.uint32)), "covariances"
            assert np.array_equal(got["rgba"], exp["rgba"]), "colours"
            if min(want, sh_degree):
                assert np.array_equal(got["sh_u8"] if level == 2 else got["sh_f16"], exp["sh"]), "spherical harmonics"
            a.close()
            table = 4096 + 1024 + 4096
        # ---- mutation pass: damaged copies must be rejected or read within bounds
        for _ in range(6):
            bad = mutate(rng, data, fmt, table)
            mutations += 1
            try:
                m = assets.SplatAsset(bad, fmt, 2)
                out = m.fill(minimum_alpha=1)
                assert out["centers"].shape[0] == m.info.splat_count
                m.close()
            except GsError:
                rejected += 1
        print(f"ok   {label}", flush=True)
    except Exception as e:
        failures += 1
        print(f"FAIL {label}: {type(e).__name__}: {str(e)[:300]}", flush=True)
print(f"soak_assets: {iters} iterations from seed {seed0}, {failures} failures; {mutations} damaged files, {rejected} rejected with GsError, "
      f"the rest read within bounds; {time.perf_counter() - t_start:.0f} s")
sys.exit(1 if failures else 0)
