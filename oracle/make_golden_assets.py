"""oracle/make_golden_assets.py — records what the REFERENCE's own asset code produces into tests/golden/assets_ref_*.npz.
Runs only where /root/reference exists (and Node): oracle/assets_ref.mjs imports src/loaders/SplatBuffer.js and
src/loaders/ply/INRIAV1PlyParser.js in place, with 'three' resolved to oracle/three_min.mjs.

Per case: a seeded INRIA-v1 .ply (written by gaussiansplats3d_amd.assets.write_ply) and, for each of
  ply   INRIAV1PlyParser.parseToUncompressedSplatBuffer (file order, level 0)
  gen0/gen1/gen2   parseToUncompressedSplatArray + SplatBuffer.generateFromUncompressedSplatArrays at compression level 0/1/2
the .ksplat bytes the reference wrote and the arrays its fill routines return (centres, scales, rotations, covariances as
fp32 and as half bits, RGBA with the alpha threshold, SH at the level SplatMesh would ask for).
usage: python -m oracle.make_golden_assets"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussiansplats3d_amd import assets  # noqa: E402

REF_SRC = "/root/reference/src"
CASES = [dict(name="sh2", n=420, rest=45, degree=2, seed=11), dict(name="sh1", n=160, rest=9, degree=1, seed=12),
         dict(name="sh0", n=150, rest=0, degree=0, seed=13)]


def make_ply(case):
    rng = np.random.default_rng(case["seed"])
    n = case["n"]
    k = int(0.62 * n)                                        # a tight cluster inside one 5-unit block: a full 256-splat bucket
    centers = np.concatenate([rng.normal(0.0, 0.25, (k, 3)) + [2.4, 2.6, -2.5], rng.uniform(-12, 12, (n - k, 3))]).astype(np.float32)
    log_s = rng.normal(-3.5, 0.9, (n, 3)).astype(np.float32)
    rot = rng.normal(size=(n, 4)).astype(np.float32)
    f_dc = rng.normal(0.0, 1.2, (n, 3)).astype(np.float32)
    opac = rng.normal(0.5, 2.5, n).astype(np.float32)
    f_rest = rng.normal(0.0, 0.35, (n, case["rest"])).astype(np.float32) if case["rest"] else None
    # edge rows: zero quaternion, negative w, saturating opacity / colour, a huge and a tiny scale, an exact block border
    rot[0] = 0.0
    rot[1] = [-0.5, 0.5, -0.5, 0.5]
    opac[2], opac[3] = 30.0, -30.0
    f_dc[4], f_dc[5] = 9.0, -9.0
    log_s[6], log_s[7] = 3.0, -12.0
    centers[8] = [5.0, -5.0, 0.0]
    return assets.write_ply(centers, log_s, rot, f_dc, opac, f_rest)


def main():
    assert os.path.isdir(REF_SRC), "reference not present"
    for case in CASES:
        ply = make_ply(case)
        with tempfile.TemporaryDirectory() as d:
            open(os.path.join(d, "in.ply"), "wb").write(ply)
            subprocess.check_call(["node", "--no-warnings", "--experimental-loader", os.path.join(ROOT, "oracle", "three_loader.mjs"),
                                   os.path.join(ROOT, "oracle", "assets_ref.mjs"), REF_SRC, os.path.join(d, "in.ply"), d,
                                   str(case["degree"]), "1"], cwd=os.path.join(ROOT, "oracle"))
            man = json.load(open(os.path.join(d, "manifest.json")))
            out = {"ply_bytes": np.frombuffer(ply, np.uint8), "manifest": np.frombuffer(json.dumps(man).encode(), np.uint8),
                   "rows": np.fromfile(os.path.join(d, "rows.f64"), np.float64).reshape(-1, man["rowLength"])}
            for tag, b in man["buffers"].items():
                n, nc = b["splatCount"], b["ncoef"]
                rd = lambda ext, dt: np.fromfile(os.path.join(d, f"{tag}_{ext}"), dt)      # noqa: E731
                out[f"{tag}_ksplat"] = np.fromfile(os.path.join(d, f"{tag}.ksplat"), np.uint8)
                out[f"{tag}_centers"] = rd("centers.f32", np.float32).reshape(n, 3)
                out[f"{tag}_scales"] = rd("scales.f32", np.float32).reshape(n, 3)
                out[f"{tag}_rotations"] = rd("rotations.f32", np.float32).reshape(n, 4)
                out[f"{tag}_cov32"] = rd("cov.f32", np.float32).reshape(n, 6)
                out[f"{tag}_cov16"] = rd("cov.u16", np.uint16).reshape(n, 6)
                out[f"{tag}_rgba"] = rd("rgba.u8", np.uint8).reshape(n, 4)
                if nc:
                    out[f"{tag}_sh"] = (rd("sh.u8", np.uint8) if b["shLevel"] == 2 else rd("sh.u16", np.uint16)).reshape(n, nc)
        path = os.path.join(ROOT, "tests", "golden", f"assets_ref_{case['name']}.npz")
        np.savez_compressed(path, **out)
        print(case["name"], {t: (b["splatCount"], b["compressionLevel"], b["shValuesChangedByIdentityTransform"])
                             for t, b in man["buffers"].items()}, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
