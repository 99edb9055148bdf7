"""oracle/asset_oracle.py — TEST INFRASTRUCTURE.  Python (IEEE double, no FMA) restatement of the reference's asset path:
INRIAV1PlyParser.parseToUncompressedSplat (src/loaders/ply/INRIAV1PlyParser.js:114-209), the level-0 row written by
SplatBuffer.writeSplatDataToSectionBuffer (src/loaders/SplatBuffer.js:1056-1113), and SplatBuffer's centre / colour /
covariance / SH fill routines without a scene transform (:221-246, :440-486, :517-575, :577-734).

Pinning: the PLY HEADER logic (field offsets, bytes per vertex, SH degree and the f_rest -> coefficient mapping) is pinned to
the reference's own PlyParserUtils.js, which is three-free and runs under Node (tests/golden/ply_header_kat.json, recorded
by oracle/make_golden_ply.py).  The rest - row decode, the .ksplat layouts of compression level 0 / 1 / 2, covariance, colour
and SH fills - is PINNED too since round 2: tests/golden/assets_ref_sh{0,1,2}.npz are recorded by oracle/make_golden_assets.py,
which imports the reference's own PlyParser / SplatBuffer modules in place under Node (ESM loader hook oracle/three_loader.mjs,
'three' -> oracle/three_min.mjs) and lets the reference write and read back .ksplat files; tests/test_assets_ref.py compares
this file and csrc/assets.hip with them bit for bit.  Small cases only."""
import math
import struct

import numpy as np


def half_trunc(v):
    """THREE.DataUtils.toHalfFloat of one number -> uint16 bits."""
    f = np.float32(min(max(float(np.float32(v)), -65504.0), 65504.0)) if v == v else np.float32(v)
    bits = struct.unpack("<I", struct.pack("<f", f))[0]
    sign = (bits >> 16) & 0x8000
    e = ((bits >> 23) & 0xFF) - 127
    m = bits & 0x7FFFFF
    if e < -24: out = 0
    elif e < -14: out = (0x0400 >> (-e - 14)) + (m >> (-e - 1))
    elif e <= 15: out = ((e + 15) << 10) + (m >> 13)
    elif e < 128: out = 0x7C00
    else: out = 0x7C00 + (m >> 13)
    return out | sign


def from_half(h):
    return float(np.array([h], np.uint16).view(np.float16)[0])


def covariance(scale, rot_xyzw):
    """SplatBuffer.computeCovariance without transform: doubles in, 6 doubles out."""
    sx, sy, sz = (float(v) for v in scale)
    x, y, z, w = (float(v) for v in rot_xyzw)
    x2, y2, z2 = x + x, y + y, z + z
    xx, xy, xz, yy, yz, zz = x * x2, x * y2, x * z2, y * y2, y * z2, z * z2
    wx, wy, wz = w * x2, w * y2, w * z2
    R = [[1 - (yy + zz), xy - wz, xz + wy], [xy + wz, 1 - (xx + zz), yz - wx], [xz - wy, yz + wx, 1 - (xx + yy)]]
    S = [[sx, 0.0, 0.0], [0.0, sy, 0.0], [0.0, 0.0, sz]]
    M = [[R[r][0] * S[0][c] + R[r][1] * S[1][c] + R[r][2] * S[2][c] for c in range(3)] for r in range(3)]
    Cm = [[M[r][0] * M[c][0] + M[r][1] * M[c][1] + M[r][2] * M[c][2] for c in range(3)] for r in range(3)]
    return [Cm[0][0], Cm[0][1], Cm[0][2], Cm[1][1], Cm[1][2], Cm[2][2]]


def ply_rows_to_level0(fields, row_floats, sh_degree):
    """fields: name -> column of the float32 matrix `row_floats` (every property float).  Returns the arrays a level-0
    buffer holds: centers f32, scales f32, rot (file order = rot_0..3) f32, rgba u8, sh f32 [n, 9|24] file order."""
    n = row_floats.shape[0]
    f_rest = sorted(int(k[7:]) for k in fields if k.startswith("f_rest_"))
    cpc = len(f_rest) / 3
    centers = np.zeros((n, 3), np.float32); scales = np.zeros((n, 3), np.float32); rot = np.zeros((n, 4), np.float32)
    rgba = np.zeros((n, 4), np.uint8)
    ncomp = {0: 0, 1: 9, 2: 24}[sh_degree]
    sh = np.zeros((n, ncomp), np.float32)
    clamp = lambda v, lo, hi: max(min(v, hi), lo)      # noqa: E731
    for i in range(n):
        g = lambda name: float(row_floats[i, fields[name]])      # noqa: E731
        centers[i] = [g("x"), g("y"), g("z")]
        scales[i] = [math.exp(g(f"scale_{k}")) for k in range(3)]
        col = [clamp(math.floor((0.5 + 0.28209479177387814 * g(f"f_dc_{k}")) * 255), 0, 255) for k in range(3)]
        op = clamp(math.floor((1 / (1 + math.exp(-g("opacity")))) * 255), 0, 255)
        rgba[i] = col + [op]
        q = [g(f"rot_{k}") for k in range(4)]
        for _ in range(2):
            l = math.sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3])
            q = [0.0, 0.0, 0.0, 1.0] if l == 0 else [v * (1 / l) for v in q]
        rot[i] = q
        if sh_degree >= 1:
            d1 = [i_ + cpc * rgb for rgb in range(3) for i_ in range(3)]
            sh[i, :9] = [g("f_rest_%d" % k) for k in d1]
            if sh_degree >= 2:
                d2 = [i_ + cpc * rgb + 3 for rgb in range(3) for i_ in range(5)]
                sh[i, 9:] = [g("f_rest_%d" % k) for k in d2]
    return centers, scales, rot, rgba, sh


def fill_from_level0(centers, scales, rot_file, rgba, sh_file, sh_degree, min_alpha=1):
    """What SplatMesh.fillSplatDataArrays yields for a level-0 buffer: cov f32 [n,6], rgba, sh half bits [n, 9|24]."""
    n = centers.shape[0]
    cov = np.zeros((n, 6), np.float32)
    for i in range(n):
        w, x, y, z = (float(v) for v in rot_file[i])
        cov[i] = covariance(scales[i], (x, y, z, w))
    out_rgba = rgba.copy()
    out_rgba[:, 3] = np.where(rgba[:, 3] >= min_alpha, rgba[:, 3], 0)
    ncomp = {0: 0, 1: 9, 2: 24}[sh_degree]
    sh = np.zeros((n, ncomp), np.uint16)
    for i in range(n):
        for c in range(3 if sh_degree >= 1 else 0):
            for ch in range(3):
                sh[i, 3 * c + ch] = half_trunc(sh_file[i, c + 3 * ch])
        for c in range(5 if sh_degree >= 2 else 0):
            for ch in range(3):
                sh[i, 9 + 3 * c + ch] = half_trunc(sh_file[i, 9 + c + 5 * ch])
    return cov, out_rgba, sh


def fill_from_ksplat(data, min_alpha=1, max_sh_degree=2):
    """SplatBuffer(bufferData) + fillSplatCenterArray / fillSplatCovarianceArray / fillSplatColorArray /
    fillSphericalHarmonicsArray (no transform), single or multiple sections.  Returns dict(centers f32, cov f32, rgba u8,
    sh (uint16 half bits for level <= 1, uint8 for level 2), sh_degree, level)."""
    u8 = np.frombuffer(data, np.uint8)
    rd = lambda fmt, off: struct.unpack_from("<" + fmt, data, off)[0]      # noqa: E731
    max_sections, max_splats, level = rd("I", 4), rd("I", 12), rd("H", 20)
    sh_min = rd("f", 36) or -1.5
    sh_max = rd("f", 40) or 1.5
    base = 4096 + 1024 * max_sections
    cb, sb, rb, hb = [12, 6, 6][level], [12, 6, 6][level], [16, 8, 8][level], [4, 2, 1][level]
    centers, cov, rgba, shs = [], [], [], []
    degrees = []
    secs = []
    for s in range(max_sections):
        h = 4096 + 1024 * s
        sec = dict(n=rd("I", h + 4), bucket_size=rd("I", h + 8), bucket_count=rd("I", h + 12), block=rd("f", h + 16),
                   bstore=rd("H", h + 20), range=rd("I", h + 24) or [1, 32767, 32767][level], full=rd("I", h + 32),
                   partial=rd("I", h + 36), deg=rd("H", h + 40))
        ncomp = {0: 0, 1: 9, 2: 24}[sec["deg"]]
        sec["bps"] = cb + sb + rb + 4 + hb * ncomp
        sec["base"] = base
        sec["buckets"] = base + 4 * sec["partial"]
        sec["data"] = base + sec["bstore"] * sec["bucket_count"] + 4 * sec["partial"]
        base = sec["data"] + sec["bps"] * sec["n"]
        degrees.append(sec["deg"])
        secs.append(sec)
    deg = min(min(degrees), max_sh_degree) if degrees else 0
    ncomp_out = {0: 0, 1: 9, 2: 24}[deg]

    def val(off, idx, sh=False):
        if level == 0:
            return rd("f", off + 4 * idx)
        if level == 1 or not sh:
            return from_half(rd("H", off + 2 * idx))
        return u8[off + idx] / 255 * (sh_max - sh_min) + sh_min

    for sec in secs:
        half_block = sec["block"] / 2.0
        sf = half_block / sec["range"]
        plens = [rd("I", sec["base"] + 4 * p) for p in range(sec["partial"])]
        for j in range(sec["n"]):
            row = sec["data"] + sec["bps"] * j
            if level == 0:
                centers.append([np.float32(rd("f", row + 4 * k)) for k in range(3)])
            else:
                span = sec["full"] * sec["bucket_size"]
                if j < span:
                    b = j // sec["bucket_size"]
                else:
                    b, start = sec["full"], span
                    for ln in plens:
                        if start <= j < start + ln:
                            break
                        start += ln
                        b += 1
                bc = [rd("f", sec["buckets"] + 12 * b + 4 * k) for k in range(3)]
                centers.append([np.float32((rd("H", row + 2 * k) - sec["range"]) * sf + bc[k]) for k in range(3)])
            srow = row + cb
            sc = [val(srow, k) for k in range(3)]
            w, x, y, z = (val(srow, 3 + k) for k in range(4))
            cov.append(covariance(sc, (x, y, z, w)))
            crow = srow + sb + rb
            a = int(u8[crow + 3])
            rgba.append([int(u8[crow]), int(u8[crow + 1]), int(u8[crow + 2]), a if a >= min_alpha else 0])
            hrow = crow + 4
            out = []

            def emit(src):
                if level == 2:
                    return int(u8[hrow + src])
                if level == 1:
                    return rd("H", hrow + 2 * src)
                return half_trunc(rd("f", hrow + 4 * src))
            if deg >= 1:
                out += [emit(c + 3 * ch) for c in range(3) for ch in range(3)]
            if deg >= 2:
                out += [emit(9 + c + 5 * ch) for c in range(5) for ch in range(3)]
            shs.append(out)
    n = len(centers)
    return dict(centers=np.array(centers, np.float32).reshape(n, 3), cov=np.array(cov, np.float64).astype(np.float32).reshape(n, 6),
                rgba=np.array(rgba, np.uint8).reshape(n, 4),
                sh=np.array(shs, np.uint8 if level == 2 else np.uint16).reshape(n, ncomp_out), sh_degree=deg, level=level)
