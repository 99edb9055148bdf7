"""Host-side mirror of the reference's scene loaders, over the native readers of libgsplat_hip (csrc/assets.hip).

Reference interface: ``PlyLoader.loadFromFileData`` / ``KSplatLoader.loadFromFileData`` -> ``SplatBuffer``
(/root/reference/src/loaders/ply/PlyLoader.js, src/loaders/ksplat/KSplatLoader.js) followed by
``SplatMesh.fillSplatDataArrays`` (src/splatmesh/SplatMesh.js:1853-1902).  ``load`` returns the arrays
``SplatMesh.build`` / the sort worker take.  The two ``write_*`` helpers produce the same file formats (used by the
tests and to stage synthetic scenes as real files); they are not part of the reference's API surface.
"""
import ctypes as C
import struct

import numpy as np

from . import _lib as L
from .util import to_half_three


class SplatAsset:
    """An opened .ply / .ksplat: ``info`` + ``fill()`` -> dict of arrays."""

    def __init__(self, data, fmt=None, spherical_harmonics_degree=2):
        self.lib = L.load()
        data = bytes(data)
        if fmt is None:
            fmt = "ply" if data[:3] == b"ply" else "ksplat"
        self.handle = C.c_void_p()
        buf = (C.c_char * len(data)).from_buffer_copy(data)
        L.check(self.lib.gs_asset_open(buf, len(data), L.GS_ASSET_PLY if fmt == "ply" else L.GS_ASSET_KSPLAT,
                                       int(spherical_harmonics_degree), C.byref(self.handle)))
        self.info = L.AssetInfo()
        L.check(self.lib.gs_asset_get_info(self.handle, C.byref(self.info)))

    def fill(self, minimum_alpha=1, half_precision_covariances=False, want_scale_rotation=False):
        n, deg = self.info.splat_count, self.info.sh_degree
        ncoef = {0: 0, 1: 9, 2: 24}[deg]
        out = {"centers": np.empty((n, 3), np.float32), "rgba": np.empty((n, 4), np.uint8), "sh_degree": deg,
               "sh_level": self.info.sh_level, "sh_range": (self.info.sh_min, self.info.sh_max)}
        cov32 = None if half_precision_covariances else np.empty((n, 6), np.float32)
        cov16 = np.empty((n, 6), np.uint16) if half_precision_covariances else None
        sh16 = np.empty((n, ncoef), np.uint16) if (ncoef and self.info.sh_level == 1) else None
        sh8 = np.empty((n, ncoef), np.uint8) if (ncoef and self.info.sh_level == 2) else None
        sc = np.empty((n, 3), np.float32) if want_scale_rotation else None
        ro = np.empty((n, 4), np.float32) if want_scale_rotation else None
        p = lambda a: a.ctypes.data if a is not None else None      # noqa: E731
        L.check(self.lib.gs_asset_fill(self.handle, int(minimum_alpha), p(out["centers"]), p(cov32), p(cov16), p(out["rgba"]),
                                       p(sh16), p(sh8), p(sc), p(ro)))
        out.update(cov=cov32, cov_f16=cov16, sh_f16=sh16, sh_u8=sh8, scales=sc, rotations=ro)
        return out

    def close(self):
        if self.handle:
            self.lib.gs_asset_close(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def load(path_or_bytes, spherical_harmonics_degree=2, minimum_alpha=1, half_precision_covariances=False):
    data = open(path_or_bytes, "rb").read() if isinstance(path_or_bytes, str) else path_or_bytes
    a = SplatAsset(data, None, spherical_harmonics_degree)
    try:
        return a.fill(minimum_alpha, half_precision_covariances)
    finally:
        a.close()


# ------------------------------------------------------------------------------------------------ writers
def write_ply(centers, log_scales, rotations_wxyz, f_dc, opacity_logit, f_rest=None, extra_uchar=None):
    """INRIA-v1 layout: x y z [nx ny nz] f_dc_0..2 f_rest_* opacity scale_0..2 rot_0..3, float32 little endian.
    f_rest: [n, 3*cpc] channel-major as the INRIA trainer writes it (all R coefficients, then G, then B)."""
    n = centers.shape[0]
    names = ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"]
    cols = [centers[:, 0], centers[:, 1], centers[:, 2], np.zeros(n), np.zeros(n), np.zeros(n), f_dc[:, 0], f_dc[:, 1], f_dc[:, 2]]
    if f_rest is not None:
        for k in range(f_rest.shape[1]):
            names.append(f"f_rest_{k}")
            cols.append(f_rest[:, k])
    names += ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
    cols += [opacity_logit, log_scales[:, 0], log_scales[:, 1], log_scales[:, 2], rotations_wxyz[:, 0], rotations_wxyz[:, 1],
             rotations_wxyz[:, 2], rotations_wxyz[:, 3]]
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % n
    header += "".join(f"property float {nm}\n" for nm in names)
    if extra_uchar is not None:
        header += "property uchar pad\n"
    header += "end_header\n"
    body = np.stack([np.asarray(c, np.float32) for c in cols], axis=1)
    if extra_uchar is None:
        return header.encode() + np.ascontiguousarray(body).tobytes()
    rec = np.zeros(n, dtype=[("f", np.float32, body.shape[1]), ("u", np.uint8)])
    rec["f"], rec["u"] = body, extra_uchar
    return header.encode() + rec.tobytes()


def write_ksplat(centers, scales, rotations_wxyz, rgba, sh_rows=None, sh_degree=0, compression_level=0, block_size=5.0,
                 bucket_size=256, sh_range=(-1.5, 1.5), scene_center=(0.0, 0.0, 0.0)):
    """One-section .ksplat following SplatBuffer.generateFromUncompressedSplatArrays / writeSplatDataToSectionBuffer
    (src/loaders/SplatBuffer.js:1056-1180, 1182-1320): level 0 = fp32 rows; levels 1/2 = uint16 bucket-relative centres,
    fp16 scale / rotation, fp16 or uint8 SH, buckets of `bucket_size` splats per `block_size`^3 block (splats are
    re-ordered bucket by bucket, like the reference).  sh_rows: float [n, 9|24] in FILE order (per degree: all R, all G,
    all B coefficients).  Returns (bytes, order) with order[k] = input row stored at position k."""
    n = centers.shape[0]
    ncomp = {0: 0, 1: 9, 2: 24}[sh_degree]
    lvl = compression_level
    bps = [44, 24, 24][lvl] + [4, 2, 1][lvl] * ncomp
    c64 = centers.astype(np.float64)
    order = np.arange(n)
    buckets_meta = b""
    n_buckets = full = 0
    partial_lengths = []
    bucket_centers = np.zeros((0, 3), np.float32)
    scale_range = 32767
    if lvl >= 1:
        mn = c64.min(axis=0) if n else np.zeros(3)
        dims = (c64.max(axis=0) - mn) if n else np.zeros(3)
        yb, zb = int(np.ceil(dims[1] / block_size)), int(np.ceil(dims[2] / block_size))
        blk = np.floor((c64 - mn) / block_size).astype(np.int64)
        ids = blk[:, 0] * (yb * zb) + blk[:, 1] * zb + blk[:, 2]
        open_b, fulls, centers_of = {}, [], {}
        for i in range(n):
            b = open_b.setdefault(int(ids[i]), [])
            if not b:
                centers_of[id(b)] = blk[i] * block_size + mn + block_size / 2.0
            b.append(i)
            if len(b) >= bucket_size:
                fulls.append(b)
                del open_b[int(ids[i])]
        # for (bucketId in obj): integer-like keys enumerate in ascending numeric order
        partial = [open_b[k] for k in sorted(open_b)]
        blist = fulls + partial
        full, partial_lengths, n_buckets = len(fulls), [len(b) for b in partial], len(blist)
        order = np.array([i for b in blist for i in b], dtype=np.int64)
        bucket_centers = np.array([centers_of[id(b)] for b in blist], np.float64).reshape(-1, 3)
        bucket_of = np.repeat(np.arange(n_buckets), [len(b) for b in blist])
        buckets_meta = np.array(partial_lengths, np.uint32).tobytes() + bucket_centers.astype(np.float32).tobytes()
    rows = bytearray()
    q = rotations_wxyz.astype(np.float64)
    q = q / np.linalg.norm(q, axis=1, keepdims=True)                       # tempRot.normalize()
    sf = scale_range / (block_size * 0.5)
    for k, i in enumerate(order):
        if lvl == 0:
            rows += np.asarray(c64[i], np.float32).tobytes() + np.asarray(scales[i], np.float32).tobytes()
            rows += np.asarray(q[i], np.float32).tobytes()
        else:
            d = c64[i] - bucket_centers[bucket_of[k]]                                          # bucketCenterDelta (doubles)
            v = np.clip(np.floor(d * sf + 0.5) + scale_range, 0, scale_range * 2 + 1)       # Math.round
            rows += v.astype(np.uint16).tobytes() + to_half_three(scales[i]).tobytes() + to_half_three(q[i]).tobytes()
        rows += np.asarray(rgba[i], np.uint8).tobytes()
        if ncomp:
            s = np.asarray(sh_rows[i], np.float64)
            if lvl == 0:
                rows += s.astype(np.float32).tobytes()
            elif lvl == 1:
                rows += to_half_three(s).tobytes()
            else:
                lo, hi = sh_range
                rows += np.clip(np.floor((np.clip(s, lo, hi) - lo) / (hi - lo) * 255), 0, 255).astype(np.uint8).tobytes()
    header = bytearray(4096)
    header[0:2] = bytes([0, 1])
    struct.pack_into("<IIII", header, 4, 1, 1, n, n)
    struct.pack_into("<H", header, 20, lvl)
    struct.pack_into("<fffff", header, 24, *scene_center, sh_range[0], sh_range[1])
    sec = bytearray(1024)
    storage = len(rows) + len(buckets_meta)
    struct.pack_into("<IIII", sec, 0, n, n, bucket_size if lvl else 0, n_buckets if lvl else 0)
    struct.pack_into("<f", sec, 16, block_size if lvl else 0.0)
    struct.pack_into("<H", sec, 20, 12 if lvl else 0)
    struct.pack_into("<IIII", sec, 24, scale_range if lvl else 0, storage, full if lvl else 0, len(partial_lengths) if lvl else 0)
    struct.pack_into("<H", sec, 40, sh_degree)
    return bytes(header) + bytes(sec) + buckets_meta + bytes(rows), order
