"""Host-side numeric helpers that mirror what the reference's JS layer does before data reaches the GPU."""
import numpy as np


def to_half_three(values):
    """THREE.DataUtils.toHalfFloat (three r160; used by /root/reference/src/loaders/SplatBuffer.js:9,469-474):
    clamp to +-65504, then TRUNCATE the fp32 mantissa via the base/shift tables (no rounding).
    Returns uint16 half bit patterns."""
    v = np.clip(np.asarray(values, dtype=np.float32), -65504.0, 65504.0)
    f = v.view(np.uint32)
    sign = (f >> 16) & 0x8000
    e = ((f >> 23) & 0xFF).astype(np.int32) - 127
    m = f & 0x007FFFFF
    out = np.zeros(f.shape, dtype=np.uint32)
    # e < -24 -> signed zero ; -24 <= e < -14 -> subnormal ; -14 <= e <= 15 -> normal ; e == 128 -> inf/nan
    sub = (e >= -24) & (e < -14)
    sh = np.where(sub, -e - 1, 13).astype(np.uint32)
    out = np.where(sub, (0x0400 >> np.clip(-e - 14, 0, 31).astype(np.uint32)) + (m >> sh), out)
    nor = (e >= -14) & (e <= 15)
    out = np.where(nor, (((e + 15).astype(np.uint32)) << 10) + (m >> 13), out)
    big = (e > 15) & (e < 128)
    out = np.where(big, 0x7C00, out)
    nan = e == 128
    out = np.where(nan, 0x7C00 + (m >> 13), out)
    return (out | sign).astype(np.uint16)


def integer_centers(centers3):
    """SplatMesh.getIntegerCenters(padFour=true), /root/reference/src/splatmesh/SplatMesh.js:1912-1926:
    Math.round(fp32 * 1000.0) in fp64 (round half up), w = 1000."""
    c = np.ascontiguousarray(centers3, dtype=np.float32).reshape(-1, 3)
    out = np.empty((c.shape[0], 4), dtype=np.int32)
    out[:, :3] = np.floor(c.astype(np.float64) * 1000.0 + 0.5).astype(np.int32)
    out[:, 3] = 1000
    return out


def float_centers(centers3):
    """SplatMesh.getFloatCenters(padFour=true), SplatMesh.js:1935-1948: w = 1.0."""
    c = np.ascontiguousarray(centers3, dtype=np.float32).reshape(-1, 3)
    return np.concatenate([c, np.ones((c.shape[0], 1), np.float32)], axis=1)
