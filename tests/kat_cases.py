"""Seeded known-answer cases for the sort seam, shared by oracle/make_golden.py (which records the
REFERENCE's answers) and the tests (which replay them).  Each case yields exactly the buffers
sortIndexes takes (/root/reference/src/worker/sorter.cpp:17-22)."""
import hashlib

import numpy as np


def _mvp(rng, scale=1.0):
    """A general (non-axis-aligned) fp64 column-major matrix; only row 3 matters to the sorter."""
    m = rng.normal(size=(4, 4)) * scale
    return np.ascontiguousarray(m.T).reshape(16)


def _int_centers(c):
    c = np.asarray(c, dtype=np.float32)
    out = np.empty((c.shape[0], 4), dtype=np.int32)
    out[:, :3] = np.floor(c.astype(np.float64) * 1000.0 + 0.5).astype(np.int32)   # Math.round
    out[:, 3] = 1000
    return out


CASES = [
    # name, n, render, sort, precision, mode, extra
    dict(name="two", n=2, render=2, sort=2, precision=16, mode="int"),
    dict(name="small", n=1000, render=1000, sort=1000, precision=16, mode="int"),
    dict(name="p10", n=5000, render=5000, sort=5000, precision=10, mode="int"),
    dict(name="p20", n=5000, render=5000, sort=5000, precision=20, mode="int"),
    dict(name="range_plus_1", n=65537, render=65537, sort=65537, precision=16, mode="int"),
    dict(name="permuted_partial", n=200000, render=150000, sort=50000, precision=16, mode="int", permute=True),
    dict(name="partial_thirds", n=30000, render=30000, sort=10000, precision=16, mode="int", permute=True),
    dict(name="duplicates", n=40000, render=40000, sort=40000, precision=16, mode="int", grid=True),
    dict(name="wraparound", n=20000, render=20000, sort=20000, precision=16, mode="int", huge=True),
    dict(name="negative_far", n=20000, render=20000, sort=20000, precision=12, mode="int", offset=-500.0),
    dict(name="million", n=1000000, render=1000000, sort=1000000, precision=16, mode="int"),
    dict(name="dynamic_int", n=30000, render=30000, sort=30000, precision=16, mode="int", dynamic=True),
    dict(name="float_static", n=30000, render=30000, sort=30000, precision=16, mode="float"),
    dict(name="float_p22", n=30000, render=30000, sort=30000, precision=22, mode="float"),
    dict(name="float_dynamic", n=30000, render=25000, sort=20000, precision=16, mode="float", dynamic=True,
         permute=True),
    dict(name="pre_int", n=30000, render=30000, sort=30000, precision=16, mode="int", precomputed=True),
    dict(name="pre_float", n=30000, render=30000, sort=30000, precision=16, mode="float", precomputed=True),
]


def make_case(case, seed=1234):
    """-> dict(indexes, centers4, mvp, sort_count, render_count, precision, use_int, dynamic,
    precomputed, scene_indexes, transforms)"""
    name = case["name"]
    rng = np.random.default_rng([seed, int(hashlib.sha256(name.encode()).hexdigest()[:8], 16)])
    n = case["n"]
    if case.get("grid"):
        c = rng.integers(-3, 4, size=(n, 3)).astype(np.float32) * 0.5          # many exactly equal keys
    elif case.get("huge"):
        # every product c*m exceeds int32 and wraps, but the keys stay clustered (spread << 2^31)
        c = (np.float32(1500.0) + rng.uniform(-0.5, 0.5, size=(n, 3))).astype(np.float32)
    else:
        c = rng.uniform(-10.0, 10.0, size=(n, 3)).astype(np.float32) + np.float32(case.get("offset", 0.0))
    use_int = case["mode"] == "int"
    if use_int:
        centers4 = _int_centers(c)
    else:
        centers4 = np.concatenate([c, np.ones((n, 1), np.float32)], axis=1)
    mvp = _mvp(rng, 3.0 if case.get("huge") else 1.0)
    if case.get("huge"):
        mvp[[2, 6, 10]] = [2.9, -3.3, 3.7]
    idx = rng.permutation(n).astype(np.uint32) if case.get("permute") else np.arange(n, dtype=np.uint32)
    idx = idx[:case["render"]].copy()
    out = dict(indexes=idx, centers4=centers4, mvp=mvp, sort_count=case["sort"], render_count=case["render"],
               precision=case["precision"], use_int=use_int, dynamic=bool(case.get("dynamic")),
               precomputed=None, scene_indexes=None, transforms=None)
    if case.get("dynamic"):
        out["scene_indexes"] = np.sort(rng.integers(0, 5, size=n)).astype(np.uint32)
        t = np.tile(np.eye(4).reshape(16), (32, 1))
        for s in range(5):
            m = np.eye(4)
            m[:3, :3] += rng.normal(size=(3, 3)) * 0.2
            m[:3, 3] = rng.normal(size=3)
            t[s] = m.T.reshape(16)
        out["transforms"] = t.astype(np.float32)
    if case.get("precomputed"):
        if use_int:
            out["precomputed"] = rng.integers(-2_000_000, 2_000_000, size=n).astype(np.int32)
        else:
            out["precomputed"] = rng.uniform(-30.0, 30.0, size=n).astype(np.float32)
    return out


def digest(arr):
    return hashlib.sha256(np.ascontiguousarray(arr).tobytes()).hexdigest()


def input_digest(args):
    h = hashlib.sha256()
    for k in ("indexes", "centers4", "precomputed", "scene_indexes", "transforms"):
        if args[k] is not None:
            h.update(np.ascontiguousarray(args[k]).tobytes())
    h.update(np.asarray(args["mvp"], dtype=np.float64).astype(np.float32).tobytes())
    return h.hexdigest()
