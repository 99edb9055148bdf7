"""Pins the sort oracle (oracle/sort_oracle.c) to the reference: the golden vectors in tests/golden were
produced by the reference's own WASM and native builds (oracle/make_golden.py)."""
import json
import os

import numpy as np
import pytest

import kat_cases
import oracle

GOLD = os.path.join(os.path.dirname(__file__), "golden")
META = json.load(open(os.path.join(GOLD, "sort_kat.json")))
SMALL = np.load(os.path.join(GOLD, "sort_kat_small.npz"))


def _kw(args):
    return {k: args[k] for k in ("sort_count", "render_count", "precision", "use_int", "dynamic", "precomputed",
                                 "scene_indexes", "transforms")}


@pytest.mark.parametrize("case", kat_cases.CASES, ids=[c["name"] for c in kat_cases.CASES])
def test_c_oracle_matches_reference_goldens(case):
    args = kat_cases.make_case(case)
    meta = META[case["name"]]
    assert kat_cases.input_digest(args) == meta["inputs"], "seeded input generator drifted from the recorded one"
    out = oracle.sort_indexes(args["indexes"], args["centers4"], args["mvp"], **_kw(args))
    assert kat_cases.digest(out) == meta["output"]
    if case["name"] in SMALL.files:
        np.testing.assert_array_equal(out, SMALL[case["name"]])


def test_degenerate_equal_keys_is_reversed_input():
    # hi == lo: WASM wraps the NaN bucket onto bucket 0 -> reversed input
    c4 = np.tile(np.array([[1, 2, 3, 1000]], np.int32), (8, 1))
    out = oracle.sort_indexes(np.arange(8, dtype=np.uint32), c4, np.arange(16.0))
    np.testing.assert_array_equal(out, SMALL["all_equal"])
    one = oracle.sort_indexes(np.zeros(1, np.uint32), c4[:1], np.arange(16.0))
    np.testing.assert_array_equal(one, SMALL["single"])


@pytest.mark.parametrize("case", [c for c in kat_cases.CASES if c["mode"] == "int" and not c.get("dynamic")
                                  and not c.get("precomputed")], ids=lambda c: c["name"])
def test_numpy_restatement_agrees(case):
    args = kat_cases.make_case(case)
    a = oracle.sort_indexes(args["indexes"], args["centers4"], args["mvp"], **_kw(args))
    b = oracle.sort_indexes_numpy(args["indexes"], args["centers4"], args["mvp"], args["sort_count"],
                                  args["render_count"], args["precision"])
    np.testing.assert_array_equal(a, b)


@pytest.mark.skipif(not oracle.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_c_oracle_matches_compiled_reference_random(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1000, 60000))
    c = rng.normal(size=(n, 3)).astype(np.float32) * 5
    ci = oracle.integer_centers(c)
    idx = rng.permutation(n).astype(np.uint32)
    render = int(rng.integers(n // 2, n + 1))
    sort = int(rng.integers(2, render + 1))
    mvp = rng.normal(size=16)
    prec = int(rng.integers(10, 21))
    a = oracle.sort_indexes(idx[:render], ci, mvp, sort, render, prec)
    b = oracle.ref_sort_indexes(idx[:render], ci, mvp, sort, render, prec)
    np.testing.assert_array_equal(a, b)


def test_integer_centers_round_half_up():
    c = np.array([[0.0005, -0.0005, 1.2345], [-1.0005, 2.5, -2.5]], np.float32)
    got = oracle.integer_centers(c)
    exp = np.floor(c.astype(np.float64) * 1000.0 + 0.5).astype(np.int32)
    np.testing.assert_array_equal(got[:, :3], exp)
    assert (got[:, 3] == 1000).all()


def test_sort_is_a_permutation_and_far_to_near():
    rng = np.random.default_rng(5)
    n = 20000
    ci = oracle.integer_centers(rng.uniform(-10, 10, (n, 3)).astype(np.float32))
    mvp = rng.normal(size=16)
    out, keys, buckets, (lo, hi), st = oracle.sort_indexes(np.arange(n, dtype=np.uint32), ci, mvp,
                                                          return_intermediates=True)
    assert st == 0
    assert np.array_equal(np.sort(out), np.arange(n))
    b_sorted = buckets[out]                          # identity index list: position == splat id
    assert (np.diff(b_sorted.astype(np.int64)) <= 0).all()


@pytest.mark.skipif(not oracle.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
def test_precision_24_overflow_is_a_reference_defect_we_flag():
    """At 24 bits (float mode's upper clamp, src/Viewer.js:208-210) fp32 rounding can map the farthest
    splat to bucket == range; the reference then indexes a counter the prefix sum never covered and
    emits a non-permutation.  We clamp to range-1 and report GSO_CLAMPED instead."""
    case = dict(name="float_p24", n=30000, render=30000, sort=30000, precision=24, mode="float")
    args = kat_cases.make_case(case)
    kw = _kw(args)
    ref = oracle.ref_sort_indexes(args["indexes"], args["centers4"], args["mvp"], **kw)
    out, _, buckets, _, status = oracle.sort_indexes(args["indexes"], args["centers4"], args["mvp"],
                                                     return_intermediates=True, **kw)
    assert np.array_equal(np.sort(out), np.arange(30000))
    if not np.array_equal(np.sort(ref), np.arange(30000)):
        assert status == 1
        assert (ref != out).sum() <= 4
    else:
        np.testing.assert_array_equal(ref, out)
