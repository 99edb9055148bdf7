"""-m gpu: the HIP sort seam against the reference's goldens and the pinned oracle, through the C ABI."""
import json
import os

import numpy as np
import pytest

import kat_cases
import oracle
from gaussiansplats3d_amd import Context, camera, create_sort_worker, util

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def ctx():
    c = Context(0)
    yield c
    c.close()


def run_worker(ctx, args):
    n = args["centers4"].shape[0]
    w = create_sort_worker(ctx, n, True, True, args["use_int"], args["dynamic"], args["precision"])
    w.post_message({"centers": args["centers4"], "sceneIndexes": args["scene_indexes"],
                    "range": {"from": 0, "to": n - 1, "count": n}})
    msg = {"modelViewProj": args["mvp"], "splatRenderCount": args["render_count"],
           "splatSortCount": args["sort_count"], "usePrecomputedDistances": args["precomputed"] is not None,
           "indexesToSort": args["indexes"], "transforms": args["transforms"],
           "precomputedDistances": args["precomputed"]}
    reply = w.post_message({"sort": msg})
    return w, reply


@pytest.mark.parametrize("case", kat_cases.CASES, ids=[c["name"] for c in kat_cases.CASES])
def test_reference_goldens_bit_exact(ctx, case):
    meta = json.load(open(os.path.join(GOLD, "sort_kat.json")))[case["name"]]
    args = kat_cases.make_case(case)
    assert kat_cases.input_digest(args) == meta["inputs"]
    w, reply = run_worker(ctx, args)
    assert reply["sortDone"] and reply["status"] == 0
    assert kat_cases.digest(reply["sortedIndexes"]) == meta["output"]
    w.terminate()


@pytest.mark.parametrize("name", ["small", "permuted_partial", "wraparound", "p20"])
def test_intermediate_keys_and_buckets_match_oracle(ctx, name):
    case = [c for c in kat_cases.CASES if c["name"] == name][0]
    args = kat_cases.make_case(case)
    w, reply = run_worker(ctx, args)
    kw = {k: args[k] for k in ("sort_count", "render_count", "precision", "use_int", "dynamic")}
    out, keys, buckets, (lo, hi), st = oracle.sort_indexes(args["indexes"], args["centers4"], args["mvp"],
                                                          return_intermediates=True, **kw)
    s0 = args["render_count"] - args["sort_count"]
    np.testing.assert_array_equal(w.debug_read(0, args["render_count"])[s0:], keys[s0:])
    np.testing.assert_array_equal(w.debug_read(1, args["render_count"])[s0:], buckets[s0:])
    assert (reply["stats"].key_min, reply["stats"].key_max) == (lo, hi)
    np.testing.assert_array_equal(reply["sortedIndexes"], out)
    w.terminate()


def test_degenerate_inputs(ctx):
    small = np.load(os.path.join(GOLD, "sort_kat_small.npz"))
    c4 = np.tile(np.array([[1, 2, 3, 1000]], np.int32), (8, 1))
    for n, key in ((8, "all_equal"), (1, "single")):
        w = create_sort_worker(ctx, n)
        w.post_message({"centers": c4[:n], "range": {"from": 0, "to": n - 1, "count": n}})
        r = w.post_message({"sort": {"modelViewProj": np.arange(16.0), "splatRenderCount": n, "splatSortCount": n}})
        np.testing.assert_array_equal(r["sortedIndexes"], small[key])
        # zero-length sort: head copy only
        r = w.post_message({"sort": {"modelViewProj": np.arange(16.0), "splatRenderCount": n, "splatSortCount": 0,
                                     "indexesToSort": np.arange(n, dtype=np.uint32)[::-1].copy()}})
        np.testing.assert_array_equal(r["sortedIndexes"], np.arange(n, dtype=np.uint32)[::-1])
        w.terminate()


def test_chunked_upload_and_count_clamp(ctx):
    rng = np.random.default_rng(3)
    n = 10000
    ci = util.integer_centers(rng.uniform(-5, 5, (n, 3)).astype(np.float32))
    w = create_sort_worker(ctx, n)
    for a in range(0, n, 3000):                       # progressive loading uploads ranges (Viewer.js:1127-1137)
        b = min(a + 3000, n)
        w.post_message({"centers": ci[a:b], "range": {"from": a, "to": b - 1, "count": b - a}})
    mvp = rng.normal(size=16)
    r = w.post_message({"sort": {"modelViewProj": mvp, "splatRenderCount": n + 500, "splatSortCount": n + 500}})
    assert r["splatRenderCount"] == n                  # SortWorker.js:100-101 clamps to the uploaded count
    np.testing.assert_array_equal(r["sortedIndexes"], oracle.sort_indexes(np.arange(n, dtype=np.uint32), ci, mvp))
    w.terminate()


@pytest.mark.parametrize("n,precision", [(5_800_000, 16), (3_000_000, 20)])
def test_full_size_properties_and_oracle(ctx, n, precision):
    """BASELINE.json full size (garden stand-in): bit-exact vs the pinned C oracle, plus size-independent
    properties: permutation, buckets non-increasing, ties in reverse input order."""
    rng = np.random.default_rng(11)
    c = rng.uniform(-8, 8, (n, 3)).astype(np.float32)
    ci = util.integer_centers(c)
    cam = camera.demo_camera("garden", 1920, 1080)
    mvp = cam.sort_mvp()
    w = create_sort_worker(ctx, n, splat_sort_distance_map_precision=precision)
    w.post_message({"centers": ci, "range": {"from": 0, "to": n - 1, "count": n}})
    r = w.post_message({"sort": {"modelViewProj": mvp, "splatRenderCount": n, "splatSortCount": n}})
    got = r["sortedIndexes"]
    exp, keys, buckets, _, st = oracle.sort_indexes(np.arange(n, dtype=np.uint32), ci, mvp, precision=precision,
                                                    return_intermediates=True)
    assert st == 0 and r["status"] == 0
    bs = buckets[got].astype(np.int64)
    assert (np.diff(bs) <= 0).all()
    same = np.diff(bs) == 0
    assert (np.diff(got.astype(np.int64))[same] < 0).all()          # equal bucket -> descending input position
    assert np.array_equal(np.bincount(got, minlength=n), np.ones(n, dtype=np.int64))
    np.testing.assert_array_equal(got, exp)
    w.terminate()


def test_ballot_ranking_fallback_is_bit_exact(monkeypatch):
    """The radix scatter ranks keys with LDS atomics only after a create-time self-test of their lane order; the portable
    ballot path (forced here) must give the same answers."""
    monkeypatch.setenv("GSPLAT_NO_LDS_ATOMIC_RANK", "1")
    c = Context(0)
    meta = json.load(open(os.path.join(GOLD, "sort_kat.json")))
    for name in ("million", "permuted_partial", "float_p22"):
        case = [k for k in kat_cases.CASES if k["name"] == name][0]
        args = kat_cases.make_case(case)
        w, reply = run_worker(c, args)
        assert kat_cases.digest(reply["sortedIndexes"]) == meta[name]["output"]
        w.terminate()
    c.close()


def test_keys_and_pass_0_histogram_in_one_launch_is_bit_exact(monkeypatch):
    """$GSPLAT_KEY_HIST_FUSED=1 (built as an opt-in): k_depth_key_hist keys the splats, meets its whole grid on one
    counter for the exact min / max and histograms pass 0 from registers.  Same list as the two kernels at sizes that give one
    chunk, several, ragged last tiles; sizes it does not take (not a multiple of 4) and partial sorts fall back silently."""
    c = Context(0)
    rng = np.random.default_rng(99)
    cam = camera.demo_camera("garden", 640, 360)
    for n in (4, 4096, 12288, 12292, 100000, 1200000, 999999):
        centers = (rng.normal(size=(n, 3)) * 3.0).astype(np.float32)
        ci = util.integer_centers(centers)
        w = create_sort_worker(c, n)
        w.post_message({"centers": ci, "range": {"from": 0, "to": n - 1, "count": n}})

        def sort(count):
            return w.post_message({"sort": {"modelViewProj": cam.sort_mvp(), "splatRenderCount": n, "splatSortCount": count,
                                            "usePrecomputedDistances": False, "indexesToSort": None, "transforms": None,
                                            "precomputedDistances": None}})["sortedIndexes"].copy()
        monkeypatch.setenv("GSPLAT_KEY_HIST_FUSED", "1")
        full = sort(n)
        np.testing.assert_array_equal(full, oracle.sort_indexes(np.arange(n, dtype=np.uint32), ci, cam.sort_mvp()), err_msg=f"n {n}")
        assert w.last_stats()[0].result_count == n
        part = sort(max(n // 2, 1))
        monkeypatch.delenv("GSPLAT_KEY_HIST_FUSED")
        np.testing.assert_array_equal(sort(n), full)
        np.testing.assert_array_equal(sort(max(n // 2, 1)), part)
        w.terminate()
    c.close()
