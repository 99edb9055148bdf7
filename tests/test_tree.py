"""Octree build (SplatTree.js) and per-sort cull (Viewer.gatherSceneNodesForSort): oracle pinned to the reference's own
code, native builder vs oracle, device gather vs oracle, scheduler logic."""
import hashlib
import json
import os
import struct

import numpy as np
import pytest

import tree_cases
from gaussiansplats3d_amd import SortScheduler, SplatTree, camera, util

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "tree_kat.json")))


@pytest.fixture(scope="module")
def ctx():
    from gaussiansplats3d_amd import Context
    c = Context(0)
    yield c
    c.close()


def _hex(v):
    return struct.pack("<d", float(v)).hex()


def _digest(leaves):
    h = hashlib.sha256()
    for lf in leaves:
        h.update(("".join(_hex(v) for v in lf["min"]) + "".join(_hex(v) for v in lf["max"]) +
                  "".join(_hex(v) for v in lf["center"]) + str(lf["depth"])).encode())
        h.update(np.asarray(lf["indexes"], dtype=np.uint32).tobytes())
    return h.hexdigest()


def _native_leaves(tree):
    bounds, centers, depths, offsets, indexes = tree.leaves()
    return [dict(min=bounds[i, :3].tolist(), max=bounds[i, 3:].tolist(), center=centers[i].tolist(), depth=int(depths[i]),
                 indexes=indexes[offsets[i]:offsets[i + 1]].tolist()) for i in range(len(depths))]


@pytest.mark.parametrize("name", tree_cases.CASES)
def test_oracle_matches_the_reference_tree(name):
    """Pins oracle/tree_oracle.py: same leaves (bounds, centres as exact doubles, depth, index lists, order) as the
    reference's createSplatTreeWorker produced for the same input (tests/golden/tree_kat.json)."""
    from oracle import tree_oracle
    case = tree_cases.make_case(name)
    g = GOLD[name]
    assert hashlib.sha256(case["centers"].tobytes()).hexdigest() == g["inputs"], "case generator drifted from the golden inputs"
    if name == "coincident":
        pytest.skip("32768-leaf degenerate tree: covered by the native builder test (pure-Python recursion is slow)")
    leaves, all_leaves = tree_oracle.build_tree(case["centers"], None, case["max_depth"], case["max_centers"])
    assert (len(leaves), all_leaves) == (g["leaves"], g["all_leaves"])
    assert _digest(leaves) == g["sha256"]


@pytest.mark.parametrize("name", tree_cases.CASES)
def test_native_builder_matches_the_reference_tree(name):
    """gs_tree_create (host-only, no GPU): bit-identical to the reference's tree."""
    case = tree_cases.make_case(name)
    g = GOLD[name]
    tree = SplatTree(None, case["max_depth"], case["max_centers"]).process_splat_mesh(case["centers"])
    info = tree.info()
    assert (info.leaves, info.all_leaves, info.splats) == (g["leaves"], g["all_leaves"], g["splats"])
    assert _digest(_native_leaves(tree)) == g["sha256"]
    tree.dispose()


def test_alpha_filter_and_first_index():
    from oracle import tree_oracle
    rng = np.random.default_rng(3)
    c = rng.normal(size=(4000, 3)).astype(np.float32)
    alpha = rng.integers(0, 4, 4000).astype(np.uint8)            # a quarter have alpha 0 -> filtered (minAlpha 1)
    tree = SplatTree(None, 8, 300).process_splat_mesh(c, alphas=alpha, min_alpha=1, first_index=1000)
    leaves, _ = tree_oracle.build_tree(c, alpha >= 1, 8, 300, first_index=1000)
    assert _digest(_native_leaves(tree)) == _digest(leaves)
    idx = np.concatenate([lf["indexes"] for lf in leaves])
    assert idx.min() >= 1000 and len(idx) == int((alpha >= 1).sum())


def test_scheduler_follows_run_splat_sort():
    cam0 = camera.demo_camera("garden", 640, 360)
    s = SortScheduler()
    # first call: lastSortViewDir = (0,0,-1), garden looks elsewhere -> angleDiff small -> partial sorts queued
    n = 100000
    view_dir = SortScheduler.view_direction(cam0)
    angle = float(np.dot(view_dir, [0, 0, -1]))
    first = s.next_sort(cam0, n, should_sort_all=False)
    expect = []
    for p in ({"t": 0.55, "f": (0.125, 0.33333, 0.75)}, {"t": 0.65, "f": (0.33333, 0.66667)}, {"t": 0.8, "f": (0.5,)}):
        if angle < p["t"]:
            expect = [int(np.floor(n * f)) for f in p["f"]]
            break
    expect.append(n)
    assert first == expect[0]
    assert s.next_sort(cam0, n, False) is None                   # sortRunning
    got = [first]
    while s.queued_sorts:
        s.sort_done()
        got.append(s.next_sort(cam0, n, False))
    assert got == expect
    s.sort_done()
    assert s.next_sort(cam0, n, False) is None                   # camera unchanged since the completed schedule
    up, pos, look = camera.DEMO_POSES["garden"]
    moved = camera.PerspectiveCamera(640, 360, np.array(pos) + [0.0, 0.0, 1.5], look, up)
    assert s.next_sort(moved, n, False) is not None              # moved >= 1.0
    s.sort_done()
    while s.queued_sorts:
        s.next_sort(moved, n, False); s.sort_done()
    assert s.next_sort(moved, n, True) is None and s.next_sort(moved, n, True, force=True) == n   # shouldSortAll


# ------------------------------------------------------------------------------------------------ sort trigger, pinned
SCHED = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "sched_kat.json")))


class _SchedCamera:
    """What SortScheduler and sort_mvp read from a camera, rebuilt from the golden's matrixWorld / projection."""

    def __init__(self, step):
        self.matrix_world = np.asarray(step["matrixWorld"], np.float64)
        self.position = self.matrix_world[12:15].copy()
        self.projection = np.asarray(step["projection"], np.float64)
        self.view = camera.invert(self.matrix_world)

    sort_mvp = camera.PerspectiveCamera.sort_mvp


@pytest.mark.parametrize("script", SCHED, ids=[s["name"] for s in SCHED])
def test_scheduler_matches_the_reference_run_splat_sort(script):
    """tests/golden/sched_kat.json: Viewer.runSplatSort's own text (src/Viewer.js:1833-1964) executed under Node over scripted
    camera paths (oracle/make_golden_sched.py).  The mirror must post a sort exactly when the reference does, with the same
    splatSortCount, and modelViewProj must agree (1e-12 relative in fp64; identical once narrowed to the fp32 the sorter reads,
    except entries that are zero up to rounding noise)."""
    s = SortScheduler(dynamic_mode=script["dynamicMode"])
    mesh_world = None if script["dynamicMode"] else np.asarray(script["meshWorld"], np.float64)
    posted = 0
    for step in script["steps"]:
        if step.get("sortDone"):
            s.sort_done()
            continue
        ref = step["ref"]
        cam = _SchedCamera(step)
        got = s.next_sort(cam, step["splatRenderCount"], step["shouldSortAll"], force=step.get("force", False),
                          force_sort_all=step.get("forceSortAll", False))
        assert (got is not None) == bool(ref["posted"]), (script["name"], step, ref)
        if got is not None:
            posted += 1
            assert got == ref["splatSortCount"] and step["splatRenderCount"] == ref["splatRenderCount"]
            want = np.array([struct.unpack("<d", bytes.fromhex(h))[0] for h in ref["modelViewProj"]])
            mvp = np.asarray(cam.sort_mvp(mesh_world), np.float64)
            np.testing.assert_allclose(mvp, want, rtol=1e-12, atol=1e-13)
            # entries that cancel to zero carry 1e-17 of noise from the inverse (LU here, cofactors in three): what the integer
            # sort reads - (int)(mvp[2 | 6 | 10] * 1000.0) of the fp32 value, sorter.cpp:41-43 - is identical
            a, b = mvp.astype(np.float32), want.astype(np.float32)
            assert np.array_equal(np.trunc(a[[2, 6, 10]].astype(np.float64) * 1000.0), np.trunc(b[[2, 6, 10]].astype(np.float64) * 1000.0))
            assert np.array_equal(a[np.abs(b) > 1e-9], b[np.abs(b) > 1e-9])
    assert posted == sum(1 for st in script["steps"] if not st.get("sortDone") and st["ref"]["posted"]) and posted >= 1


# ------------------------------------------------------------------------------------------------ cull, pinned
GATHER = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "gather_kat.json")))


def _gather_cases():
    return [(name, k) for name in GATHER for k in range(len(GATHER[name]["cameras"]))]


def _model_view(cam):
    return np.array([struct.unpack("<d", bytes.fromhex(h))[0] for h in cam["modelView"]])


class _Dims:
    def __init__(self, cam):
        self.width, self.height = cam["width"], cam["height"]


@pytest.mark.parametrize("name,k", _gather_cases())
def test_cull_oracle_matches_the_reference_gather(name, k):
    """Pins oracle/tree_oracle.py::gather: same splatRenderCount and the same index list as the reference's own
    Viewer.gatherSceneNodesForSort text produced under Node (oracle/gather_ref.mjs, tests/golden/gather_kat.json)."""
    from oracle import tree_oracle
    case = tree_cases.make_case(name)
    cam = GATHER[name]["cameras"][k]
    leaves, _ = tree_oracle.build_tree(case["centers"], None, case["max_depth"], case["max_centers"])
    assert len(leaves) == GATHER[name]["leaves"]
    got = tree_oracle.gather(leaves, _model_view(cam), cam["fov"], cam["width"], cam["height"], cam["gatherAll"])
    assert len(got) == cam["splatRenderCount"]
    assert hashlib.sha256(got.tobytes()).hexdigest() == cam["sha256"]
    if cam["indexes"] is not None:
        np.testing.assert_array_equal(got, np.array(cam["indexes"], np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("name,k", _gather_cases())
def test_device_gather_matches_the_reference_gather(ctx, name, k):
    """The device cull (gs_tree_gather) against the same reference-recorded lists."""
    case = tree_cases.make_case(name)
    cam = GATHER[name]["cameras"][k]
    tree = SplatTree(ctx, case["max_depth"], case["max_centers"]).process_splat_mesh(case["centers"])
    got = tree.gather_scene_nodes_for_sort(_Dims(cam), gather_all_nodes=cam["gatherAll"], fov_deg=cam["fov"],
                                           model_view=_model_view(cam))
    assert got["splatRenderCount"] == cam["splatRenderCount"]
    assert hashlib.sha256(got["indexesToSort"].tobytes()).hexdigest() == cam["sha256"]
    tree.dispose()


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("pose,gather_all", [("garden", False), ("truck", False), ("bonsai", False), ("garden", True)])
def test_device_gather_matches_oracle(ctx, pose, gather_all):
    from oracle import tree_oracle
    case = tree_cases.make_case("clusters40k")
    c = case["centers"]
    tree = SplatTree(ctx, 8, 300).process_splat_mesh(c)
    cam = camera.demo_camera(pose, 1920, 1080)
    leaves, _ = tree_oracle.build_tree(c, None, 8, 300)
    expect = tree_oracle.gather(leaves, cam.view, 50.0, 1920, 1080, gather_all)
    got = tree.gather_scene_nodes_for_sort(cam, gather_all_nodes=gather_all)
    assert got["splatRenderCount"] == len(expect)
    if not gather_all and pose != "bonsai":               # the bonsai pose looks at the whole cloud from outside
        assert 0 < len(expect) < 40000, "the case should cull something but not everything"
    np.testing.assert_array_equal(got["indexesToSort"], expect)
    tree.dispose()


@pytest.mark.gpu
def test_gather_then_sort_on_device_matches_reference_pipeline(ctx):
    """cull -> indexesToSort (device) -> sort, against oracle cull + the reference sorter on the host list;
    also a partial sort (splatSortCount < splatRenderCount)."""
    import oracle
    from oracle import tree_oracle
    from gaussiansplats3d_amd import create_sort_worker
    case = tree_cases.make_case("clusters40k")
    c = case["centers"]
    n = c.shape[0]
    ci = util.integer_centers(c)
    cam = camera.demo_camera("garden", 1280, 720)
    tree = SplatTree(ctx, 8, 300).process_splat_mesh(c)
    w = create_sort_worker(ctx, n)
    w.post_message({"centers": ci, "range": {"from": 0, "to": n - 1, "count": n}})
    leaves, _ = tree_oracle.build_tree(c, None, 8, 300)
    idx = tree_oracle.gather(leaves, cam.view, 50.0, 1280, 720)
    r = tree.gather_scene_nodes_for_sort(cam, sort_worker=w, to_host=False)
    R = r["splatRenderCount"]
    assert R == len(idx)
    for sort_count in (R, R // 3):
        reply = w.sort_gathered(cam.sort_mvp(), sort_count)
        expect = oracle.sort_indexes(idx, ci, cam.sort_mvp(), sort_count=sort_count, render_count=R)
        np.testing.assert_array_equal(reply["sortedIndexes"], expect)
    w.terminate()
    tree.dispose()


@pytest.mark.gpu
@pytest.mark.parametrize("frustum_cull", [False, True])
def test_asynchronous_gather_sorts_and_draws_without_a_host_round_trip(ctx, frustum_cull):
    """gs_tree_gather(render_count = NULL): splatRenderCount never reaches the host - the plan kernel leaves it on the device,
    the sort takes the list's length from there, the draw takes the sorted list's length from the sort.  The sorted list
    (its real length from the sort's statistics) equals the reference sorter on the oracle's gathered list, the frame
    equals the frame of the synchronous path; with the per-splat frustum cull on top the list is the same one minus the
    splats the cull drops, and the frame does not change."""
    import oracle
    from oracle import tree_oracle
    import helpers
    from gaussiansplats3d_amd import SplatMesh, create_sort_worker
    scene = helpers.small_scene(30000, 1, seed=61)
    c, n = scene.centers, scene.count
    ci = util.integer_centers(c)
    cam = camera.demo_camera("garden", 640, 360)
    tree = SplatTree(ctx, 8, 200).process_splat_mesh(c)
    w = create_sort_worker(ctx, n)
    w.post_message({"centers": ci, "range": {"from": 0, "to": n - 1, "count": n}})
    mesh = SplatMesh(ctx, n, 1).build(scene.centers, scene.cov, scene.rgba, scene.sh)
    mesh.set_camera(cam)
    leaves, _ = tree_oracle.build_tree(c, None, 8, 200)
    idx = tree_oracle.gather(leaves, cam.view, 50.0, 640, 360)
    assert 0 < len(idx) < n, "the case should cull something"
    expect = oracle.sort_indexes(idx, ci, cam.sort_mvp())
    # synchronous path: the frame to reproduce
    r = tree.gather_scene_nodes_for_sort(cam, sort_worker=w, to_host=False)
    assert r["splatRenderCount"] == len(idx)
    w.sort_gathered(cam.sort_mvp(), keep_on_device=True)
    mesh.use_sorter_result(w, r["splatRenderCount"])
    want = mesh.render()[0]
    assert want[..., 3].any()
    # asynchronous path
    w.set_frustum_cull(frustum_cull)
    for _ in range(2):                                       # twice: the device-side tables must be left clean
        r = tree.gather_scene_nodes_for_sort(cam, sort_worker=w, to_host=False, asynchronous=True)
        assert r["countOnDevice"] and r["splatRenderCount"] == int(tree.info().splats) >= len(idx)
        w.sort_gathered(cam.sort_mvp(), keep_on_device=True)
        mesh.use_sorter_result(w, r["splatRenderCount"])
        got = mesh.render()[0]
        np.testing.assert_array_equal(got, want)
    st, _ = w.last_stats()
    got_list = w.debug_read(2, int(st.result_count))       # (positions beyond the result's length are undefined)
    if frustum_cull:
        kept_sorted, keep = oracle.culled_sort(idx, ci, cam.sort_mvp())
        np.testing.assert_array_equal(got_list, kept_sorted)
        assert st.result_count == int(keep.sum()) < len(idx)
    else:
        assert st.result_count == len(idx)
        np.testing.assert_array_equal(got_list, expect)
    with pytest.raises(Exception):                           # a partial sort needs splatRenderCount on the host
        tree.gather_scene_nodes_for_sort(cam, sort_worker=w, to_host=False, asynchronous=True)
        w.sort_gathered(cam.sort_mvp(), 100, keep_on_device=True)
    w.set_frustum_cull(False)
    w.terminate()
    mesh.dispose()
    tree.dispose()


@pytest.mark.gpu
def test_deferred_gather_copy_float_sorter_and_lifecycle(ctx, monkeypatch):
    """gs_tree_gather only plans when a static sorter is its only reader; the sorter copies the lists itself, fused with its key
    kernel (sorter.hip, k_tree_copy_keys).  The float sorter goes through the same kernel; the list must equal the one of the
    immediate copy (GSPLAT_TREE_NO_DEFER=1) and the reference sorter on the oracle's gathered list.  Lifecycle: a second gather
    before the sort supersedes the first; a tree destroyed before the sort leaves the sorter without a list (an error, not a
    dangling pointer); a second sorter gathering from the same tree makes the first one copy at once."""
    import oracle
    from oracle import tree_oracle
    from gaussiansplats3d_amd import create_sort_worker
    case = tree_cases.make_case("clusters40k")
    c = case["centers"]
    n = c.shape[0]
    cf = np.concatenate([c.astype(np.float32), np.ones((n, 1), np.float32)], axis=1)
    cams = [camera.demo_camera("garden", 1280, 720), camera.orbit_cameras("garden", 1280, 720, 6)[2]]
    tree = SplatTree(ctx, 8, 300).process_splat_mesh(c)
    leaves, _ = tree_oracle.build_tree(c, None, 8, 300)
    w = create_sort_worker(ctx, n, integer_based_sort=False)
    w.post_message({"centers": cf, "range": {"from": 0, "to": n - 1, "count": n}})
    lists = {}
    for defer in (True, False):
        if not defer:
            monkeypatch.setenv("GSPLAT_TREE_NO_DEFER", "1")
        for k, cam in enumerate(cams):
            idx = tree_oracle.gather(leaves, cam.view, 50.0, 1280, 720)
            if defer and k == 1:                              # an unconsumed gather for another camera first: superseded
                tree.gather_scene_nodes_for_sort(cams[0], sort_worker=w, to_host=False)
            r = tree.gather_scene_nodes_for_sort(cam, sort_worker=w, to_host=False)
            assert r["splatRenderCount"] == len(idx)
            reply = w.sort_gathered(cam.sort_mvp())
            expect = oracle.sort_indexes(idx, cf, cam.sort_mvp(), use_int=False)
            np.testing.assert_array_equal(reply["sortedIndexes"], expect)
            lists[(defer, k)] = reply["sortedIndexes"]
        if not defer:
            monkeypatch.delenv("GSPLAT_TREE_NO_DEFER")
    for k in range(len(cams)):
        np.testing.assert_array_equal(lists[(True, k)], lists[(False, k)])
    # two sorters, one tree: the second gather makes the first sorter copy its (still planned) list at once
    w2 = create_sort_worker(ctx, n, integer_based_sort=False)
    w2.post_message({"centers": cf, "range": {"from": 0, "to": n - 1, "count": n}})
    tree.gather_scene_nodes_for_sort(cams[0], sort_worker=w, to_host=False)
    tree.gather_scene_nodes_for_sort(cams[1], sort_worker=w2, to_host=False)
    np.testing.assert_array_equal(w.sort_gathered(cams[0].sort_mvp())["sortedIndexes"], lists[(True, 0)])
    np.testing.assert_array_equal(w2.sort_gathered(cams[1].sort_mvp())["sortedIndexes"], lists[(True, 1)])
    # the tree goes away under a planned gather: the sort is refused
    tree.gather_scene_nodes_for_sort(cams[0], sort_worker=w, to_host=False)
    tree.dispose()
    with pytest.raises(Exception):
        w.sort_gathered(cams[0].sort_mvp())
    w.terminate()
    w2.terminate()


@pytest.mark.gpu
@pytest.mark.parametrize("name", tree_cases.CASES)
def test_device_builder_matches_the_reference_tree(ctx, name):
    """gs_tree_create with a context builds the octree ON THE DEVICE (level-synchronous memberships, first-visitor claim,
    stable sort): the same leaves as the reference's createSplatTreeWorker, bit for bit - bounds, centres, depths, index
    lists, order, node and leaf counts (tests/golden/tree_kat.json), including centres on split planes (memberships in
    several children), the depth limit and the 32768-leaf degenerate tree."""
    case = tree_cases.make_case(name)
    g = GOLD[name]
    tree = SplatTree(ctx, case["max_depth"], case["max_centers"]).process_splat_mesh(case["centers"])
    info = tree.info()
    assert (info.leaves, info.all_leaves, info.splats) == (g["leaves"], g["all_leaves"], g["splats"])
    assert _digest(_native_leaves(tree)) == g["sha256"]
    tree.dispose()


@pytest.mark.gpu
def test_device_builder_equals_the_host_builder_on_a_large_filtered_scene(ctx, monkeypatch):
    """300 k clustered centres, an alpha filter and a first_index: device build == host build (GSPLAT_TREE_HOST_BUILD=1),
    every field; and the device gather works on the device-built tree."""
    rng = np.random.default_rng(77)
    k = rng.uniform(-4, 4, size=(200, 3))
    c = (k[rng.integers(0, 200, 300000)] + rng.normal(size=(300000, 3)) * 0.15).astype(np.float32)
    c[::1000] = np.round(c[::1000])                          # some centres on coarse planes
    alphas = rng.integers(0, 256, 300000).astype(np.uint8)
    dev = SplatTree(ctx, 8, 1000).process_splat_mesh(c, alphas=alphas, min_alpha=40, first_index=1234)
    monkeypatch.setenv("GSPLAT_TREE_HOST_BUILD", "1")
    host = SplatTree(ctx, 8, 1000).process_splat_mesh(c, alphas=alphas, min_alpha=40, first_index=1234)
    monkeypatch.delenv("GSPLAT_TREE_HOST_BUILD")
    a, b = dev.info(), host.info()
    assert (a.leaves, a.all_leaves, a.nodes, a.splats) == (b.leaves, b.all_leaves, b.nodes, b.splats) and a.leaves > 100
    for x, y in zip(dev.leaves(), host.leaves()):
        np.testing.assert_array_equal(x, y)
    cam = camera.demo_camera("garden", 1920, 1080)
    ga = dev.gather_scene_nodes_for_sort(cam)
    gb = host.gather_scene_nodes_for_sort(cam)
    assert ga["splatRenderCount"] == gb["splatRenderCount"] > 0
    np.testing.assert_array_equal(ga["indexesToSort"], gb["indexesToSort"])
    dev.dispose()
    host.dispose()
