"""The Node N-API binding (node/gsplat_addon.c + node/gsplat.js): CPU tier checks it builds, loads and exposes the
reference-shaped interface; the GPU tier pushes real sorts through createSortWorker's message protocol."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

import kat_cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NODE_DIR = os.path.join(ROOT, "node")
pytestmark = pytest.mark.skipif(shutil.which("node") is None, reason="node is not installed")


def _built():
    subprocess.check_call(["make", "-C", NODE_DIR], stdout=subprocess.DEVNULL)
    return os.path.join(NODE_DIR, "gsplat_addon.node")


def test_addon_loads_and_exports_the_seams():
    _built()
    js = ("const g=require('./gsplat.js');"
          "console.log(JSON.stringify({k:Object.keys(g.addon).sort(),abi:g.addon.abiVersion,"
          "w:typeof g.createSortWorker,m:typeof g.SplatMeshHIP,args:g.createSortWorker.length}))")
    out = subprocess.check_output(["node", "-e", js], cwd=NODE_DIR, text=True)
    info = json.loads(out.strip().splitlines()[-1])
    assert info["abi"] == 5 and info["w"] == "function" and info["m"] == "function"
    for name in ("contextCreate", "sorterCreate", "sorterUploadCenters", "sorterSort", "meshCreate", "meshUpload",
                 "meshRender", "sorterDestroy", "meshDestroy", "contextDestroy", "deviceCount", "sorterBindMesh",
                 "sorterSetFrustumCull", "sorterSortGathered", "treeCreate", "treeGather", "assetLoad", "meshSetScenes",
                 "meshUploadSceneIndexes", "meshUploadShU8", "meshSetDestination", "meshSetDrawMode"):
        assert name in info["k"]
    assert info["args"] == 5          # five positional parameters before the defaulted precision, like the reference


def test_js_half_float_matches_host_mirror():
    _built()
    from gaussiansplats3d_amd.util import to_half_three
    vals = [0.0, 1.0, -1.0, 0.1, 65504.0, 1e-5, 6e-8, 1e5, 3.14159, -2.71828e-3, 0.33333, 123.456, -7e-6]
    js = f"const g=require('./gsplat.js');console.log(JSON.stringify({json.dumps(vals)}.map(g.toHalfFloat)))"
    got = json.loads(subprocess.check_output(["node", "-e", js], cwd=NODE_DIR, text=True))
    np.testing.assert_array_equal(np.array(got, dtype=np.uint16), to_half_three(np.array(vals, np.float32)))


def _write_case(path, args):
    n = args["centers4"].shape[0]
    hdr = np.array([n, args["render_count"], args["sort_count"], 1 << args["precision"], int(args["use_int"]),
                    int(args["dynamic"]), int(args["precomputed"] is not None), 0], dtype=np.uint32)
    with open(path, "wb") as f:
        f.write(hdr.tobytes())
        f.write(args["indexes"].tobytes())
        f.write(np.ascontiguousarray(args["centers4"]).tobytes())
        f.write(np.asarray(args["mvp"], dtype=np.float64).astype(np.float32).tobytes())
        if args["dynamic"]:
            f.write(args["scene_indexes"].tobytes())
            f.write(args["transforms"].tobytes())
        if args["precomputed"] is not None:
            f.write(args["precomputed"].tobytes())


@pytest.mark.gpu
@pytest.mark.parametrize("name,mode", [("small", "copy"), ("permuted_partial", "shared"), ("dynamic_int", "copy"),
                                        ("float_dynamic", "shared"), ("pre_int", "copy")])
def test_sort_through_the_js_protocol_is_bit_exact(tmp_path, name, mode):
    _built()
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", "sort_kat.json")))[name]
    case = [c for c in kat_cases.CASES if c["name"] == name][0]
    args = kat_cases.make_case(case)
    inp, outp = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    _write_case(inp, args)
    cmd = ["node", "sort_via_js.js", inp, outp] + (["shared"] if mode == "shared" else [])
    res = subprocess.run(cmd, cwd=NODE_DIR, capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    got = np.fromfile(outp, dtype=np.uint32)
    assert kat_cases.digest(got) == meta["output"]


@pytest.mark.gpu
def test_sort_seam_is_asynchronous_like_the_reference_worker(tmp_path):
    """postMessage returns before sortDone fires (src/worker/SortWorker.js:62-80 replies from a worker thread), messages are
    handled in posting order, and both results are the reference's."""
    _built()
    import oracle
    case = [c for c in kat_cases.CASES if c["name"] == "permuted_partial"][0]
    args = kat_cases.make_case(case)
    inp, out_a, out_b = str(tmp_path / "in.bin"), str(tmp_path / "a.bin"), str(tmp_path / "b.bin")
    _write_case(inp, args)
    res = subprocess.run(["node", "async_via_js.js", inp, out_a, out_b], cwd=NODE_DIR, capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    info = json.loads(res.stdout.strip().splitlines()[-1])
    assert info["repliesAtReturn"] == 0, "a sortDone fired before postMessage returned"
    assert info["events"] == ["posted", "sortDone1", "sortDone2"]
    kw = dict(sort_count=args["sort_count"], render_count=args["render_count"], precision=args["precision"])
    np.testing.assert_array_equal(np.fromfile(out_a, dtype=np.uint32),
                                  oracle.sort_indexes(args["indexes"], args["centers4"], args["mvp"], **kw))
    np.testing.assert_array_equal(np.fromfile(out_b, dtype=np.uint32),
                                  oracle.sort_indexes(args["indexes"], args["centers4"], info["mvpB"], **kw))


@pytest.mark.gpu
def test_frustum_culled_sort_through_the_js_protocol(tmp_path):
    """worker.setFrustumCull(true): the sortDone reply carries the kept list and its length as splatRenderCount."""
    import oracle
    import helpers
    from gaussiansplats3d_amd import camera, util
    _built()
    scene = helpers.small_scene(20000, 0, seed=77)
    rng = np.random.default_rng(78)
    scene.centers[:8000] += rng.normal(size=(8000, 3)).astype(np.float32) * 6.0          # push many out of the frustum
    ci = util.integer_centers(scene.centers)
    cam = camera.demo_camera("garden", 1280, 720)
    idx = np.arange(scene.count, dtype=np.uint32)
    args = {"centers4": ci, "render_count": scene.count, "sort_count": scene.count, "precision": 16, "use_int": True,
            "dynamic": False, "precomputed": None, "indexes": idx, "mvp": cam.sort_mvp()}
    inp, outp = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    _write_case(inp, args)
    res = subprocess.run(["node", "sort_via_js.js", inp, outp, "cull"], cwd=NODE_DIR, capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    reply = json.loads(res.stdout.strip().splitlines()[-1])
    expect, keep = oracle.culled_sort(idx, ci, cam.sort_mvp())
    assert 0 < keep.sum() < scene.count and reply["splatRenderCount"] == len(expect)
    np.testing.assert_array_equal(np.fromfile(outp, dtype=np.uint32), expect)


def test_tree_and_asset_bindings_on_the_host(tmp_path):
    """treeCreate (host-only tree, ctx = null) and assetLoad through Node: same answers as the Python mirrors."""
    _built()
    import tree_cases
    import asset_cases
    from gaussiansplats3d_amd import assets
    case = tree_cases.make_case("clusters40k")
    cpath = str(tmp_path / "c.f32")
    case["centers"].astype(np.float32).tofile(cpath)
    out = subprocess.check_output(["node", "tree_asset_via_js.js", "tree", cpath, str(case["max_depth"]), str(case["max_centers"])],
                                  cwd=NODE_DIR, text=True)
    info = json.loads(out.strip().splitlines()[-1])
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "tree_kat.json")))["clusters40k"]
    assert (info["leaves"], info["allLeaves"], info["splats"]) == (gold["leaves"], gold["all_leaves"], gold["splats"])
    data, _ = asset_cases.make_case("sh2_45")
    ppath, opath = str(tmp_path / "a.ply"), str(tmp_path / "a.bin")
    open(ppath, "wb").write(data)
    meta = json.loads(subprocess.check_output(["node", "tree_asset_via_js.js", "asset", ppath, "1", "2", opath],
                                              cwd=NODE_DIR, text=True).strip().splitlines()[-1])
    exp = assets.load(data, 2)
    assert (meta["splatCount"], meta["shDegree"], meta["compressionLevel"], meta["shLevel"]) == (64, 2, 0, 1)
    want = exp["centers"].tobytes() + exp["cov"].tobytes() + exp["rgba"].tobytes() + exp["sh_f16"].tobytes()
    assert open(opath, "rb").read() == want


@pytest.mark.gpu
def test_tree_gather_through_node_matches_oracle(tmp_path):
    _built()
    import tree_cases
    from oracle import tree_oracle
    from gaussiansplats3d_amd import camera
    case = tree_cases.make_case("clusters40k")
    cam = camera.demo_camera("garden", 1920, 1080)
    cpath, mpath, opath = str(tmp_path / "c.f32"), str(tmp_path / "mv.f64"), str(tmp_path / "o.u32")
    case["centers"].astype(np.float32).tofile(cpath)
    np.asarray(cam.view, np.float64).tofile(mpath)
    res = subprocess.run(["node", "tree_asset_via_js.js", "tree", cpath, "8", "300", "gather", mpath, "1920", "1080", opath],
                         cwd=NODE_DIR, capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    leaves, _ = tree_oracle.build_tree(case["centers"], None, 8, 300)
    expect = tree_oracle.gather(leaves, cam.view, 50.0, 1920, 1080)
    info = json.loads(res.stdout.strip().splitlines()[-1])
    assert info["renderCount"] == len(expect)
    np.testing.assert_array_equal(np.fromfile(opath, dtype=np.uint32), expect)


@pytest.mark.gpu
@pytest.mark.parametrize("ortho,fade,effects,dest,rop8", [(False, False, False, False, 0), (True, False, False, False, 0), (False, True, True, False, 0),
                                                          (False, False, False, True, 0), (False, False, False, True, 1), (False, False, False, False, 2)])
def test_mesh_draw_through_the_js_shim_matches_the_python_mirror(tmp_path, ortho, fade, effects, dest, rop8):
    """SplatMeshHIP (node/gsplat.js -> N-API -> C ABI) and SplatMesh (ctypes -> C ABI) must draw the same pixels,
    including the orthographic, fade-in and per-scene opacity / visibility uniforms, and a destination (drop-in mode's depth test
    against the host's geometry + its colour, gs_mesh_set_destination) and the RGBA8-per-splat draw modes (setRop8: gs_mesh_set_draw_mode)."""
    import helpers
    import oracle
    from gaussiansplats3d_amd import Context, SplatMesh, camera, util
    _built()
    scene = helpers.small_scene(2500, 1, seed=88)
    n, W, H = scene.count, 320, 180
    up, pos, look = camera.DEMO_POSES["garden"]
    cam = camera.OrthographicCamera(W, H, pos, look, up, zoom=40.0) if ortho else camera.demo_camera("garden", W, H)
    order = oracle.sort_indexes(np.arange(n, dtype=np.uint32), util.integer_centers(scene.centers), cam.sort_mvp())
    sidx = (np.arange(n) % 3).astype(np.uint32)
    opacity, visible = np.array([1.0, 0.5, 1.0], np.float32), np.array([1, 1, 0], np.uint32)
    center = scene.centers.mean(axis=0).astype(np.float32)
    radius = float(np.median(np.linalg.norm(scene.centers - center, axis=1)))
    # the Python mirror
    ctx = Context(0)
    mesh = SplatMesh(ctx, n, 1, enable_optional_effects=effects)
    mesh.build(scene.centers, scene.cov, scene.rgba, scene.sh, scene_indexes=sidx if effects else None)
    if effects:
        mesh.set_scenes(opacity=opacity.tolist(), visible=visible.tolist())
    if fade:
        mesh.set_fade_in(center, radius)
    mesh.set_camera(cam)
    mesh.update_render_indexes(order, n)
    rng = np.random.default_rng(8)
    dst_depth = (0.9 + 0.1 * rng.random((H, W))).astype(np.float32)      # around the scene's own window depths
    dst_colour = rng.integers(0, 256, size=(H, W, 4), dtype=np.uint8)
    if dest:
        mesh.set_destination(depth=dst_depth, rgba=dst_colour, depth_unorm24=True)
    if rop8:
        mesh.set_draw_mode(rop8=True, full=rop8 == 2)
    expect, _ = mesh.render()
    mesh.dispose()
    ctx.close()
    # the same draw through Node
    fx, fy = cam.focal()
    flags = (1 if ortho else 0) | (2 if fade else 0) | (4 if effects else 0) | (8 if dest else 0) | (16 if rop8 == 1 else 0) | (32 if rop8 == 2 else 0)
    nsc = 3 if effects else 1
    hdr = np.array([n, 1, W, H, flags, nsc, 0, 0], np.uint32)
    parts = [hdr, scene.centers.astype(np.float32), scene.cov.astype(np.float32), scene.rgba, scene.sh.view(np.uint16), order, sidx,
             np.asarray(cam.model_view(), np.float64).astype(np.float32), np.asarray(cam.projection, np.float64).astype(np.float32),
             np.asarray(cam.position, np.float32), np.array([fx, fy], np.float32), np.array([getattr(cam, "zoom", 1.0)], np.float32),
             center, np.array([radius], np.float32), opacity[:nsc], visible[:nsc]] + ([dst_depth, dst_colour] if dest else [])
    inp, outp = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(inp, "wb") as f:
        for p in parts:
            f.write(np.ascontiguousarray(p).tobytes())
    res = subprocess.run(["node", "render_via_js.js", inp, outp], cwd=NODE_DIR, capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    got = np.fromfile(outp, dtype=np.uint8).reshape(H, W, 4)
    assert got.any()
    np.testing.assert_array_equal(got, expect)
    assert json.loads(res.stdout.strip().splitlines()[-1])["stripIdentical"], "StripGroup(1).renderStrip differs from render()"
