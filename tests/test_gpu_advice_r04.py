"""-m gpu: regression tests for the round-4 advisor findings - each scenario fails or misbehaves on the round-4 library.
The sort's A/B switches ($GSPLAT_NO_SORT_PACK, $GSPLAT_NO_SORT_CHUNK) are read once per process, so the scenarios run in child
processes, once per switch."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import helpers
import oracle
from gaussiansplats3d_amd import Context, GsError, SplatMesh, camera, create_sort_worker, util

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

EMPTY_LIST_SCRIPT = r"""
import json, sys
sys.path.insert(0, %(here)r); sys.path.insert(0, %(root)r)
import numpy as np
import helpers, oracle
from oracle import tree_oracle
from gaussiansplats3d_amd import Context, SplatMesh, SplatTree, camera, create_sort_worker, util
ctx = Context(0)
n = 70000                                                  # > 4 chunks of the chunk-staged scatter
scene = helpers.small_scene(n, 0, seed=91)
cam = camera.demo_camera("garden", 320, 200)
pos, look = np.array(camera.DEMO_POSES["garden"][1]), np.array(camera.DEMO_POSES["garden"][2])
fwd = (look - pos) / np.linalg.norm(look - pos)
away = camera.PerspectiveCamera(320, 200, tuple(pos - 30.0 * fwd), tuple(pos - 60.0 * fwd), camera.DEMO_POSES["garden"][0])   # far behind, looking the other way
ci = util.integer_centers(scene.centers)
w = create_sort_worker(ctx, n)
w.post_message({"centers": ci, "range": {"from": 0, "to": n - 1, "count": n}})
mesh = SplatMesh(ctx, n, 0).build(scene.centers, scene.cov, scene.rgba, None)
mesh.use_sorter_result(w, n)
identity = np.arange(n, dtype=np.uint32)
def full_sort_is_exact(c):
    reply = w.post_message({"sort": {"modelViewProj": c.sort_mvp(), "splatRenderCount": n, "splatSortCount": n}})
    return bool(np.array_equal(reply["sortedIndexes"], oracle.sort_indexes(identity, ci, c.sort_mvp())))
out = {"first": full_sort_is_exact(cam)}                   # leaves row 0 of the offset table full of counts
# (1) a visibility-culled sort with nothing visible
w.set_visibility_cull(True)
mesh.set_camera(away)
mesh.project()
reply = w.post_message({"sort": {"modelViewProj": away.sort_mvp(), "splatRenderCount": n, "splatSortCount": n}})
out["vis_empty"] = int(reply["stats"].result_count)
frame, st = mesh.render()
out["vis_frame_empty"] = bool(not frame.any()) and int(st.visible_splats) == 0
w.set_visibility_cull(False)
out["after_vis"] = full_sort_is_exact(cam)
# (2) an asynchronous octree gather that keeps no leaf, with and without the fused frustum cull
tree = SplatTree(ctx, 8, 200).process_splat_mesh(scene.centers)
leaves, _ = tree_oracle.build_tree(scene.centers, None, 8, 200)
out["oracle_gather_away"] = int(len(tree_oracle.gather(leaves, away.view, 50.0, 320, 200)))
for fc in (False, True):
    w.set_frustum_cull(fc)
    out["warm_%%d" %% fc] = full_sort_is_exact(cam) if not fc else True
    tree.gather_scene_nodes_for_sort(away, sort_worker=w, to_host=False, asynchronous=True)
    w.sort_gathered(away.sort_mvp(), keep_on_device=True)
    st, _ = w.last_stats()
    out["gather_empty_%%d" %% fc] = int(st.result_count)
    mesh.set_camera(away)
    mesh.use_sorter_result(w, int(tree.info().splats))
    frame, _ = mesh.render()
    out["gather_frame_empty_%%d" %% fc] = bool(not frame.any())
w.set_frustum_cull(False)
out["last"] = full_sort_is_exact(cam)
print(json.dumps(out))
tree.dispose(); w.terminate(); mesh.dispose(); ctx.close()
"""


@pytest.mark.parametrize("switch", [None, "GSPLAT_NO_SORT_CHUNK", "GSPLAT_NO_SORT_PACK", "GSPLAT_NO_LDS_ATOMIC_RANK"])
def test_an_empty_device_side_list_after_a_full_sort_writes_nothing(switch):
    """workgroup 0 of the chunk-staged scatter stays alive for an empty list and used to read the PREVIOUS
    sort's row 0 of the offset table as its own digit counts - storing that many never-written staging words through offsets of
    a zero table.  Reached by a visibility-culled sort with nothing visible and by an asynchronous gather that keeps no leaf."""
    env = dict(os.environ)
    if switch:
        env[switch] = "1"
    run = subprocess.run([sys.executable, "-c", EMPTY_LIST_SCRIPT % {"here": HERE, "root": ROOT}], env=env, capture_output=True,
                         text=True, timeout=600)
    assert run.returncode == 0, run.stderr[-3000:]
    out = json.loads(run.stdout.strip().splitlines()[-1])
    assert out["oracle_gather_away"] == 0, "the pose must cull every leaf"
    assert out == {**out, "first": True, "vis_empty": 0, "vis_frame_empty": True, "after_vis": True, "gather_empty_0": 0,
                   "gather_empty_1": 0, "gather_frame_empty_0": True, "gather_frame_empty_1": True, "last": True}, out


@pytest.fixture(scope="module")
def ctx():
    c = Context(0)
    yield c
    c.close()


def test_packed_sort_sizes_its_payload_field_from_the_mesh_not_from_the_centres(ctx):
    """a bound sorter's payload is a position in the MESH's storage order (< mesh.uploaded), which needs more
    bits than sorter.uploaded - 1 when fewer centres than mesh splats have been uploaded."""
    n_mesh, n_sort = 5000, 1900                                # 13 bits of payload, 11 bits of list index
    scene = helpers.small_scene(n_mesh, 0, seed=17)
    cam = camera.demo_camera("garden", 256, 144)
    ci = util.integer_centers(scene.centers)
    mesh = SplatMesh(ctx, n_mesh, 0).build(scene.centers, scene.cov, scene.rgba, None)
    w = create_sort_worker(ctx, n_mesh)
    w.post_message({"centers": ci[:n_sort], "range": {"from": 0, "to": n_sort - 1, "count": n_sort}})
    mesh.use_sorter_result(w, n_sort)                          # binds: payloads become mesh positions
    mesh.set_camera(cam)
    expect = oracle.sort_indexes(np.arange(n_sort, dtype=np.uint32), ci[:n_sort], cam.sort_mvp())
    reply = w.post_message({"sort": {"modelViewProj": cam.sort_mvp(), "splatRenderCount": n_sort, "splatSortCount": n_sort}})
    np.testing.assert_array_equal(reply["sortedIndexes"], expect)
    w.sort_on_device(cam.sort_mvp(), n_sort)
    got, _ = mesh.render()
    c, cov, rgba, sh = helpers.oracle_inputs(scene)
    ocam = oracle.make_camera(cam.model_view(), cam.projection, cam.position, cam.width, cam.height, 0, 0)
    fb, q, amb, _ = oracle.render(ocam, c, cov, rgba, sh, expect)
    print(helpers.compare_frames(got, fb, amb, "partial centres, bound sorter"))
    w.terminate()
    mesh.dispose()


def test_fork_join_flag_reaches_the_library_and_is_refused_with_one_stream():
    """Context(fork_join=True) with single_stream left at None silently made a pipelined context."""
    c = Context(0, fork_join=True)
    c.close()
    with pytest.raises(ValueError):
        Context(0, single_stream=True, fork_join=True)


def test_a_culled_sort_between_a_planned_gather_and_its_sort_does_not_poison_the_keep_mask(ctx):
    """the planned gather zeroes the keep mask for its fused copy; a frustum-culled sort of another list in between
    rewrites it."""
    from oracle import tree_oracle
    from gaussiansplats3d_amd import SplatTree
    scene = helpers.small_scene(30000, 0, seed=61)
    n = scene.count
    ci = util.integer_centers(scene.centers)
    cam = camera.demo_camera("garden", 640, 360)
    other = camera.orbit_cameras("garden", 640, 360, 6)[3]
    tree = SplatTree(ctx, 8, 200).process_splat_mesh(scene.centers)
    leaves, _ = tree_oracle.build_tree(scene.centers, None, 8, 200)
    idx = tree_oracle.gather(leaves, cam.view, 50.0, 640, 360)
    w = create_sort_worker(ctx, n)
    w.post_message({"centers": ci, "range": {"from": 0, "to": n - 1, "count": n}})
    w.set_frustum_cull(True)
    tree.gather_scene_nodes_for_sort(cam, sort_worker=w, to_host=False)                    # planned: the mask is zeroed
    some = np.random.default_rng(3).permutation(n).astype(np.uint32)[: n // 2]
    between = w.post_message({"sort": {"modelViewProj": other.sort_mvp(), "splatRenderCount": some.size, "splatSortCount": some.size,
                                       "indexesToSort": some}})
    exp_between, _ = oracle.culled_sort(some, ci, other.sort_mvp())
    np.testing.assert_array_equal(between["sortedIndexes"], exp_between)
    reply = w.sort_gathered(cam.sort_mvp())
    expect, keep = oracle.culled_sort(idx, ci, cam.sort_mvp())
    np.testing.assert_array_equal(reply["sortedIndexes"], expect)
    w.terminate()
    tree.dispose()


def test_rop8_verification_refuses_a_pending_projection(ctx):
    """gs_mesh_project for the next frame swaps the record sets under the last draw's lists."""
    scene = helpers.small_scene(2000, 0, seed=5)
    cam = camera.demo_camera("garden", 128, 80)
    mesh = SplatMesh(ctx, scene.count, 0).build(scene.centers, scene.cov, scene.rgba, None)
    mesh.set_camera(cam)
    mesh.update_render_indexes(np.arange(scene.count, dtype=np.uint32), scene.count)
    mesh.render()
    mesh.rop8_window(0, 0, 32, 32)
    mesh.project()
    with pytest.raises(GsError):
        mesh.rop8_window(0, 0, 32, 32)
    mesh.render()
    mesh.rop8_window(0, 0, 32, 32)
    mesh.dispose()
