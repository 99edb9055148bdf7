"""-m gpu: GS_CAM_DEPTH_SLABS, the two-level composite (csrc/tile_blend.hip).  A pixel's composite is defined as the fold, in slab
order, of per-slab composites that each start at T = 1; a splat's slab is the top bits of its sort bucket.  Checked here:
the slab-mode frame meets the SAME tolerance against the fp32 oracle as the default frame, differs from it by rounding only,
is reproduced bit for bit by strips (with the visibility-culled sort every rank of a multi-GPU draw uses), does not depend on
the list-bin size, survives deep non-saturating piles (many slabs per bin) and fully opaque near slabs (farther slabs give up),
and degrades to one slab for host-supplied index lists."""
import os

import numpy as np
import pytest

import helpers
import oracle
from gaussiansplats3d_amd import Context, SplatMesh, camera, create_sort_worker, scenes, util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = Context(0)
    yield c
    c.close()


def _deep_pile(n, seed):
    """Most splats nearly transparent and stacked along the view direction: lists tens of thousands deep that never saturate,
    spread over the whole depth range (every slab is populated), plus an opaque cluster near the camera."""
    scene = helpers.small_scene(n, 1, seed, scale=0.05)
    rng = np.random.default_rng(seed + 1)
    scene.rgba[:, 3] = np.clip(np.round(255.0 / (1.0 + np.exp(-rng.normal(-3.5, 0.7, size=n)))), 1, 255).astype(np.uint8)
    pos, look = np.array(camera.DEMO_POSES["garden"][1]), np.array(camera.DEMO_POSES["garden"][2])
    fwd = (look - pos) / np.linalg.norm(look - pos)
    k = n // 10
    scene.centers[:k] = (pos + fwd * rng.uniform(0.6, 1.0, size=(k, 1)) + rng.normal(size=(k, 3)) * 0.08 +
                         np.array([0.35, 0.0, 0.0])).astype(np.float32)
    scene.rgba[:k, 3] = 255                                  # an opaque blob close to the camera: its slab saturates some bins
    return scene


def _draw(ctx, scene, cam, slabs, tile_rows=None, vis_cull=False, list_shift=None, monkeypatch=None):
    n = scene.count
    if list_shift is not None:
        monkeypatch.setenv("GSPLAT_LIST_SHIFT", str(list_shift))
    mesh = SplatMesh(ctx, n, scene.sh_degree, scene.cov_half, depth_slabs=slabs)
    if list_shift is not None:
        monkeypatch.delenv("GSPLAT_LIST_SHIFT")
    mesh.build(scene.centers, scene.cov, scene.rgba, scene.sh if scene.sh_degree else None)
    mesh.set_camera(cam)
    w = create_sort_worker(ctx, n)
    w.post_message({"centers": util.integer_centers(scene.centers), "range": {"from": 0, "to": n - 1, "count": n}})
    mesh.use_sorter_result(w, n)
    w.set_visibility_cull(vis_cull)
    frames = []
    for rows in ([None] if tile_rows is None else tile_rows):
        if vis_cull:
            mesh.project(rows)
        w.sort_on_device(cam.sort_mvp(), n)
        f, st = mesh.render(tile_rows=rows)
        frames.append(f)
    w.terminate()
    mesh.dispose()
    return frames, st


@pytest.mark.parametrize("make", ["scene_like", "deep_pile"])
def test_slab_mode_meets_the_oracle_tolerance_and_differs_by_rounding_only(ctx, make):
    W, H = 480, 270
    cam = camera.demo_camera("garden", W, H)
    scene = scenes.scene_like(60000, 1, 777) if make == "scene_like" else _deep_pile(40000, 31)
    (plain,), st0 = _draw(ctx, scene, cam, False)
    (slab,), st1 = _draw(ctx, scene, cam, True)
    assert st0.visible_splats == st1.visible_splats and plain[..., 3].any()
    diff = np.abs(plain.astype(int) - slab.astype(int))
    assert diff.max() <= 1 and (diff > 0).mean() < 0.01, (diff.max(), (diff > 0).mean())
    ci = util.integer_centers(scene.centers)
    order = oracle.sort_indexes(np.arange(scene.count, dtype=np.uint32), ci, cam.sort_mvp())
    c, cov, rgba, sh = helpers.oracle_inputs(scene)
    ocam = oracle.make_camera(cam.model_view(), cam.projection, cam.position, W, H, scene.sh_degree, scene.sh_degree)
    fb, q, amb, frags = oracle.render(ocam, c, cov, rgba, sh, order)
    print(helpers.compare_frames(slab, fb, amb, f"slab mode, {make}"))
    if make == "deep_pile":                                  # the pile really is deep: most of the lists get walked
        assert st1.splats_walked > 20 * W * H / 256


def test_slab_mode_strips_equal_the_full_frame_bit_for_bit(ctx, monkeypatch):
    """Strips (with the per-rank visibility-culled sort) and both list-bin sizes: a pixel's value depends only on its own ordered
    splats and their slabs."""
    W, H = 480, 270
    cam = camera.demo_camera("garden", W, H)
    scene = _deep_pile(40000, 32)
    rows = (H + 15) // 16
    (full,), _ = _draw(ctx, scene, cam, True)
    strips, _ = _draw(ctx, scene, cam, True, tile_rows=[(0, 5), (5, 6), (6, 11), (11, rows)], vis_cull=True)
    np.testing.assert_array_equal(np.concatenate(strips, axis=0), full)
    for shift in (1, 3):                                     # 32-px and 128-px list bins
        (f,), _ = _draw(ctx, scene, cam, True, list_shift=shift, monkeypatch=monkeypatch)
        np.testing.assert_array_equal(f, full)


def test_host_index_lists_fold_as_one_slab(ctx):
    """No sorter, no buckets: every entry is slab 0, the fold is the single fold.  The slab executors freeze a pixel at
    saturation (T <= 1e-4 -> 0) while the default kernel lets it run until its quadrant retires: fp32 rounding below 1e-4 of
    full scale, i.e. at most one 8-bit step on a rounding boundary."""
    W, H = 320, 200
    cam = camera.demo_camera("garden", W, H)
    scene = helpers.small_scene(8000, 1, seed=5)
    order = oracle.sort_indexes(np.arange(scene.count, dtype=np.uint32), util.integer_centers(scene.centers), cam.sort_mvp())
    frames = []
    for slabs in (False, True):
        mesh = SplatMesh(ctx, scene.count, 1, depth_slabs=slabs).build(scene.centers, scene.cov, scene.rgba, scene.sh)
        mesh.set_camera(cam)
        mesh.update_render_indexes(order, scene.count)
        frames.append(mesh.render()[0])
        mesh.dispose()
    d = np.abs(frames[0].astype(np.int32) - frames[1].astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() < 0.01, (int(d.max()), float((d > 0).mean()))


def test_slab_parallel_bins_give_the_same_bits_as_the_sequential_fold():
    """The first slab-mode draw of a mesh has no statistics: every bin is folded by one workgroup (MODE_SEQ).  From the second
    draw on the bins that cost far more than the mean are drawn by one workgroup per depth slab and merged by k_slab_fold
    (MODE_PART).  Same arithmetic, different executors: the frames must be identical, and the scene must really have deep bins."""
    import subprocess
    import sys
    # (the limits are read once per process: a child with a low bar, so that this small scene has "deep" bins)
    env = dict(os.environ, GSPLAT_DEEP_FACTOR="2", GSPLAT_DEEP_MIN="1000", GS_SLAB_CHILD="1")
    if not os.environ.get("GS_SLAB_CHILD"):
        out = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", __file__ + "::test_slab_parallel_bins_give_the_same_bits_as_the_sequential_fold"],
                             env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-1000:]
        return
    ctx = Context(0)
    W, H = 480, 270
    cam = camera.demo_camera("garden", W, H)
    n = 260000
    scene = _deep_pile(n, 33)
    rng = np.random.default_rng(9)
    pos, look = np.array(camera.DEMO_POSES["garden"][1]), np.array(camera.DEMO_POSES["garden"][2])
    fwd = (look - pos) / np.linalg.norm(look - pos)
    k = n // 2                                              # half of the splats in a thin translucent column: a few very deep bins
    scene.centers[-k:] = (pos + fwd * rng.uniform(1.5, 9.0, size=(k, 1)) + rng.normal(size=(k, 3)) * 0.03).astype(np.float32)
    scene.rgba[-k:, 3] = 2                                  # alpha 2/255: thousands of them before a pixel saturates
    mesh = SplatMesh(ctx, n, scene.sh_degree, depth_slabs=True).build(scene.centers, scene.cov, scene.rgba, scene.sh)
    mesh.set_camera(cam)
    w = create_sort_worker(ctx, n)
    w.post_message({"centers": util.integer_centers(scene.centers), "range": {"from": 0, "to": n - 1, "count": n}})
    mesh.use_sorter_result(w, n)
    frames, deep, cost = [], [], []
    for _ in range(4):
        w.sort_on_device(cam.sort_mvp(), n)
        frames.append(mesh.render()[0])
        deep.append(len(mesh.deep_bins()))
        c = mesh.blend_bin_stats()[..., 1].astype(np.int64)
        cost.append((int(c.mean()), int(c.max())))
    assert deep[0] == 0 and deep[1] > 0 and deep[2] > 0, (deep, cost)
    for f in frames[1:]:
        np.testing.assert_array_equal(f, frames[0])
    # ... and strips of that state still reproduce it
    rows = (H + 15) // 16
    strips = []
    for r in ((0, 7), (7, rows)):
        w.sort_on_device(cam.sort_mvp(), n)
        strips.append(mesh.render(tile_rows=r)[0])
    np.testing.assert_array_equal(np.concatenate(strips, axis=0), frames[0])
    w.terminate()
    mesh.dispose()
    ctx.close()
