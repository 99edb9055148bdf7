"""-m gpu: the chunked composite and the deep pass (csrc/tile_blend.hip).  The value of a pixel is defined per 16x16 quadrant:
the quadrant's ordered survivors are cut into chunks of 1024, each chunk composites from T = 1, the chunks are merged near -> far.
Below one chunk that is the plain composite (every other GPU test); here the lists are thousands of splats deep and do not
saturate.  Checked: the per-bin kernel (which closes chunks itself) and the deep pass (one wave per quadrant and chunk + a fold)
produce the SAME bits; strips of a multi-GPU draw (with the per-rank visibility-culled sort) and both list-bin sizes reproduce the
full frame bit for bit; the frame meets the same tolerance against the fp32 oracle as every other frame; lists too long for the
deep pass's tables stay with the per-bin kernel."""
import numpy as np
import pytest

import helpers
import oracle
from gaussiansplats3d_amd import Context, SplatMesh, camera, create_sort_worker, util

pytestmark = pytest.mark.gpu
CHUNK = 1024


@pytest.fixture(scope="module")
def ctx():
    c = Context(0)
    yield c
    c.close()


def _pile(n, seed, lateral=0.12, alpha_max=3):
    """A column of nearly transparent splats along the view direction: a few quadrants see thousands of them and never saturate
    (alpha 1/255 .. 3/255), the rest of the frame sees an ordinary scene."""
    scene = helpers.small_scene(n, 1, seed, scale=0.05)
    rng = np.random.default_rng(seed + 1)
    pos, look = np.array(camera.DEMO_POSES["garden"][1]), np.array(camera.DEMO_POSES["garden"][2])
    fwd = (look - pos) / np.linalg.norm(look - pos)
    k = (n * 4) // 5
    scene.centers[:k] = (pos + fwd * rng.uniform(1.5, 6.0, size=(k, 1)) + rng.normal(size=(k, 3)) * lateral).astype(np.float32)
    scene.rgba[:k, 3] = rng.integers(1, alpha_max + 1, size=k).astype(np.uint8)
    return scene


class Rig:
    def __init__(self, ctx, scene, cam, list_shift=None, monkeypatch=None):
        n = scene.count
        if list_shift is not None:
            monkeypatch.setenv("GSPLAT_LIST_SHIFT", str(list_shift))
        self.mesh = SplatMesh(ctx, n, scene.sh_degree, scene.cov_half)
        if list_shift is not None:
            monkeypatch.delenv("GSPLAT_LIST_SHIFT")
        self.mesh.build(scene.centers, scene.cov, scene.rgba, scene.sh if scene.sh_degree else None)
        self.mesh.set_camera(cam)
        self.w = create_sort_worker(ctx, n)
        self.w.post_message({"centers": util.integer_centers(scene.centers), "range": {"from": 0, "to": n - 1, "count": n}})
        self.mesh.use_sorter_result(self.w, n)
        self.cam, self.n = cam, n

    def draw(self, rows=None, vis_cull=False):
        self.w.set_visibility_cull(vis_cull)
        if vis_cull:
            self.mesh.project(rows)
        self.w.sort_on_device(self.cam.sort_mvp(), self.n)
        return self.mesh.render(tile_rows=rows)

    def close(self):
        self.w.terminate()
        self.mesh.dispose()


def test_deep_pass_and_per_bin_kernel_produce_the_same_bits(ctx):
    W, H = 480, 270
    cam = camera.demo_camera("garden", W, H)
    rig = Rig(ctx, _pile(60000, 41), cam)
    rig.mesh.set_deep_pass(False)
    plain, st0 = rig.draw()
    info0 = rig.mesh.deep_pass_info()
    assert len(info0["bins"]) == 0 and not info0["pool_exhausted"]
    assert info0["chunks_closed_by_bins"] >= 4, info0          # the pile really is several chunks deep
    assert st0.splats_walked > 8 * CHUNK
    rig.mesh.set_deep_pass(True)
    frames = [rig.draw() for _ in range(3)]                    # statistics -> candidates (mapped host word) -> the deep pass runs
    info = rig.mesh.deep_pass_info()
    assert len(info["bins"]) >= 1 and info["candidates"] >= len(info["bins"]), info
    for f, st in frames:
        np.testing.assert_array_equal(f, plain)
    # the quadrants the pass took over no longer close chunks in the per-bin kernel; its statistics keep the bins selected
    assert info["chunks_closed_by_bins"] < info0["chunks_closed_by_bins"]
    again, st = rig.draw()
    np.testing.assert_array_equal(again, plain)
    assert len(rig.mesh.deep_pass_info()["bins"]) >= 1
    # (the waves of the pass cannot know that the chunks in front of theirs already saturated the quadrant: they composite what
    # the merge then ignores - never less than the per-bin kernel walks)
    assert int(st.splats_walked) >= int(st0.splats_walked)
    rig.close()


def test_chunked_frame_meets_the_oracle_tolerance(ctx):
    W, H = 320, 200
    cam = camera.demo_camera("garden", W, H)
    scene = _pile(30000, 43, lateral=0.06)
    rig = Rig(ctx, scene, cam)
    for _ in range(3):
        frame, st = rig.draw()
    assert len(rig.mesh.deep_pass_info()["bins"]) >= 1
    rig.close()
    ci = util.integer_centers(scene.centers)
    order = oracle.sort_indexes(np.arange(scene.count, dtype=np.uint32), ci, cam.sort_mvp())
    c, cov, rgba, sh = helpers.oracle_inputs(scene)
    ocam = oracle.make_camera(cam.model_view(), cam.projection, cam.position, W, H, scene.sh_degree, scene.sh_degree)
    fb, q, amb, frags = oracle.render(ocam, c, cov, rgba, sh, order)
    print(helpers.compare_frames(frame, fb, amb, "deep pile, chunked composite"))


def test_strips_and_list_bin_sizes_reproduce_the_full_frame_bit_for_bit(ctx, monkeypatch):
    """A pixel's value depends only on its quadrant's own ordered survivors: not on the strip a rank draws (odd tile-row cuts that
    split bins), not on the list-bin size, not on which executor took the bin."""
    W, H = 480, 270
    cam = camera.demo_camera("garden", W, H)
    scene = _pile(60000, 47)
    rows = (H + 15) // 16
    cuts = [(0, 5), (5, 6), (6, 11), (11, rows)]
    rig = Rig(ctx, scene, cam)
    for _ in range(3):
        full, _ = rig.draw()
    assert len(rig.mesh.deep_pass_info()["bins"]) >= 1
    for _ in range(3):                                         # every strip warms its own statistics up to its own deep pass
        strips = [rig.draw(r, vis_cull=True)[0] for r in cuts]
        np.testing.assert_array_equal(np.concatenate(strips, axis=0), full)
    rig.close()
    for shift in (1, 3):                                       # 32-px and 128-px list bins
        r2 = Rig(ctx, scene, cam, list_shift=shift, monkeypatch=monkeypatch)
        for _ in range(3):
            f, _ = r2.draw()
            np.testing.assert_array_equal(f, full)
        r2.close()


def test_lists_longer_than_the_deep_tables_stay_with_the_per_bin_kernel(ctx, monkeypatch):
    """More than 65536 entries in one 128-px list: the bins of that list are named by the selection but every kernel of the deep
    pass leaves them alone."""
    W, H = 256, 144
    cam = camera.demo_camera("garden", W, H)
    scene = _pile(90000, 53, lateral=0.03, alpha_max=1)
    rig = Rig(ctx, scene, cam, list_shift=3, monkeypatch=monkeypatch)
    rig.mesh.set_deep_pass(False)
    plain, st = rig.draw()
    assert int(rig.mesh.bin_entry_counts().max()) > 65536
    rig.mesh.set_deep_pass(True)
    for _ in range(3):
        f, _ = rig.draw()
        np.testing.assert_array_equal(f, plain)
    rig.close()


def test_an_exhausted_partial_pool_is_flagged_and_still_composites(tmp_path):
    """The per-bin kernel closes chunks into a finite pool.  With the pool shrunk to 2 slots ($GSPLAT_POOL_SLOTS, read once per
    process: a child process) most quadrants of the pile go on as one long chunk: the draw says so and the frame is the same
    composite up to fp32 rounding."""
    import json
    import os
    import subprocess
    import sys
    script = f"""
import json, sys
sys.path.insert(0, {os.path.dirname(os.path.abspath(__file__))!r}); sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})
import numpy as np
import test_gpu_deep as T
from gaussiansplats3d_amd import Context, camera
c = Context(0)
rig = T.Rig(c, T._pile(60000, 41), camera.demo_camera("garden", 480, 270))
rig.mesh.set_deep_pass(False)
f, st = rig.draw()
np.save({str(tmp_path / 'frame.npy')!r}, f)
info = rig.mesh.deep_pass_info()
info["stats_flags"] = int(st.flags)                  # gs_render_stats.flags: the draw itself says so (GS_DRAW_POOL_EXHAUSTED)
print(json.dumps(info, default=lambda o: o.tolist() if hasattr(o, 'tolist') else o))
rig.close(); c.close()
"""
    infos = []
    for slots in ("2", None):
        env = dict(os.environ)
        if slots:
            env["GSPLAT_POOL_SLOTS"] = slots
        out = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        infos.append((json.loads(out.stdout.strip().splitlines()[-1]), np.load(tmp_path / "frame.npy")))
    (small, f_small), (full, f_full) = infos
    assert small["pool_exhausted"] and not full["pool_exhausted"] and full["chunks_closed_by_bins"] >= 4
    assert (small["stats_flags"] & 1) == 1 and (full["stats_flags"] & 1) == 0
    d = np.abs(f_small.astype(np.int32) - f_full.astype(np.int32))
    assert d.max() <= 1, int(d.max())
