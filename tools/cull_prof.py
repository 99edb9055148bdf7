"""Cull-on frames only (octree gather -> sort of the gathered list -> draw), for rocprofv3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gaussiansplats3d_amd import Context, SplatMesh, SplatTree, camera, create_sort_worker, scenes, util
cfg = scenes.CONFIGS["C3"]
scene = scenes.make_config_scene("C3")
cam = camera.demo_camera(cfg["pose"], cfg["width"], cfg["height"])
N = scene.count
ctx = Context(0)
w = create_sort_worker(ctx, N)
w.post_message({"centers": util.integer_centers(scene.centers), "range": {"from": 0, "to": N - 1, "count": N}})
mesh = SplatMesh(ctx, N, scene.sh_degree).build(scene.centers, scene.cov, scene.rgba, scene.sh)
mesh.set_camera(cam)
tree = SplatTree(ctx, 8, 1000).process_splat_mesh(scene.centers, alphas=scene.rgba[:, 3])
import time
w.set_frustum_cull(True)                                     # as bench.py's cull_on column: the per-splat cull on top
splats = int(tree.info().splats)
FRAMES = 40
for k in range(FRAMES + 4):
    if k == 4:
        ctx.synchronize(); t0 = time.perf_counter()
    tree.gather_scene_nodes_for_sort(cam, sort_worker=w, to_host=False, asynchronous=True)
    w.sort_gathered(cam.sort_mvp(), keep_on_device=True)
    mesh.use_sorter_result(w, splats)
    mesh.render(to_host=False, want_stats=False)
ctx.synchronize()
print("cull-on frame: %.4f ms (%d frames, asynchronous gather + fused frustum cull)" % ((time.perf_counter() - t0) / FRAMES * 1e3, FRAMES))
