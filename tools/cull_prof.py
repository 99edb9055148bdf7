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
for _ in range(12):
    r = tree.gather_scene_nodes_for_sort(cam, sort_worker=w, to_host=False)
    w.sort_gathered(cam.sort_mvp(), keep_on_device=True)
    mesh.use_sorter_result(w, r["splatRenderCount"])
    mesh.render(to_host=False, want_stats=False)
ctx.synchronize()
print("R =", r["splatRenderCount"])
