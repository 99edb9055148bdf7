"""Frames with the per-splat frustum cull fused into the sort, for rocprofv3 (GSPLAT_SERIAL=1 gives clean per-kernel times)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussiansplats3d_amd import Context, SplatMesh, camera, create_sort_worker, scenes, util
cfg = scenes.CONFIGS["C3"]
scene = scenes.make_config_scene("C3")
cam = camera.demo_camera(cfg["pose"], cfg["width"], cfg["height"])
N = scene.count
ctx = Context(0)
w = create_sort_worker(ctx, N)
w.post_message({"centers": util.integer_centers(scene.centers), "range": {"from": 0, "to": N - 1, "count": N}})
mesh = SplatMesh(ctx, N, scene.sh_degree).build(scene.centers, scene.cov, scene.rgba, scene.sh)
mesh.set_camera(cam)
w.set_frustum_cull(True)
mesh.use_sorter_result(w, N)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    w.sort_on_device(cam.sort_mvp(), N)
    mesh.render(to_host=False, want_stats=False)
ctx.synchronize()
print("kept =", w.last_stats()[0].result_count)
