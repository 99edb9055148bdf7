"""Distribution of per-bin entry counts for the C3 stand-in (tail analysis of k_tile_blend)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
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
w.sort_on_device(cam.sort_mvp(), N)
mesh.use_sorter_result(w, N)
fb, st = mesh.render()
c = mesh.bin_entry_counts().astype(np.int64).ravel()
print("bins", c.size, "entries", c.sum(), "mean", c.mean(), "p50", np.percentile(c, 50), "p90", np.percentile(c, 90), "p99", np.percentile(c, 99), "max", c.max())
a = fb[..., 3]
print("alpha: saturated(255) frac", (a == 255).mean(), "zero frac", (a == 0).mean(), "mean", a.mean())
