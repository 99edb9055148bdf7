import sys, os, time
sys.path.insert(0, '/root/repo')
import numpy as np
from gaussiansplats3d_amd import Context, SplatMesh, camera, create_sort_worker, scenes, util
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
cfg = scenes.CONFIGS[name]
scene = scenes.make_config_scene(name)
cam = camera.demo_camera(cfg["pose"], cfg["width"], cfg["height"])
N = scene.count
for mode in ("single", "fork", "default", "single", "fork"):
    ctx = Context(0, single_stream=(mode == "single"), fork_join=(mode == "fork"))
    w = create_sort_worker(ctx, N)
    w.post_message({"centers": util.integer_centers(scene.centers), "range": {"from": 0, "to": N - 1, "count": N}})
    mesh = SplatMesh(ctx, N, scene.sh_degree, scene.cov_half).build(scene.centers, scene.cov, scene.rgba, scene.sh if scene.sh_degree else None)
    mesh.set_camera(cam); mesh.use_sorter_result(w, N)
    mvp = cam.sort_mvp()
    for _ in range(5):
        w.sort_on_device(mvp, N); mesh.render(to_host=False, want_stats=True)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(60):
        w.sort_on_device(mvp, N); mesh.render(to_host=False, want_stats=False)
    ctx.synchronize()
    print(f"{name} {mode:8s} {(time.perf_counter()-t0)/60*1e3:.4f} ms/frame", flush=True)
    w.terminate(); mesh.dispose(); ctx.close()
