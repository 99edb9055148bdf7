"""What one rank of an N-GPU strip-sharded draw costs, measured on ONE GPU: for every N the ranks' strips (balanced from
the probe frame's per-row entry counts, as bench.py does) are drawn one after the other, each as `steps` pipelined
frames of full sort + strip draw; the slowest rank bounds the N-GPU frame (the RGBA8 gather over xGMI is not included).
usage: python tools/strip_scaling.py [C3|C5] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gaussiansplats3d_amd import Context, SplatMesh, camera, create_sort_worker, scenes, util
from gaussiansplats3d_amd import dist as gdist
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
cfg = scenes.CONFIGS[name]
W, H = cfg["width"], cfg["height"]
scene = scenes.make_config_scene(name)
cam = camera.demo_camera(cfg["pose"], W, H)
N = scene.count
ctx = Context(0)
w = create_sort_worker(ctx, N)
w.post_message({"centers": util.integer_centers(scene.centers), "range": {"from": 0, "to": N - 1, "count": N}})
mesh = SplatMesh(ctx, N, scene.sh_degree, scene.cov_half).build(scene.centers, scene.cov, scene.rgba, scene.sh if scene.sh_degree else None)
mesh.set_camera(cam)
mvp = cam.sort_mvp()
w.sort_on_device(mvp, N)
mesh.use_sorter_result(w, N)
mesh.render(to_host=False, want_stats=True)
row_cost = mesh.tile_row_costs()
base = None
for world in (1, 2, 4, 8):
    strips = gdist.balanced_row_strips(row_cost, world) if world > 1 else [(0, (H + 15) // 16)]
    per_rank = []
    for s in strips:
        for _ in range(3):
            w.sort_on_device(mvp, N); mesh.render(tile_rows=s if world > 1 else None, to_host=False, want_stats=False)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            w.sort_on_device(mvp, N); mesh.render(tile_rows=s if world > 1 else None, to_host=False, want_stats=False)
        ctx.synchronize()
        per_rank.append((time.perf_counter() - t0) / steps * 1e3)
    worst = max(per_rank)
    base = base or worst
    print(f"{name} N={world}: slowest rank {worst:.4f} ms (ranks {', '.join('%.3f' % t for t in per_rank)}) -> "
          f"{N / worst / 1e3:.0f} Msplats/s, speed-up {base / worst:.2f}x without the gather")
