"""Per-rank cost of the strip-sharded frame for N = 1, 2, 4, 8, measured on ONE GPU: every rank's frame (vertex stage with
its strip -> visibility-culled sort -> bin -> blend of the strip) is run in turn on a single-stream context and the slowest
rank is reported, i.e. what bench.py --gpus N would take without the framebuffer gather.
usage: python tools/strip_scaling.py [C3|C5|...] [steps] [N:r]     (N:r = only rank r of N, e.g. under rocprofv3)"""
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
ctx = Context(0, single_stream=True)
w = create_sort_worker(ctx, N)
w.post_message({"centers": util.integer_centers(scene.centers), "range": {"from": 0, "to": N - 1, "count": N}})
mesh = SplatMesh(ctx, N, scene.sh_degree, scene.cov_half).build(scene.centers, scene.cov, scene.rgba, scene.sh if scene.sh_degree else None)
mesh.set_camera(cam)
mvp = cam.sort_mvp()
for _ in range(2):
    w.sort_on_device(mvp, N)
    mesh.use_sorter_result(w, N)
    mesh.render(to_host=False, want_stats=True)
row_cost = mesh.tile_row_costs()


def timed(strip, culled):
    w.set_visibility_cull(culled)
    def frame():
        if culled:
            mesh.project(strip)
        w.sort_on_device(mvp, N)
        mesh.render(tile_rows=strip, to_host=False, want_stats=False)
    for _ in range(3):
        frame()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        frame()
    ctx.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


if len(sys.argv) > 3:
    n, r = (int(v) for v in sys.argv[3].split(":"))
    strip = gdist.balanced_row_strips(row_cost, n, align=int(os.environ.get("GS_STRIP_ALIGN", "2")))[r] if n > 1 else None
    print(f"{name} rank {r} of {n}: strip rows {strip}: {timed(strip, True):.4f} ms/frame over {steps} frames (+3 warm-up, +2 full)")
    sys.exit(0)
base = timed(None, False)
print(f"{name} full sort, 1 GPU (the headline path): {base:.4f} ms/frame")
one = None
for n in (1, 2, 4, 8):
    strips = gdist.balanced_row_strips(row_cost, n, align=int(os.environ.get("GS_STRIP_ALIGN", "2"))) if n > 1 else [None]
    ms = [timed(s, True) for s in strips]
    kept = []
    slow = max(ms)
    one = one or slow
    print(f"{name} N={n}: slowest rank {slow:.4f} ms (ranks {', '.join('%.3f' % v for v in ms)}) -> "
          f"{N / (slow * 1e-3) / 1e6:.0f} Msplats/s, speed-up {one / slow:.2f}x over N=1 of the same path, "
          f"{base / slow:.2f}x over the full-sort frame, without the gather")
