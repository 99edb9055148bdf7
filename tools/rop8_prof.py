"""GS_DRAW_ROP8 frames only (sort -> vertex stage -> bin -> entry sort -> k_tile_blend_rop8), for rocprofv3.
usage: python tools/rop8_prof.py [C3] [bounded|full|fp32] [frames=40]   (fp32: the default draw mode, for comparison)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussiansplats3d_amd import Context, SplatMesh, camera, create_sort_worker, scenes, util
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
full = len(sys.argv) > 2 and sys.argv[2] == "full"
FRAMES = int(sys.argv[3]) if len(sys.argv) > 3 else 40
cfg = scenes.CONFIGS[name]
scene = scenes.make_config_scene(name)
cam = camera.demo_camera(cfg["pose"], cfg["width"], cfg["height"])
N = scene.count
ctx = Context(0, single_stream=True)
w = create_sort_worker(ctx, N)
w.post_message({"centers": util.integer_centers(scene.centers), "range": {"from": 0, "to": N - 1, "count": N}})
mesh = SplatMesh(ctx, N, scene.sh_degree, scene.cov_half).build(scene.centers, scene.cov, scene.rgba, scene.sh if scene.sh_degree else None)
mesh.set_camera(cam)
mesh.use_sorter_result(w, N)
fp32 = len(sys.argv) > 2 and sys.argv[2] == "fp32"
if not fp32:
    mesh.set_draw_mode(rop8=True, full=full)
for _ in range(3):                                  # settle the list-bin size, the bin order and the deep pass
    w.sort_on_device(cam.sort_mvp(), N)
    mesh.render(to_host=False, want_stats=True)
for k in range(FRAMES + 4):
    if k == 4:
        ctx.synchronize(); t0 = time.perf_counter()
    w.sort_on_device(cam.sort_mvp(), N)
    mesh.render(to_host=False, want_stats=False)
ctx.synchronize()
print("%s %s frame: %.4f ms (%d frames)" % (name, "fp32" if fp32 else "ROP8 full-walk" if full else "ROP8 bounded", (time.perf_counter() - t0) / FRAMES * 1e3, FRAMES))
