import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from gaussiansplats3d_amd import Context, SplatMesh, create_sort_worker, camera, scenes, util
name = "C3S"
cfg = scenes.CONFIGS[name]
scene = scenes.make_config_scene(name)
W, H = cfg["width"], cfg["height"]
N = scene.count
ctx = Context(0, single_stream=True)
w = create_sort_worker(ctx, N)
w.post_message({"centers": util.integer_centers(scene.centers), "range": {"from": 0, "to": N - 1, "count": N}})
mesh = SplatMesh(ctx, N, scene.sh_degree, scene.cov_half).build(scene.centers, scene.cov, scene.rgba, scene.sh)
mesh.use_sorter_result(w, N)
cams = camera.orbit_cameras(cfg["pose"], W, H, 360)
print("cold mesh: 8 frames of a moving camera without any synchronisation, then 120 more (tools/motion_ab.py's sequence)")
mcams = camera.orbit_cameras(cfg["pose"], W, H, 1440)
for rep in range(3):
    for c in mcams[:8]:
        mesh.set_camera(c); w.sort_on_device(c.sort_mvp(), N); mesh.render(to_host=False, want_stats=False)
    ctx.synchronize()
    print("  after the 8:", {k: (v.size if hasattr(v, "size") else v) for k, v in mesh.deep_pass_info().items()})
    t0 = time.perf_counter()
    for c in mcams[:120]:
        mesh.set_camera(c); w.sort_on_device(c.sort_mvp(), N); mesh.render(to_host=False, want_stats=False)
    ctx.synchronize()
    print("  %.4f ms/frame" % ((time.perf_counter() - t0) / 120 * 1e3), {k: (v.size if hasattr(v, "size") else v) for k, v in mesh.deep_pass_info().items()})
print("fixed pose, synchronised every frame")
mesh.set_camera(cams[0]); mvp = cams[0].sort_mvp()
for k in range(8):
    w.sort_on_device(mvp, N); _, st = mesh.render(to_host=False, want_stats=True)
    d = mesh.deep_pass_info()
    print(k, "frame %.3f blend %.3f bin %.3f entry-sort %.3f proj %.3f visible %d entries %d walked %d deep bins %d cand %d" % (st.device_ms, st.blend_ms, st.bin_ms, st.tile_sort_ms, st.project_ms, st.visible_splats, st.tile_entries, st.splats_walked, d["bins"].size, d["candidates"]))
print("moving 1 deg/frame, synchronised every frame")
for k in range(1, 24):
    c = cams[k]; mesh.set_camera(c); mvp = c.sort_mvp()
    w.sort_on_device(mvp, N); _, st = mesh.render(to_host=False, want_stats=True)
    d = mesh.deep_pass_info()
    print(k, "frame %.3f blend %.3f bin %.3f entry-sort %.3f proj %.3f visible %d entries %d walked %d deep bins %d cand %d" % (st.device_ms, st.blend_ms, st.bin_ms, st.tile_sort_ms, st.project_ms, st.visible_splats, st.tile_entries, st.splats_walked, d["bins"].size, d["candidates"]))
print("moving, free-running 60 frames")
ctx.synchronize(); t0 = time.perf_counter()
for k in range(24, 84):
    c = cams[k]; mesh.set_camera(c); w.sort_on_device(c.sort_mvp(), N); mesh.render(to_host=False, want_stats=False)
ctx.synchronize(); print("%.4f ms/frame" % ((time.perf_counter() - t0) / 60 * 1e3))
print("pose 84 fixed, free-running 60 frames")
c = cams[84]; mesh.set_camera(c)
for rep in range(2):
    ctx.synchronize(); t0 = time.perf_counter()
    for k in range(60):
        w.sort_on_device(c.sort_mvp(), N); mesh.render(to_host=False, want_stats=False)
    ctx.synchronize(); print("%.4f ms/frame" % ((time.perf_counter() - t0) / 60 * 1e3))
