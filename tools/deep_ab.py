"""The deep pass on / off on the same box (same pixels either way): one-stream frame time, isolated stage medians, the pass's own
numbers, and a check that the two frames are identical.
usage: python tools/deep_ab.py "C3S C3T C3 C2" [frames]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from gaussiansplats3d_amd import Context, SplatMesh, camera, create_sort_worker, scenes, util

names = (sys.argv[1] if len(sys.argv) > 1 else "C3S").split()
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ctx = Context(0, single_stream=True)
for name in names:
    cfg = scenes.CONFIGS[name]
    scene = scenes.make_config_scene("C3" if name == "C5" else name)
    cam = camera.demo_camera(cfg["pose"], cfg["width"], cfg["height"])
    N = scene.count
    w = create_sort_worker(ctx, N)
    w.post_message({"centers": util.integer_centers(scene.centers), "range": {"from": 0, "to": N - 1, "count": N}})
    mesh = SplatMesh(ctx, N, scene.sh_degree, scene.cov_half).build(scene.centers, scene.cov, scene.rgba, scene.sh if scene.sh_degree else None)
    mesh.set_camera(cam)
    mesh.use_sorter_result(w, N)
    mvp = cam.sort_mvp()
    ref = None
    for deep in (False, True, False, True):
        mesh.set_deep_pass(deep)
        for _ in range(4):
            w.sort_on_device(mvp, N)
            img, _ = mesh.render(to_host=True, want_stats=True)
        info = mesh.deep_pass_info()
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(frames):
            w.sort_on_device(mvp, N)
            mesh.render(to_host=False, want_stats=False)
        ctx.synchronize()
        ms = (time.perf_counter() - t0) / frames * 1e3
        st = {"bin": [], "esort": [], "blend": []}
        ctx.set_stage_timing(True)
        for _ in range(7):
            w.sort_on_device(mvp, N)
            _, r = mesh.render(to_host=False, want_stats=True)
            st["bin"].append(r.bin_ms); st["esort"].append(r.tile_sort_ms); st["blend"].append(r.blend_ms)
        ctx.set_stage_timing(False)
        diff = ""
        if ref is None:
            ref = img
        else:
            d = np.abs(ref.astype(int) - img.astype(int))
            diff = " | vs the first frame: max %d LSB, %d channel values differ" % (d.max(), int((d > 0).sum()))
        bs = mesh.blend_bin_stats()[..., 1].astype(np.int64) // 2
        print("%-4s deep pass %-3s frame %.4f ms = %7.1f Msplats/s | bin %.4f esort %.4f blend %.4f | walked %d | bins in the pass %d "
              "(over the threshold %d), chunks closed by per-bin workgroups %d, pool exhausted %s | pairs per bin mean %d max %d%s" %
              (name, "on" if deep else "off", ms, N / ms / 1e3, np.median(st["bin"]), np.median(st["esort"]), np.median(st["blend"]),
               r.splats_walked, len(info["bins"]), info["candidates"], info["chunks_closed_by_bins"], info["pool_exhausted"],
               bs.mean(), bs.max(), diff), flush=True)
    w.terminate(); mesh.dispose()
    del scene
