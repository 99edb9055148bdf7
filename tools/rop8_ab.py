"""The ROP8 draw modes (GS_DRAW_ROP8 = bounded walk, GS_DRAW_ROP8_FULL) with their bins ordered by the previous draw of the same mode
and view, and in row-major order ($GSPLAT_NO_BLEND_ORDER, read per draw): ms per frame, the blend's share, the frame's CRC.

usage: python tools/rop8_ab.py "C3 C2" """
import os
import sys
import time
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gaussiansplats3d_amd import Context, SplatMesh, create_sort_worker, camera, scenes, util
for name in sys.argv[1].split():
    cfg = scenes.CONFIGS[name]
    scene = scenes.make_config_scene(name)
    W, H = cfg["width"], cfg["height"]
    cam = camera.demo_camera(cfg["pose"], W, H)
    N = scene.count
    ctx = Context(0, single_stream=True)
    w = create_sort_worker(ctx, N)
    w.post_message({"centers": util.integer_centers(scene.centers), "range": {"from": 0, "to": N - 1, "count": N}})
    mesh = SplatMesh(ctx, N, scene.sh_degree, scene.cov_half).build(scene.centers, scene.cov, scene.rgba, scene.sh if scene.sh_degree else None)
    mesh.set_camera(cam)
    mesh.use_sorter_result(w, N)
    mvp = cam.sort_mvp()
    for rnd in range(2):
        for full in (False, True):
            for env in ("", "1"):
                if env: os.environ["GSPLAT_NO_BLEND_ORDER"] = "1"
                else: os.environ.pop("GSPLAT_NO_BLEND_ORDER", None)
                mesh.set_draw_mode(rop8=True, full=full)
                for _ in range(3):
                    w.sort_on_device(mvp, N); img, st = mesh.render(to_host=True, want_stats=True)
                ctx.synchronize(); t0 = time.perf_counter()
                F = 10 if full else 40
                for _ in range(F):
                    w.sort_on_device(mvp, N); mesh.render(to_host=False, want_stats=False)
                ctx.synchronize()
                print(name, "full" if full else "bounded", "no-order" if env else "order   ", "%.4f ms/frame blend %.4f crc %08x" % ((time.perf_counter() - t0) / F * 1e3, st.blend_ms, zlib.crc32(img.tobytes())), flush=True)
    os.environ.pop("GSPLAT_NO_BLEND_ORDER", None)
    w.terminate(); mesh.dispose(); ctx.close()
