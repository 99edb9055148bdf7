"""Frame time and stage times per forced list-bin size (GSPLAT_LIST_SHIFT is read per mesh at creation): one process, one scene.
usage: python tools/list_shift_ab.py "C3 C2" "2 3 4 5" """
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gaussiansplats3d_amd import Context, SplatMesh, camera, create_sort_worker, scenes, util
for name in sys.argv[1].split():
    cfg = scenes.CONFIGS[name]
    scene = scenes.make_config_scene("C3" if name == "C5" else name)
    cam = camera.demo_camera(cfg["pose"], cfg["width"], cfg["height"])
    N = scene.count
    for shift in sys.argv[2].split():
        if shift == "auto":
            os.environ.pop("GSPLAT_LIST_SHIFT", None)
        else:
            os.environ["GSPLAT_LIST_SHIFT"] = shift
        ctx = Context(0, single_stream=True)
        w = create_sort_worker(ctx, N)
        w.post_message({"centers": util.integer_centers(scene.centers), "range": {"from": 0, "to": N - 1, "count": N}})
        mesh = SplatMesh(ctx, N, scene.sh_degree, scene.cov_half).build(scene.centers, scene.cov, scene.rgba, scene.sh if scene.sh_degree else None)
        mesh.set_camera(cam); mesh.use_sorter_result(w, N)
        mvp = cam.sort_mvp()
        for _ in range(4):
            w.sort_on_device(mvp, N); mesh.render(to_host=False, want_stats=True)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(40):
            w.sort_on_device(mvp, N); mesh.render(to_host=False, want_stats=False)
        ctx.synchronize()
        ms = (time.perf_counter() - t0) / 40 * 1e3
        ctx.set_stage_timing(True)
        st = []
        for _ in range(7):
            w.sort_on_device(mvp, N)
            _, r = mesh.render(to_host=False, want_stats=True)
            st.append((r.bin_ms, r.tile_sort_ms, r.blend_ms))
        b, e, bl = np.median(np.array(st), axis=0)
        print(f"{name} list shift {shift:>4s} ({int(r.list_bin_px)} px): frame {ms:.4f} ms | bin {b:.4f} esort {e:.4f} blend {bl:.4f} | entries {int(r.tile_entries)} scanned {int(r.entries_scanned)}", flush=True)
        w.terminate(); mesh.dispose(); ctx.close()
