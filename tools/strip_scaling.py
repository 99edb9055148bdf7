"""Per-rank cost of the strip-sharded frame for N = 1, 2, 4, 8, measured on ONE GPU: every rank's frame (vertex stage with
its strip -> visibility-culled sort -> bin -> blend of the strip) is run in turn on a single-stream context and the slowest
rank is reported, i.e. what bench.py --gpus N would take without the framebuffer gather.
usage: python tools/strip_scaling.py [C3|C5|...] [steps] [N:r]     (N:r = only rank r of N, e.g. under rocprofv3)
       python tools/strip_scaling.py [C3|C5|...] [steps] sm       the SORT-MIDDLE split of DESIGN.md 7 costed from measured parts
                                                                    (8 ranks; the device path of that split is not built)"""
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
ctx = Context(0, single_stream=not os.environ.get("GS_STRIP_STREAMS"))   # $GS_STRIP_STREAMS=1: the default context (sorter / vertex stage on streams of their own), as bench.py --gpus N uses
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


def sort_middle_parts(n_ranks=8, link_GBps=float(os.environ.get("GS_LINK_GBPS", "77"))):
    """Per-rank frame of the sort-middle split at `n_ranks`, from parts measured on this one GPU:
       source      rank r projects and keys only the N / n splats it owns (original-index range): a mesh + sorter holding exactly
                   those splats: vertex stage over the full frame + min / max + survivor compaction (measured as project +
                   visibility-culled sort of that sub-scene: an UPPER bound, the source needs no radix passes);
       exchange    8-byte all-reduce of the key range (~15 us assumed) + all-to-all-v: rank d receives V_d x 48 bytes, 1 / n of it
                   over each of its links (bytes per link / link rate + 20 us of latency assumed);
       destination two stable radix passes over the V_d survivors of its strip (measured: a sorter holding exactly those splats,
                   full sort) + bin + entry sort + blend of the strip (measured stage times of the strip's draw)."""
    strips = gdist.balanced_row_strips(row_cost, n_ranks, align=int(os.environ.get("GS_STRIP_ALIGN", "2")))
    per = (N + n_ranks - 1) // n_ranks
    rows = []
    ctx.set_stage_timing(True)
    for r in range(n_ranks):
        b, e = r * per, min((r + 1) * per, N)
        # --- source side: the owned sub-scene
        sub = slice(b, e)
        w8 = create_sort_worker(ctx, e - b)
        w8.post_message({"centers": util.integer_centers(scene.centers[sub]), "range": {"from": 0, "to": e - b - 1, "count": e - b}})
        m8 = SplatMesh(ctx, e - b, scene.sh_degree, scene.cov_half).build(scene.centers[sub], scene.cov[sub], scene.rgba[sub],
                                                                         scene.sh[sub] if scene.sh_degree else None)
        m8.set_camera(cam)
        m8.use_sorter_result(w8, e - b)
        w8.set_visibility_cull(True)
        def src():
            m8.project(None)
            w8.sort_on_device(mvp, e - b)
        for _ in range(3):
            src()
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            src()
        ctx.synchronize()
        t_src = (time.perf_counter() - t0) / steps * 1e3
        w8.terminate(); m8.dispose()
        # --- destination side: the strip's survivors (from the full mesh's own vertex stage for that strip)
        strip = strips[r]
        w.set_visibility_cull(True)
        mesh.project(strip)
        w.sort_on_device(mvp, N)
        _, st = mesh.render(tile_rows=strip, to_host=False, want_stats=True)
        ss, _ = w.last_stats()
        V = int(ss.result_count)
        draw_ms = float(st.bin_ms + st.tile_sort_ms + st.blend_ms)
        ids = np.sort(w.debug_read(2, V).astype(np.int64)) if V else np.zeros(0, np.int64)
        wd = create_sort_worker(ctx, max(V, 1))
        if V:
            wd.post_message({"centers": util.integer_centers(scene.centers[ids]), "range": {"from": 0, "to": V - 1, "count": V}})
            for _ in range(3):
                wd.sort_on_device(mvp, V)
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                wd.sort_on_device(mvp, V)
            ctx.synchronize()
            t_sort = (time.perf_counter() - t0) / steps * 1e3
        else:
            t_sort = 0.0
        wd.terminate()
        w.set_visibility_cull(False)
        a2a = V * 48.0 / n_ranks / (link_GBps * 1e9) * 1e3 + 0.020
        total = t_src + 0.015 + a2a + t_sort + draw_ms
        rows.append((r, strip, V, t_src, a2a, t_sort, draw_ms, total))
        print(f"{name} sort-middle rank {r}/{n_ranks}: strip {strip} V={V}: source {t_src:.4f} + all-reduce 0.015 + all-to-all {a2a:.4f} "
              f"+ radix(V) {t_sort:.4f} + bin/entry sort/blend {draw_ms:.4f} = {total:.4f} ms", flush=True)
    ctx.set_stage_timing(False)
    return rows


if len(sys.argv) > 3 and sys.argv[3] == "sm":
    base = timed(None, False)
    rows = sort_middle_parts(8)
    slow = max(r[-1] for r in rows)
    strips8 = gdist.balanced_row_strips(row_cost, 8, align=int(os.environ.get("GS_STRIP_ALIGN", "2")))
    strip_ms = max(timed(s, True) for s in strips8)
    print(f"{name} 1 GPU full-sort frame {base:.4f} ms | 8 ranks, strips only (measured, slowest rank): {strip_ms:.4f} ms = {base / strip_ms:.2f}x | "
          f"8 ranks, sort-middle (from parts, slowest rank): {slow:.4f} ms = {base / slow:.2f}x   (framebuffer gather excluded in both; "
          f"link {os.environ.get('GS_LINK_GBPS', '77')} GB/s assumed)")
    sys.exit(0)
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
