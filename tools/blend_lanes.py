"""What fraction of the blend's evaluated pixel lanes can hit anything: k_tile_blend built with -DGS_BLEND_PROFILE
(gaussiansplats3d_amd/csrc/libgsplat_hip_blendprof.so, `make -C gaussiansplats3d_amd/csrc blendprof`; selected through
GSPLAT_HIP_LIB) counts, per draw, the lanes it evaluates (128 per evaluated half quadrant), the lanes whose pixel passes the
fragment shader's `A <= 8` test, and those that pass it on a pixel still accumulating (T > 0).

usage: GSPLAT_HIP_LIB=.../libgsplat_hip_blendprof.so python tools/blend_lanes.py [C3 C3T C2 C5 C3S ...]   (one JSON line each)"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from gaussiansplats3d_amd import Context, SplatMesh, _lib, camera, create_sort_worker, scenes, util

WORDS, MAX_BINS = 14, 40960


def measure(ctx, lib, name, scene=None):
    cfg = scenes.CONFIGS[name]
    scene = scene if scene is not None else scenes.make_config_scene(name)
    cam = camera.demo_camera(cfg["pose"], cfg["width"], cfg["height"])
    N = scene.count
    w = create_sort_worker(ctx, N)
    w.post_message({"centers": util.integer_centers(scene.centers), "range": {"from": 0, "to": N - 1, "count": N}})
    mesh = SplatMesh(ctx, N, scene.sh_degree, scene.cov_half).build(scene.centers, scene.cov, scene.rgba,
                                                                    scene.sh if scene.sh_degree else None)
    mesh.set_camera(cam)
    mesh.use_sorter_result(w, N)
    for _ in range(3):                                      # settle the list-bin size and the entry capacity
        w.sort_on_device(cam.sort_mvp(), N)
        _, st = mesh.render(to_host=False, want_stats=True)
    ctx.synchronize()
    bins = min(((cfg["width"] + 31) // 32) * ((cfg["height"] + 31) // 32), MAX_BINS)
    buf = np.zeros((bins, WORDS), dtype=np.uint64)
    assert lib.gs_debug_blend_prof(buf.ctypes.data, bins) == 0
    ev, kept, useful, halves = (int(buf[:, k].sum()) for k in (8, 9, 10, 11))
    pairs = int(buf[:, 4:8].sum())
    out = {"config": name, "splats": N, "visible": int(st.visible_splats), "blend_ms": round(float(st.blend_ms), 4),
           "pairs_walked": pairs, "halves_evaluated": halves, "halves_per_pair": round(halves / max(pairs, 1), 4),
           "lanes_evaluated": ev, "lanes_kept": kept, "lanes_useful": useful,
           "lanes_kept_frac": round(kept / max(ev, 1), 4), "lanes_useful_frac": round(useful / max(ev, 1), 4),
           # what the same draw evaluated before halves could be skipped: 256 lanes per (splat, quadrant) pair
           "lanes_kept_frac_of_whole_quadrants": round(kept / max(256 * pairs, 1), 4),
           # a walk by 8x8 blocks (four 16-lane groups of a wave on their own block's survivors, in step between two saturation
           # tests): iterations it would take per (splat, quadrant) pair walked today, and blocks a pair reaches (of 4)
           "block_walk_iterations": int(buf[:, 12].sum()), "block_walk_iterations_per_pair": round(int(buf[:, 12].sum()) / max(pairs, 1), 4),
           "blocks_per_pair": round(int(buf[:, 13].sum()) / max(pairs, 1), 4),
           "bins_sampled": bins, "entries_scanned": int(st.entries_scanned)}
    w.terminate()
    mesh.dispose()
    return out


def main():
    names = sys.argv[1:] or ["C3"]
    lib = _lib.load()
    if not hasattr(lib, "gs_debug_blend_prof"):
        raise SystemExit("blend_lanes.py: this library was not built with -DGS_BLEND_PROFILE (set GSPLAT_HIP_LIB)")
    lib.gs_debug_blend_prof.argtypes = [C.c_void_p, C.c_uint]
    ctx = Context(0)
    cache = {}
    for name in names:
        key = "C3" if name == "C5" else name
        if key not in cache:
            cache.clear()
            cache[key] = scenes.make_config_scene(key)
        print(json.dumps(measure(ctx, lib, name, cache[key])), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
