"""Per-workgroup timeline of k_bin_count / k_bin_emit (variant built with -DGS_BIN_PROFILE, selected through GSPLAT_HIP_LIB).
usage: python tools/bin_profile.py [C3] [N:r]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gaussiansplats3d_amd import Context, SplatMesh, camera, create_sort_worker, scenes, util, _lib
from gaussiansplats3d_amd import dist as gdist
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
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
for _ in range(3):
    w.sort_on_device(cam.sort_mvp(), N)
    mesh.render(to_host=False, want_stats=True)
strip = None
if len(sys.argv) > 2:
    n, r = (int(v) for v in sys.argv[2].split(":"))
    strip = gdist.balanced_row_strips(mesh.tile_row_costs(), n)[r]
    w.set_visibility_cull(True)
for _ in range(4):
    if strip is not None:
        mesh.project(strip)
    w.sort_on_device(cam.sort_mvp(), N)
    mesh.render(tile_rows=strip, to_host=False, want_stats=False)
ctx.synchronize()
buf = np.zeros((2, 2048, 4), dtype=np.uint64)
lib = _lib.load()
lib.gs_debug_bin_prof.argtypes = [C.c_void_p]
assert lib.gs_debug_bin_prof(buf.ctypes.data) == 0
for k, kname in enumerate(("k_bin_count", "k_bin_emit")):
    b = buf[k].astype(np.int64)
    used = b[:, 0] > 0
    t0, t1, t2, out = b[used, 0], b[used, 1], b[used, 2], b[used, 3]
    t2 = np.where(t2 > 0, t2, t1)          # workgroups that returned after the prologue
    s = t0.min()
    print(f"{name} {kname} strip={strip}: workgroups {used.sum()}  span {(t2.max() - s) / 100:.1f} us | first start..last start "
          f"{(t0.max() - s) / 100:.1f} us | prologue mean {((t1 - t0) / 100).mean():.2f} max {((t1 - t0) / 100).max():.2f} us | "
          f"body mean {((t2 - t1) / 100).mean():.2f} p90 {np.percentile((t2 - t1) / 100, 90):.2f} max {((t2 - t1) / 100).max():.2f} us | "
          f"out mean {out.mean():.0f} max {out.max()}")
    edges = np.linspace(0, (t2.max() - s) / 100, 13)
    for a, e in zip(edges[:-1], edges[1:]):
        act = (((t0 - s) / 100 < e) & ((t2 - s) / 100 > a)).sum()
        print(f"   {a:6.1f}-{e:6.1f} us active workgroups {act}")
    late = np.argsort(t2)[-5:][::-1]
    ids = np.nonzero(used)[0]
    for i in late:
        print(f"   block {ids[i]} start {(t0[i] - s) / 100:.1f} prologue end {(t1[i] - s) / 100:.1f} end {(t2[i] - s) / 100:.1f} out {out[i]}")
