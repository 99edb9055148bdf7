"""Per-bin timeline of k_tile_blend (variant built with -DGS_BLEND_PROFILE, selected through GSPLAT_HIP_LIB).
usage: python tools/blend_profile.py [C3] [r0:r1]      r0:r1 = only the strip of 16-px tile rows [r0, r1) (a rank's frame)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gaussiansplats3d_amd import Context, SplatMesh, camera, create_sort_worker, scenes, util, _lib
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
rows = tuple(int(v) for v in sys.argv[2].split(":")) if len(sys.argv) > 2 else None
cfg = scenes.CONFIGS[name]
scene = scenes.make_config_scene(name)
cam = camera.demo_camera(cfg["pose"], cfg["width"], cfg["height"])
N = scene.count
ctx = Context(0)
w = create_sort_worker(ctx, N)
w.post_message({"centers": util.integer_centers(scene.centers), "range": {"from": 0, "to": N - 1, "count": N}})
mesh = SplatMesh(ctx, N, scene.sh_degree, scene.cov_half).build(scene.centers, scene.cov, scene.rgba, scene.sh if scene.sh_degree else None)
mesh.set_camera(cam)
mesh.use_sorter_result(w, N)
for _ in range(5):
    w.sort_on_device(cam.sort_mvp(), N)
    _, st = mesh.render(tile_rows=rows, to_host=True, want_stats=True)       # (to_host: the deep pass's verdict reaches the next draw)
ctx.synchronize()
info = mesh.deep_pass_info()
bins = ((cfg["width"] + 31) // 32) * ((cfg["height"] + 31) // 32)
if rows:
    bins = ((cfg["width"] + 31) // 32) * ((rows[1] * 16 + 31) // 32 - (rows[0] * 16) // 32)
bins = min(bins, 40960)
buf = np.zeros((bins, 14), dtype=np.uint64)     # BLEND_PROF_WORDS per bin (tile_blend.hip)
lib = _lib.load()
lib.gs_debug_blend_prof.argtypes = [C.c_void_p, C.c_uint]
assert lib.gs_debug_blend_prof(buf.ctypes.data, bins) == 0
t0, t1, n = buf[:, 0].astype(np.int64), buf[:, 1].astype(np.int64), buf[:, 2].astype(np.int64)
walked = buf[:, 4:8].astype(np.int64)
batches = buf[:, 3].astype(np.int64)
need = (n + 255) // 256
print("batches staged per bin: mean %.2f (of %.2f in the list), p50 %d p90 %d max %d; bins that stop early: %.1f %%; entries staged / entries in lists: %.3f" %
      (batches.mean(), need.mean(), np.percentile(batches, 50), np.percentile(batches, 90), batches.max(),
       100.0 * (batches < need).mean(), np.minimum(batches * 256, n).sum() / max(n.sum(), 1)))
start = t0.min()
dur = (t1 - t0) / 100.0          # us (100 MHz)
rel0, rel1 = (t0 - start) / 100.0, (t1 - start) / 100.0
print(f"{name}: blend_ms(stats)={st.blend_ms:.4f} bins={bins} span={rel1.max():.1f} us  sum(dur)={dur.sum():.0f} us  "
      f"mean={dur.mean():.2f} max={dur.max():.1f} p99={np.percentile(dur, 99):.1f} p90={np.percentile(dur, 90):.1f}")
print("list length: mean %.0f max %d | walked per wave: mean %.0f max %d | walked/len %.3f" %
      (n.mean(), n.max(), walked.mean(), walked.max(), walked.sum() / max(4 * n.sum(), 1)))
# concurrency over time
edges = np.linspace(0, rel1.max(), 23)
for a, b in zip(edges[:-1], edges[1:]):
    act = ((rel0 < b) & (rel1 > a)).sum()
    print(f"  {a:6.1f}-{b:6.1f} us  active bins {act}")
late = np.argsort(rel1)[-8:]
for i in late[::-1]:
    print(f"  bin {i} ({i % ((cfg['width']+31)//32)},{i // ((cfg['width']+31)//32)}) start {rel0[i]:.1f} end {rel1[i]:.1f} dur {dur[i]:.1f} n {n[i]} walked {walked[i].tolist()}")

# the deep pass's units (one wave per bin, quadrant, chunk), if the last draw ran it
if len(info["bins"]) and hasattr(lib, "gs_debug_deep_prof"):
    CM, D = 32, 512
    ub = np.zeros((D * CM * 4, 4), dtype=np.uint64)
    lib.gs_debug_deep_prof.argtypes = [C.c_void_p]
    assert lib.gs_debug_deep_prof(ub.ctypes.data) == 0
    live = ub[:, 1] > 0
    u0, u1 = (ub[live, 0].astype(np.int64) - start) / 100.0, (ub[live, 1].astype(np.int64) - start) / 100.0
    ud, uw, un = u1 - u0, ub[live, 2].astype(np.int64), ub[live, 3].astype(np.int64)
    # (bins the pass drew keep their old words in the per-bin table: drop them from the per-bin numbers)
    drawn = np.ones(bins, bool); drawn[info["bins"][info["bins"] < bins]] = False
    print("deep pass: %d bins, %d units that composited something; unit duration us: mean %.1f p50 %.1f p90 %.1f max %.1f; start: first %.1f "
          "last %.1f; end of the last unit %.1f us | windows scanned per unit mean %.0f max %d | survivors per unit mean %.0f max %d" %
          (len(info["bins"]), live.sum(), ud.mean(), np.percentile(ud, 50), np.percentile(ud, 90), ud.max(), u0.min(), u0.max(), u1.max(),
           uw.mean(), uw.max(), un.mean(), un.max()))
    print("per-bin workgroups of the same launch: first start %.1f, last end %.1f us" % (rel0[drawn].min(), rel1[drawn].max()))
    slow = np.argsort(ud)[-6:]
    idx = np.nonzero(live)[0]
    for k in slow[::-1]:
        u = idx[k]
        print("  unit d=%d c=%d q=%d: %.1f us, start %.1f, windows %d, survivors %d" % (u // (4 * CM), (u // 4) % CM, u % 4, ud[k], u0[k], uw[k], un[k]))
