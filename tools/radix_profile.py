"""Phase timeline of the depth sort's scatter kernels (variant built with -DGS_RADIX_PROFILE -fgpu-rdc is NOT needed: the
profile array lives in sorter.hip; selected through GSPLAT_HIP_LIB).  usage: python tools/radix_profile.py [C3] [frames]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gaussiansplats3d_amd import Context, camera, create_sort_worker, scenes, util, _lib
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
cfg = scenes.CONFIGS[name]
scene = scenes.make_config_scene(name)
cam = camera.demo_camera(cfg["pose"], cfg["width"], cfg["height"])
N = scene.count
ctx = Context(0, single_stream=True)
w = create_sort_worker(ctx, N)
w.post_message({"centers": util.integer_centers(scene.centers), "range": {"from": 0, "to": N - 1, "count": N}})
frames = len(sys.argv) > 2 and sys.argv[2] == "frames"      # whole frames (sort -> draw) instead of sorts alone
if frames:
    from gaussiansplats3d_amd import SplatMesh
    mesh = SplatMesh(ctx, N, scene.sh_degree, scene.cov_half).build(scene.centers, scene.cov, scene.rgba, scene.sh if scene.sh_degree else None)
    mesh.set_camera(cam)
    w.sort_on_device(cam.sort_mvp(), N)
    mesh.use_sorter_result(w, N)
for _ in range(40 if frames else 4):
    w.sort_on_device(cam.sort_mvp(), N)
    if frames:
        mesh.render(to_host=False, want_stats=False)
ctx.synchronize()
buf = np.zeros((2, 1024, 10), dtype=np.uint64)
lib = _lib.load()
lib.gs_debug_radix_prof.argtypes = [C.c_void_p]
assert lib.gs_debug_radix_prof(buf.ctypes.data) == 0
names = ["prologue (offset tables)", "tile 0: fetch + decode", "zero + rank (LDS atomics)", "digit offsets (block scan)", "reorder in LDS", "stores"]
for slot in range(2):
    b = buf[slot].astype(np.int64)
    used = b[:, 0] > 0
    b = b[used]
    s = b[:, 0].min()
    print(f"{name} scatter pass {slot}: workgroups {used.sum()}  kernel span {(b[:, 7].max() - s) / 100:.1f} us, tiles per workgroup {b[:, 8].mean():.2f}, "
          f"workgroup lifetime mean {((b[:, 7] - b[:, 0]) / 100).mean():.1f} us, last start {(b[:, 0].max() - s) / 100:.1f} us")
    for k, n in enumerate(names):
        d = (b[:, k + 1] - b[:, k]) / 100.0
        print(f"    {n:30s} mean {d.mean():6.2f}  p90 {np.percentile(d, 90):6.2f}  max {d.max():6.2f} us")
    rest = (b[:, 7] - b[:, 6]) / 100.0
    print(f"    {'the remaining tiles':30s} mean {rest.mean():6.2f}  (per tile {(rest / np.maximum(b[:, 8] - 1, 1)).mean():.2f}) us")
b0, b1 = buf[0].astype(np.int64), buf[1].astype(np.int64)
u0, u1 = b0[:, 0] > 0, b1[:, 0] > 0
if u0.any() and u1.any():
    print(f"last workgroup of pass 0 done -> first workgroup of pass 1 starts: {(b1[u1, 0].min() - b0[u0, 7].max()) / 100:.1f} us "
          f"(between them: the end of pass 0 = write-back of its dirty L2 lines, the pass-1 histogram kernel ~5 us, two launches)")
