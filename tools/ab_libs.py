"""Same-box A/B of several builds of libgsplat_hip.so in ONE process (box-to-box variance is ~10 %): every scene is
generated once, every library is dlopen'ed under its own path and measured in turn, `rounds` times interleaved.

usage: python tools/ab_libs.py "C3 C3T" lib_a.so lib_b.so ... [--frames 40] [--rounds 2]
prints per (config, library): one-stream ms per frame (unbracketed region), the isolated-frame stage medians and the CRC-32 of the
frame (libraries that must draw the same bits can be compared at a glance)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from gaussiansplats3d_amd import _lib, camera, scenes, util


def use_library(path):
    """(an older build may lack entry points added since: they are dropped from the table for that library only)"""
    import ctypes
    _lib._lib = None
    _lib.LIB_PATH = os.path.abspath(path)
    probe = ctypes.CDLL(_lib.LIB_PATH)
    if not hasattr(use_library, "all_symbols"):
        use_library.all_symbols = dict(_lib.SYMBOLS)
    _lib.SYMBOLS.clear()
    _lib.SYMBOLS.update({k: v for k, v in use_library.all_symbols.items() if hasattr(probe, k)})
    return _lib.load()


def measure(scene, cfg, frames):
    from gaussiansplats3d_amd import Context, SplatMesh, create_sort_worker
    cam = camera.demo_camera(cfg["pose"], cfg["width"], cfg["height"])
    N = scene.count
    ctx = Context(0, single_stream=True)
    w = create_sort_worker(ctx, N)
    w.post_message({"centers": util.integer_centers(scene.centers), "range": {"from": 0, "to": N - 1, "count": N}})
    mesh = SplatMesh(ctx, N, scene.sh_degree, scene.cov_half).build(scene.centers, scene.cov, scene.rgba,
                                                                    scene.sh if scene.sh_degree else None)
    mesh.set_camera(cam)
    mesh.use_sorter_result(w, N)
    mvp = cam.sort_mvp()
    for _ in range(3):
        w.sort_on_device(mvp, N)
        mesh.render(to_host=False, want_stats=True)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(frames):
        w.sort_on_device(mvp, N)
        mesh.render(to_host=False, want_stats=False)
    ctx.synchronize()
    ms = (time.perf_counter() - t0) / frames * 1e3
    st = {"sort": [], "project": [], "bin": [], "esort": [], "blend": []}
    ctx.set_stage_timing(True)
    for _ in range(9):
        w.sort_on_device(mvp, N)
        _, r = mesh.render(to_host=False, want_stats=True)
        s, _ = w.last_stats()
        st["sort"].append(s.device_ms); st["project"].append(r.project_ms); st["bin"].append(r.bin_ms)
        st["esort"].append(r.tile_sort_ms); st["blend"].append(r.blend_ms)
    out = {k: float(np.median(v)) for k, v in st.items()}
    out["ms"] = ms
    import zlib
    w.sort_on_device(mvp, N)
    img, _ = mesh.render(to_host=True, want_stats=True)
    out["crc"] = zlib.crc32(img.tobytes())
    out["halves"] = int(getattr(r, "halves_evaluated", 0))
    out["walked"] = int(r.splats_walked)
    w.terminate(); mesh.dispose(); ctx.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("configs")
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--frames", type=int, default=40)
    ap.add_argument("--rounds", type=int, default=2)
    a = ap.parse_args()
    for name in a.configs.split():
        cfg = scenes.CONFIGS[name]
        scene = scenes.make_config_scene("C3" if name == "C5" else name)
        for rnd in range(a.rounds):
            for item in a.libs:
                env, lib = item.split("|") if "|" in item else ("", item)       # `ENV=V,ENV=V|lib.so`: that library under those switches
                sets = dict(kv.split("=", 1) for kv in env.split(",") if kv)
                old = {k: os.environ.get(k) for k in sets}
                os.environ.update(sets)
                use_library(lib)
                r = measure(scene, cfg, a.frames)
                for k, v in old.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
                lib = (env + " " if env else "") + os.path.basename(lib)
                print("%-4s %-40s frame %.4f ms | sort %.4f project %.4f bin %.4f esort %.4f blend %.4f | walked %d halves %d | frame crc %08x" %
                      (name, lib, r["ms"], r["sort"], r["project"], r["bin"], r["esort"], r["blend"],
                       r["walked"], r["halves"], r["crc"]), flush=True)
        del scene


if __name__ == "__main__":
    main()
