"""Same-box A/B of the DEPTH SORT alone across several builds of libgsplat_hip.so, in ONE process.

usage: python tools/sort_ab.py "C3 C4" lib_a.so lib_b.so ... [--sorts 30] [--rounds 2] [--check]

Per (config, library): median / min of gs_sort_stats.device_ms over `sorts` full sorts of the config's splat count (static
integer mode, 16-bit buckets, the sorter bound to a mesh so that the payload is the mesh's storage position, as in a frame),
and a CRC of the sorted list, which must agree across the libraries (--check additionally compares it with the CPU oracle's
list: the reference's order, through tests/tools/sort_reference_crc.py).  Only the centres of a config's scene are generated (the same RNG stream as
scenes.make_config_scene), the mesh gets unit covariances: nothing but the sort is measured here.

GSPLAT_SORT_AB_MARK=1 prints a line per library with the number of sorts it ran, in order - tools/sort_ab_trace.py cuts a
rocprofv3 kernel trace of this script into per-library kernel tables with it."""
import argparse
import os
import sys
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from gaussiansplats3d_amd import _lib, camera, scenes, util


def use_library(path):
    import ctypes
    _lib._lib = None
    _lib.LIB_PATH = os.path.abspath(path)
    probe = ctypes.CDLL(_lib.LIB_PATH)
    if not hasattr(use_library, "all_symbols"):
        use_library.all_symbols = dict(_lib.SYMBOLS)
    _lib.SYMBOLS.clear()
    _lib.SYMBOLS.update({k: v for k, v in use_library.all_symbols.items() if hasattr(probe, k)})
    return _lib.load()


def config_centers(name):
    """The centres scenes.make_config_scene(name) would produce (same seeds, same RNG draws up to the centres)."""
    c = scenes.CONFIGS[name]
    n = c["n"]
    num = 3 if name in ("C5", "C3T") else int(name[1:])
    rng = np.random.default_rng(scenes.SEED_BASE + num)
    if name == "C4":
        return rng.uniform(-10.0, 10.0, size=(n, 3)).astype(np.float32)
    n_cl = int(round(0.8 * n))
    k = 4096
    cl_c = rng.uniform(-4.0, 4.0, size=(k, 3))
    cl_s = np.exp(rng.uniform(np.log(0.02), np.log(0.4), size=k))
    which = rng.integers(0, k, size=n_cl)
    centers = np.empty((n, 3), dtype=np.float32)
    centers[:n_cl] = (cl_c[which] + rng.normal(size=(n_cl, 3)) * cl_s[which, None]).astype(np.float32)
    centers[n_cl:] = rng.uniform(-8.0, 8.0, size=(n - n_cl, 3)).astype(np.float32)
    return centers[rng.permutation(n)]


def measure(centers, cfg, sorts, bound=True):
    from gaussiansplats3d_amd import Context, SplatMesh, create_sort_worker
    cam = camera.demo_camera(cfg["pose"], cfg["width"], cfg["height"])
    N = centers.shape[0]
    ctx = Context(0, single_stream=True)
    w = create_sort_worker(ctx, N)
    w.post_message({"centers": util.integer_centers(centers), "range": {"from": 0, "to": N - 1, "count": N}})
    mesh = None
    if bound:
        cov = np.tile(np.array([1e-4, 0, 0, 1e-4, 0, 1e-4], dtype=np.float32), (N, 1))
        rgba = np.full((N, 4), 255, dtype=np.uint8)
        mesh = SplatMesh(ctx, N, 0, False).build(centers, cov, rgba, None)
        mesh.use_sorter_result(w, N)
    mvp = cam.sort_mvp()
    ctx.set_stage_timing(True)                    # device_ms of sorts that stay on the device
    for _ in range(3):
        w.sort_on_device(mvp, N)
    ctx.synchronize()
    ms = []
    for _ in range(sorts):
        w.sort_on_device(mvp, N)
        s, _ = w.last_stats()
        ms.append(s.device_ms)
    out = w.debug_read(2, N)                      # the caller's splat indexes, whatever the payload was
    crc = zlib.crc32(out.tobytes())
    w.terminate()
    if mesh is not None:
        mesh.dispose()
    ctx.close()
    return float(np.median(ms)), float(np.min(ms)), crc, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("configs")
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--sorts", type=int, default=30)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--unbound", action="store_true", help="no mesh: the payload is the splat index itself")
    a = ap.parse_args()
    mark = os.environ.get("GSPLAT_SORT_AB_MARK")
    bad = 0
    for name in a.configs.split():
        cfg = scenes.CONFIGS[name]
        centers = config_centers(name)
        want = None
        if a.check:
            # (the checker lives with the tests: tests/tools/sort_reference_crc.py runs the sort oracle, this tool only gets the CRC)
            sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "tools"))
            from sort_reference_crc import reference_crc
            want = reference_crc(name, centers)
        res = {}
        for rnd in range(a.rounds):
            for lib in a.libs:
                env, path = lib.split("|") if "|" in lib else ("", lib)      # `ENV=V,ENV=V|lib.so`: that library under those switches
                sets = dict(kv.split("=", 1) for kv in env.split(",") if kv)
                old_env = {k: os.environ.get(k) for k in sets}
                os.environ.update(sets)
                use_library(path)
                med, mn, crc, _ = measure(centers, cfg, a.sorts, bound=not a.unbound)
                for k, v in old_env.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
                res.setdefault(lib, []).append((med, mn, crc))
                if mark:
                    print(f"MARK {name} {os.path.basename(lib)} sorts={a.sorts + 3}", flush=True)
        crcs = {r[0][2] for r in res.values()}
        for lib in a.libs:
            r = res[lib]
            ok = "" if want is None else ("  == oracle" if r[0][2] == want else "  != ORACLE")
            bad += (want is not None and r[0][2] != want)
            print(f"{name:4s} {lib.replace(os.path.dirname(lib.split(chr(124))[-1]) + os.sep, ''):40s} sort median " + " / ".join(f"{x[0]:.4f}" for x in r) + " ms   min " +
                  " / ".join(f"{x[1]:.4f}" for x in r) + f"   crc {r[0][2]:08x}{ok}", flush=True)
        if len(crcs) != 1:
            print(f"{name}: THE LIBRARIES DISAGREE ({len(crcs)} different sorted lists)", flush=True)
            bad += 1
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
