"""Same-box A/B of the blend's schedule under camera motion: several builds of libgsplat_hip.so (and / or environment switches) in
ONE process, per (config, library): the fixed demo pose back to back, the 60-pose orbit as a MOVING camera (a new pose every frame,
nothing synchronised in between, two laps), and the orbit's isolated frames (median / max; the first pose is drawn without any
statistics of that view).

usage: python tools/orbit_ab.py "C3 C3S" lib_a.so lib_b.so ... [--frames 40] [--rounds 2]
       an item `ENV=VALUE,ENV=VALUE|lib.so` runs that library under those environment switches (read when the mesh is created or
       per draw, e.g. GSPLAT_NO_BLEND_ORDER=1)."""
import argparse
import os
import sys
import time
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from gaussiansplats3d_amd import camera, scenes, util
from ab_libs import use_library


def measure(scene, cfg, frames):
    from gaussiansplats3d_amd import Context, SplatMesh, create_sort_worker
    W, H = cfg["width"], cfg["height"]
    cam = camera.demo_camera(cfg["pose"], W, H)
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
    fixed = (time.perf_counter() - t0) / frames * 1e3
    cams = camera.orbit_cameras(cfg["pose"], W, H, 60)
    mvps = [c.sort_mvp() for c in cams]
    iso, first = [], []
    for c, m in zip(cams, mvps):
        mesh.set_camera(c)
        ctx.synchronize()
        t0 = time.perf_counter()                     # the first frame of a pose: whatever the engine keeps between frames
        w.sort_on_device(m, N)                       # describes ANOTHER view (it may also grow the entry buffer)
        mesh.render(to_host=False, want_stats=True)
        ctx.synchronize()
        first.append((time.perf_counter() - t0) * 1e3)
        t0 = time.perf_counter()
        w.sort_on_device(m, N)
        mesh.render(to_host=False, want_stats=False)
        ctx.synchronize()
        iso.append((time.perf_counter() - t0) * 1e3)
    # every 5th pose as a FIXED pose, back to back: what the orbit would cost if motion itself were free
    per_pose = []
    for c, m in list(zip(cams, mvps))[::5]:
        mesh.set_camera(c)
        for _ in range(3):
            w.sort_on_device(m, N)
            mesh.render(to_host=False, want_stats=False)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            w.sort_on_device(m, N)
            mesh.render(to_host=False, want_stats=False)
        ctx.synchronize()
        per_pose.append((time.perf_counter() - t0) / 20 * 1e3)
    ctx.synchronize()
    t0 = time.perf_counter()
    for lap in range(2):
        for c, m in zip(cams, mvps):
            mesh.set_camera(c)
            w.sort_on_device(m, N)
            mesh.render(to_host=False, want_stats=False)
    ctx.synchronize()
    moving = (time.perf_counter() - t0) / (2 * len(cams)) * 1e3
    mesh.set_camera(cam)
    w.sort_on_device(mvp, N)
    img, _ = mesh.render(to_host=True, want_stats=True)
    out = {"fixed": fixed, "moving": moving, "iso_median": float(np.median(iso)), "iso_mean": float(np.mean(iso)),
           "iso_max": float(np.max(iso)), "first_median": float(np.median(first)), "first_max": float(np.max(first)),
           "pose_fixed_mean": float(np.mean(per_pose)), "pose_fixed_max": float(np.max(per_pose)),
           "crc": zlib.crc32(img.tobytes())}
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
                env, lib = item.split("|") if "|" in item else ("", item)
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
                print("%-4s %-44s fixed %.4f ms | moving camera %.4f ms (the same poses held fixed, back to back: mean %.4f max %.4f) | isolated frames median %.4f mean %.4f max %.4f | first frame of a pose "
                      "median %.4f max %.4f | demo-pose frame crc %08x" %
                      (name, (env + " " if env else "") + os.path.basename(lib), r["fixed"], r["moving"], r["pose_fixed_mean"], r["pose_fixed_max"], r["iso_median"], r["iso_mean"],
                       r["iso_max"], r["first_median"], r["first_max"], r["crc"]), flush=True)
        del scene


if __name__ == "__main__":
    main()
