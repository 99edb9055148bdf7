"""How far may the camera move between two frames before the previous frame's blend statistics stop being a useful order for this
frame's bins?  One library, one process: the orbit at several angular steps per frame (a new pose EVERY frame, nothing synchronised in
between), drawn (a) with the engine's own gate and (b) with $GSPLAT_BLEND_ORDER_STALE=1 (always the previous draw's order) and (c) with
$GSPLAT_NO_BLEND_ORDER=1 (row-major always).

usage: python tools/motion_ab.py "C3 C2" [--steps "0.25 0.5 1 2 3 6"] [--frames 120]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from gaussiansplats3d_amd import Context, SplatMesh, camera, create_sort_worker, scenes, util


def pan_cameras(name, width, height, deg, frames):
    """The demo camera turning about its own position (its up axis), `deg` per frame, swinging +-20 degrees about the demo pose."""
    up, pos, look = (np.asarray(v, dtype=np.float64) for v in camera.DEMO_POSES[name])
    axis = up / np.linalg.norm(up)
    rel = look - pos
    out = []
    a, step = 0.0, np.radians(deg)
    for k in range(frames):
        r = rel * np.cos(a) + np.cross(axis, rel) * np.sin(a) + axis * np.dot(axis, rel) * (1.0 - np.cos(a))
        out.append(camera.PerspectiveCamera(width, height, tuple(pos), tuple(pos + r), tuple(up)))
        if abs(a + step) > np.radians(20.0):
            step = -step
        a += step
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("configs")
    ap.add_argument("--steps", default="0.25 0.5 1 2 3 6")
    ap.add_argument("--frames", type=int, default=120)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--pan", action="store_true", help="rotate the camera about its own position instead of orbiting its look-at point")
    a = ap.parse_args()
    for name in a.configs.split():
        cfg = scenes.CONFIGS[name]
        scene = scenes.make_config_scene(name)
        W, H = cfg["width"], cfg["height"]
        N = scene.count
        ctx = Context(0, single_stream=True)
        w = create_sort_worker(ctx, N)
        w.post_message({"centers": util.integer_centers(scene.centers), "range": {"from": 0, "to": N - 1, "count": N}})
        mesh = SplatMesh(ctx, N, scene.sh_degree, scene.cov_half).build(scene.centers, scene.cov, scene.rgba,
                                                                        scene.sh if scene.sh_degree else None)
        mesh.use_sorter_result(w, N)
        for deg in (float(x) for x in a.steps.split()):
            poses = int(round(360.0 / deg))
            cams = (pan_cameras(cfg["pose"], W, H, deg, a.frames) if a.pan else camera.orbit_cameras(cfg["pose"], W, H, poses)[: a.frames])
            mvps = [c.sort_mvp() for c in cams]
            for rnd in range(a.rounds):
                row = []
                for label, env in (("gate", {}), ("stale", {"GSPLAT_BLEND_ORDER_STALE": "1"}), ("row-major", {"GSPLAT_NO_BLEND_ORDER": "1"})):
                    for k in ("GSPLAT_BLEND_ORDER_STALE", "GSPLAT_NO_BLEND_ORDER"):
                        os.environ.pop(k, None)
                    os.environ.update(env)
                    for c, m in list(zip(cams, mvps))[:8]:                # warm: buffers grown, statistics of the neighbourhood
                        mesh.set_camera(c); w.sort_on_device(m, N); mesh.render(to_host=False, want_stats=False)
                    ctx.synchronize()
                    t0 = time.perf_counter()
                    for c, m in zip(cams, mvps):
                        mesh.set_camera(c); w.sort_on_device(m, N); mesh.render(to_host=False, want_stats=False)
                    ctx.synchronize()
                    row.append("%s %.4f" % (label, (time.perf_counter() - t0) / len(cams) * 1e3))
                print("%-4s %s %5.2f deg/frame (%d frames): %s ms/frame" % (name, "pan  " if a.pan else "orbit", deg, len(cams), " | ".join(row)), flush=True)
        for k in ("GSPLAT_BLEND_ORDER_STALE", "GSPLAT_NO_BLEND_ORDER"):
            os.environ.pop(k, None)
        w.terminate(); mesh.dispose(); ctx.close()
        del scene


if __name__ == "__main__":
    main()
