"""What k_project costs when (a) the whole frame is drawn, (b) one strip of eight, (c) the camera looks away from the scene
(every splat fails the frustum test): the floor of the vertex stage.  HIP events around the kernel (gs_mesh_kernel_time)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gaussiansplats3d_amd import Context, SplatMesh, camera, scenes
from gaussiansplats3d_amd import dist as gdist
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
cfg = scenes.CONFIGS[name]
W, H = cfg["width"], cfg["height"]
scene = scenes.make_config_scene(name)
N = scene.count
ctx = Context(0, single_stream=True, stage_timing=True)
mesh = SplatMesh(ctx, N, scene.sh_degree, scene.cov_half).build(scene.centers, scene.cov, scene.rgba, scene.sh if scene.sh_degree else None)
up, pos, look = camera.DEMO_POSES[cfg["pose"]]
cams = {"demo pose": camera.demo_camera(cfg["pose"], W, H),
        "looking away": camera.PerspectiveCamera(W, H, pos, tuple(2 * np.asarray(pos) - np.asarray(look)), up)}
order = np.arange(N, dtype=np.uint32)


def measure(label, cam, strip):
    mesh.set_camera(cam)
    mesh.update_render_indexes(order, N)
    for _ in range(3):
        mesh.render(tile_rows=strip, to_host=False, want_stats=True)
    mesh.kernel_time(1, reset=True)            # (timed draws: the whole vertex stage - block test + mask reset + k_project - is clock 1)
    vis = 0
    for _ in range(20):
        _, st = mesh.render(tile_rows=strip, to_host=False, want_stats=True)
        vis = st.visible_splats
    ms, n = mesh.kernel_time(1, reset=True)
    print(f"{name} k_project, {label:28s} strip={strip}: {ms / n * 1e3:7.1f} us  visible {vis}")


measure("demo pose", cams["demo pose"], None)
rows = (H + 15) // 16
measure("demo pose, middle eighth", cams["demo pose"], (rows * 4 // 8, rows * 5 // 8))
measure("looking away", cams["looking away"], None)
