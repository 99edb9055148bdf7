"""Stage times and blend statistics of the full C5 frame and of each of its eight balanced strips (stage events on, isolated
draws): what a rank's draw costs beside its share of the blend.  usage: python tools/strip_blend.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gaussiansplats3d_amd import Context, SplatMesh, camera, create_sort_worker, scenes, util
from gaussiansplats3d_amd import dist as gdist
cfg = scenes.CONFIGS["C5"]; W, H = cfg["width"], cfg["height"]
scene = scenes.make_config_scene("C5"); cam = camera.demo_camera(cfg["pose"], W, H); N = scene.count
ctx = Context(0, single_stream=True)
w = create_sort_worker(ctx, N)
w.post_message({"centers": util.integer_centers(scene.centers), "range": {"from": 0, "to": N - 1, "count": N}})
mesh = SplatMesh(ctx, N, scene.sh_degree, scene.cov_half).build(scene.centers, scene.cov, scene.rgba, scene.sh)
mesh.set_camera(cam)
w.sort_on_device(cam.sort_mvp(), N); mesh.use_sorter_result(w, N)
for _ in range(3):
    w.sort_on_device(cam.sort_mvp(), N); _, st = mesh.render(to_host=False, want_stats=True)
print("full: blend %.4f ms entries %d scanned %d walked %d visible %d bin %.4f esort %.4f" % (st.blend_ms, st.tile_entries, st.entries_scanned, st.splats_walked, st.visible_splats, st.bin_ms, st.tile_sort_ms))
strips = gdist.balanced_row_strips(mesh.tile_row_costs(), 8)
tot = 0
for s in strips:
    for _ in range(3):
        _, st = mesh.render(tile_rows=s, to_host=False, want_stats=True)
    tot += st.blend_ms
    print("strip %s: blend %.4f ms entries %d scanned %d walked %d visible %d bin %.4f esort %.4f" % (s, st.blend_ms, st.tile_entries, st.entries_scanned, st.splats_walked, st.visible_splats, st.bin_ms, st.tile_sort_ms))
print("sum of strip blends %.4f" % tot)
