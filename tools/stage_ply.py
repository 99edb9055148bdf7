"""Stage a BASELINE-sized INRIA-v1 .ply on disk so that the whole path FILE -> native reader -> sort -> draw runs at full size
(no capture ships with this repository, and the GPU box has no network): the stand-in scene of a configuration, re-expressed in
the trainer's own parameters (log-scales, quaternions, SH-3 coefficients, opacity logits: 62 floats per splat like the real
garden.ply) and written with `assets.write_ply`.  The CONTENT is synthetic - the same seeded distribution as the stand-in - the
FORMAT, the size and the reader are the real ones.

usage: python tools/stage_ply.py C3 /tmp/gsdata       -> /tmp/gsdata/garden.ply   (5.8 M splats, 1.44 GB)
       GS_DATA_DIR=/tmp/gsdata python bench.py ...     the bench line then says "data": "file:garden.ply" """
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from gaussiansplats3d_amd import assets, scenes

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
out_dir = sys.argv[2] if len(sys.argv) > 2 else "/tmp/gsdata"
name = scenes.REAL_FILES[cfg]
c = scenes.CONFIGS[cfg]
os.environ.pop("GS_DATA_DIR", None)
t0 = time.perf_counter()
stand_in = scenes.make_config_scene(cfg)
n = stand_in.count
rng = np.random.default_rng(scenes.SEED_BASE + 700 + int(cfg[1:2]))
# the stand-in's own parameter distributions (scenes._covariances / _appearance), in the trainer's terms
log_s = np.clip(rng.normal(np.log(0.015), 0.7, size=(n, 3)), np.log(1e-3), np.log(0.5)).astype(np.float32)
rot = rng.normal(size=(n, 4)).astype(np.float32)
opacity = rng.normal(0.5, 2.0, size=n).astype(np.float32)
f_dc = ((rng.integers(0, 256, size=(n, 3)) / 255.0 - 0.5) / 0.28209479177387814).astype(np.float32)   # colour = 0.5 + SH_C0 * f_dc
f_rest = rng.normal(0.0, 0.1, size=(n, 45)).astype(np.float32) if c["sh"] else None                  # SH-3 in the file, like the trainer's
data = assets.write_ply(stand_in.centers, log_s, rot, f_dc, opacity, f_rest)
os.makedirs(out_dir, exist_ok=True)
path = os.path.join(out_dir, name)
with open(path, "wb") as f:
    f.write(data)
t1 = time.perf_counter()
print(f"staged {path}: {n} splats, {len(data) / 1e6:.0f} MB, {len(data) // n} bytes per splat, in {t1 - t0:.1f} s")
del data
# the reader, timed (what scenes.load_real_scene runs)
t0 = time.perf_counter()
arr = assets.load(path, spherical_harmonics_degree=c["sh"])
t1 = time.perf_counter()
print(f"native reader: {arr['centers'].shape[0]} splats kept (alpha >= 1/255), SH degree {int(arr['sh_degree'])}, "
      f"{t1 - t0:.2f} s = {os.path.getsize(path) / (t1 - t0) / 1e9:.2f} GB/s of file")
