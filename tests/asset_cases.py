"""Seeded .ply inputs of the asset known-answer tests (shared by oracle/make_golden_ply.py and tests/test_assets.py)."""
import numpy as np

from gaussiansplats3d_amd import assets

CASES = {"sh2_45": 45, "sh1_9": 9, "sh0": 0, "sh2_24": 24, "sh1_with_uchar": 9, "odd_27": 27}


def make_case(name):
    n_rest = CASES[name]
    rng = np.random.default_rng(1000 + n_rest + len(name))
    n = 64
    cols = dict(centers=rng.normal(size=(n, 3)).astype(np.float32), log_scales=rng.normal(-4, 1, size=(n, 3)).astype(np.float32),
                rotations=rng.normal(size=(n, 4)).astype(np.float32), f_dc=rng.normal(0, 1.2, size=(n, 3)).astype(np.float32),
                opacity=rng.normal(0, 3, size=n).astype(np.float32),
                f_rest=rng.normal(0, 0.2, size=(n, n_rest)).astype(np.float32) if n_rest else None)
    cols["rotations"][3] = 0.0                        # a zero quaternion: normalize() -> (0,0,0,1)
    extra = rng.integers(0, 256, n).astype(np.uint8) if name == "sh1_with_uchar" else None
    data = assets.write_ply(cols["centers"], cols["log_scales"], cols["rotations"], cols["f_dc"], cols["opacity"], cols["f_rest"], extra)
    return data, cols
