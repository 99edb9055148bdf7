"""CRC-32 of the reference-ordered list of a bench configuration's full sort (the sort oracle on the configuration's integer
centres and demo camera): what `tools/sort_ab.py --check` compares every library's list with.  The oracle stays on the tests'
side of the tree; the A/B tool only receives the number."""
import os
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np

import oracle
from gaussiansplats3d_amd import camera, scenes, util


def reference_crc(cfg_name, centers):
    cfg = scenes.CONFIGS[cfg_name]
    cam = camera.demo_camera(cfg["pose"], cfg["width"], cfg["height"])
    order = oracle.sort_indexes(np.arange(centers.shape[0], dtype=np.uint32), util.integer_centers(centers), cam.sort_mvp())
    return zlib.crc32(order.tobytes())
