"""Seeded inputs of the octree known-answer tests (shared by oracle/make_golden_tree.py and tests/test_tree.py)."""
import numpy as np


def make_case(name):
    rng = np.random.default_rng(abs(hash(name)) % (2 ** 31) if False else sum(ord(ch) for ch in name) * 7919)
    if name == "gauss5k":
        c = rng.normal(size=(5000, 3)).astype(np.float32)
        return dict(centers=c, max_depth=8, max_centers=1000)
    if name == "clusters40k":
        k = rng.uniform(-4, 4, size=(40, 3))
        c = (k[rng.integers(0, 40, 40000)] + rng.normal(size=(40000, 3)) * 0.2).astype(np.float32)
        return dict(centers=c, max_depth=8, max_centers=1000)
    if name == "grid_on_split_planes":      # quantised coordinates: many centres sit exactly on split planes
        c = rng.integers(-8, 9, size=(6000, 3)).astype(np.float32) * 0.25
        return dict(centers=c, max_depth=8, max_centers=200)
    if name == "deep_small_nodes":          # depth limit reached: leaves deeper than max_depth keep > max_centers
        c = (rng.normal(size=(3000, 3)) * 1e-3).astype(np.float32)
        c[:10] += 5.0
        return dict(centers=c, max_depth=3, max_centers=50)
    if name == "tiny":
        return dict(centers=np.array([[0, 0, 0], [1, 2, 3], [-1, 0.5, 2]], np.float32), max_depth=8, max_centers=2)
    if name == "coincident":                # degenerate box: every centre identical
        return dict(centers=np.tile(np.array([[0.5, -0.25, 2.0]], np.float32), (40, 1)), max_depth=4, max_centers=10)
    raise KeyError(name)


CASES = ["gauss5k", "clusters40k", "grid_on_split_planes", "deep_small_nodes", "tiny", "coincident"]
