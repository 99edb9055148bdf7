"""CPU tier: the raster oracle's destination semantics - the reference's `depthTest: true, depthWrite: false` + NormalBlending over
what other scene geometry drew first (/root/reference/src/splatmesh/SplatMaterial3D.js:72-73, draw order src/Viewer.js:1610-1616,
drop-in mode src/DropInViewer.js:34-42).  The depth test is per (splat, pixel) and nothing is written back, so it is pinned by
construction: a pixel's value under a destination depth is the composite of exactly the splats whose centre depth passes
LessEqualDepth at that pixel, over the destination colour."""
import numpy as np

import helpers
import oracle
from gaussiansplats3d_amd import camera


def _setup(n=600, w=96, h=64, seed=3):
    scene = helpers.small_scene(n, 0, seed=seed)
    cam = camera.demo_camera("garden", w, h)
    c, cov, rgba, sh = helpers.oracle_inputs(scene)
    ocam = oracle.make_camera(cam.model_view(), cam.projection, cam.position, w, h, 0, 0)
    order = np.arange(n, dtype=np.uint32)
    return ocam, (c, cov, rgba, sh), order, w, h


def test_binary_depth_selects_per_pixel_between_the_frame_and_the_destination():
    ocam, s, order, w, h = _setup()
    rng = np.random.default_rng(1)
    dst = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
    full, *_ = oracle.render(ocam, *s, order, dst_rgba=dst)
    mask = rng.random((h, w)) < 0.5
    depth = np.where(mask, np.float32(1.0), np.float32(0.0)).astype(np.float32)   # 1: everything passes (z_w <= 1), 0: nothing does
    got, *_ = oracle.render(ocam, *s, order, depth=depth, dst_rgba=dst)
    expect = np.where(mask[..., None], full, dst.astype(np.float32) * np.float32(1.0 / 255.0))
    assert np.array_equal(got, expect)
    # and without a destination colour the untouched pixels are the Viewer's clear colour
    got0, *_ = oracle.render(ocam, *s, order, depth=depth)
    full0, *_ = oracle.render(ocam, *s, order)
    assert np.array_equal(got0, np.where(mask[..., None], full0, 0.0))


def test_a_constant_depth_draws_exactly_the_splats_in_front_of_it():
    ocam, s, order, w, h = _setup()
    p = oracle.project(ocam, *s)
    zw = (p["ndcz"] * np.float32(0.5) + np.float32(0.5)).astype(np.float32)
    vis = p["visible"] == 1
    cut = np.float32(np.median(zw[vis]))
    depth = np.full((h, w), cut, dtype=np.float32)
    got, *_ = oracle.render(ocam, *s, order, depth=depth)
    front = order[(zw <= cut)]                                  # LessEqualDepth: a splat AT the stored depth is drawn
    expect, *_ = oracle.render(ocam, *s, front)
    assert 0 < front.size < order.size
    assert np.array_equal(got, expect)


def test_unorm24_compares_what_a_24_bit_depth_buffer_stores():
    """round(z * (2^24 - 1)) on both sides, restated here in numpy fp64: the frame is the composite of exactly the splats whose
    24-bit value is <= the stored depth's.  (Where the two modes can disagree - two fp32 depths inside one 24-bit step - only
    exists below z = 0.5, i.e. nearer than twice the near plane; the stored depth sits on a splat's own value, so the edge counts.)"""
    ocam, s, order, w, h = _setup()
    p = oracle.project(ocam, *s)
    zw = (p["ndcz"] * np.float32(0.5) + np.float32(0.5)).astype(np.float32)
    vis = np.flatnonzero(p["visible"] == 1)
    k = vis[np.argsort(zw[vis])[vis.size // 2]]
    q = lambda z: np.floor(z.astype(np.float64) * 16777215.0 + 0.5)
    depth = np.full((h, w), zw[k], dtype=np.float32)
    u24, *_ = oracle.render(ocam, *s, order, depth=depth, depth_unorm24=True)
    expect, *_ = oracle.render(ocam, *s, order[q(zw) <= q(np.float32(zw[k]).reshape(1))[0]])
    assert np.array_equal(u24, expect)


def test_windows_see_the_same_destination_as_the_full_frame():
    ocam, s, order, w, h = _setup()
    rng = np.random.default_rng(2)
    dst = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
    depth = rng.random((h, w)).astype(np.float32)
    full, *_ = oracle.render(ocam, *s, order, depth=depth, dst_rgba=dst)
    wins = [(8, 4, 32, 24), (60, 40, 36, 24)]
    crops, _ = oracle.render_windows(ocam, *s, order, windows=wins, depth=depth, dst_rgba=dst)
    for (x0, y0, ww, hh), (fb, _amb) in zip(wins, crops):
        assert np.array_equal(fb, full[y0:y0 + hh, x0:x0 + ww])
