"""oracle — CPU checkers for the sort-and-rasterize hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package; the product (``gaussiansplats3d_amd``) never does.

* :func:`sort_indexes` / :func:`integer_centers` — ``sort_oracle.c``, restating
  ``/root/reference/src/worker/sorter.cpp:17-168`` and ``src/splatmesh/SplatMesh.js:1912-1926``.
* :func:`sort_indexes_numpy` — independent numpy restatement of the static integer path
  (SURVEY.md Appendix A.1), used to cross-check the C oracle.
* :func:`ref_sort_indexes` — the REFERENCE's own ``sorter_no_simd.cpp`` compiled into
  ``oracle/_ref/libsorter_ref.so`` by ``oracle/Makefile`` (present only after ``make -C oracle``
  where ``/root/reference`` exists; the prebuilt file travels to the GPU box).
* :func:`project` / :func:`render` — ``raster_oracle.c``, restating the GLSL in
  ``src/splatmesh/SplatMaterial.js`` + ``SplatMaterial3D.js``; PINNED to the reference's own shader text executed on the
  CPU (tests/golden/raster_ref.npz, see the file header).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

_u32p = C.POINTER(C.c_uint32)
_i32p = C.POINTER(C.c_int32)
_f32p = C.POINTER(C.c_float)
_u8p = C.POINTER(C.c_uint8)


def build(force=False):
    """Compile liboracle.so (and _ref/libsorter_ref.so when /root/reference exists)."""
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("sort_oracle.c", "raster_oracle.c")]
    stale = force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    ref_so = os.path.join(_HERE, "_ref", "libsorter_ref.so")
    ref_wasm = os.path.join(_HERE, "_ref", "sorter_no_simd_non_shared.wasm")
    if (force or not os.path.exists(ref_so) or not os.path.exists(ref_wasm)) and os.path.exists("/root/reference/src/worker/sorter_no_simd.cpp"):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


def _lib():
    global _LIB
    if _LIB is None:
        build()
        _LIB = C.CDLL(os.path.join(_HERE, "liboracle.so"))
        _LIB.gso_sort_indexes.restype = C.c_int
        _LIB.gro_render.restype = C.c_uint64
    return _LIB


def have_ref():
    return os.path.exists(os.path.join(_HERE, "_ref", "libsorter_ref.so"))


def _ref():
    global _REF
    if _REF is None:
        build()
        _REF = C.CDLL(os.path.join(_HERE, "_ref", "libsorter_ref.so"))
        _REF.sortIndexes.restype = None
    return _REF


def wasm_path():
    """The reference's prebuilt scalar, non-shared-memory WASM sorter: in place under /root/reference, or the copy
    oracle/Makefile leaves in oracle/_ref/ (what the GPU box sees); None when neither exists."""
    for p in ("/root/reference/src/worker/sorter_no_simd_non_shared.wasm",
              os.path.join(_HERE, "_ref", "sorter_no_simd_non_shared.wasm")):
        if os.path.exists(p):
            return p
    return None


def wasm_sort_timing(indexes, centers4, mvp, repeat=5, precision=16):
    """Times the reference's WASM sorter under Node on a static integer full sort (oracle/wasm_ref.js: frequencies zeroed,
    process.hrtime around exports.sortIndexes, as src/worker/SortWorker.js:53-60).  Returns {ms_min, ms_mean, repeat, node}
    or None when the module or Node is missing."""
    import shutil
    import tempfile
    wasm = wasm_path()
    if wasm is None or shutil.which("node") is None:
        return None
    indexes = np.ascontiguousarray(indexes, dtype=np.uint32)
    centers4 = np.ascontiguousarray(centers4, dtype=np.int32)
    n, r = centers4.shape[0], indexes.shape[0]
    hdr = np.array([n, r, r, 1 << precision, 1, 0, 0, 0], dtype=np.uint32)
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, "in.bin"), "wb") as f:
            f.write(hdr.tobytes())
            f.write(indexes.tobytes())
            f.write(centers4.tobytes())
            f.write(np.asarray(mvp, dtype=np.float64).astype(np.float32).tobytes())
        out = subprocess.check_output(["node", os.path.join(_HERE, "wasm_ref.js"), wasm, os.path.join(d, "in.bin"),
                                       os.path.join(d, "out.bin"), str(int(repeat))], text=True)
    j = __import__("json").loads(out.strip().splitlines()[-1])
    return {"ms_min": float(j["ms"]), "ms_mean": float(j["ms_mean"]), "repeat": int(j["repeat"]), "node": j.get("node", "")}


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


# --------------------------------------------------------------------------- sort
def integer_centers(centers3):
    """SplatMesh.getIntegerCenters(padFour=True): fp32[n,3] -> int32[n,4]."""
    c = np.ascontiguousarray(centers3, dtype=np.float32).reshape(-1, 3)
    out = np.empty((c.shape[0], 4), dtype=np.int32)
    _lib().gso_integer_centers(_p(c, _f32p), C.c_uint32(c.shape[0]), _p(out, _i32p))
    return out


def _sort_args(indexes, centers4, mvp, sort_count, render_count, precomputed, scene_indexes, transforms, use_int):
    indexes = np.ascontiguousarray(indexes, dtype=np.uint32)
    centers4 = np.ascontiguousarray(centers4, dtype=np.int32 if use_int else np.float32)
    mvp = np.ascontiguousarray(np.asarray(mvp, dtype=np.float64).astype(np.float32).reshape(16))
    if render_count is None:
        render_count = indexes.shape[0]
    if sort_count is None:
        sort_count = render_count
    if precomputed is not None:
        precomputed = np.ascontiguousarray(precomputed, dtype=np.int32 if use_int else np.float32)
    if scene_indexes is not None:
        scene_indexes = np.ascontiguousarray(scene_indexes, dtype=np.uint32)
    if transforms is not None:
        transforms = np.ascontiguousarray(transforms, dtype=np.float32)
    return indexes, centers4, mvp, int(sort_count), int(render_count), precomputed, scene_indexes, transforms


def sort_indexes(indexes, centers4, mvp, sort_count=None, render_count=None, precision=16,
                 use_int=True, dynamic=False, precomputed=None, scene_indexes=None, transforms=None,
                 return_intermediates=False):
    """C oracle for sortIndexes.  centers4: int32[n,4] (or fp32[n,4]); mvp: 16 column-major values.

    Returns uint32[render_count]; with return_intermediates also (keys, buckets, (lo, hi), status).
    """
    (indexes, centers4, mvp, sort_count, render_count, precomputed, scene_indexes,
     transforms) = _sort_args(indexes, centers4, mvp, sort_count, render_count, precomputed, scene_indexes,
                              transforms, use_int)
    out = np.zeros(render_count, dtype=np.uint32)
    keys = np.zeros(render_count, dtype=np.int32)
    buckets = np.zeros(render_count, dtype=np.int32)
    lohi = np.zeros(2, dtype=np.int32)
    st = _lib().gso_sort_indexes(
        _p(indexes, _u32p), centers4.ctypes.data_as(C.c_void_p),
        precomputed.ctypes.data_as(C.c_void_p) if precomputed is not None else None,
        _p(mvp, _f32p), _p(out, _u32p), _p(scene_indexes, _u32p), _p(transforms, _f32p),
        C.c_uint32(1 << precision), C.c_uint32(sort_count), C.c_uint32(render_count),
        C.c_int(precomputed is not None), C.c_int(bool(use_int)), C.c_int(bool(dynamic)),
        _p(keys, _i32p), _p(buckets, _i32p), _p(lohi, _i32p))
    if st < 0:
        raise MemoryError("sort oracle")
    if return_intermediates:
        return out, keys, buckets, (int(lohi[0]), int(lohi[1])), st
    return out


def ref_sort_indexes(indexes, centers4, mvp, sort_count=None, render_count=None, precision=16,
                     use_int=True, dynamic=False, precomputed=None, scene_indexes=None, transforms=None):
    """The reference's own compiled sorter (oracle/_ref).  Guards the hi==lo case, where the native
    build of the reference source dereferences a wild pointer: callers must not pass
    inputs whose keys are all equal."""
    (indexes, centers4, mvp, sort_count, render_count, precomputed, scene_indexes,
     transforms) = _sort_args(indexes, centers4, mvp, sort_count, render_count, precomputed, scene_indexes,
                              transforms, use_int)
    n = centers4.shape[0]
    rng = 1 << precision
    out = np.zeros(max(render_count, 1), dtype=np.uint32)
    mapped = np.zeros(max(render_count, 1), dtype=np.int32)
    freq = np.zeros(2 * rng, dtype=np.uint32)          # SortWorker.js:137-138 allocates 2*range, zeroed :53-55
    dummy_u = np.zeros(1, dtype=np.uint32)
    dummy_f = np.zeros(16, dtype=np.float32)
    _ref().sortIndexes(
        _p(indexes, _u32p), centers4.ctypes.data_as(C.c_void_p),
        (precomputed if precomputed is not None else mapped).ctypes.data_as(C.c_void_p),
        _p(mapped, _i32p), _p(freq, _u32p), _p(mvp, _f32p), _p(out, _u32p),
        _p(scene_indexes if scene_indexes is not None else dummy_u, _u32p),
        _p(transforms if transforms is not None else dummy_f, _f32p),
        C.c_uint32(rng), C.c_uint32(sort_count), C.c_uint32(render_count), C.c_uint32(n),
        C.c_bool(precomputed is not None), C.c_bool(bool(use_int)), C.c_bool(bool(dynamic)))
    return out[:render_count]


def sort_indexes_numpy(indexes, int_centers4, mvp, sort_count=None, render_count=None, precision=16):
    """numpy restatement of the static integer path: independent of the C oracle."""
    idx = np.ascontiguousarray(indexes, dtype=np.uint32)
    ci = np.ascontiguousarray(int_centers4, dtype=np.int32)
    R = idx.shape[0] if render_count is None else int(render_count)
    Rs = R if sort_count is None else int(sort_count)
    s0 = R - Rs
    m32 = np.asarray(mvp, dtype=np.float64).astype(np.float32).reshape(16)
    m = np.trunc(m32[[2, 6, 10]].astype(np.float64) * 1000.0).astype(np.int64)
    tail = idx[s0:R]
    c = ci[tail].astype(np.int64)
    # int32 wrap-around of every product and sum == arithmetic mod 2^32
    d = (c[:, 0] * m[0] + c[:, 1] * m[1] + c[:, 2] * m[2]) & 0xFFFFFFFF
    d = d.astype(np.uint32).view(np.int32)
    out = idx[:R].copy()
    if Rs == 0:
        return out
    lo, hi = int(d.min()), int(d.max())
    rng = 1 << precision
    if hi == lo:
        b = np.zeros(Rs, dtype=np.int64)
    else:
        range_map = np.float32(rng - 1) / (np.float32(hi) - np.float32(lo))
        diff = ((d.astype(np.int64) - lo) & 0xFFFFFFFF).astype(np.uint32).view(np.int32)
        b = np.trunc(diff.astype(np.float32) * range_map).astype(np.int64)
        b = np.clip(b, 0, rng - 1)
    order = np.argsort(b, kind="stable")            # ascending bucket, ties in input order
    out[s0:R] = tail[order][::-1]                   # reversed: descending bucket, ties reversed
    return out


def frustum_keep(mvp, centers4, indexes=None, use_int=True):
    """numpy fp32 restatement of the engine's per-splat frustum cull (include/gsplat_hip.h, gs_sorter_set_frustum_cull;
    no counterpart in the reference).  Every operation is a separate fp32 multiply / add in the kernel's order
    (numpy never contracts).  Returns bool[len(indexes)]: list positions the culled sort keeps."""
    m = np.asarray(mvp, dtype=np.float64).astype(np.float32).reshape(16)
    c = np.ascontiguousarray(centers4)
    if indexes is not None:
        c = c[np.asarray(indexes, dtype=np.int64)]
    if use_int:
        xyz = [c[:, k].astype(np.int32).astype(np.float32) * np.float32(0.001) for k in range(3)]
    else:
        xyz = [c[:, k].astype(np.float32) for k in range(3)]
    q = [((m[r] * xyz[0] + m[4 + r] * xyz[1]) + m[8 + r] * xyz[2]) + m[12 + r] for r in range(4)]
    for v in q:
        assert v.dtype == np.float32
    lim_xy = np.float32(1.25) * q[3] + np.float32(0.01)
    lim_z = np.float32(1.01) * q[3] + np.float32(0.01)
    drop = (np.abs(q[0]) > lim_xy) | (np.abs(q[1]) > lim_xy) | (q[2] < -lim_z) | (q[2] > lim_z)
    return ~drop


def culled_sort(indexes, centers4, mvp, precision=16, use_int=True):
    """What a frustum-culled full sort must return: the reference's sorted list (keys, range and buckets over every
    list position) with the dropped positions removed.  Returns (sorted_kept uint32[], keep bool[])."""
    full = sort_indexes(indexes, centers4, mvp, precision=precision, use_int=use_int)
    keep = frustum_keep(mvp, centers4, indexes, use_int)
    idx = np.asarray(indexes, dtype=np.uint32)
    # the same splat index may appear at several list positions: keep / drop is per splat, so filtering by index is exact
    kept_of_splat = np.zeros(int(np.asarray(centers4).shape[0]), dtype=bool)
    kept_of_splat[idx[keep]] = True
    return full[kept_of_splat[full]], keep


# --------------------------------------------------------------------------- raster
class Camera(C.Structure):
    _fields_ = [("view", C.c_float * 16), ("proj", C.c_float * 16), ("cam_pos", C.c_float * 3),
                ("focal", C.c_float * 2), ("viewport", C.c_float * 2), ("splat_scale", C.c_float),
                ("kernel2d", C.c_float), ("max_splat_px", C.c_float), ("inv_focal_adj", C.c_float),
                ("sh_degree", C.c_int32), ("sh_stored", C.c_int32), ("antialiased", C.c_int32),
                ("point_cloud", C.c_int32),
                # shader permutations (see raster_oracle.c)
                ("orthographic", C.c_int32), ("ortho_zoom", C.c_float), ("fade_in", C.c_int32),
                ("scene_center", C.c_float * 3), ("fade_start", C.c_float), ("effects", C.c_int32),
                ("dynamic", C.c_int32), ("sh8", C.c_int32), ("scene_count", C.c_int32),
                ("view_matrix", C.c_float * 16), ("transforms", (C.c_float * 16) * 32),
                ("inv_cam_pos", (C.c_float * 3) * 32), ("opacity", C.c_float * 32), ("visible", C.c_int32 * 32),
                ("sh8_min", C.c_float * 32), ("sh8_max", C.c_float * 32)]


SPLAT2D = np.dtype([("visible", np.int32), ("cx", np.float32), ("cy", np.float32), ("b1x", np.float32),
                    ("b1y", np.float32), ("b2x", np.float32), ("b2y", np.float32), ("r", np.float32),
                    ("g", np.float32), ("b", np.float32), ("a", np.float32), ("ndcz", np.float32)])


def make_camera(view, proj, cam_pos, width, height, sh_degree=0, sh_stored=0, splat_scale=1.0,
                kernel2d=0.3, max_splat_px=1024.0, focal_adjustment=1.0, antialiased=False,
                point_cloud=False):
    """Uniforms as Viewer.updateSplatMesh / SplatMesh.updateUniforms derive them (dpr = 1)."""
    cam = Camera()
    v32 = np.asarray(view, dtype=np.float64).astype(np.float32).reshape(16)
    p32 = np.asarray(proj, dtype=np.float64).astype(np.float32).reshape(16)
    cam.view[:] = v32.tolist()
    cam.proj[:] = p32.tolist()
    cam.cam_pos[:] = np.asarray(cam_pos, dtype=np.float32).tolist()
    p64 = np.asarray(proj, dtype=np.float64).reshape(16)
    cam.focal[0] = p64[0] * 0.5 * width * focal_adjustment      # Viewer.js:662-665,673
    cam.focal[1] = p64[5] * 0.5 * height * focal_adjustment
    cam.viewport[0] = float(width)
    cam.viewport[1] = float(height)
    cam.splat_scale = splat_scale
    cam.kernel2d = kernel2d
    cam.max_splat_px = max_splat_px
    cam.inv_focal_adj = 1.0 / focal_adjustment
    cam.sh_degree = sh_degree
    cam.sh_stored = sh_stored
    cam.antialiased = int(antialiased)
    cam.point_cloud = int(point_cloud)
    return cam


def _scene_args(centers, cov, rgba, sh):
    centers = np.ascontiguousarray(centers, dtype=np.float32).reshape(-1, 3)
    cov = np.ascontiguousarray(cov, dtype=np.float32).reshape(-1, 6)
    rgba = np.ascontiguousarray(rgba, dtype=np.uint8).reshape(-1, 4)
    if sh is not None:
        sh = np.ascontiguousarray(sh, dtype=np.float32).reshape(centers.shape[0], -1)
    return centers, cov, rgba, sh


def set_scenes(cam, view_matrix=None, transforms=None, camera_position=None, opacity=None, visible=None,
               sh8_range=None, dynamic=False, effects=False):
    """Per-scene uniforms of the dynamic / optional-effects / 8-bit-SH shader permutations.  transforms: list of
    column-major 16-vectors; inverse(transform) * cameraPosition is evaluated here in fp64."""
    n = len(transforms) if transforms is not None else (len(opacity) if opacity is not None else 1)
    cam.scene_count = n
    cam.dynamic = int(dynamic)
    cam.effects = int(effects)
    if view_matrix is not None:
        cam.view_matrix[:] = np.asarray(view_matrix, np.float64).astype(np.float32).reshape(16).tolist()
    for s_ in range(n):
        if transforms is not None:
            t = np.asarray(transforms[s_], np.float64).reshape(16)
            cam.transforms[s_][:] = t.astype(np.float32).tolist()
            if camera_position is not None:
                inv = np.linalg.inv(t.reshape(4, 4).T)
                p = inv @ np.array([*np.asarray(camera_position, np.float64), 1.0])
                cam.inv_cam_pos[s_][:] = (p[:3]).astype(np.float32).tolist()
        cam.opacity[s_] = 1.0 if opacity is None else float(min(max(opacity[s_], 0.0), 1.0))
        cam.visible[s_] = 1 if visible is None else int(bool(visible[s_]))
        if sh8_range is not None:
            cam.sh8_min[s_], cam.sh8_max[s_] = float(sh8_range[s_][0]), float(sh8_range[s_][1])
    return cam


def _set_scene_idx(scene_indexes):
    if scene_indexes is None:
        _lib().gro_set_scene_indexes(None)
        return None
    si = np.ascontiguousarray(scene_indexes, dtype=np.uint32)
    _lib().gro_set_scene_indexes(si.ctypes.data_as(_u32p))
    return si


def project(cam, centers, cov, rgba, sh=None, order=None, scene_indexes=None):
    centers, cov, rgba, sh = _scene_args(centers, cov, rgba, sh)
    _keep = _set_scene_idx(scene_indexes)  # noqa: F841
    if order is not None:
        order = np.ascontiguousarray(order, dtype=np.uint32)
    count = centers.shape[0] if order is None else order.shape[0]
    out = np.zeros(count, dtype=SPLAT2D)
    _lib().gro_project(C.byref(cam), _p(centers, _f32p), _p(cov, _f32p), _p(rgba, _u8p), _p(sh, _f32p),
                       _p(order, _u32p), C.c_uint32(count), out.ctypes.data_as(C.c_void_p))
    return out


def _set_destination(depth, depth_unorm24, W, H):
    """The destination depth of the next render call (gro_set_destination_depth); returns the array to keep alive."""
    if depth is None:
        _lib().gro_set_destination_depth(None, 0)
        return None
    d = np.ascontiguousarray(depth, dtype=np.float32).reshape(H, W)
    _lib().gro_set_destination_depth(d.ctypes.data_as(_f32p), int(bool(depth_unorm24)))
    return d


def _dst_colour(dst_rgba, W, H):
    """RGBA8 [H, W, 4] destination colour -> the float framebuffer the composite starts from (None: the Viewer's clear colour)."""
    if dst_rgba is None:
        return np.zeros((H, W, 4), dtype=np.float32)
    return np.ascontiguousarray(np.asarray(dst_rgba, dtype=np.uint8).reshape(H, W, 4).astype(np.float32) * np.float32(1.0 / 255.0))


def render(cam, centers, cov, rgba, sh=None, order=None, rop8=False, amb_eps=1e-3, scene_indexes=None, depth=None,
           depth_unorm24=False, dst_rgba=None):
    """Returns (fb float32[H,W,4] row0=bottom, rgba8 uint8[H,W,4], ambig uint8[H,W], fragments).
    depth: float32 [H, W] window depth other scene geometry left (`depthTest: true, depthWrite: false`,
    SplatMaterial3D.js:72-73; LessEqualDepth); depth_unorm24: compare as a 24-bit depth buffer; dst_rgba: uint8 [H, W, 4] the
    colour the splats are blended over (draw order src/Viewer.js:1610-1616)."""
    centers, cov, rgba, sh = _scene_args(centers, cov, rgba, sh)
    _keep = _set_scene_idx(scene_indexes)  # noqa: F841
    if order is not None:
        order = np.ascontiguousarray(order, dtype=np.uint32)
    count = centers.shape[0] if order is None else order.shape[0]
    W, H = int(cam.viewport[0]), int(cam.viewport[1])
    fb = _dst_colour(dst_rgba, W, H)
    amb = np.zeros((H, W), dtype=np.uint8)
    _keep_depth = _set_destination(depth, depth_unorm24, W, H)  # noqa: F841
    try:
        frags = _lib().gro_render(C.byref(cam), _p(centers, _f32p), _p(cov, _f32p), _p(rgba, _u8p),
                                  _p(sh, _f32p), _p(order, _u32p), C.c_uint32(count), C.c_int(int(rop8)),
                                  C.c_float(amb_eps), _p(fb, _f32p), _p(amb, _u8p))
    finally:
        _set_destination(None, False, W, H)
    q = np.empty((H, W, 4), dtype=np.uint8)
    _lib().gro_quantize(_p(fb, _f32p), C.c_uint64(fb.size), _p(q, _u8p))
    return fb, q, amb, int(frags)


def render_windows(cam, centers, cov, rgba, sh=None, order=None, windows=(), rop8=False, amb_eps=1e-3, scene_indexes=None,
                   depth=None, depth_unorm24=False, dst_rgba=None, error_bounds=None):
    """Crops of the full frame: `windows` = [(x0, y0, w, h)] in GL window coordinates (row 0 = bottom).  Every splat of
    `order` is projected once and composited into the windows it reaches; window k's pixels equal pixels
    [y0:y0+h, x0:x0+w] of :func:`render`.  `sh` may be float32 [n, 9|24] or IEEE-half bits / float16 (kept as stored:
    the full-size scenes need no fp32 copy).  Returns [(fb float32[h,w,4], ambig uint8[h,w])] and the fragment count."""
    centers = np.ascontiguousarray(centers, dtype=np.float32).reshape(-1, 3)
    cov = np.ascontiguousarray(cov, dtype=np.float32).reshape(-1, 6)
    rgba = np.ascontiguousarray(rgba, dtype=np.uint8).reshape(-1, 4)
    sh_f16 = 0
    if sh is not None:
        sh = np.ascontiguousarray(sh)
        if sh.dtype in (np.float16, np.uint16):
            sh = sh.view(np.uint16)
            sh_f16 = 1
        else:
            sh = np.ascontiguousarray(sh, dtype=np.float32)
        sh = sh.reshape(centers.shape[0], -1)
    _keep = _set_scene_idx(scene_indexes)  # noqa: F841
    if order is not None:
        order = np.ascontiguousarray(order, dtype=np.uint32)
    count = centers.shape[0] if order is None else order.shape[0]
    wins = np.ascontiguousarray(np.asarray(windows, dtype=np.int32).reshape(-1, 4))
    n = wins.shape[0]
    W, H = int(cam.viewport[0]), int(cam.viewport[1])
    if dst_rgba is None:
        fbs = [np.zeros((int(h), int(w), 4), dtype=np.float32) for _, _, w, h in wins]
    else:                                                        # (depth / dst_rgba: the FULL frame's, as in render)
        full = _dst_colour(dst_rgba, W, H)
        fbs = [np.ascontiguousarray(full[int(y0):int(y0 + h), int(x0):int(x0 + w)]) for x0, y0, w, h in wins]
        assert all(f.shape[:2] == (int(h), int(w)) for f, (_, _, w, h) in zip(fbs, wins)), "windows must lie inside the frame"
    _keep_depth = _set_destination(depth, depth_unorm24, W, H)  # noqa: F841
    ambs = [np.zeros((int(h), int(w)), dtype=np.uint8) for _, _, w, h in wins]
    fb_ptrs = (C.c_void_p * n)(*[f.ctypes.data for f in fbs])
    amb_ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in ambs])
    lib = _lib()
    if error_bounds is not None:
        # error_bounds: a list the caller passes in; it receives one float32 [h, w] plane per window - how far (in 1/255) an RGBA8
        # target that rounds after every splat can sit from the exact composite of that pixel (raster_oracle.c, g_err_bound)
        assert n <= 64
        del error_bounds[:]
        error_bounds.extend(np.zeros((int(h), int(w)), dtype=np.float32) for _, _, w, h in wins)
        eb_ptrs = (C.c_void_p * n)(*[e.ctypes.data for e in error_bounds])
        lib.gro_set_error_bound_planes(C.c_uint32(n), eb_ptrs)
    lib.gro_render_windows.restype = C.c_uint64
    frags = lib.gro_render_windows(C.byref(cam), _p(centers, _f32p), _p(cov, _f32p), _p(rgba, _u8p),
                                   sh.ctypes.data_as(C.c_void_p) if sh is not None else None, C.c_int(sh_f16),
                                   _p(order, _u32p), C.c_uint32(count), C.c_int(int(rop8)), C.c_float(amb_eps),
                                   C.c_uint32(n), wins.ctypes.data_as(C.POINTER(C.c_int32)), fb_ptrs, amb_ptrs)
    _set_destination(None, False, W, H)
    lib.gro_set_error_bound_planes(C.c_uint32(0), None)
    return list(zip(fbs, ambs)), int(frags)
