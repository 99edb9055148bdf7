"""Host-side mirror of the reference's SplatMesh render seam, over the HIP engine.

Reference interface (/root/reference/src/splatmesh/SplatMesh.js): ``build`` (:306-405) uploads scene data
(``setupDataTextures`` :637-898), ``updateRenderIndexes(globalIndexes, renderSplatCount)`` (:1228-1235),
``updateUniforms(renderDimensions, focalX, focalY, ortho, zoom, inverseFocalAdjustment)`` (:1248-1280),
``setSplatScale`` / ``setPointCloudModeEnabled`` (:1282-1300); the draw itself is
``renderer.render(splatMesh, camera)`` (src/Viewer.js:1616).  Here ``render`` returns the RGBA8 framebuffer
(row 0 = bottom, GL convention) instead of drawing into a WebGL canvas.
"""
import ctypes as C

import numpy as np

from . import _lib as L
from .util import to_half_three


class SplatMesh:
    def __init__(self, context, max_splat_count, spherical_harmonics_degree=0, half_precision_covariances=False,
                 antialiased=False, kernel_2d_size=0.3, max_screen_space_splat_size=1024.0, splat_scale=1.0,
                 point_cloud_mode=False, spherical_harmonics_8bit=False, dynamic_mode=False,
                 enable_optional_effects=False):
        self.ctx = context
        self.lib = context.lib
        self.max_splat_count = int(max_splat_count)
        self.sh_degree = int(spherical_harmonics_degree)
        self.half_cov = bool(half_precision_covariances)
        self.antialiased = bool(antialiased)
        self.kernel_2d_size = float(kernel_2d_size)
        self.max_screen_space_splat_size = float(max_screen_space_splat_size)
        self.splat_scale = float(splat_scale)
        self.point_cloud_mode = bool(point_cloud_mode)
        self.sh_8bit = bool(spherical_harmonics_8bit)          # sphericalHarmonics8BitMode (compression level 2)
        self.dynamic_mode = bool(dynamic_mode)
        self.enable_optional_effects = bool(enable_optional_effects)
        self.fade_in = None                                    # (sceneCenter, visibleRegionFadeStartRadius) or None
        self.splat_count = 0
        self.render_count = 0
        self._indexes = None          # host indexes from updateRenderIndexes
        self._sorter = None           # or a SortWorker whose result is device resident
        self._cam = L.Camera()
        self.last_status = 0
        self.handle = C.c_void_p()
        L.check(self.lib.gs_mesh_create(context.handle, self.max_splat_count, self.sh_degree,
                                        (L.GS_MESH_COV_HALF if self.half_cov else 0) |
                                        (L.GS_MESH_SH_U8 if self.sh_8bit else 0), C.byref(self.handle)))
        context._adopt(self)

    # -- build / data upload ------------------------------------------------------------------------
    def build(self, centers, covariances, colors, spherical_harmonics=None, start=0, scene_indexes=None):
        """fillSplatDataArrays output -> device planes.  covariances: float32 [n,6]; narrowed with
        THREE.DataUtils.toHalfFloat semantics when half_precision_covariances (SplatBuffer.js:469-474).
        spherical_harmonics: float16 (or uint16 bit patterns) [n, 9|24], coefficient-major RGB triples."""
        c = np.ascontiguousarray(centers, dtype=np.float32).reshape(-1, 3)
        n = c.shape[0]
        cov = np.ascontiguousarray(covariances, dtype=np.float32).reshape(n, 6)
        rgba = np.ascontiguousarray(colors, dtype=np.uint8).reshape(n, 4)
        cov16 = to_half_three(cov) if self.half_cov else None
        sh = None
        sh8 = None
        ncoef = 9 if self.sh_degree == 1 else 24
        if self.sh_degree > 0 and self.sh_8bit:
            sh8 = np.ascontiguousarray(np.asarray(spherical_harmonics, dtype=np.uint8).reshape(n, ncoef))
        elif self.sh_degree > 0:
            sh = np.ascontiguousarray(spherical_harmonics)
            if sh.dtype != np.uint16:
                sh = sh.astype(np.float16).view(np.uint16)
            sh = np.ascontiguousarray(sh.reshape(n, 9 if self.sh_degree == 1 else 24))
        L.check(self.lib.gs_mesh_upload(self.handle, int(start), n, c.ctypes.data,
                                        None if self.half_cov else cov.ctypes.data,
                                        cov16.ctypes.data if self.half_cov else None, rgba.ctypes.data,
                                        sh.ctypes.data if sh is not None else None))
        if sh8 is not None:
            L.check(self.lib.gs_mesh_upload_sh_u8(self.handle, int(start), n, sh8.ctypes.data))
        if scene_indexes is not None:
            si = np.ascontiguousarray(scene_indexes, dtype=np.uint32).reshape(n)
            L.check(self.lib.gs_mesh_upload_scene_indexes(self.handle, int(start), n, si.ctypes.data))
        self.splat_count = max(self.splat_count, int(start) + n)
        return self

    def set_scenes(self, transforms=None, camera_position=None, opacity=None, visible=None, sh8_range=None):
        """Per-scene uniforms: uniforms.transforms (dynamicMode), sceneOpacity / sceneVisibility
        (enableOptionalEffects), sphericalHarmonics8BitCompressionRangeMin/Max.  transforms: column-major 16-vectors;
        inverse(transform) * cameraPosition (the dynamic-mode SH view origin) is evaluated here in fp64."""
        n = max(len(x) for x in (transforms, opacity, visible, sh8_range) if x is not None)
        sp = L.SceneParams()
        sp.scene_count = n
        for s_ in range(n):
            t = np.eye(4).T.reshape(16) if transforms is None else np.asarray(transforms[s_], np.float64).reshape(16)
            sp.transforms[s_][:] = t.astype(np.float32).tolist()
            if camera_position is not None:
                p = np.linalg.inv(t.reshape(4, 4).T) @ np.array([*np.asarray(camera_position, np.float64), 1.0])
                sp.inv_cam_pos[s_][:] = [*p[:3].astype(np.float32).tolist(), 1.0]
            sp.opacity[s_] = 1.0 if opacity is None else float(min(max(opacity[s_], 0.0), 1.0))     # clamp(), SplatMesh.js:1271
            sp.visible[s_] = 1 if visible is None else int(bool(visible[s_]))
            if sh8_range is not None:
                sp.sh8_min[s_], sp.sh8_max[s_] = float(sh8_range[s_][0]), float(sh8_range[s_][1])
        L.check(self.lib.gs_mesh_set_scenes(self.handle, C.byref(sp)))
        return self

    def set_fade_in(self, scene_center=None, visible_region_fade_start_radius=0.0):
        """fadeInComplete == 0 with these uniforms (SplatMesh.updateVisibleRegionFadeDistance); None = complete."""
        self.fade_in = None if scene_center is None else (np.asarray(scene_center, np.float32), float(visible_region_fade_start_radius))

    def get_splat_count(self):
        return self.splat_count

    # -- per-sort / per-frame state -------------------------------------------------------------------
    def update_render_indexes(self, global_indexes, render_splat_count):
        """SplatMesh.updateRenderIndexes: host Uint32Array of sorted global indexes (drawn back to front)."""
        self._indexes = np.ascontiguousarray(global_indexes, dtype=np.uint32)
        self._sorter = None
        self.render_count = int(render_splat_count)

    def use_sorter_result(self, sort_worker, render_splat_count):
        """Device-resident alternative: draw the result the sort worker left in HBM."""
        if sort_worker is not None and getattr(sort_worker, "_bound_mesh", None) is not self:
            # from its next sort on, the worker hands over positions in this mesh's storage order (no per-frame translation)
            L.check(self.lib.gs_sorter_bind_mesh(sort_worker.handle, self.handle))
            sort_worker._bound_mesh = self
        self._sorter = sort_worker
        self._indexes = None
        self.render_count = int(render_splat_count)

    def update_uniforms(self, render_dimensions, focal_x, focal_y, orthographic_mode=False, orthographic_zoom=1.0,
                        inverse_focal_adjustment=1.0, model_view=None, projection=None, camera_position=None,
                        spherical_harmonics_degree=None, view_matrix=None):
        """SplatMesh.updateUniforms + three's built-in modelViewMatrix / projectionMatrix / cameraPosition."""
        cam = self._cam
        cam.ortho_zoom = float(orthographic_zoom)
        cam.width, cam.height = int(render_dimensions[0]), int(render_dimensions[1])
        cam.focal[0], cam.focal[1] = float(focal_x), float(focal_y)
        cam.inv_focal_adj = float(inverse_focal_adjustment)
        cam.splat_scale = self.splat_scale
        cam.kernel2d = self.kernel_2d_size
        cam.max_splat_px = self.max_screen_space_splat_size
        cam.sh_degree = self.sh_degree if spherical_harmonics_degree is None else int(spherical_harmonics_degree)
        cam.flags = ((L.GS_CAM_ANTIALIASED if self.antialiased else 0) | (L.GS_CAM_POINT_CLOUD if self.point_cloud_mode else 0) |
                     (L.GS_CAM_ORTHOGRAPHIC if orthographic_mode else 0) | (L.GS_CAM_DYNAMIC if self.dynamic_mode else 0) |
                     (L.GS_CAM_SCENE_EFFECTS if self.enable_optional_effects else 0) |
                     (L.GS_CAM_FADE_IN if self.fade_in is not None else 0))
        if self.fade_in is not None:
            cam.scene_center[:] = self.fade_in[0].tolist()
            cam.fade_start_radius = self.fade_in[1]
        if view_matrix is not None:
            cam.view_matrix[:] = np.asarray(view_matrix, dtype=np.float64).astype(np.float32).reshape(16).tolist()
        if model_view is not None:
            cam.view[:] = np.asarray(model_view, dtype=np.float64).astype(np.float32).reshape(16).tolist()
        if projection is not None:
            cam.proj[:] = np.asarray(projection, dtype=np.float64).astype(np.float32).reshape(16).tolist()
        if camera_position is not None:
            cam.cam_pos[:] = np.asarray(camera_position, dtype=np.float32).tolist()

    def set_camera(self, camera, focal_adjustment=1.0, mesh_world=None, spherical_harmonics_degree=None):
        """Viewer.updateSplatMesh (src/Viewer.js:651-677) for a camera.PerspectiveCamera."""
        fx, fy = camera.focal(focal_adjustment)
        ortho = bool(getattr(camera, "is_orthographic", False))
        self.update_uniforms((camera.width, camera.height), fx, fy, ortho, getattr(camera, "zoom", 1.0), 1.0 / focal_adjustment,
                             camera.model_view(mesh_world), camera.projection, camera.position,
                             spherical_harmonics_degree, view_matrix=camera.view)

    def set_splat_scale(self, splat_scale=1.0):
        self.splat_scale = float(splat_scale)
        self._cam.splat_scale = self.splat_scale

    def set_point_cloud_mode_enabled(self, enabled):
        self.point_cloud_mode = bool(enabled)

    def strip_shape(self, tile_rows=None):
        cam = self._cam
        rows_total = (cam.height + L.GS_TILE - 1) // L.GS_TILE
        r0, r1 = (0, rows_total) if tile_rows is None else tile_rows
        y0, y1 = r0 * L.GS_TILE, min(r1 * L.GS_TILE, cam.height)
        return max(y1 - y0, 0), cam.width

    def project(self, tile_rows=None):
        """The vertex stage on its own (gs_mesh_project) for the camera set by ``set_camera`` / ``update_uniforms`` and the
        strip `tile_rows`; the next ``render`` with the same camera and strip consumes it instead of projecting again."""
        cam = self._cam
        cam.tile_row_begin, cam.tile_row_end = (0, 0) if tile_rows is None else (int(tile_rows[0]), int(tile_rows[1]))
        L.check(self.lib.gs_mesh_project(self.handle, C.byref(cam)))

    def render(self, tile_rows=None, out_device_ptr=None, want_stats=True, to_host=True):
        """renderer.render(splatMesh, camera).  Returns (uint8[h,w,4] or None, RenderStats or None).
        tile_rows=(begin,end): render only those 16-px tile rows (multi-GPU strips)."""
        cam = self._cam
        cam.tile_row_begin, cam.tile_row_end = (0, 0) if tile_rows is None else (int(tile_rows[0]), int(tile_rows[1]))
        h, w = self.strip_shape(tile_rows)
        out = np.empty((h, w, 4), dtype=np.uint8) if to_host else None
        stats = L.RenderStats() if want_stats else None
        idx = self._indexes
        # GS_WARN_FRAME_TRUNCATED: an earlier asynchronous draw overflowed its entry buffer (buffers grown since)
        self.last_status = L.check(self.lib.gs_mesh_render(
            self.handle, C.byref(cam), idx.ctypes.data if idx is not None else None,
            self._sorter.handle if self._sorter is not None else None, self.render_count,
            out.ctypes.data if out is not None else None, C.c_void_p(out_device_ptr) if out_device_ptr else None,
            C.byref(stats) if stats is not None else None))
        return out, stats

    def set_destination(self, depth=None, rgba=None, depth_unorm24=False, depth_device_ptr=None, rgba_device_ptr=None,
                        size=None):
        """What the following draws are depth-tested against and blended over - the reference's `depthTest: true,
        depthWrite: false` + NormalBlending over whatever the host's scene drew first (SplatMaterial3D.js:72-73,
        src/Viewer.js:1610-1616, src/DropInViewer.js:34-42).  depth: float32 [H, W] window depth (row 0 = bottom), rgba: uint8
        [H, W, 4]; or device pointers of the same layouts with size = (width, height).  No arguments: back to a cleared target."""
        if depth is None and rgba is None and depth_device_ptr is None and rgba_device_ptr is None:
            L.check(self.lib.gs_mesh_set_destination(self.handle, None))
            return self
        d = L.Destination()
        keep = []
        if depth is not None and depth_device_ptr:
            raise ValueError("pass the destination depth on the host OR on the device, not both")
        if rgba is not None and rgba_device_ptr:
            raise ValueError("pass the destination colour on the host OR on the device, not both")
        shapes = []                                  # (width, height) every input implies: they must agree
        if size is not None:
            if len(size) != 2:
                raise ValueError("size = (width, height)")
            shapes.append((int(size[0]), int(size[1])))
        if depth is not None:
            a = np.ascontiguousarray(depth, dtype=np.float32)
            if a.ndim != 2:
                raise ValueError("depth must be a float32 array of shape [H, W], got %r" % (a.shape,))
            shapes.append((a.shape[1], a.shape[0]))
            d.depth_host = a.ctypes.data
            keep.append(a)
        if rgba is not None:
            a = np.ascontiguousarray(rgba, dtype=np.uint8)
            if a.ndim != 3 or a.shape[2] != 4:
                raise ValueError("rgba must be a uint8 array of shape [H, W, 4], got %r" % (a.shape,))
            shapes.append((a.shape[1], a.shape[0]))
            d.rgba_host = a.ctypes.data
            keep.append(a)
        if not shapes:
            raise ValueError("device pointers need size = (width, height)")
        if any(sh != shapes[0] for sh in shapes):
            raise ValueError("the destination's depth, colour and size disagree: %r" % (shapes,))
        size = shapes[0]
        if depth_device_ptr:
            d.depth_dev = int(depth_device_ptr)
        if rgba_device_ptr:
            d.rgba_dev = int(rgba_device_ptr)
        d.width, d.height = int(size[0]), int(size[1])
        d.flags = L.GS_DEST_DEPTH_UNORM24 if depth_unorm24 else 0
        L.check(self.lib.gs_mesh_set_destination(self.handle, C.byref(d)))
        return self

    def debug_set_entry_capacity(self, capacity):
        """Test hook: resize the entry buffers (the overflow -> regrow path of a draw)."""
        L.check(self.lib.gs_mesh_debug_set_entry_capacity(self.handle, int(capacity)))

    def last_stats(self):
        stats = L.RenderStats()
        L.check(self.lib.gs_mesh_last_stats(self.handle, C.byref(stats)))
        return stats

    def kernel_time(self, which=0, reset=True):
        """(summed device ms, launches) of one kernel since the last reset; which 0 = k_project."""
        total, n = C.c_double(0.0), C.c_uint32(0)
        L.check(self.lib.gs_mesh_kernel_time(self.handle, int(which), 1 if reset else 0, C.byref(total), C.byref(n)))
        return float(total.value), int(n.value)

    def debug_records(self, count=None):
        """Vertex-stage outputs of the last draw: (records uint32 [n,8], rects uint32 [n,2], visible bool [n]).
        Records and rects are only defined where `visible` is set."""
        n = self.splat_count if count is None else count
        recs = np.empty((n, 8), dtype=np.uint32)
        rects = np.empty((n, 2), dtype=np.uint32)
        words = (n + 63) // 64
        mask = np.empty(words, dtype=np.uint64)
        L.check(self.lib.gs_mesh_debug_read(self.handle, 0, recs.ctypes.data, n))
        L.check(self.lib.gs_mesh_debug_read(self.handle, 1, rects.ctypes.data, n))
        L.check(self.lib.gs_mesh_debug_read(self.handle, 3, mask.ctypes.data, words))
        vis = np.unpackbits(mask.view(np.uint8), bitorder="little")[:n].astype(bool)
        return recs, rects, vis

    def bin_entry_counts(self, tile_rows=None):
        """Entries per list bin of the last draw, shaped [list_rows, lists_x] (`tile_rows`: the strip it drew)."""
        cam = self._cam
        LB = int(self.last_stats().list_bin_px)
        bins_x = (cam.width + LB - 1) // LB
        rows_total = (cam.height + L.GS_TILE - 1) // L.GS_TILE
        r0, r1 = (0, rows_total) if tile_rows is None else tile_rows
        y0, y1 = r0 * L.GS_TILE, min(r1 * L.GS_TILE, cam.height)
        b0, b1 = y0 // LB, (max(y1, y0) + LB - 1) // LB if y1 > y0 else y0 // LB
        rng = np.empty(((b1 - b0) * bins_x, 2), dtype=np.uint32)
        if rng.shape[0]:
            L.check(self.lib.gs_mesh_debug_read(self.handle, 2, rng.ctypes.data, rng.shape[0]))
        cnt = np.where(rng[:, 1] > rng[:, 0], rng[:, 1] - rng[:, 0], 0).astype(np.uint32)   # untouched: (~0, 0)
        return cnt.reshape(b1 - b0, bins_x)

    def blend_bin_stats(self):
        """Per 32-px blend bin of the last FULL-frame draw: (entries scanned, 2 x (splat, quadrant) pairs composited),
        [bin_rows, bins_x, 2]."""
        cam = self._cam
        bx, by = (cam.width + L.GS_BIN - 1) // L.GS_BIN, (cam.height + L.GS_BIN - 1) // L.GS_BIN
        out = np.zeros((by * bx, 2), dtype=np.uint32)
        L.check(self.lib.gs_mesh_debug_read(self.handle, 4, out.ctypes.data, by * bx))
        return out.reshape(by, bx, 2)

    def set_draw_mode(self, rop8=False, full=False):
        """How the following draws composite: the fp32 front-to-back composite rounded once (default), or - rop8=True - the
        reference's RGBA8 render target as a GPU executes it: back to front, every channel rounded to 8 bits after every splat
        (SplatMaterial3D.js:65-75; gs_mesh_set_draw_mode).  GS_DRAW_ROP8 walks the splats in front of each quadrant's saturation depth
        (T <= 1e-6) - ~4x the fp32 blend; full=True (GS_DRAW_ROP8_FULL) walks every list to its end (~70x)."""
        mode = L.GS_DRAW_FP32 if not rop8 else (L.GS_DRAW_ROP8_FULL if full else L.GS_DRAW_ROP8)
        L.check(self.lib.gs_mesh_set_draw_mode(self.handle, mode))
        return self

    def set_deep_pass(self, enabled):
        """Scheduling only (the pixels do not change): whether very deep bins may be composited by one wave per quadrant and
        chunk instead of by one workgroup."""
        L.check(self.lib.gs_mesh_set_deep_pass(self.handle, 1 if enabled else 0))

    def deep_bins(self):
        """The 32-px bins the last draw composited through the deep pass (one wave per quadrant and chunk; chosen from the
        statistics of the draw before it)."""
        return self.deep_pass_info()["bins"]

    def deep_pass_info(self):
        """{bins drawn by the deep pass, bins over its threshold, chunk partials the per-bin kernel closed itself, whether its
        partial pool ran out} of the last draw."""
        out = np.zeros(4 + 512, dtype=np.uint32)               # 4 words + at most GS_DEEP_MAX_BINS = 512 bins
        L.check(self.lib.gs_mesh_debug_read(self.handle, 5, out.ctypes.data, out.size))
        return {"bins": out[4:4 + int(out[0])].copy(), "candidates": int(out[1]), "chunks_closed_by_bins": int(out[2]),
                "pool_exhausted": bool(out[3])}

    def view_share(self):
        """{visible, projected: of the last full-frame draw whose verdict has reached the host (a mapped word, no statistics call
        needed), block_test: where the last vertex stage ran its block test - 1 a kernel of its own, 0 in every workgroup, 2 nowhere}.
        Scheduling only: frames do not depend on it."""
        out = np.zeros(3, dtype=np.uint32)
        L.check(self.lib.gs_mesh_debug_read(self.handle, 6, out.ctypes.data, 3))
        return {"visible": int(out[0]), "projected": int(out[1]), "block_test": int(out[2])}

    def tile_row_costs(self):
        """Work estimate per 16-px tile row of the last FULL-frame draw (used to balance multi-GPU strips).  The blend is
        what a strip mostly pays for and its cost is what it WALKS before its pixels saturate, not the length of its lists:
        per 32-px bin row, the half tiles it evaluated (0.55 of a whole-tile walk each) + an eighth of the entries it staged;
        binning / entry sorting add the list entries of the row (a quarter each).  Rows inherit an even share of the bin /
        list-bin row they lie in."""
        cam = self._cam
        rows_total = (cam.height + L.GS_TILE - 1) // L.GS_TILE
        ratio = int(self.last_stats().list_bin_px) // L.GS_TILE
        entries = np.repeat(self.bin_entry_counts().sum(axis=1).astype(np.float64) / ratio, ratio)[:rows_total]
        st = self.blend_bin_stats().astype(np.float64).sum(axis=1)          # [bin_rows, 2]
        blend = np.repeat((0.55 * st[:, 1] + st[:, 0] / 8.0) / 2.0, 2)[:rows_total]
        if blend.shape[0] < rows_total:
            blend = np.pad(blend, (0, rows_total - blend.shape[0]))
        return blend + 0.25 * entries

    def rop8_window(self, x0, y0, width, height):
        """Verification: the window [x0, x0+width) x [y0, y0+height) of the LAST draw composited the reference's way - back to
        front into an RGBA8 target, every channel rounded to 8 bits after every splat (gs_mesh_debug_rop8;
        SplatMaterial3D.js:65-75).  uint8 [height, width, 4], row 0 = y0 (GL orientation)."""
        out = np.empty((int(height), int(width), 4), dtype=np.uint8)
        L.check(self.lib.gs_mesh_debug_rop8(self.handle, int(x0), int(y0), int(width), int(height), out.ctypes.data))
        return out

    def dispose(self):
        if self.handle:
            self.lib.gs_mesh_destroy(self.handle)
            self.handle = C.c_void_p()

    close = dispose

    def __del__(self):
        try:
            self.dispose()
        except Exception:
            pass
