"""Host-side mirror of the reference's SplatTree + the Viewer's per-sort cull, over the native/HIP engine.

Reference interface: ``SplatMesh.buildSplatTree(minAlphas)`` -> ``new SplatTree(8, 1000).processSplatMesh(mesh,
filterFunc)`` (/root/reference/src/splatmesh/SplatMesh.js:231-280, src/splattree/SplatTree.js:296-420) and
``Viewer.gatherSceneNodesForSort(gatherAllNodes)`` (src/Viewer.js:1969-2077), plus the sort trigger / partial-sort
schedule of ``Viewer.runSplatSort`` (src/Viewer.js:1833-1964) as :class:`SortScheduler`.
"""
import ctypes as C

import numpy as np

from . import _lib as L
from . import camera as cam_math


class SplatTree:
    """``SplatTree(maxDepth, maxCentersPerNode)``; ``process_splat_mesh`` builds it (native C++, reference arithmetic),
    ``gather_scene_nodes_for_sort`` runs the per-sort cull on the device."""

    def __init__(self, context=None, max_depth=8, max_centers_per_node=1000):
        self.ctx = context
        self.lib = context.lib if context is not None else L.load()
        self.max_depth = int(max_depth)
        self.max_centers_per_node = int(max_centers_per_node)
        self.handle = C.c_void_p()
        self.splat_count = 0
        self._splats = None           # splats held by the leaves (info().splats), cached for asynchronous gathers
        if context is not None:
            context._adopt(self)

    def process_splat_mesh(self, centers, alphas=None, min_alpha=1, first_index=0):
        """centers float32 [n,3] (SplatMesh.getSplatCenter); alphas uint8 [n] with the filter
        ``alpha >= minAlpha`` of SplatMesh.js:239-244 (minAlpha defaults to 1)."""
        self.dispose()
        c = np.ascontiguousarray(centers, dtype=np.float32).reshape(-1, 3)
        keep = None
        if alphas is not None:
            keep = np.ascontiguousarray(np.asarray(alphas).reshape(-1) >= min_alpha, dtype=np.uint8)
        L.check(self.lib.gs_tree_create(self.ctx.handle if self.ctx is not None else None, c.ctypes.data,
                                        keep.ctypes.data if keep is not None else None, c.shape[0], int(first_index),
                                        self.max_depth, self.max_centers_per_node, C.byref(self.handle)))
        self.splat_count = c.shape[0]
        self._splats = None
        return self

    def info(self):
        info = L.TreeInfo()
        L.check(self.lib.gs_tree_get_info(self.handle, C.byref(info)))
        return info

    def leaves(self):
        """(bounds float64 [L,6], centers float64 [L,3], depths uint32 [L], offsets uint32 [L+1], indexes uint32 [S]) —
        subTree.nodesWithIndexes in order."""
        info = self.info()
        n, s = info.leaves, info.splats
        bounds = np.empty((n, 6), np.float64); centers = np.empty((n, 3), np.float64)
        depths = np.empty(n, np.uint32); offsets = np.empty(n + 1, np.uint32); indexes = np.empty(s, np.uint32)
        L.check(self.lib.gs_tree_read(self.handle, bounds.ctypes.data, centers.ctypes.data, depths.ctypes.data,
                                      offsets.ctypes.data, indexes.ctypes.data))
        return bounds, centers, depths, offsets, indexes

    def gather_scene_nodes_for_sort(self, camera, sort_worker=None, gather_all_nodes=False, mesh_world=None,
                                    fov_deg=cam_math.THREE_FOV_DEG, to_host=True, model_view=None, asynchronous=False):
        """Viewer.gatherSceneNodesForSort: returns {'splatRenderCount', 'shouldSortAll', 'indexesToSort'}.  With a
        sort worker the list is written into its device buffer (``sort_worker.sort_gathered`` consumes it).
        asynchronous=True (needs a sort worker, no host list): nothing returns to the host - splatRenderCount stays on the
        device next to the list and the reply carries the tree's splat count as its upper bound."""
        gp = L.GatherParams()
        if model_view is not None:        # inverse(camera.matrixWorld) * splatMesh.matrixWorld, already multiplied (fp64)
            mv = np.asarray(model_view, np.float64)
        else:
            mv = np.asarray(camera.view if mesh_world is None else cam_math.multiply(camera.view, mesh_world), np.float64)
        gp.model_view[:] = mv.reshape(16).tolist()
        gp.fov_y_deg = float(fov_deg)
        gp.render_width, gp.render_height = float(camera.width), float(camera.height)
        gp.gather_all = 1 if gather_all_nodes else 0
        count = C.c_uint32(0)
        if asynchronous:
            if self._splats is None:
                self._splats = int(self.info().splats)
            L.check(self.lib.gs_tree_gather(self.handle, C.byref(gp), sort_worker.handle, None, None))
            sort_worker.gathered_count = self._splats
            return {"splatRenderCount": self._splats, "shouldSortAll": False, "indexesToSort": None, "countOnDevice": True}
        out = np.empty(self.info().splats, dtype=np.uint32) if to_host else None
        L.check(self.lib.gs_tree_gather(self.handle, C.byref(gp), sort_worker.handle if sort_worker is not None else None,
                                        C.byref(count), out.ctypes.data if out is not None else None))
        if sort_worker is not None:
            sort_worker.gathered_count = int(count.value)
        return {"splatRenderCount": int(count.value), "shouldSortAll": False,
                "indexesToSort": out[:count.value] if out is not None else None}

    def dispose(self):
        if self.handle:
            self.lib.gs_tree_destroy(self.handle)
            self.handle = C.c_void_p()

    close = dispose

    def __del__(self):
        try:
            self.dispose()
        except Exception:
            pass


# Viewer.runSplatSort's partial-sort table, src/Viewer.js:1843-1856
PARTIAL_SORTS = (
    {"angleThreshold": 0.55, "sortFractions": (0.125, 0.33333, 0.75)},
    {"angleThreshold": 0.65, "sortFractions": (0.33333, 0.66667)},
    {"angleThreshold": 0.8, "sortFractions": (0.5,)},
)


class SortScheduler:
    """The decision logic of Viewer.runSplatSort (src/Viewer.js:1858-1961) without the Viewer: when a sort is due
    (view direction dot <= 0.99 or position moved >= 1.0 since the last completed schedule) and how many of the
    nearest splats each queued sort covers (partial sorts first, then the full list)."""

    def __init__(self, dynamic_mode=False):
        self.dynamic_mode = bool(dynamic_mode)
        self.last_sort_view_dir = np.array([0.0, 0.0, -1.0])
        self.last_sort_view_pos = np.zeros(3)
        self.queued_sorts = []
        self.sort_running = False

    @staticmethod
    def view_direction(camera):
        """(0,0,-1).applyQuaternion(camera.quaternion) = -Z axis of matrixWorld."""
        m = np.asarray(camera.matrix_world, dtype=np.float64).reshape(4, 4).T
        return -m[:3, 2]

    def next_sort(self, camera, splat_render_count, should_sort_all, force=False, force_sort_all=False):
        """Returns None when no sort is needed, else the splatSortCount of the sort to post now."""
        if self.sort_running:
            return None
        sort_view_dir = self.view_direction(camera)
        angle_diff = float(np.dot(sort_view_dir, self.last_sort_view_dir))
        position_diff = float(np.linalg.norm(np.asarray(camera.position, np.float64) - self.last_sort_view_pos))
        if not force and not self.dynamic_mode and not self.queued_sorts:
            if not (angle_diff <= 0.99 or position_diff >= 1.0):
                return None
        should_sort_all = should_sort_all or force_sort_all
        if not self.queued_sorts:
            if self.dynamic_mode or should_sort_all:
                self.queued_sorts.append(splat_render_count)
            else:
                for partial in PARTIAL_SORTS:
                    if angle_diff < partial["angleThreshold"]:
                        for fraction in partial["sortFractions"]:
                            self.queued_sorts.append(int(np.floor(splat_render_count * fraction)))
                        break
                self.queued_sorts.append(splat_render_count)
        sort_count = min(self.queued_sorts.pop(0), splat_render_count)
        self.sort_running = True
        if not self.queued_sorts:
            self.last_sort_view_pos = np.asarray(camera.position, np.float64).copy()
            self.last_sort_view_dir = sort_view_dir.copy()
        return sort_count

    def sort_done(self):
        self.sort_running = False
