"""Host-side mirror of the reference's sort worker, over the HIP engine.

Reference interface: ``createSortWorker(splatCount, useSharedMemory, enableSIMDInSort, integerBasedSort,
dynamicMode, splatSortDistanceMapPrecision)`` and its message protocol
(/root/reference/src/worker/SortWorker.js:83-256), driven by ``Viewer.setupSortWorker`` / ``runSplatSort``
(/root/reference/src/Viewer.js:1235-1300, 1833-1964).  Same message names, same argument meaning; the worker
thread, the WASM module flavours and the shared WebAssembly.Memory disappear — the "memory" is HBM.
"""
import ctypes as C

import numpy as np

from . import _lib as L

DEFAULT_PRECISION = 16     # Constants.DefaultSplatSortDistanceMapPrecision, src/Constants.js:3


class SortWorker:
    """``worker = create_sort_worker(...)``; ``worker.post_message({...})``; replies arrive on ``worker.onmessage``
    (a callable taking the reply dict) and are also returned."""

    def __init__(self, context, splat_count, integer_based_sort=True, dynamic_mode=False,
                 precision=DEFAULT_PRECISION):
        self.ctx = context
        self.lib = context.lib
        self.max_splat_count = int(splat_count)          # Viewer.js:1285 writes worker.maxSplatCount
        self.integer_based_sort = bool(integer_based_sort)
        self.dynamic_mode = bool(dynamic_mode)
        self.precision = int(precision)
        self.onmessage = None
        self.uploaded_splat_count = 0
        self.gathered_count = 0
        self.frustum_cull = False
        self.visibility_cull = False
        flags = (L.GS_SORT_INTEGER if integer_based_sort else 0) | (L.GS_SORT_DYNAMIC if dynamic_mode else 0)
        self.handle = C.c_void_p()
        L.check(self.lib.gs_sorter_create(context.handle, self.max_splat_count, flags, self.precision,
                                          C.byref(self.handle)))
        context._adopt(self)
        self._reply({"sortSetupPhase1Complete": True})   # SortWorker.js:181-195

    # -- protocol ---------------------------------------------------------------------------------
    def post_message(self, msg):
        if "centers" in msg:
            return self._on_centers(msg)
        if "sort" in msg:
            return self._on_sort(msg["sort"])
        raise ValueError("unknown sort-worker message: " + ", ".join(msg.keys()))

    def _reply(self, msg):
        if self.onmessage:
            self.onmessage(msg)
        return msg

    def _on_centers(self, msg):
        """{centers, sceneIndexes, range:{from,to,count}} — SortWorker.js:84-98."""
        rng = msg["range"]
        dt = np.int32 if self.integer_based_sort else np.float32
        centers = np.ascontiguousarray(np.asarray(msg["centers"]).view(dt) if isinstance(msg["centers"], np.ndarray)
                                       else np.frombuffer(msg["centers"], dtype=dt)).reshape(-1)
        count = int(rng["count"])
        if centers.size != 4 * count:
            raise ValueError("centers must hold 4 values per splat (padFour layout)")
        scene = None
        if self.dynamic_mode:
            scene = np.ascontiguousarray(msg["sceneIndexes"], dtype=np.uint32).reshape(-1)
        L.check(self.lib.gs_sorter_upload_centers(self.handle, int(rng["from"]), count, centers.ctypes.data,
                                                  scene.ctypes.data if scene is not None else None))
        self.uploaded_splat_count = max(self.uploaded_splat_count, int(rng["from"]) + count)
        return None

    def _on_sort(self, s, keep_on_device=False):
        """{sort:{modelViewProj, splatRenderCount, splatSortCount, usePrecomputedDistances, indexesToSort,
        transforms, precomputedDistances}} -> {sortDone, splatSortCount, splatRenderCount, sortTime,
        sortedIndexes} — SortWorker.js:31-81, 99-115."""
        render = min(int(s.get("splatRenderCount") or 0), self.uploaded_splat_count)
        sort = min(int(s.get("splatSortCount") or 0), self.uploaded_splat_count)
        mvp = np.ascontiguousarray(np.asarray(s["modelViewProj"], dtype=np.float64).astype(np.float32))  # :54
        idx = s.get("indexesToSort")
        if idx is not None:
            idx = np.ascontiguousarray(idx, dtype=np.uint32)
            if idx.size < render:                    # the C side copies splatRenderCount entries
                raise ValueError(f"indexesToSort holds {idx.size} entries, splatRenderCount is {render}")
        pre = None
        if s.get("usePrecomputedDistances"):
            pre = np.ascontiguousarray(s["precomputedDistances"],
                                       dtype=np.int32 if self.integer_based_sort else np.float32)
            if pre.size < self.uploaded_splat_count:  # one distance per uploaded splat (SortWorker.js:125-178)
                raise ValueError(f"precomputedDistances holds {pre.size} values for {self.uploaded_splat_count} splats")
        tr = None
        if self.dynamic_mode:
            tr = np.zeros(16 * L.GS_MAX_SCENES, dtype=np.float32)
            t_in = np.asarray(s["transforms"], dtype=np.float32).reshape(-1)
            tr[:t_in.size] = t_in
        keep = keep_on_device or s.get("keepOnDevice", False)
        out = None if keep else np.empty(render, dtype=np.uint32)
        stats = L.SortStats()
        st = L.check(self.lib.gs_sorter_sort(
            self.handle, mvp.ctypes.data, idx.ctypes.data if idx is not None else None, sort, render,
            pre.ctypes.data if pre is not None else None, tr.ctypes.data if tr is not None else None,
            out.ctypes.data if out is not None else None, None if keep else C.byref(stats)))
        reply = {"sortDone": True, "splatSortCount": sort, "splatRenderCount": render,
                 "sortTime": float(stats.device_ms), "status": st}
        if out is not None:
            # under the per-splat frustum cull the list holds only the kept splats (stats.result_count of them)
            reply["sortedIndexes"] = out[:stats.result_count] if (self.frustum_cull or self.visibility_cull) else out
            reply["stats"] = stats
        return self._reply(reply)

    # -- extras the HIP engine makes possible -------------------------------------------------------
    def sort_on_device(self, mvp, render_count, sort_count=None, indexes=None):
        """Enqueue a sort whose result stays in HBM for SplatMesh.render (no host round trip)."""
        return self._on_sort({"modelViewProj": mvp, "splatRenderCount": render_count,
                              "splatSortCount": render_count if sort_count is None else sort_count,
                              "indexesToSort": indexes, "transforms": np.tile(np.eye(4, dtype=np.float32).reshape(16), 32)},
                             keep_on_device=True)

    def sort_gathered(self, mvp, sort_count=None, keep_on_device=False):
        """Sort the device-resident indexesToSort list (and splatRenderCount) the last
        ``SplatTree.gather_scene_nodes_for_sort(..., sort_worker=self)`` produced."""
        mvp = np.ascontiguousarray(np.asarray(mvp, dtype=np.float64).astype(np.float32))
        tr = np.tile(np.eye(4, dtype=np.float32).reshape(16), L.GS_MAX_SCENES) if self.dynamic_mode else None
        stats = L.SortStats()
        # render count is held by the library; ask for the sorted list only when the caller wants it on the host
        out = None if keep_on_device else np.empty(self.gathered_count, dtype=np.uint32)
        sc = 0xFFFFFFFF if sort_count is None else int(sort_count)
        st = L.check(self.lib.gs_sorter_sort_gathered(self.handle, mvp.ctypes.data, sc, None,
                                                      tr.ctypes.data if tr is not None else None,
                                                      out.ctypes.data if out is not None else None,
                                                      None if keep_on_device else C.byref(stats)))
        if out is not None and (self.frustum_cull or self.visibility_cull):
            out = out[:stats.result_count]
        return {"sortDone": True, "status": st, "sortTime": float(stats.device_ms), "sortedIndexes": out, "stats": stats,
                "splatRenderCount": self.gathered_count}

    def set_frustum_cull(self, enable=True):
        """Fuse a per-splat frustum cull into full sorts (gs_sorter_set_frustum_cull): the result is the reference's
        sorted list restricted to the splats that can reach the frame for this modelViewProj."""
        L.check(self.lib.gs_sorter_set_frustum_cull(self.handle, 1 if enable else 0))
        self.frustum_cull = bool(enable)

    def set_visibility_cull(self, enable=True):
        """Exact per-splat cull (gs_sorter_set_visibility_cull): a full sort keeps the list positions whose splat survived
        ``SplatMesh.project`` of this frame's camera (and strip).  Needs ``mesh.use_sorter_result(self, ...)`` (the bind)
        and, every frame, ``mesh.project()`` before the sort."""
        L.check(self.lib.gs_sorter_set_visibility_cull(self.handle, 1 if enable else 0))
        self.visibility_cull = bool(enable)

    def keep_bits(self, count):
        """Test hook: the cull's keep flag per list position of the last sort."""
        words = np.empty((count + 31) // 32, dtype=np.uint32)
        L.check(self.lib.gs_sorter_debug_read(self.handle, 3, words.ctypes.data, words.size))
        return np.unpackbits(words.view(np.uint8), bitorder="little")[:count].astype(bool)

    def last_stats(self):
        stats = L.SortStats()
        st = L.check(self.lib.gs_sorter_last_stats(self.handle, C.byref(stats)))
        return stats, st

    def debug_read(self, what, count):
        out = np.empty(count, dtype=np.uint32 if what == 2 else np.int32)
        L.check(self.lib.gs_sorter_debug_read(self.handle, what, out.ctypes.data, count))
        return out

    def terminate(self):                                   # Viewer.js:1311
        if self.handle:
            self.lib.gs_sorter_destroy(self.handle)
            self.handle = C.c_void_p()

    close = terminate

    def __del__(self):
        try:
            self.terminate()
        except Exception:
            pass


def create_sort_worker(context, splat_count, use_shared_memory=True, enable_simd_in_sort=True,
                       integer_based_sort=True, dynamic_mode=False, splat_sort_distance_map_precision=DEFAULT_PRECISION):
    """Same argument list as the reference's createSortWorker (SortWorker.js:202-203); the two WASM-only flags
    are accepted and ignored."""
    del use_shared_memory, enable_simd_in_sort
    return SortWorker(context, splat_count, integer_based_sort, dynamic_mode, splat_sort_distance_map_precision)
