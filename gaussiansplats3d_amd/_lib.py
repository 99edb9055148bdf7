"""ctypes binding of libgsplat_hip.so (the C ABI in include/gsplat_hip.h).

There is no CPU fallback: importing works anywhere (so the symbol table can be checked on a CPU box), but every
compute entry point needs a gfx950 device and raises :class:`GsError` otherwise.
"""
import ctypes as C
import os
import subprocess
import atexit
import weakref

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("GSPLAT_HIP_LIB") or os.path.join(CSRC, "libgsplat_hip.so")   # override: A/B of builds
BLENDPROF_LIB_PATH = os.path.join(CSRC, "libgsplat_hip_blendprof.so")   # -DGS_BLEND_PROFILE build (tools/blend_lanes.py)

GS_OK, GS_WARN_KEY_CLAMPED, GS_WARN_FRAME_TRUNCATED = 0, 1, 2
GS_ERR_INVALID, GS_ERR_HIP, GS_ERR_NOMEM, GS_ERR_CAPACITY, GS_ERR_UNSUPPORTED = -1, -2, -3, -4, -5
GS_SORT_INTEGER, GS_SORT_DYNAMIC = 1, 2
GS_MESH_COV_HALF, GS_MESH_SH_U8, GS_MESH_KEEP_ORDER = 1, 2, 4
GS_CAM_ANTIALIASED, GS_CAM_POINT_CLOUD, GS_CAM_ORTHOGRAPHIC, GS_CAM_FADE_IN, GS_CAM_SCENE_EFFECTS, GS_CAM_DYNAMIC = 1, 2, 4, 8, 16, 32
GS_CTX_SINGLE_STREAM, GS_CTX_STAGE_TIMING, GS_CTX_FORK_JOIN = 1, 2, 4
GS_TILE = 16
GS_BIN = 32          # blend workgroups are per 32-px bin (2x2 tiles); entry lists per list bin (RenderStats.list_bin_px)
GS_MAX_SCENES = 32


class GsError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"libgsplat_hip status {status}: {message}")
        self.status = status


class SortStats(C.Structure):
    _fields_ = [("device_ms", C.c_float), ("key_min", C.c_int32), ("key_max", C.c_int32), ("clamped", C.c_uint32),
                ("passes", C.c_uint32), ("result_count", C.c_uint32)]


class Camera(C.Structure):
    _fields_ = [("view", C.c_float * 16), ("proj", C.c_float * 16), ("cam_pos", C.c_float * 3),
                ("focal", C.c_float * 2), ("width", C.c_uint32), ("height", C.c_uint32), ("splat_scale", C.c_float),
                ("kernel2d", C.c_float), ("max_splat_px", C.c_float), ("inv_focal_adj", C.c_float),
                ("sh_degree", C.c_uint32), ("flags", C.c_uint32), ("tile_row_begin", C.c_uint32),
                ("tile_row_end", C.c_uint32),
                ("ortho_zoom", C.c_float), ("scene_center", C.c_float * 3), ("fade_start_radius", C.c_float),
                ("view_matrix", C.c_float * 16)]


class SceneParams(C.Structure):
    _fields_ = [("scene_count", C.c_uint32), ("pad", C.c_uint32), ("transforms", (C.c_float * 16) * 32),
                ("inv_cam_pos", (C.c_float * 4) * 32), ("opacity", C.c_float * 32), ("visible", C.c_uint32 * 32),
                ("sh8_min", C.c_float * 32), ("sh8_max", C.c_float * 32)]


class RenderStats(C.Structure):
    _fields_ = [("device_ms", C.c_float), ("project_ms", C.c_float), ("bin_ms", C.c_float),
                ("tile_sort_ms", C.c_float), ("blend_ms", C.c_float), ("visible_splats", C.c_uint32),
                ("tile_entries", C.c_uint64), ("entry_capacity", C.c_uint32), ("overflowed", C.c_uint32),
                ("tiles16", C.c_uint64), ("list_bin_px", C.c_uint32), ("flags", C.c_uint32),
                ("entries_scanned", C.c_uint64), ("splats_walked", C.c_uint64), ("halves_evaluated", C.c_uint64)]


class Destination(C.Structure):
    """gs_destination: what the splats are depth-tested against and blended over (gs_mesh_set_destination)."""
    _fields_ = [("depth_host", C.c_void_p), ("depth_dev", C.c_void_p), ("rgba_host", C.c_void_p), ("rgba_dev", C.c_void_p),
                ("width", C.c_uint32), ("height", C.c_uint32), ("flags", C.c_uint32), ("pad", C.c_uint32)]


GS_DEST_DEPTH_UNORM24 = 1
GS_DRAW_FP32, GS_DRAW_ROP8, GS_DRAW_ROP8_FULL = 0, 1, 2


class TreeInfo(C.Structure):
    _fields_ = [("leaves", C.c_uint32), ("all_leaves", C.c_uint32), ("nodes", C.c_uint32), ("splats", C.c_uint32),
                ("scene_min", C.c_double * 3), ("scene_max", C.c_double * 3)]


class GatherParams(C.Structure):
    _fields_ = [("model_view", C.c_double * 16), ("fov_y_deg", C.c_double), ("render_width", C.c_double),
                ("render_height", C.c_double), ("gather_all", C.c_uint32), ("pad", C.c_uint32)]


class AssetInfo(C.Structure):
    _fields_ = [("splat_count", C.c_uint32), ("sh_degree", C.c_uint32), ("compression_level", C.c_uint32),
                ("sh_level", C.c_uint32), ("scene_center", C.c_float * 3), ("sh_min", C.c_float), ("sh_max", C.c_float)]


GS_ASSET_PLY, GS_ASSET_KSPLAT = 1, 2

# every symbol include/gsplat_hip.h declares: (restype, argtypes)
_VP = C.c_void_p
SYMBOLS = {
    "gs_last_error": (C.c_char_p, []),
    "gs_abi_version": (C.c_int, []),
    "gs_mesh_debug_set_entry_capacity": (C.c_int, [_VP, C.c_uint32]),
    "gs_device_count": (C.c_int, []),
    "gs_context_create": (C.c_int, [C.c_int, _VP, C.POINTER(_VP)]),
    "gs_context_create_ex": (C.c_int, [C.c_int, _VP, C.c_uint32, C.POINTER(_VP)]),
    "gs_context_destroy": (None, [_VP]),
    "gs_context_synchronize": (C.c_int, [_VP]),
    "gs_context_set_stage_timing": (C.c_int, [_VP, C.c_int]),
    "gs_sorter_create": (C.c_int, [_VP, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(_VP)]),
    "gs_sorter_destroy": (None, [_VP]),
    "gs_sorter_upload_centers": (C.c_int, [_VP, C.c_uint32, C.c_uint32, _VP, _VP]),
    "gs_sorter_sort": (C.c_int, [_VP, _VP, _VP, C.c_uint32, C.c_uint32, _VP, _VP, _VP, C.POINTER(SortStats)]),
    "gs_sorter_sort_gathered": (C.c_int, [_VP, _VP, C.c_uint32, _VP, _VP, _VP, C.POINTER(SortStats)]),
    "gs_sorter_bind_mesh": (C.c_int, [_VP, _VP]),
    "gs_sorter_set_frustum_cull": (C.c_int, [_VP, C.c_int]),
    "gs_sorter_set_visibility_cull": (C.c_int, [_VP, C.c_int]),
    "gs_sorter_debug_read": (C.c_int, [_VP, C.c_int, _VP, C.c_uint32]),
    "gs_asset_open": (C.c_int, [_VP, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(_VP)]),
    "gs_asset_close": (None, [_VP]),
    "gs_asset_get_info": (C.c_int, [_VP, C.POINTER(AssetInfo)]),
    "gs_asset_fill": (C.c_int, [_VP, C.c_uint32, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "gs_tree_create": (C.c_int, [_VP, _VP, _VP, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(_VP)]),
    "gs_tree_destroy": (None, [_VP]),
    "gs_tree_get_info": (C.c_int, [_VP, C.POINTER(TreeInfo)]),
    "gs_tree_read": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP]),
    "gs_tree_gather": (C.c_int, [_VP, C.POINTER(GatherParams), _VP, C.POINTER(C.c_uint32), _VP]),
    "gs_sorter_last_stats": (C.c_int, [_VP, C.POINTER(SortStats)]),
    "gs_mesh_create": (C.c_int, [_VP, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(_VP)]),
    "gs_mesh_destroy": (None, [_VP]),
    "gs_mesh_upload": (C.c_int, [_VP, C.c_uint32, C.c_uint32, _VP, _VP, _VP, _VP, _VP]),
    "gs_mesh_upload_sh_u8": (C.c_int, [_VP, C.c_uint32, C.c_uint32, _VP]),
    "gs_mesh_upload_scene_indexes": (C.c_int, [_VP, C.c_uint32, C.c_uint32, _VP]),
    "gs_mesh_set_scenes": (C.c_int, [_VP, C.POINTER(SceneParams)]),
    "gs_mesh_project": (C.c_int, [_VP, C.POINTER(Camera)]),
    "gs_mesh_set_destination": (C.c_int, [_VP, C.POINTER(Destination)]),
    "gs_group_unique_id": (C.c_int, [_VP]),
    "gs_group_create": (C.c_int, [_VP, _VP, C.c_uint32, C.c_uint32, C.POINTER(_VP)]),
    "gs_group_destroy": (None, [_VP]),
    "gs_group_gather_strips": (C.c_int, [_VP, _VP, _VP, C.c_uint32, _VP, _VP, C.c_uint32]),
    "gs_group_set_overlap": (C.c_int, [_VP, C.c_int]),
    "gs_group_wait": (C.c_int, [_VP]),
    "gs_group_render_gather": (C.c_int, [_VP, _VP, C.POINTER(Camera), _VP, _VP, C.c_uint32, _VP, _VP, C.c_uint32, _VP]),
    "gs_mesh_render": (C.c_int, [_VP, C.POINTER(Camera), _VP, _VP, C.c_uint32, _VP, _VP, C.POINTER(RenderStats)]),
    "gs_mesh_debug_read": (C.c_int, [_VP, C.c_int, _VP, C.c_uint32]),
    "gs_mesh_debug_rop8": (C.c_int, [_VP, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _VP]),
    "gs_mesh_set_deep_pass": (C.c_int, [_VP, C.c_int]),
    "gs_mesh_set_draw_mode": (C.c_int, [_VP, C.c_uint32]),
    "gs_mesh_last_stats": (C.c_int, [_VP, C.POINTER(RenderStats)]),
    "gs_mesh_kernel_time": (C.c_int, [_VP, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint32)]),
}

_lib = None


def build(force=False):
    """Compile csrc/*.hip for gfx950 into csrc/libgsplat_hip.so (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", CSRC, "-j8", "all", "blendprof"]     # + the lane-counting measurement build of the blend
    if force:
        subprocess.check_call(["make", "-C", CSRC, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return LIB_PATH


def load():
    """dlopen the library and type every entry point.  Fails loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GsError(GS_ERR_HIP, f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                      "(there is no CPU fallback)")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(status):
    if status < 0:
        raise GsError(status, load().gs_last_error().decode("utf-8", "replace"))
    return status


_live_contexts = weakref.WeakSet()


def _close_live_contexts():
    """Interpreter exit: destroy contexts (and their sorters / meshes) while the HIP runtime is still up.  Objects that are
    only finalised during module teardown would call hipFree / hipEventDestroy after the runtime's own atexit handler."""
    for ctx in list(_live_contexts):
        try:
            ctx.close()
        except Exception:
            pass


atexit.register(_close_live_contexts)


class Context:
    """One per GPU (gs_context).  `stream`: a raw hipStream_t (e.g. torch.cuda.current_stream().cuda_stream)."""

    def __init__(self, device=0, stream=None, single_stream=None, stage_timing=False, fork_join=False):
        """single_stream: True = sorts and draws share one stream (a frame is strictly sort -> draw), False = the sorter
        and the vertex stage get streams of their own (the reference's worker-thread shape), None = $GSPLAT_SERIAL.
        stage_timing: time the stages of every sort / draw, not only of those that return statistics (GS_CTX_STAGE_TIMING)."""
        self.lib = load()
        self.handle = _VP()
        st = _VP(stream) if stream else None
        if fork_join and single_stream:
            raise ValueError("fork_join needs the streams single_stream=True removes")
        if single_stream is None and fork_join:
            # (this combination used to fall through to gs_context_create and silently drop the flag)
            serial = os.environ.get("GSPLAT_SERIAL", "")
            if serial not in ("", "0"):
                raise ValueError("fork_join cannot be combined with $GSPLAT_SERIAL")
            single_stream = False
        if single_stream is None:
            check(self.lib.gs_context_create(int(device), st, C.byref(self.handle)))
        else:
            check(self.lib.gs_context_create_ex(int(device), st, (GS_CTX_SINGLE_STREAM if single_stream else 0) |
                                                (GS_CTX_FORK_JOIN if fork_join else 0), C.byref(self.handle)))
        if stage_timing:
            self.set_stage_timing(True)
        self.device = int(device)
        self._children = weakref.WeakSet()     # sorters / meshes: must be destroyed before the context
        _live_contexts.add(self)

    def _adopt(self, child):
        self._children.add(child)

    def synchronize(self):
        check(self.lib.gs_context_synchronize(self.handle))

    def set_stage_timing(self, enable):
        check(self.lib.gs_context_set_stage_timing(self.handle, int(bool(enable))))

    def close(self):
        if self.handle:
            for child in list(self._children):   # a child outliving its context would be a use-after-free
                child.close()
            self.lib.gs_context_destroy(self.handle)
            self.handle = _VP()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
