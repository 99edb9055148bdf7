/*
 * gsplat_addon.c — thin Node N-API binding over the C ABI of include/gsplat_hip.h.
 * JavaScript callers keep the reference's interfaces (node/gsplat.js mirrors createSortWorker and the
 * SplatMesh render seam); this file only marshals typed arrays into plain pointers.  No compute lives here.
 * Build: make -C node   (gcc, /usr/include/node/node_api.h; links ../gaussiansplats3d_amd/csrc/libgsplat_hip.so)
 */
#ifndef NAPI_VERSION
#define NAPI_VERSION 6            /* napi_set_instance_data */
#endif
#include <node_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <pthread.h>

#include "../include/gsplat_hip.h"

/* The C ABI wants calls on one context serialised by the caller.  JavaScript is single-threaded, but sorterSortAsync runs
 * gs_sorter_sort on a libuv pool thread (the reference's sort runs in a Web Worker, src/worker/SortWorker.js), so every
 * library call of this addon takes a lock: a draw issued while a sort is in flight waits for it (~1 ms), never races it.
 * The lock belongs to the addon INSTANCE (one per JavaScript thread: the main thread and every worker_threads Worker get
 * their own), not to the process: ranks of a multi-GPU group that live in one process as Workers, each with its own
 * context, must not serialise on each other - ncclCommInitRank and the root's gather block until every rank has called,
 * and a process-wide lock held across them would deadlock two ranks of one process.  Objects (contexts, sorters, meshes)
 * cannot cross JavaScript threads (napi externals are not transferable), so per-instance is per-context. */
typedef struct { pthread_mutex_t lock; } addon_state;
static addon_state* state_of(napi_env env) {
    void* d = NULL;
    napi_get_instance_data(env, &d);
    return (addon_state*)d;
}
static void state_free(napi_env env, void* data, void* hint) {
    (void)env; (void)hint;
    pthread_mutex_destroy(&((addon_state*)data)->lock);
    free(data);
}
#define LOCKED(stmt) do { pthread_mutex_t* l_ = &state_of(env)->lock; pthread_mutex_lock(l_); stmt; pthread_mutex_unlock(l_); } while (0)

#define NAPI_OK(call)                                                        \
    do {                                                                     \
        if ((call) != napi_ok) {                                             \
            napi_throw_error(env, NULL, "N-API call failed: " #call);        \
            return NULL;                                                     \
        }                                                                    \
    } while (0)

static napi_value throw_gs(napi_env env, int status) {
    char msg[640];
    snprintf(msg, sizeof msg, "libgsplat_hip status %d: %s", status, gs_last_error());
    napi_throw_error(env, "GS_ERROR", msg);
    return NULL;
}

/* typed array / ArrayBuffer / null -> pointer (+ byte length) */
static int get_bytes(napi_env env, napi_value v, void** data, size_t* bytes) {
    napi_valuetype t;
    *data = NULL;
    *bytes = 0;
    if (napi_typeof(env, v, &t) != napi_ok) return 0;
    if (t == napi_null || t == napi_undefined) return 1;
    bool is = false;
    if (napi_is_typedarray(env, v, &is) == napi_ok && is) {
        napi_typedarray_type tt;
        size_t len, off;
        napi_value ab;
        if (napi_get_typedarray_info(env, v, &tt, &len, data, &ab, &off) != napi_ok) return 0;
        static const size_t w[] = {1, 1, 1, 2, 2, 4, 4, 4, 8, 8, 8};
        *bytes = len * w[tt];
        return 1;
    }
    if (napi_is_arraybuffer(env, v, &is) == napi_ok && is) return napi_get_arraybuffer_info(env, v, data, bytes) == napi_ok;
    return 0;
}

static void* get_external(napi_env env, napi_value v) {
    void* p = NULL;
    napi_valuetype t;
    if (napi_typeof(env, v, &t) != napi_ok || t != napi_external) return NULL;
    napi_get_value_external(env, v, &p);
    return p;
}

static uint32_t get_u32(napi_env env, napi_value v) {
    uint32_t x = 0;
    napi_get_value_uint32(env, v, &x);
    return x;
}
static double get_f64(napi_env env, napi_value v) {
    double x = 0;
    napi_get_value_double(env, v, &x);
    return x;
}
static napi_value num(napi_env env, double v) {
    napi_value r;
    napi_create_double(env, v, &r);
    return r;
}
static void set(napi_env env, napi_value obj, const char* k, double v) { napi_set_named_property(env, obj, k, num(env, v)); }

#define ARGS(n)                                                             \
    size_t argc = n;                                                        \
    napi_value argv[n];                                                     \
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));          \
    if (argc < n) { napi_throw_type_error(env, NULL, "too few arguments"); return NULL; }

static napi_value DeviceCount(napi_env env, napi_callback_info info) {
    (void)info;
    int n = gs_device_count();
    if (n < 0) return throw_gs(env, n);
    return num(env, n);
}

static napi_value ContextCreate(napi_env env, napi_callback_info info) {
    ARGS(1)
    gs_context* ctx = NULL;
    int st;

    LOCKED(st = gs_context_create((int)get_u32(env, argv[0]), NULL, &ctx));
    if (st < 0) return throw_gs(env, st);
    napi_value r;
    NAPI_OK(napi_create_external(env, ctx, NULL, NULL, &r));
    return r;
}
static napi_value ContextDestroy(napi_env env, napi_callback_info info) {
    ARGS(1)
    LOCKED(gs_context_destroy((gs_context*)get_external(env, argv[0])));
    return NULL;
}

/* sorterCreate(ctx, maxSplatCount, flags, precisionBits) */
static napi_value SorterCreate(napi_env env, napi_callback_info info) {
    ARGS(4)
    gs_sorter* s = NULL;
    int st;

    LOCKED(st = gs_sorter_create((gs_context*)get_external(env, argv[0]), get_u32(env, argv[1]), get_u32(env, argv[2]),
                              get_u32(env, argv[3]), &s));
    if (st < 0) return throw_gs(env, st);
    napi_value r;
    NAPI_OK(napi_create_external(env, s, NULL, NULL, &r));
    return r;
}
static napi_value SorterDestroy(napi_env env, napi_callback_info info) {
    ARGS(1)
    LOCKED(gs_sorter_destroy((gs_sorter*)get_external(env, argv[0])));
    return NULL;
}
/* sorterUploadCenters(sorter, from, count, centers(Int32Array|Float32Array|ArrayBuffer), sceneIndexes|null) */
static napi_value SorterUploadCenters(napi_env env, napi_callback_info info) {
    ARGS(5)
    void *c, *sc;
    size_t cb, sb;
    if (!get_bytes(env, argv[3], &c, &cb) || !get_bytes(env, argv[4], &sc, &sb)) { napi_throw_type_error(env, NULL, "centers / sceneIndexes"); return NULL; }
    const uint32_t count = get_u32(env, argv[2]);
    if (cb < (size_t)count * 16 || (sc && sb < (size_t)count * 4)) { napi_throw_range_error(env, NULL, "buffer shorter than count"); return NULL; }
    int st;

    LOCKED(st = gs_sorter_upload_centers((gs_sorter*)get_external(env, argv[0]), get_u32(env, argv[1]), count, c, (const uint32_t*)sc));
    if (st < 0) return throw_gs(env, st);
    return NULL;
}
/* sorterSort(sorter, mvp Float32Array(16), indexes|null, sortCount, renderCount, precomputed|null, transforms|null,
 *            out Uint32Array|null) -> {status, sortTime, keyMin, keyMax, clamped, passes} */
static napi_value SorterSort(napi_env env, napi_callback_info info) {
    ARGS(8)
    void *mvp, *idx, *pre, *tr, *out;
    size_t mb, ib, pb, tb, ob;
    if (!get_bytes(env, argv[1], &mvp, &mb) || mb < 64 || !get_bytes(env, argv[2], &idx, &ib) ||
        !get_bytes(env, argv[5], &pre, &pb) || !get_bytes(env, argv[6], &tr, &tb) || !get_bytes(env, argv[7], &out, &ob)) {
        napi_throw_type_error(env, NULL, "sorterSort: bad buffer argument");
        return NULL;
    }
    const uint32_t sortc = get_u32(env, argv[3]), renderc = get_u32(env, argv[4]);
    if ((idx && ib < (size_t)renderc * 4) || (out && ob < (size_t)renderc * 4) || (tr && tb < 16 * 4 * GS_MAX_SCENES)) {
        napi_throw_range_error(env, NULL, "sorterSort: buffer shorter than renderCount");
        return NULL;
    }
    gs_sort_stats stats;
    memset(&stats, 0, sizeof stats);
    int st;

    LOCKED(st = gs_sorter_sort((gs_sorter*)get_external(env, argv[0]), (const float*)mvp, (const uint32_t*)idx, sortc, renderc,
                            pre, (const float*)tr, (uint32_t*)out, out ? &stats : NULL));
    if (st < 0) return throw_gs(env, st);
    napi_value r;
    NAPI_OK(napi_create_object(env, &r));
    set(env, r, "status", st);
    set(env, r, "sortTime", stats.device_ms);
    set(env, r, "keyMin", stats.key_min);
    set(env, r, "keyMax", stats.key_max);
    set(env, r, "clamped", stats.clamped);
    set(env, r, "passes", stats.passes);
    set(env, r, "resultCount", stats.result_count);
    return r;
}

/* sorterSortAsync(sorter, mvp, indexes|null, sortCount, renderCount, precomputed|null, transforms|null, out|null, callback)
 * The same sort on a libuv pool thread: returns at once, callback(error|null, {status, sortTime, ...}) runs on the main
 * thread when the result is in `out` - the shape of the reference's worker: postMessage returns immediately, sortDone
 * arrives later (src/worker/SortWorker.js:62-80 -> src/Viewer.js:1243-1264).  The typed arrays are referenced until then. */
typedef struct {
    napi_async_work work;
    napi_ref cb, keep[3];
    gs_sorter* sorter;
    pthread_mutex_t* lock;          /* the addon instance's lock (sort_execute has no usable env) */
    float mvp[16];
    float transforms[16 * GS_MAX_SCENES];
    int has_tr;
    const uint32_t* idx;
    const void* pre;
    uint32_t* out;
    uint32_t sortc, renderc;
    gs_sort_stats stats;
    int status;
    char err[512];
} sort_job;

static void sort_execute(napi_env env, void* data) {
    (void)env;
    sort_job* j = (sort_job*)data;
    pthread_mutex_lock(j->lock);
    j->status = gs_sorter_sort(j->sorter, j->mvp, j->idx, j->sortc, j->renderc, j->pre, j->has_tr ? j->transforms : NULL, j->out,
                               j->out ? &j->stats : NULL);
    if (j->status < 0) snprintf(j->err, sizeof j->err, "libgsplat_hip status %d: %s", j->status, gs_last_error());   /* thread-local */
    pthread_mutex_unlock(j->lock);
}

static void sort_complete(napi_env env, napi_status status, void* data) {
    sort_job* j = (sort_job*)data;
    napi_value cb, undef, args[2], r;
    napi_get_undefined(env, &undef);
    napi_get_reference_value(env, j->cb, &cb);
    if (status != napi_ok || j->status < 0) {
        napi_value msg;
        napi_create_string_utf8(env, status != napi_ok ? "sorterSortAsync: the work item was cancelled" : j->err, NAPI_AUTO_LENGTH, &msg);
        napi_create_error(env, NULL, msg, &args[0]);
        args[1] = undef;
    } else {
        napi_get_null(env, &args[0]);
        napi_create_object(env, &r);
        set(env, r, "status", j->status);
        set(env, r, "sortTime", j->stats.device_ms);
        set(env, r, "keyMin", j->stats.key_min);
        set(env, r, "keyMax", j->stats.key_max);
        set(env, r, "clamped", j->stats.clamped);
        set(env, r, "passes", j->stats.passes);
        set(env, r, "resultCount", j->stats.result_count);
        args[1] = r;
    }
    napi_delete_reference(env, j->cb);
    for (int k = 0; k < 3; k++)
        if (j->keep[k]) napi_delete_reference(env, j->keep[k]);
    napi_delete_async_work(env, j->work);
    free(j);
    napi_value ignored;
    napi_call_function(env, undef, cb, 2, args, &ignored);
}

static napi_value SorterSortAsync(napi_env env, napi_callback_info info) {
    ARGS(9)
    void *mvp, *idx, *pre, *tr, *out;
    size_t mb, ib, pb, tb, ob;
    if (!get_bytes(env, argv[1], &mvp, &mb) || mb < 64 || !get_bytes(env, argv[2], &idx, &ib) ||
        !get_bytes(env, argv[5], &pre, &pb) || !get_bytes(env, argv[6], &tr, &tb) || !get_bytes(env, argv[7], &out, &ob)) {
        napi_throw_type_error(env, NULL, "sorterSortAsync: bad buffer argument");
        return NULL;
    }
    const uint32_t sortc = get_u32(env, argv[3]), renderc = get_u32(env, argv[4]);
    if ((idx && ib < (size_t)renderc * 4) || (out && ob < (size_t)renderc * 4) || (tr && tb < 16 * 4 * GS_MAX_SCENES)) {
        napi_throw_range_error(env, NULL, "sorterSortAsync: buffer shorter than renderCount");
        return NULL;
    }
    napi_valuetype ft;
    if (napi_typeof(env, argv[8], &ft) != napi_ok || ft != napi_function) {
        napi_throw_type_error(env, NULL, "sorterSortAsync: callback");
        return NULL;
    }
    sort_job* j = (sort_job*)calloc(1, sizeof *j);
    if (!j) { napi_throw_error(env, NULL, "out of memory"); return NULL; }
    j->sorter = (gs_sorter*)get_external(env, argv[0]);
    j->lock = &state_of(env)->lock;
    memcpy(j->mvp, mvp, 64);
    if (tr) { memcpy(j->transforms, tr, sizeof j->transforms); j->has_tr = 1; }
    j->idx = (const uint32_t*)idx; j->pre = pre; j->out = (uint32_t*)out;
    j->sortc = sortc; j->renderc = renderc;
    napi_create_reference(env, argv[8], 1, &j->cb);
    const int held[3] = {2, 5, 7};                         /* indexes, precomputed, out: alive until the callback */
    for (int k = 0; k < 3; k++) {
        napi_valuetype t;
        napi_typeof(env, argv[held[k]], &t);
        if (t == napi_object) napi_create_reference(env, argv[held[k]], 1, &j->keep[k]);
    }
    napi_value name;
    napi_create_string_utf8(env, "gsplat.sorterSort", NAPI_AUTO_LENGTH, &name);
    if (napi_create_async_work(env, NULL, name, sort_execute, sort_complete, j, &j->work) != napi_ok ||
        napi_queue_async_work(env, j->work) != napi_ok) {
        napi_throw_error(env, NULL, "sorterSortAsync: could not queue the work item");
        return NULL;
    }
    return NULL;
}

/* meshCreate(ctx, maxSplatCount, shDegree, flags) */
static napi_value MeshCreate(napi_env env, napi_callback_info info) {
    ARGS(4)
    gs_mesh* m = NULL;
    int st;

    LOCKED(st = gs_mesh_create((gs_context*)get_external(env, argv[0]), get_u32(env, argv[1]), get_u32(env, argv[2]), get_u32(env, argv[3]), &m));
    if (st < 0) return throw_gs(env, st);
    napi_value r;
    NAPI_OK(napi_create_external(env, m, NULL, NULL, &r));
    return r;
}
static napi_value MeshDestroy(napi_env env, napi_callback_info info) {
    ARGS(1)
    LOCKED(gs_mesh_destroy((gs_mesh*)get_external(env, argv[0])));
    return NULL;
}
/* meshUpload(mesh, from, count, centers F32, covF32|null, covF16(Uint16)|null, rgba U8, shF16(Uint16)|null) */
static napi_value MeshUpload(napi_env env, napi_callback_info info) {
    ARGS(8)
    void* p[5];
    size_t b[5];
    for (int i = 0; i < 5; i++)
        if (!get_bytes(env, argv[3 + i], &p[i], &b[i])) { napi_throw_type_error(env, NULL, "meshUpload: bad buffer"); return NULL; }
    const uint32_t count = get_u32(env, argv[2]);
    if (b[0] < (size_t)count * 12 || b[3] < (size_t)count * 4 || (p[1] && b[1] < (size_t)count * 24) || (p[2] && b[2] < (size_t)count * 12)) {
        napi_throw_range_error(env, NULL, "meshUpload: buffer shorter than count");
        return NULL;
    }
    int st;

    LOCKED(st = gs_mesh_upload((gs_mesh*)get_external(env, argv[0]), get_u32(env, argv[1]), count, (const float*)p[0], (const float*)p[1],
                            (const uint16_t*)p[2], (const uint8_t*)p[3], (const uint16_t*)p[4]));
    if (st < 0) return throw_gs(env, st);
    return NULL;
}
/* meshRender(mesh, cam{view,proj,camPos,focal,width,height,splatScale,kernel2d,maxSplatPx,invFocalAdj,shDegree,flags,
 *            tileRowBegin,tileRowEnd}, sortedIndexes|null, sorter|null, renderCount, out Uint8Array) -> stats object */
/* camera object -> gs_camera; throws and returns 0 on a malformed object */
static int parse_camera(napi_env env, napi_value obj, gs_camera* out) {
    gs_camera cam;
    memset(&cam, 0, sizeof cam);
    napi_value v;
    void* d;
    size_t nb;
#define F32ARR(key, dst, n)                                                                              \
    if (napi_get_named_property(env, obj, key, &v) != napi_ok || !get_bytes(env, v, &d, &nb) || nb < (n) * 4) { \
        napi_throw_type_error(env, NULL, "camera." key); return 0; }                                     \
    memcpy(dst, d, (n) * 4);
    F32ARR("view", cam.view, 16)
    F32ARR("proj", cam.proj, 16)
    F32ARR("camPos", cam.cam_pos, 3)
    F32ARR("focal", cam.focal, 2)
#define NUM(key, dst, type) if (napi_get_named_property(env, obj, key, &v) != napi_ok) { napi_throw_type_error(env, NULL, "camera." key); return 0; } dst = (type)get_f64(env, v);
    NUM("width", cam.width, uint32_t)
    NUM("height", cam.height, uint32_t)
    NUM("splatScale", cam.splat_scale, float)
    NUM("kernel2d", cam.kernel2d, float)
    NUM("maxSplatPx", cam.max_splat_px, float)
    NUM("invFocalAdj", cam.inv_focal_adj, float)
    NUM("shDegree", cam.sh_degree, uint32_t)
    NUM("flags", cam.flags, uint32_t)
    NUM("tileRowBegin", cam.tile_row_begin, uint32_t)
    NUM("tileRowEnd", cam.tile_row_end, uint32_t)
    /* optional: the uniforms of the orthographic / fade-in / dynamic permutations (absent = zero) */
    {
        bool has = false;
        if (napi_has_named_property(env, obj, "orthoZoom", &has) == napi_ok && has) { NUM("orthoZoom", cam.ortho_zoom, float) }
        if (napi_has_named_property(env, obj, "fadeStartRadius", &has) == napi_ok && has) { NUM("fadeStartRadius", cam.fade_start_radius, float) }
        if (napi_has_named_property(env, obj, "sceneCenter", &has) == napi_ok && has) { F32ARR("sceneCenter", cam.scene_center, 3) }
        if (napi_has_named_property(env, obj, "viewMatrix", &has) == napi_ok && has) { F32ARR("viewMatrix", cam.view_matrix, 16) }
    }
#undef F32ARR
#undef NUM
    *out = cam;
    return 1;
}

static napi_value MeshRender(napi_env env, napi_callback_info info) {
    ARGS(6)
    gs_camera cam;
    if (!parse_camera(env, argv[1], &cam)) return NULL;
    void *idx, *out;
    size_t ib, ob;
    if (!get_bytes(env, argv[2], &idx, &ib) || !get_bytes(env, argv[5], &out, &ob)) { napi_throw_type_error(env, NULL, "meshRender: bad buffer"); return NULL; }
    const uint32_t renderc = get_u32(env, argv[4]);
    if (idx && ib < (size_t)renderc * 4) { napi_throw_range_error(env, NULL, "sortedIndexes shorter than renderCount"); return NULL; }
    gs_render_stats stats;
    memset(&stats, 0, sizeof stats);
    int st;

    LOCKED(st = gs_mesh_render((gs_mesh*)get_external(env, argv[0]), &cam, (const uint32_t*)idx, (gs_sorter*)get_external(env, argv[3]),
                            renderc, (uint8_t*)out, NULL, &stats));
    if (st < 0) return throw_gs(env, st);
    napi_value r;
    NAPI_OK(napi_create_object(env, &r));
    set(env, r, "deviceMs", stats.device_ms);
    set(env, r, "projectMs", stats.project_ms);
    set(env, r, "binMs", stats.bin_ms);
    set(env, r, "tileSortMs", stats.tile_sort_ms);
    set(env, r, "blendMs", stats.blend_ms);
    set(env, r, "visibleSplats", stats.visible_splats);
    set(env, r, "tileEntries", (double)stats.tile_entries);
    set(env, r, "tiles16", (double)stats.tiles16);
    set(env, r, "listBinPx", stats.list_bin_px);
    set(env, r, "overflowed", stats.overflowed);
    set(env, r, "flags", (double)stats.flags);                 /* GS_DRAW_POOL_EXHAUSTED = 1 */
    set(env, r, "entriesScanned", (double)stats.entries_scanned);
    set(env, r, "splatsWalked", (double)stats.splats_walked);
    set(env, r, "halvesEvaluated", (double)stats.halves_evaluated);
    return r;
}

/* meshUploadShU8(mesh, from, count, sh Uint8Array(9|24 per splat)) — 8-bit SH of a GS_MESH_SH_U8 mesh */
static napi_value MeshUploadShU8(napi_env env, napi_callback_info info) {
    ARGS(4)
    void* p;
    size_t b;
    if (!get_bytes(env, argv[3], &p, &b) || !p) { napi_throw_type_error(env, NULL, "meshUploadShU8: bad buffer"); return NULL; }
    const uint32_t count = get_u32(env, argv[2]);
    if (b < (size_t)count * 9) { napi_throw_range_error(env, NULL, "meshUploadShU8: buffer shorter than count"); return NULL; }
    int st;
    LOCKED(st = gs_mesh_upload_sh_u8((gs_mesh*)get_external(env, argv[0]), get_u32(env, argv[1]), count, (const uint8_t*)p));
    if (st < 0) return throw_gs(env, st);
    return NULL;
}
/* meshUploadSceneIndexes(mesh, from, count, Uint32Array) — sceneIndexesTexture, SplatMesh.js:881-897 */
static napi_value MeshUploadSceneIndexes(napi_env env, napi_callback_info info) {
    ARGS(4)
    void* p;
    size_t b;
    if (!get_bytes(env, argv[3], &p, &b) || !p) { napi_throw_type_error(env, NULL, "meshUploadSceneIndexes: bad buffer"); return NULL; }
    const uint32_t count = get_u32(env, argv[2]);
    if (b < (size_t)count * 4) { napi_throw_range_error(env, NULL, "meshUploadSceneIndexes: buffer shorter than count"); return NULL; }
    int st;

    LOCKED(st = gs_mesh_upload_scene_indexes((gs_mesh*)get_external(env, argv[0]), get_u32(env, argv[1]), count, (const uint32_t*)p));
    if (st < 0) return throw_gs(env, st);
    return NULL;
}
/* meshSetScenes(mesh, {sceneCount, transforms F32(16*n), invCamPos F32(4*n), opacity F32(n), visible U32(n), sh8Min F32(n),
 *               sh8Max F32(n)}) — the per-scene uniforms of SplatMesh.updateUniforms (SplatMesh.js:1263-1276) */
static napi_value MeshSetScenes(napi_env env, napi_callback_info info) {
    ARGS(2)
    static gs_scene_params sp;                 /* 2.9 KB: keep it off the stack of the JS thread's callback */
    memset(&sp, 0, sizeof sp);
    napi_value v;
    NAPI_OK(napi_get_named_property(env, argv[1], "sceneCount", &v));
    sp.scene_count = get_u32(env, v);
    if (sp.scene_count < 1 || sp.scene_count > GS_MAX_SCENES) { napi_throw_range_error(env, NULL, "meshSetScenes: sceneCount"); return NULL; }
    for (uint32_t i = 0; i < GS_MAX_SCENES; i++) {
        sp.opacity[i] = 1.0f; sp.visible[i] = 1u; sp.sh8_min[i] = -1.0f; sp.sh8_max[i] = 1.0f;
        for (int k = 0; k < 4; k++) sp.transforms[i][5 * k] = 1.0f;
    }
    const struct { const char* key; void* dst; size_t per; } fields[] = {
        {"transforms", sp.transforms, 64}, {"invCamPos", sp.inv_cam_pos, 16}, {"opacity", sp.opacity, 4},
        {"visible", sp.visible, 4},        {"sh8Min", sp.sh8_min, 4},         {"sh8Max", sp.sh8_max, 4}};
    for (size_t f = 0; f < sizeof fields / sizeof fields[0]; f++) {
        bool has = false;
        NAPI_OK(napi_has_named_property(env, argv[1], fields[f].key, &has));
        if (!has) continue;
        void* d;
        size_t nb;
        NAPI_OK(napi_get_named_property(env, argv[1], fields[f].key, &v));
        if (!get_bytes(env, v, &d, &nb) || !d) continue;
        const size_t want = fields[f].per * sp.scene_count;
        if (nb < want) { napi_throw_range_error(env, NULL, "meshSetScenes: array shorter than sceneCount"); return NULL; }
        memcpy(fields[f].dst, d, want);
    }
    int st;

    LOCKED(st = gs_mesh_set_scenes((gs_mesh*)get_external(env, argv[0]), &sp));
    if (st < 0) return throw_gs(env, st);
    return NULL;
}

/* meshProject(mesh, camera): the vertex stage on its own (gs_mesh_project) */
static napi_value MeshProject(napi_env env, napi_callback_info info) {
    ARGS(2)
    gs_camera cam;
    if (!parse_camera(env, argv[1], &cam)) return NULL;
    int st;
    LOCKED(st = gs_mesh_project((gs_mesh*)get_external(env, argv[0]), &cam));
    if (st < 0) return throw_gs(env, st);
    return NULL;
}
static napi_value SorterSetVisibilityCull(napi_env env, napi_callback_info info) {
    ARGS(2)
    int st;
    LOCKED(st = gs_sorter_set_visibility_cull((gs_sorter*)get_external(env, argv[0]), (int)get_u32(env, argv[1])));
    if (st < 0) return throw_gs(env, st);
    return NULL;
}
/* groupUniqueId() -> Uint8Array(128) (ncclGetUniqueId: call on one rank, hand the bytes to the others) */
static napi_value GroupUniqueId(napi_env env, napi_callback_info info) {
    (void)info;
    void* data;
    napi_value ab, ta;
    NAPI_OK(napi_create_arraybuffer(env, GS_GROUP_ID_BYTES, &data, &ab));
    int st;
    LOCKED(st = gs_group_unique_id((uint8_t*)data));
    if (st < 0) return throw_gs(env, st);
    NAPI_OK(napi_create_typedarray(env, napi_uint8_array, GS_GROUP_ID_BYTES, ab, 0, &ta));
    return ta;
}
/* groupCreate(ctx, id Uint8Array(128)|null, worldSize, rank) */
static napi_value GroupCreate(napi_env env, napi_callback_info info) {
    ARGS(4)
    void* id;
    size_t nb;
    if (!get_bytes(env, argv[1], &id, &nb) || (id && nb < GS_GROUP_ID_BYTES)) { napi_throw_type_error(env, NULL, "groupCreate: id"); return NULL; }
    gs_group* g = NULL;
    int st;
    LOCKED(st = gs_group_create((gs_context*)get_external(env, argv[0]), (const uint8_t*)id, get_u32(env, argv[2]), get_u32(env, argv[3]), &g));
    if (st < 0) return throw_gs(env, st);
    napi_value r;
    NAPI_OK(napi_create_external(env, g, NULL, NULL, &r));
    return r;
}
static napi_value GroupDestroy(napi_env env, napi_callback_info info) {
    ARGS(1)
    LOCKED(gs_group_destroy((gs_group*)get_external(env, argv[0])));
    return NULL;
}
/* groupSetOverlap(group, enabled): the strip transfer of frame k beside the draw of frame k + 1 (gs_group_set_overlap) */
static napi_value GroupSetOverlap(napi_env env, napi_callback_info info) {
    ARGS(2)
    int st;
    LOCKED(st = gs_group_set_overlap((gs_group*)get_external(env, argv[0]), (int)get_u32(env, argv[1])));
    if (st < 0) return throw_gs(env, st);
    return NULL;
}
/* groupWait(group): every transfer issued so far has completed */
static napi_value GroupWait(napi_env env, napi_callback_info info) {
    ARGS(1)
    int st;
    LOCKED(st = gs_group_wait((gs_group*)get_external(env, argv[0])));
    if (st < 0) return throw_gs(env, st);
    return NULL;
}
/* meshSetDrawMode(mesh, mode): 0 = the fp32 composite rounded once, 1 = the reference's RGBA8 target, rounded after every splat,
 * over the splats in front of the saturation depth, 2 = the same over every list to its end (gs_mesh_set_draw_mode) */
static napi_value MeshSetDrawMode(napi_env env, napi_callback_info info) {
    size_t argc = 2; napi_value argv[2];
    napi_get_cb_info(env, info, &argc, argv, NULL, NULL);
    int st;
    LOCKED(st = gs_mesh_set_draw_mode((gs_mesh*)get_external(env, argv[0]), get_u32(env, argv[1])));
    if (st < 0) return throw_gs(env, st);
    return NULL;
}

/* meshSetDeepPass(mesh, enabled): scheduling only, the pixels do not change (gs_mesh_set_deep_pass) */
static napi_value MeshSetDeepPass(napi_env env, napi_callback_info info) {
    ARGS(2)
    int st;
    LOCKED(st = gs_mesh_set_deep_pass((gs_mesh*)get_external(env, argv[0]), (int)get_u32(env, argv[1])));
    if (st < 0) return throw_gs(env, st);
    return NULL;
}
/* groupRenderGather(group, mesh, camera, sortedIndexes Uint32Array|null, sorter|null, renderCount, rowBegin Uint32Array,
 *                   rowEnd Uint32Array, root, out Uint8Array|null) */
static napi_value GroupRenderGather(napi_env env, napi_callback_info info) {
    ARGS(10)
    gs_camera cam;
    if (!parse_camera(env, argv[2], &cam)) return NULL;
    void *idx, *rb, *re, *out;
    size_t ib, rbb, reb, ob;
    const uint32_t renderc = get_u32(env, argv[5]);
    if (!get_bytes(env, argv[3], &idx, &ib) || !get_bytes(env, argv[6], &rb, &rbb) || !get_bytes(env, argv[7], &re, &reb) ||
        !get_bytes(env, argv[9], &out, &ob) || !rb || !re || rbb != reb || (out && ob < (size_t)cam.width * cam.height * 4) ||
        (idx && ib < (size_t)renderc * 4)) {
        napi_throw_type_error(env, NULL, "groupRenderGather: index list / row tables / output buffer");
        return NULL;
    }
    int st;
    LOCKED(st = gs_group_render_gather((gs_group*)get_external(env, argv[0]), (gs_mesh*)get_external(env, argv[1]), &cam,
                                       (const uint32_t*)idx, (gs_sorter*)get_external(env, argv[4]), renderc, (const uint32_t*)rb,
                                       (const uint32_t*)re, get_u32(env, argv[8]), (uint8_t*)out));
    if (st < 0) return throw_gs(env, st);
    return num(env, st);
}

/* sorterBindMesh(sorter, mesh|null) */
static napi_value SorterBindMesh(napi_env env, napi_callback_info info) {
    ARGS(2)
    int st;

    LOCKED(st = gs_sorter_bind_mesh((gs_sorter*)get_external(env, argv[0]), (gs_mesh*)get_external(env, argv[1])));
    if (st < 0) return throw_gs(env, st);
    return NULL;
}
/* sorterSetFrustumCull(sorter, enable) */
static napi_value SorterSetFrustumCull(napi_env env, napi_callback_info info) {
    ARGS(2)
    int st;

    LOCKED(st = gs_sorter_set_frustum_cull((gs_sorter*)get_external(env, argv[0]), (int)get_u32(env, argv[1])));
    if (st < 0) return throw_gs(env, st);
    return NULL;
}
/* sorterSortGathered(sorter, mvp Float32Array(16), sortCount, out Uint32Array|null) -> {status, sortTime} */
static napi_value SorterSortGathered(napi_env env, napi_callback_info info) {
    ARGS(4)
    void *mvp, *out;
    size_t mb, ob;
    if (!get_bytes(env, argv[1], &mvp, &mb) || mb < 64 || !get_bytes(env, argv[3], &out, &ob)) {
        napi_throw_type_error(env, NULL, "sorterSortGathered: bad buffer argument");
        return NULL;
    }
    gs_sort_stats stats;
    memset(&stats, 0, sizeof stats);
    int st;

    LOCKED(st = gs_sorter_sort_gathered((gs_sorter*)get_external(env, argv[0]), (const float*)mvp, get_u32(env, argv[2]), NULL, NULL,
                                     (uint32_t*)out, out ? &stats : NULL));
    if (st < 0) return throw_gs(env, st);
    napi_value r;
    NAPI_OK(napi_create_object(env, &r));
    set(env, r, "status", st);
    set(env, r, "sortTime", stats.device_ms);
    set(env, r, "resultCount", stats.result_count);
    return r;
}

/* treeCreate(ctx|null, centers Float32Array(3n), keep Uint8Array|null, count, firstIndex, maxDepth, maxCentersPerNode) */
static napi_value TreeCreate(napi_env env, napi_callback_info info) {
    ARGS(7)
    void *c, *k;
    size_t cb, kb;
    if (!get_bytes(env, argv[1], &c, &cb) || !get_bytes(env, argv[2], &k, &kb)) { napi_throw_type_error(env, NULL, "treeCreate: bad buffer"); return NULL; }
    const uint32_t count = get_u32(env, argv[3]);
    if (cb < (size_t)count * 12 || (k && kb < count)) { napi_throw_range_error(env, NULL, "treeCreate: buffer shorter than count"); return NULL; }
    gs_tree* t = NULL;
    int st;

    LOCKED(st = gs_tree_create((gs_context*)get_external(env, argv[0]), (const float*)c, (const uint8_t*)k, count, get_u32(env, argv[4]),
                            get_u32(env, argv[5]), get_u32(env, argv[6]), &t));
    if (st < 0) return throw_gs(env, st);
    napi_value r;
    NAPI_OK(napi_create_external(env, t, NULL, NULL, &r));
    return r;
}
static napi_value TreeDestroy(napi_env env, napi_callback_info info) {
    ARGS(1)
    LOCKED(gs_tree_destroy((gs_tree*)get_external(env, argv[0])));
    return NULL;
}
/* treeInfo(tree) -> {leaves, allLeaves, nodes, splats} */
static napi_value TreeInfo(napi_env env, napi_callback_info info) {
    ARGS(1)
    gs_tree_info ti;
    int st;

    LOCKED(st = gs_tree_get_info((gs_tree*)get_external(env, argv[0]), &ti));
    if (st < 0) return throw_gs(env, st);
    napi_value r;
    NAPI_OK(napi_create_object(env, &r));
    set(env, r, "leaves", ti.leaves);
    set(env, r, "allLeaves", ti.all_leaves);
    set(env, r, "nodes", ti.nodes);
    set(env, r, "splats", ti.splats);
    return r;
}
/* treeRead(tree) -> {bounds Float64Array(6L) = min xyz | max xyz, centers Float64Array(3L), depths Uint32Array(L),
 *                    offsets Uint32Array(L+1), indexes Uint32Array(S)}: the leaves in nodesWithIndexes order (gs_tree_read) */
static napi_value TreeRead(napi_env env, napi_callback_info info) {
    ARGS(1)
    gs_tree* t = (gs_tree*)get_external(env, argv[0]);
    gs_tree_info ti;
    int st;
    LOCKED(st = gs_tree_get_info(t, &ti));
    if (st < 0) return throw_gs(env, st);
    const size_t L = ti.leaves, S = ti.splats;
    void *bnd, *ctr, *dep, *off, *idx;
    napi_value ab[5], ta[5], r;
    NAPI_OK(napi_create_arraybuffer(env, 48 * L, &bnd, &ab[0]));
    NAPI_OK(napi_create_arraybuffer(env, 24 * L, &ctr, &ab[1]));
    NAPI_OK(napi_create_arraybuffer(env, 4 * L, &dep, &ab[2]));
    NAPI_OK(napi_create_arraybuffer(env, 4 * (L + 1), &off, &ab[3]));
    NAPI_OK(napi_create_arraybuffer(env, 4 * S, &idx, &ab[4]));
    LOCKED(st = gs_tree_read(t, (double*)bnd, (double*)ctr, (uint32_t*)dep, (uint32_t*)off, (uint32_t*)idx));
    if (st < 0) return throw_gs(env, st);
    NAPI_OK(napi_create_typedarray(env, napi_float64_array, 6 * L, ab[0], 0, &ta[0]));
    NAPI_OK(napi_create_typedarray(env, napi_float64_array, 3 * L, ab[1], 0, &ta[1]));
    NAPI_OK(napi_create_typedarray(env, napi_uint32_array, L, ab[2], 0, &ta[2]));
    NAPI_OK(napi_create_typedarray(env, napi_uint32_array, L + 1, ab[3], 0, &ta[3]));
    NAPI_OK(napi_create_typedarray(env, napi_uint32_array, S, ab[4], 0, &ta[4]));
    NAPI_OK(napi_create_object(env, &r));
    static const char* names[5] = {"bounds", "centers", "depths", "offsets", "indexes"};
    for (int k = 0; k < 5; k++) NAPI_OK(napi_set_named_property(env, r, names[k], ta[k]));
    return r;
}
/* treeGather(tree, modelView Float64Array(16), fovYDeg, renderWidth, renderHeight, gatherAll, sorter|null,
 *            out Uint32Array|null) -> splatRenderCount */
static napi_value TreeGather(napi_env env, napi_callback_info info) {
    ARGS(8)
    gs_gather_params gp;
    memset(&gp, 0, sizeof gp);
    void *mv, *out;
    size_t mb, ob;
    if (!get_bytes(env, argv[1], &mv, &mb) || mb < 128 || !get_bytes(env, argv[7], &out, &ob)) {
        napi_throw_type_error(env, NULL, "treeGather: modelView must be a Float64Array(16)");
        return NULL;
    }
    memcpy(gp.model_view, mv, 128);
    gp.fov_y_deg = get_f64(env, argv[2]);
    gp.render_width = get_f64(env, argv[3]);
    gp.render_height = get_f64(env, argv[4]);
    gp.gather_all = get_u32(env, argv[5]);
    gs_tree_info ti;
    gs_tree* t = (gs_tree*)get_external(env, argv[0]);
    if (gs_tree_get_info(t, &ti) < 0) return throw_gs(env, GS_ERR_INVALID);
    if (out && ob < (size_t)ti.splats * 4) { napi_throw_range_error(env, NULL, "treeGather: out shorter than the tree's splat count"); return NULL; }
    uint32_t render_count = 0;
    int st;

    LOCKED(st = gs_tree_gather(t, &gp, (gs_sorter*)get_external(env, argv[6]), &render_count, (uint32_t*)out));
    if (st < 0) return throw_gs(env, st);
    return num(env, render_count);
}

/* assetLoad(bytes ArrayBuffer|Uint8Array, format 1=ply 2=ksplat, maxShDegree, minAlpha, halfCov)
 *   -> {splatCount, shDegree, compressionLevel, shLevel, shMin, shMax, centers F32, cov F32|U16, rgba U8, sh U16|U8|null} */
static napi_value AssetLoad(napi_env env, napi_callback_info info) {
    ARGS(5)
    void* data;
    size_t nb;
    if (!get_bytes(env, argv[0], &data, &nb) || !data) { napi_throw_type_error(env, NULL, "assetLoad: bytes"); return NULL; }
    gs_asset* a = NULL;
    int st;

    LOCKED(st = gs_asset_open(data, nb, get_u32(env, argv[1]), get_u32(env, argv[2]), &a));
    if (st < 0) return throw_gs(env, st);
    gs_asset_info ai;
    gs_asset_get_info(a, &ai);
    const uint32_t n = ai.splat_count, ncoef = ai.sh_degree == 0 ? 0 : (ai.sh_degree == 1 ? 9 : 24);
    const uint32_t half = get_u32(env, argv[4]);
    napi_value r, ab, ta;
    void *centers, *cov, *rgba, *sh = NULL;
    NAPI_OK(napi_create_object(env, &r));
#define NEWARR(key, type, elems, esize, ptr)                                                       NAPI_OK(napi_create_arraybuffer(env, (size_t)(elems) * (esize), &ptr, &ab));                   NAPI_OK(napi_create_typedarray(env, type, (size_t)(elems), ab, 0, &ta));                        napi_set_named_property(env, r, key, ta);
    NEWARR("centers", napi_float32_array, (size_t)n * 3, 4, centers)
    if (half) { NEWARR("cov", napi_uint16_array, (size_t)n * 6, 2, cov) } else { NEWARR("cov", napi_float32_array, (size_t)n * 6, 4, cov) }
    NEWARR("rgba", napi_uint8_array, (size_t)n * 4, 1, rgba)
    if (ncoef) {
        if (ai.sh_level == 2) { NEWARR("sh", napi_uint8_array, (size_t)n * ncoef, 1, sh) } else { NEWARR("sh", napi_uint16_array, (size_t)n * ncoef, 2, sh) }
    }
    LOCKED(st = gs_asset_fill(a, get_u32(env, argv[3]), (float*)centers, half ? NULL : (float*)cov, half ? (uint16_t*)cov : NULL, (uint8_t*)rgba,
                       (ncoef && ai.sh_level != 2) ? (uint16_t*)sh : NULL, (ncoef && ai.sh_level == 2) ? (uint8_t*)sh : NULL, NULL, NULL));
    LOCKED(gs_asset_close(a));
    if (st < 0) return throw_gs(env, st);
    set(env, r, "splatCount", n);
    set(env, r, "shDegree", ai.sh_degree);
    set(env, r, "compressionLevel", ai.compression_level);
    set(env, r, "shLevel", ai.sh_level);
    set(env, r, "shMin", ai.sh_min);
    set(env, r, "shMax", ai.sh_max);
    return r;
}

/* meshSetDestination(mesh, depth Float32Array(W*H)|null, rgba Uint8Array(4*W*H)|null, width, height, flags): what the following
 * draws are depth-tested against and blended over (gs_mesh_set_destination; SplatMaterial3D.js:72-73, src/Viewer.js:1610-1616,
 * src/DropInViewer.js:34-42).  Both null: back to a cleared target without a depth test. */
static napi_value MeshSetDestination(napi_env env, napi_callback_info info) {
    ARGS(6)
    void *depth, *rgba;
    size_t db, rb;
    if (!get_bytes(env, argv[1], &depth, &db) || !get_bytes(env, argv[2], &rgba, &rb)) {
        napi_throw_type_error(env, NULL, "meshSetDestination: depth must be a Float32Array or null, rgba a Uint8Array or null");
        return NULL;
    }
    gs_destination d;
    memset(&d, 0, sizeof d);
    d.width = get_u32(env, argv[3]);
    d.height = get_u32(env, argv[4]);
    d.flags = get_u32(env, argv[5]);
    const size_t px = (size_t)d.width * d.height;
    if ((depth && db < px * 4) || (rgba && rb < px * 4)) {
        napi_throw_range_error(env, NULL, "meshSetDestination: buffer shorter than width * height");
        return NULL;
    }
    d.depth_host = (const float*)depth;
    d.rgba_host = (const uint8_t*)rgba;
    int st;
    LOCKED(st = gs_mesh_set_destination((gs_mesh*)get_external(env, argv[0]), (depth || rgba) ? &d : NULL));
    if (st < 0) return throw_gs(env, st);
    return NULL;
}
static napi_value Init(napi_env env, napi_value exports) {
    addon_state* state = (addon_state*)calloc(1, sizeof *state);
    if (!state || pthread_mutex_init(&state->lock, NULL) != 0 || napi_set_instance_data(env, state, state_free, NULL) != napi_ok) {
        free(state);
        napi_throw_error(env, NULL, "gsplat addon: could not create the instance state");
        return NULL;
    }
    static const struct { const char* name; napi_callback fn; } fns[] = {
        {"deviceCount", DeviceCount},       {"contextCreate", ContextCreate}, {"contextDestroy", ContextDestroy},
        {"sorterCreate", SorterCreate},     {"sorterDestroy", SorterDestroy}, {"sorterUploadCenters", SorterUploadCenters},
        {"sorterSort", SorterSort},         {"sorterSortAsync", SorterSortAsync}, {"meshCreate", MeshCreate},       {"meshDestroy", MeshDestroy},
        {"meshUpload", MeshUpload},         {"meshRender", MeshRender},       {"meshUploadShU8", MeshUploadShU8},
        {"meshUploadSceneIndexes", MeshUploadSceneIndexes},                   {"meshSetScenes", MeshSetScenes},
        {"sorterBindMesh", SorterBindMesh}, {"sorterSetFrustumCull", SorterSetFrustumCull}, {"sorterSortGathered", SorterSortGathered},
        {"treeCreate", TreeCreate},         {"treeDestroy", TreeDestroy},     {"treeInfo", TreeInfo},
        {"treeGather", TreeGather},         {"treeRead", TreeRead},         {"assetLoad", AssetLoad},
        {"meshProject", MeshProject},       {"sorterSetVisibilityCull", SorterSetVisibilityCull},
        {"groupUniqueId", GroupUniqueId},   {"groupCreate", GroupCreate},     {"groupDestroy", GroupDestroy},
        {"groupRenderGather", GroupRenderGather}, {"groupSetOverlap", GroupSetOverlap}, {"groupWait", GroupWait},
        {"meshSetDeepPass", MeshSetDeepPass}, {"meshSetDestination", MeshSetDestination}, {"meshSetDrawMode", MeshSetDrawMode},
    };
    for (size_t i = 0; i < sizeof fns / sizeof fns[0]; i++) {
        napi_value f;
        if (napi_create_function(env, fns[i].name, NAPI_AUTO_LENGTH, fns[i].fn, NULL, &f) != napi_ok) return NULL;
        napi_set_named_property(env, exports, fns[i].name, f);
    }
    napi_value ver;
    napi_create_int32(env, gs_abi_version(), &ver);
    napi_set_named_property(env, exports, "abiVersion", ver);
    return exports;
}
NAPI_MODULE(NODE_GYP_MODULE_NAME, Init)
