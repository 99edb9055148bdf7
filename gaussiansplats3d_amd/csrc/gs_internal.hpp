// gs_internal.hpp — shared host/device declarations of libgsplat_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <new>
#include <utility>
#include <vector>

#include "../../include/gsplat_hip.h"

// ---------------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------------
void gs_set_error(const char* fmt, ...);

#define GS_HIP(expr)                                                                              \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            gs_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return GS_ERR_HIP;                                                                    \
        }                                                                                         \
    } while (0)

#define GS_REQUIRE(cond, msg)                                    \
    do {                                                         \
        if (!(cond)) {                                           \
            gs_set_error("invalid argument: %s", msg);           \
            return GS_ERR_INVALID;                               \
        }                                                        \
    } while (0)

#define GS_TRY(expr)                 \
    do {                             \
        int _s = (expr);             \
        if (_s < 0) return _s;       \
    } while (0)

// ---------------------------------------------------------------------------------------------------
// device memory
// ---------------------------------------------------------------------------------------------------
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int alloc(size_t n) {
        release();
        if (n == 0) n = 16;
        hipError_t e = hipMalloc(&p, n);
        if (e != hipSuccess) {
            p = nullptr;
            gs_set_error("hipMalloc(%zu) failed: %s", n, hipGetErrorString(e));
            return GS_ERR_NOMEM;
        }
        bytes = n;
        return GS_OK;
    }
    int ensure(size_t n) { return n <= bytes ? GS_OK : alloc(n); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <class T>
    T* as() const { return reinterpret_cast<T*>(p); }
    ~DevBuf() { release(); }
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
};

// ---------------------------------------------------------------------------------------------------
// radix sort geometry (radix.hpp)
// ---------------------------------------------------------------------------------------------------
constexpr uint32_t GS_BIN_SHIFT = 1;                       // a blend workgroup owns a bin of 2x2 tiles of 16 px = 32x32 px
constexpr uint32_t GS_BIN = GS_TILE << GS_BIN_SHIFT;
// The entry lists are per LIST BIN of (16 << GS_LIST_SHIFT) px.  The blend reads only the head of a list before its pixels
// saturate (C3: the first 256 of ~2900 entries per 32-px bin, tools/blend_profile.py), so emitting and sorting per-32-px
// entries mostly produced entries nobody read.  128-px list bins cut the entries 3.7x and their sort to ONE 8-bit pass at
// 1080p (15 x 9 = 135 lists); a blend workgroup scans its parent list and keeps what touches its own 32-px block.
// That only pays when splats are large enough to share lists: a scene of many tiny splats (C4: 16 M splats of ~1.5 tiles)
// gains no entries and makes every blend workgroup scan 16 bins' worth of them, so the size is chosen per mesh from the
// statistics of its last measured draw (gs_mesh::list_shift): 128 px when a visible splat covers >= 3 tiles, else 32 px.
constexpr uint32_t GS_LIST_SHIFT_LARGE = 3;                // 128-px list bins
constexpr uint32_t GS_LIST_SHIFT_SMALL = GS_BIN_SHIFT;     // 32-px list bins = one list per blend workgroup
constexpr float GS_LIST_TILES_PER_SPLAT = 3.0f;
// ... and 256- / 512-px list bins when splats are larger still: what matters is the list bin's area against the splat's (a bin
// scans the entries of its whole list): 128 px at 10 tiles per splat (C3) is 6x, and lists of 36x (C2 at 256 px) or 100x (C3T at
// 512 px) make the blend scan itself to a standstill (r04 tools/list_shift_ab.py: C2 0.281 -> 0.372 -> 0.947 ms at 128 / 256 / 512
// px, C3T 0.598 -> 0.906 -> 2.20).  So the size follows the splats at <= ~12x: 256 px from 21 tiles per visible splat, 512 px
// from 80 (an 8K frame of the garden stand-in covers 101: entries 5.96 M -> 2.19 M, its 2040 lists become 135 and its two-pass
// entry sort one pass (0.082 -> 0.022 ms), the binner 0.091 -> 0.056 ms, the blend scans 24 % more (0.654 -> 0.686 ms): frame
// 0.903 -> 0.837 ms; a rank of eight 0.267 -> 0.228 ms).
constexpr uint32_t GS_LIST_SHIFT_BIG = 4, GS_LIST_SHIFT_HUGE = 5;          // 256- / 512-px list bins
constexpr float GS_LIST_TILES_PER_SPLAT_BIG = 21.0f, GS_LIST_TILES_PER_SPLAT_HUGE = 80.0f;
// CHUNKED COMPOSITE (tile_blend.hip).  The value of a pixel is DEFINED per 16x16 quadrant as a two-level fold: the quadrant's
// ordered survivors (the entries of its list whose exact reach test includes the quadrant) are cut into chunks (below), each
// chunk is composited front to back from T = 1, C = 0, and the chunks are merged near -> far (C = fma(T, C_c, C); T = T * T_c).
// A quadrant with <= 1024 survivors (all quadrants of the BASELINE configurations but a few dozen of C3T) is one chunk and the
// merge is exact, so
// this IS the plain front-to-back composite there; a quadrant that is thousands of splats deep and does not saturate (a surface
// seen at a grazing angle) can be composited by many waves at once - the "over" operator is associative - and the frame does not
// depend on who did it.  The last chunk (index GS_CHUNKS_MAX - 1) is unbounded.
constexpr uint32_t GS_CHUNKS_MAX = 32;
// Chunk sizes: 1024, then 4 x 256, 4 x 512, then 1024 each (boundaries at 1024, 1280, ... 2048, 2560, ... 4096, 5120, ...; the
// last chunk, index 31, is unbounded: 26.6 k survivors are covered by bounded chunks).  A chunk stops by ITS OWN transmittance
// (that is what makes it independent of the chunks in front of it), and the merge stops at chunk boundaries - so a quadrant that
// saturates inside a later chunk is composited up to that chunk's end.  The first chunk is long so that the quadrants of ordinary
// frames never leave it; the chunks right behind it are short because that is where it costs: C3T's deepest quadrants saturate
// after ~1100-1500 survivors and their bins END the blend launch - with a second chunk of 1024 they walked to 2048 and the C3T
// blend went 0.39 -> 0.51 ms; with 512 everywhere C3T walked 17 % more splats; with 2048 first, the first unit of every deep
// quadrant was the long pole of the deep pass (profiles/r03y_*).  Deep quadrants want long chunks (fewer partials, less overshoot
// relative to their depth), hence the growth.
__host__ __device__ constexpr uint32_t gs_chunk_size(uint32_t c) { return c == 0u ? 1024u : c < 5u ? 256u : c < 9u ? 512u : 1024u; }
__host__ __device__ constexpr uint32_t gs_chunk_first(uint32_t c) {
    return c == 0u ? 0u : c < 5u ? 1024u + (c - 1u) * 256u : c < 9u ? 2048u + (c - 5u) * 512u : 4096u + (c - 9u) * 1024u;
}
__host__ __device__ constexpr uint32_t gs_chunk_count(uint32_t s) {
    return s == 0u ? 0u : s <= 1024u ? 1u : s <= 2048u ? 1u + (s - 1024u + 255u) / 256u : s <= 4096u ? 5u + (s - 2048u + 511u) / 512u
         : (9u + (s - 4096u + 1023u) / 1024u < GS_CHUNKS_MAX ? 9u + (s - 4096u + 1023u) / 1024u : GS_CHUNKS_MAX);
}
static_assert(gs_chunk_first(5) == 2048u && gs_chunk_first(9) == 4096u && gs_chunk_count(1025) == 2u && gs_chunk_count(2048) == 5u &&
              gs_chunk_count(2049) == 6u && gs_chunk_count(4096) == 9u && gs_chunk_count(4097) == 10u && gs_chunk_count(1u << 20) == GS_CHUNKS_MAX,
              "chunk table");
constexpr uint32_t GS_CHUNK = 1024;                         // the first (and the typical deep) chunk
// The deep pass: the <= GS_DEEP_MAX_BINS bins that cost most in the previous draw (and more than a threshold) are scanned once
// (k_deep_scan: exact quadrant masks of every list entry + survivor counts per GS_DEEP_RLEN entries) and composited by one wave per
// (bin, quadrant, chunk); k_deep_fold merges.  Lists longer than GS_DEEP_LIST_CAP stay with the one-workgroup-per-bin kernel, which
// closes chunks itself (partials in a pool).
constexpr uint32_t GS_DEEP_MAX_BINS = 512, GS_DEEP_LIST_CAP = 65536, GS_DEEP_RLEN = 1024, GS_DEEP_RANGES = GS_DEEP_LIST_CAP / GS_DEEP_RLEN;
constexpr uint32_t GS_DEEP_SCAN_WGS = 32;                   // k_deep_scan workgroups per deep bin (each strides over the ranges)
constexpr uint32_t GS_DEEP_UNITS = GS_DEEP_MAX_BINS * 4u * GS_CHUNKS_MAX;   // (bin, quadrant, chunk) waves of the deep pass
constexpr uint32_t GS_POOL_SLOTS = 16384;                   // chunk partials (4 KB each) the per-bin kernel may close per draw
constexpr uint32_t GS_ENT_SLOT_MASK = (1u << 28) - 1u;     // deep_ent word = record slot | quadrant mask << 28
constexpr uint32_t GS_DEEP_NONE = 0xFFFFFFFFu;
// flag words of a draw (gs_mesh::deep_flags): [0] deep bins of this draw  [1] bins over the threshold (mirrored to the host: the
// NEXT draw launches the deep pass when this is non-zero)  [2] next pool slot  [3] pool exhausted  [4] units of the deep pass's
// work list  [5] the next unit a wave of the pass takes | deep_list [DEEP_MAX] |
// deep_of [bins]
constexpr uint32_t GS_FLAG_COUNT = 0, GS_FLAG_CAND = 1, GS_FLAG_POOL_NEXT = 2, GS_FLAG_POOL_OVER = 3, GS_FLAG_UNITS = 4, GS_FLAG_UNIT_NEXT = 5, GS_FLAG_LIST = 8,
                   GS_FLAG_OF = GS_FLAG_LIST + GS_DEEP_MAX_BINS;

#ifndef RADIX_TILE_CFG
#define RADIX_TILE_CFG 4096
#endif
constexpr int RADIX_TILE = RADIX_TILE_CFG;                // keys per workgroup iteration
// Rows of a radix pass's offset table = the most workgroups (chunks) a pass may have.  The chunk-staged passes of the depth
// sort (radix.hpp) want <= 3 tiles per workgroup: 2048 rows cover 25 M keys.
#ifndef RADIX_MAX_BLOCKS_CFG
#define RADIX_MAX_BLOCKS_CFG 2048
#endif
constexpr int RADIX_MAX_BLOCKS = RADIX_MAX_BLOCKS_CFG;
// Grid cap of the tile-at-a-time scatter kernel.  Every workgroup starts by summing rows of the two-level offset table, so fewer,
// longer workgroups win: same-box A/B of the isolated C3 depth sort (r04a/b) 256: 0.0806 ms, 384: 0.0798, 512: 0.0733,
// 1024: 0.0822, 2048: 0.093 (r01e, whole frames: 2048: 0.498, 1024: 0.477, 512: 0.471, 384: 0.489, 256: 0.483).
#ifndef RADIX_TILE_GRID_CFG
#define RADIX_TILE_GRID_CFG 512
#endif
constexpr int RADIX_TILE_GRID = RADIX_TILE_GRID_CFG;
static_assert(RADIX_TILE_GRID <= RADIX_MAX_BLOCKS, "the tile kernel's grid needs a table row per workgroup");
constexpr int RADIX_BINS = 256;
constexpr int RADIX_MAX_PASSES = 4;
constexpr int RADIX_GROUP = 32;                           // workgroups per group row of the two-level offset table (radix.hpp)
constexpr int RADIX_MAX_GROUPS = RADIX_MAX_BLOCKS / RADIX_GROUP;
constexpr int RADIX_TOTAL_WORDS = RADIX_MAX_PASSES * RADIX_MAX_GROUPS * RADIX_BINS;

struct RadixScratch {
    DevBuf block_hist;    // uint32 [RADIX_MAX_BLOCKS][RADIX_BINS]  (workgroup-major: 1 KiB coalesced rows)
    DevBuf digit_total;   // uint32 [RADIX_MAX_PASSES][RADIX_MAX_GROUPS][RADIX_BINS]: digit counts per group of 32 workgroups
    int init() {
        GS_TRY(block_hist.alloc(sizeof(uint32_t) * RADIX_BINS * RADIX_MAX_BLOCKS));
        GS_TRY(digit_total.alloc(sizeof(uint32_t) * RADIX_TOTAL_WORDS));
        return GS_OK;
    }
};

// ---------------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------------
// Streams of one context (the reference runs its sort in a Web Worker, concurrently with rendering,
// /root/reference/src/worker/SortWorker.js; here the "worker thread" is a HIP stream):
//   stream  the caller-visible stream: binning, tile sort, blend, every copy back to the host
//   aux     vertex stage (k_project) of a draw, forked from / joined into `stream` with events
// Each gs_sorter additionally owns a private stream; its result is joined into `stream` where a draw consumes it.
struct gs_mesh;
struct gs_context {
    int device = 0;
    bool wide_entry_keys = false;      // GSPLAT_WIDE_ENTRY_KEYS=1: 32-bit list keys even below 65536 lists (test hook)
    hipStream_t stream = nullptr;
    hipStream_t aux = nullptr;
    bool own_stream = false;
    bool stage_events = false;            // GS_CTX_STAGE_TIMING / GSPLAT_STAGE_EVENTS=1: bracket the stages of EVERY sort and
                                          // draw with timing events.  Otherwise only calls that are handed a stats pointer
                                          // (and so synchronise anyway) are bracketed: an event record is a barrier packet
                                          // on the stream, and the nine of a frame cost 35 us of a 0.33 ms frame (r02m)
    uint32_t kernel_sample = 8;           // GSPLAT_KERNEL_SAMPLE: on a single-stream context k_project is bracketed with events
                                          // every n-th draw (0 = never) for gs_mesh_kernel_time; every draw costs 1.5 % (r02m)
    bool serial = false;                  // GSPLAT_SERIAL=1: everything on `stream` (debugging / per-stage timing)
    bool fork_join = false;               // GS_CTX_FORK_JOIN: sorts wait for what `stream` holds when they are called (serial frames)
    hipEvent_t ev_fork = nullptr;         // ... recorded on `stream` by every sort of such a context
    int cu_count = 256;
    bool lds_atomic_lane_order = false;   // self-test result: ds_add_rtn serves same-address lanes in lane order
    RadixScratch radix;                   // scratch of the create-time self-test
    std::vector<gs_mesh*> live_meshes;    // so that a sorter bound to a mesh never dereferences a destroyed one
};

struct RadixExec {                        // where and with what scratch a radix pass runs
    hipStream_t stream;
    RadixScratch* scratch;
    bool atomic_rank;
};

struct ScopedDevice {
    int prev = -1;
    explicit ScopedDevice(int dev) {
        (void)hipGetDevice(&prev);
        if (prev != dev) (void)hipSetDevice(dev);
        else prev = -1;
    }
    ~ScopedDevice() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

// ---------------------------------------------------------------------------------------------------
// sorter object (sorter.hip)
// ---------------------------------------------------------------------------------------------------
constexpr uint32_t SORT_SHARDS = 32;   // min / max words per sort: workgroup b reduces into shard b % 32
struct SortFrame {            // device-resident per-sort scalars
    int32_t key_min[SORT_SHARDS];   // atomicMin targets, initialised to +2147483640 (sorter.cpp:25)
    int32_t key_max[SORT_SHARDS];   // atomicMax targets, initialised to -2147483640 (sorter.cpp:24)
    __host__ __device__ int32_t lo() const {
        int32_t v = key_min[0];
        for (uint32_t k = 1; k < SORT_SHARDS; k++) v = key_min[k] < v ? key_min[k] : v;
        return v;
    }
    __host__ __device__ int32_t hi() const {
        int32_t v = key_max[0];
        for (uint32_t k = 1; k < SORT_SHARDS; k++) v = key_max[k] > v ? key_max[k] : v;
        return v;
    }
    uint32_t clamped;
    uint32_t kept;            // frustum-cull variant: list positions that survive = length of the sorted result
};

// A planned octree gather as the sorter's fused copy + key kernel sees it (tree.hip -> sorter.hip): per LEAF the first slot of
// its index list in indexesToSort (0xFFFFFFFF = culled), its length and where its indexes start in the leaf-major list.
struct gs_tree;
struct TreeGatherView {
    uint64_t tree_uid;            // identifies the tree (and so the leaf-major order) for the sorter's per-tree caches
    uint32_t leaves, tree_splats;
    const uint32_t *leaf_offset, *leaf_count, *leaf_begin, *leaf_indexes, *totals;   // totals: {splats gathered, leaves kept}
};
void gs_tree_view(gs_tree* t, TreeGatherView* v);
int gs_tree_copy_plain(gs_tree* t, uint32_t* out_dev, hipStream_t st);   // the kept leaves' index lists -> indexesToSort
void gs_tree_forget_sorter(gs_tree* t, gs_sorter* s);                    // the sorter consumed (or dropped) the pending gather

struct gs_sorter {
    gs_context* ctx = nullptr;
    uint32_t max_count = 0, flags = 0, precision = 16, uploaded = 0;
    // SoA planes of the AoS x4 centres the worker receives (int32 or float bit patterns)
    DevBuf cx, cy, cz, cw, scene_idx;
    DevBuf caos;               // the AoS x4 centres as uploaded (16-byte gathers for index-list sorts)
    DevBuf staging;            // upload staging (AoS) / host index list / precomputed distances
    DevBuf idx_in;             // indexesToSort on device
    DevBuf precomputed;
    DevBuf keys;               // int32 depth key per list position (mappedDistances, phase A)
    DevBuf keyA, keyB, valA, valB;   // radix ping-pong
    DevBuf sorted;             // uint32 [render_count]: the sortDone payload, stays resident for the mesh
    DevBuf frame;              // SortFrame [2]: the sort in flight uses one, its first kernel resets the other
    uint32_t frame_index = 0;
    DevBuf scene_rows;         // per-scene key coefficients (dynamic mode)
    DevBuf debug;
    RadixScratch radix;
    hipStream_t stream = nullptr;      // the "worker thread": sorts run here, concurrently with draws on ctx->stream
    bool own_stream = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed_sort = false;              // the last sort recorded ev0 (see gs_context::stage_events)
    hipEvent_t ev_consumed = nullptr;  // recorded on ctx->stream by a draw once it has read `sorted`
    bool consumer_pending = false;
    gs_mesh* bound_mesh = nullptr;     // gs_sorter_bind_mesh: results are positions in this mesh's storage order
    gs_mesh* result_mesh = nullptr;    // ... as it was when the last sort ran (nullptr = plain splat indexes)
    const uint32_t* result_unmap = nullptr;
    uint32_t result_payload_max = 0;   // largest payload the last sort could emit (clamp of the host-visible un-mapping)
    uint32_t gathered = 0;             // splatRenderCount of the list gs_tree_gather left in idx_in (an upper bound when
                                       // gathered_on_device: the asynchronous gather leaves the real count in gathered_dev)
    bool has_gathered = false;
    bool gathered_on_device = false;
    DevBuf gathered_dev;               // uint32: splatRenderCount of that list, written by the gather's plan
    gs_tree* pending_tree = nullptr;   // gs_tree_gather planned a gather whose lists this sorter has still to copy (deferred so
                                       // that a full sort of a static scene can fuse the copy with its key kernel)
    bool pending_keep_zeroed = false;  // ... and zeroed this sorter's keep mask for a fused per-splat cull
    uint32_t centers_version = 0;      // bumped by gs_sorter_upload_centers (invalidates the leaf-major caches below)
    DevBuf pay_in;                     // uint32 [max]: the gathered list's payloads (positions in the bound mesh, or the indexes)
    DevBuf leaf_centers, leaf_pos;     // the centres / payloads of the tree's splats in ITS leaf-major order: the fused copy
                                       // streams them instead of gathering 16 bytes per list entry at random
    uint64_t leaf_cache_tree = 0;      // what the two caches were built for
    uint32_t leaf_cache_centers = 0, leaf_cache_layout = 0, leaf_cache_uploaded = 0;
    const void* leaf_cache_mesh = nullptr;
    uint32_t last_render = 0, last_sort = 0, last_passes = 0;
    bool last_identity = true;
    bool has_result = false;
    bool frustum_cull = false;         // gs_sorter_set_frustum_cull
    bool visibility_cull = false;      // gs_sorter_set_visibility_cull
    bool last_culled = false;          // the resident result holds only the kept splats; its length lives in result_frame
    const SortFrame* result_frame = nullptr;
    DevBuf keep_mask;                  // 1 bit per list position (frustum-cull variant)
    DevBuf chunk_counts;               // survivors per chunk of the identity list (visibility-cull variant)
    DevBuf key_sync;                   // uint32 [2]: {arrivals, time-outs} of k_depth_key_hist's barrier across the grid ($GSPLAT_KEY_HIST_FUSED)
    uint32_t key_sync_base = 0;        // arrivals before the next launch
    bool key_sync_used = false;
    DevBuf mask_copy;                  // the bound mesh's visibility mask as the last visibility-culled sort consumed it
    bool last_vis_culled = false;
};

// ---------------------------------------------------------------------------------------------------
// mesh object (mesh.hip + project.hip + tile_bin.hip + tile_blend.hip)
// ---------------------------------------------------------------------------------------------------
// Vertex-stage record consumed by the blend (32 B, one aligned sector per gather):
//   cx, cy         centre in pixels (GL window coordinates, row 0 = bottom)
//   ax, ay, bx, by K * e1 / |b1|, K * e2 / |b2| with K^2 = 4*log2(e): power = (a.d)^2 + (b.d)^2 =
//                  0.5*log2(e)*A, alpha = exp2(-power) * a0; discard when power > 4*log2(e) (A > 8)
//   c0             r | g << 16 (unorm16)
//   c1             b | a << 16 (unorm16; a = rgba8.a/255, or the antialias-compensated alpha)
struct __attribute__((aligned(32))) SplatRec {
    float cx, cy, ax, ay;
    float bx, by;
    uint32_t c0, c1;
};
static_assert(sizeof(SplatRec) == 32, "SplatRec must be one 32-byte sector");

struct RenderFrame {          // device-resident per-draw scalars
    uint32_t visible;         // splats passing the vertex-stage rejects
    uint32_t entries_lo;      // total tile entries D (uint64 split for atomics-free writes)
    uint32_t entries_hi;
    uint32_t overflow;        // D exceeded the entry capacity
    uint32_t entry_count;     // min(D, capacity): what the tile sort and blend consume
    uint32_t tiles16_lo;      // sum over visible splats of 16x16 tiles touched (the D of SURVEY.md section 8d)
    uint32_t tiles16_hi;
    uint32_t pad;
};

struct ProjectParams {
    float view[16];
    float proj[16];
    float cam_pos[3];
    float focal_x, focal_y;
    float width, height;
    float splat_scale, kernel2d, max_splat_px, inv_focal_adj;
    uint32_t sh_degree;       // degree evaluated
    uint32_t sh_stored;       // degree stored
    uint32_t cov_half;
    uint32_t flags;
    uint32_t tiles_x, tiles_y;     // 16-px tile grid (the unit of vertex-stage rects and of the multi-GPU strips)
    uint32_t row_begin, row_end;   // 16-px tile rows rendered by this rank
    // shader permutations (k_project<true>)
    float view_matrix[16];
    float ortho_zoom, fade_start;
    float scene_center[3];
    uint32_t scene_count, sh_u8;
    uint32_t bins_x;               // 32-px bin grid: one 256-thread workgroup blends a bin
    uint32_t bin_row_begin, bin_row_end;
    uint32_t list_shift;           // a list bin is (16 << list_shift) px
    uint32_t lists_x;              // list-bin grid: the unit of the entry lists and of the entry sort's keys
    uint32_t list_row_begin, list_row_end;
    uint32_t y0, y1;               // pixel rows [y0, y1) of this rank's strip
    uint32_t count;
    uint32_t block_cull;           // 1: whole 256-splat storage blocks are tested first (project.hip); 0 for per-scene transforms
    float mv_row_norm[3];          // |row r of mat3(view)| * (1 + 1e-6): bounds |T0|, |T1| of the strip pre-test (project.hip)
    uint32_t depth_mode;           // destination depth test (gs_mesh_set_destination): 0 = off, 1 = fp32 compare, 2 = as a 24-bit buffer
};

struct gs_mesh {
    gs_context* ctx = nullptr;
    uint32_t list_shift = GS_LIST_SHIFT_LARGE;     // list-bin size of the next draw (mesh_collect_stats re-evaluates it)
    uint32_t drawn_list_shift = GS_LIST_SHIFT_LARGE;   // ... of the last draw (what tile_ranges / the statistics refer to)
    ProjectParams last_pp = {};                    // the last draw's geometry (gs_mesh_debug_rop8 walks its lists)
    ProjectParams stats_pp = {};                   // ... of the draw that wrote blend_stats (set once that draw is enqueued)
    bool stats_pp_valid = false;
    double centre_sum[3] = {0.0, 0.0, 0.0}, centre_sq = 0.0;   // over every finite centre ever uploaded: where the scene is and how large
    uint64_t centre_n = 0;                          // (tile_bin.hip: how far a camera moved, in screen heights)
    int forced_list_shift = -1;                    // GSPLAT_LIST_SHIFT (A/B and tests)
    uint32_t max_count = 0, sh_degree = 0, flags = 0, uploaded = 0;
    // SoA planes
    DevBuf px, py, pz;         // float centres
    DevBuf covA, covB;         // fp32: float4 + float2 ; fp16: uint2 + uint
    DevBuf cov_bound;          // float: an upper bound of the covariance's spectral radius (largest absolute row sum), written
                               // at upload; lets a rank of a multi-GPU draw drop splats that cannot reach its strip before it
                               // fetches their covariance
    DevBuf block_box;          // float [ceil(n/256)][8]: per storage block {min xyz, max xyz of the centres, max cov_bound, -}:
                               // lets k_project drop a whole block (frustum, strip) without reading its centres
    DevBuf rgba;               // uint32
    DevBuf sh0, sh1, sh2;      // fp16: uint4 planes (SH2: 3 planes; SH1: sh0 = uint4, sh1 = uint)
                               // u8  : sh0 = uint4 (bytes 0..15), sh1 = uint2 (bytes 16..23, SH2 only)
    DevBuf perm;               // uint32 [n]: original splat index -> internal (Morton-ordered) position
    DevBuf inv_perm;           // uint32 [n]: internal position -> original splat index
    std::vector<std::pair<uint32_t, uint32_t>> slotted;   // [begin, end) ranges of splats that own storage slots (disjoint, sorted)
    bool reorder = true;
    uint32_t layout_version = 0;   // bumped whenever splats receive storage slots (perm changes): a sorter's per-tree payload cache
    bool no_block_cull = false;    // GSPLAT_NO_BLOCK_CULL=1 (A/B and tests)
    bool no_block_list = false;    // GSPLAT_NO_BLOCK_LIST=1: every k_project workgroup tests its own block (the round-4 shape; A/B and tests)
    bool block_test_always = false;   // GSPLAT_BLOCK_TEST_ALWAYS=1: the separate block test whatever the scene looks like (A/B and tests)
    uint32_t measured_visible = 0, measured_count = 0;   // of the last full-frame draw whose statistics were read (like list_shift:
                                   // chosen from the last measured draw): more than 60 % visible -> no separate block test
    bool translate = true;     // this draw's index list is in the caller's numbering (needs perm)
    DevBuf scene_idx;          // uint32 per splat (allocated by gs_mesh_upload_scene_indexes)
    DevBuf scene_dev;          // gs_scene_params on the device
    bool has_scenes = false;
    uint32_t scene_count = 1;
    DevBuf staging;
    // per-draw
    DevBuf recs;               // SplatRec [n]  survivors compacted inside each 256-splat block (project.hip)
    DevBuf zrec;               // float [n]     the survivor's window-space centre depth, same slots (only while a destination
                               //               depth is set: the blend's depth test, gs_mesh_set_destination)
    DevBuf rects;              // uint2 [n]     tile rect per survivor, same slots
    DevBuf vis_mask;           // uint64 [4*ceil(n/256)]  1 = splat survived the vertex stage and touches a pixel
    DevBuf block_any;          // uint8 [ceil(n/256)]      1 = some splat of the 256-splat block survived the vertex stage
    DevBuf vis32;              // uint2 [8*ceil(n/256)]   {the same mask per 32 splats, slot of its first visible splat}
    DevBuf prect;              // uint2 [n]: per splat POSITION of a live block {x0:12|y0:12|slot in block:8, x1:12|y1:12|visible:1}:
                               //            the binner's one gather per sorted index (project.hip)
    DevBuf vis_orig;           // uint32 [ceil(n/32)]     the mask by ORIGINAL splat index (gs_mesh_project only: feeds the
                               //                         visibility-culled sort)
    DevBuf cidx;               // uint32 [render_count] record slots of the visible splats in traversal order (compacted per workgroup)
    DevBuf order;              // uint32 [render_count] when the caller supplies host indexes
    DevBuf rect_q;             // uint2 [render_count] their rects, same layout as cidx
    DevBuf coff;               // uint32 [render_count] first entry slot of each, relative to its binning workgroup
    DevBuf bin_sums;           // uint32 [3][BIN_MAX_BLOCKS]: entries | visible splats | 16-px tiles per workgroup
    DevBuf bin_scan;           // k_bin_fused's scan across the grid: uint64 granules {draw serial, value} [3][BIN_MAX_BLOCKS] per slice,
                               // [3][BIN_MAX_BLOCKS / 32] per group of slices, then one uint32 `fail` word (tile_bin.hip)
    bool bin_scan_ready = false;
    DevBuf ekeyA, ekeyB, evalA, evalB;   // tile entries ping-pong (key = tile id, val = splat index)
    DevBuf tile_ranges;        // uint2 [bins]
    DevBuf frame;              // RenderFrame
    DevBuf fb;                 // internal RGBA8 framebuffer
    DevBuf blend_stats;        // uint2 [blend workgroups of the last draw]: {entries staged, half quadrants evaluated}, then
                               // uint32 [the same]: (splat, quadrant) pairs walked
    uint32_t blend_bins = 0, blend_row_begin = 0, blend_width = 0;    // the bins the last draw blended
    DevBuf deep_flags;         // uint32 words, layout above (GS_FLAG_*)
    DevBuf chunk_pool;         // float4 [GS_POOL_SLOTS][256]: chunk partials closed by the per-bin kernel
    DevBuf deep_ent;           // uint32 [GS_DEEP_MAX_BINS][GS_DEEP_LIST_CAP]: slot | quadrant mask << 28 per list entry (deep pass)
    DevBuf deep_cnt;           // uint32 [GS_DEEP_MAX_BINS][GS_DEEP_RANGES][4]: survivors per range and quadrant
    DevBuf deep_partial;       // float4 [GS_DEEP_UNITS][256]: {C, T} of every (deep bin, quadrant, chunk)
    DevBuf deep_work;          // uint32 [GS_DEEP_UNITS]: the (bin, quadrant, chunk) units that exist, packed (k_deep_plan)
    uint32_t draw_mode = 0;    // GS_DRAW_FP32 | GS_DRAW_ROP8 (gs_mesh_set_draw_mode)
    int32_t blend_stats_mode = 0;    // the draw mode (GS_DRAW_*) that wrote blend_stats: they schedule a later draw of the SAME mode only
    bool no_deep = false;      // GSPLAT_NO_DEEP: never launch the deep pass (the per-bin kernel draws everything)
    bool deep_pass = false;    // this draw runs the deep pass (decided in gs_launch_binning)
    DevBuf blend_order;        // uint32 [blend bins]: this draw's bins by descending cost in the previous draw (k_bin_emit)
    bool blend_order_valid = false;
    RadixScratch radix;
    uint32_t entry_capacity = 0;
    uint32_t sorted_buf = 0;   // ping-pong buffer index holding the tile-sorted entries of the last draw
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_p0 = nullptr, ev_p1 = nullptr;   // around k_project on ctx->aux (= ring slot of the last draw)
    // ring of event pairs around k_project: per-launch durations over a whole timed region without a sync per frame
    static constexpr int TIMING_RING = 128;
    hipEvent_t ring0[TIMING_RING] = {}, ring1[TIMING_RING] = {};
    bool ring_used[TIMING_RING] = {};
    bool ring_whole[TIMING_RING] = {};             // the slot brackets the whole vertex stage (a timed draw), not k_project alone
    uint32_t ring_next = 0;
    double proj_sum_ms = 0.0, stage_sum_ms = 0.0;  // k_project alone (sampled draws) | block test + mask memset + k_project (timed draws)
    uint32_t proj_launches = 0, stage_launches = 0;
    hipEvent_t ev_done = nullptr;                  // end of the last draw that read THIS set of vertex-stage outputs (on ctx->stream)
    // The vertex stage's outputs exist twice on a context with streams of its own: the vertex stage of frame k + 1 writes the set
    // frame k - 1 drew from while frame k is still binned and blended from the other one (mesh_project swaps the two; every field
    // that belongs to a set is swapped with it).  `alt` is the set that is NOT current.
    struct ProjSet {
        DevBuf recs, zrec, rects, vis_mask, block_any, vis32, prect, vis_orig;
        hipEvent_t ev_done = nullptr;
        bool drawn = false, vis_orig_dirty = true, vis_orig_lazy = false;
        uint32_t vis_orig_count = 0;
    } alt;
    bool two_sets = false;                         // alt is allocated and in use
    bool set_drawn = false;                        // a draw has read the current set (ev_done is recorded)
    gs_render_stats last = {};
    bool has_draw = false;
    uint32_t last_count = 0;
    uint32_t* mirror_host = nullptr;   // mapped pinned words written by every draw's k_bin_emit: [0..3] {serial, overflow, entries lo, hi},
                                       // [4] bins over the deep pass's threshold, [5] their share of the last walk in 1 / 1024, [8..11] {serial, visible splats, 16-px tiles lo, hi}
    uint32_t* mirror_dev = nullptr;
    uint32_t draw_serial = 0, healed_serial = 0, adapted_serial = 0;
    uint32_t grow_entries_to = 0;         // entry capacity wanted before the next draw (the list bins became smaller: mesh.hip)
    uint32_t full_serial[8] = {0, 0, 0, 0, 0, 0, 0, 0}, full_count[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // the last full-frame draws: serial (slot = serial & 7) and splats projected
    uint32_t project_serial = 0;
    uint32_t last_project_mode = 1;       // gs_launch_project: 1 = k_block_test + k_project, 0 = the test in every workgroup, 2 = no test
    bool derive_orig_mask = false;        // a bound sorter builds vis_orig itself from vis_mask + its position map (sorter.hip,
                                          // k_mask_derive_count): full-frame gs_mesh_project calls skip the per-survivor atomics
    bool vis_orig_lazy = false;           // ... and the pending projection did: vis_orig holds nothing yet
    bool vis_orig_dirty = true;           // vis_orig may hold bits (cleared by the sort that consumes it, see k_mask_compact)
    uint32_t vis_orig_count = 0;          // splats the last gs_mesh_project looked at
    bool timed_project = false;           // the last vertex stage was bracketed with ev_p0 / ev_p1
    bool timed_draw = false;              // the last draw recorded its stage events (see gs_context::stage_events)
    uint32_t truncated_draws = 0;      // asynchronous draws that overflowed the entry buffer (noticed after the fact)
    bool projection_pending = false;   // gs_mesh_project ran; the next gs_mesh_render with the same camera consumes it
    gs_camera projected_cam = {};
    uint32_t projected_depth_mode = 0; // ... and it wrote zrec for this depth mode (a destination set in between re-projects)
    // destination of the draws (gs_mesh_set_destination): what the splats are depth-tested against and blended over
    DevBuf dest_depth_own, dest_rgba_own;          // copies of host-side inputs
    const float* dest_depth = nullptr;             // float [dest_h][dest_w] window depth, or nullptr
    const uint32_t* dest_rgba = nullptr;           // RGBA8 [dest_h][dest_w], or nullptr
    uint32_t dest_w = 0, dest_h = 0, dest_flags = 0;
    const float* drawn_dest_depth = nullptr;       // ... as the last draw saw it (gs_mesh_debug_rop8 refuses a destination that
    const uint32_t* drawn_dest_rgba = nullptr;     // changed since: it combines the mesh's destination with the last draw's geometry)
    uint32_t drawn_dest_w = 0, drawn_dest_h = 0, drawn_dest_flags = 0;
};

int gs_selftest_lds_atomic_order(gs_context* ctx, bool* ok);

// payload maps of a mesh for a bound sorter (nullptr when the mesh keeps upload order or covers fewer splats)
const uint32_t* gs_mesh_payload_map(gs_mesh* m, uint32_t splats);
const uint32_t* gs_mesh_payload_unmap(gs_mesh* m);

// kernels' host launchers ---------------------------------------------------------------------------
// (ev_before / ev_after, nullable, recorded on ctx->aux: around the whole vertex stage - block test, mask memset, k_project - when
// whole_stage (a timed draw's project_ms), else right around k_project itself, so that gs_mesh_kernel_time, the bench's live
// roofline clock over the untimed frames of its timed region, times that one kernel)
int gs_launch_project(gs_mesh* m, const ProjectParams& pp, bool orig_mask, hipEvent_t ev_before = nullptr, hipEvent_t ev_after = nullptr,
                      bool whole_stage = false);
int gs_launch_binning(gs_mesh* m, const ProjectParams& pp, const uint32_t* order_dev, gs_sorter* sorter, uint32_t render_count);
int gs_launch_blend(gs_mesh* m, const ProjectParams& pp, uint8_t* out_dev);
int gs_launch_rop8_window(gs_mesh* m, const ProjectParams& pp, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, uint32_t* out_dev);
