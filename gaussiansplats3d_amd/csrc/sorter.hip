// sorter.hip — the SORT SEAM on gfx950: replaces the reference's Web-Worker + WASM counting sort
// (/root/reference/src/worker/SortWorker.js:31-81 -> sorter.cpp:17-168) with
//   k_depth_key      phase A  per-splat int32 view-depth key + device-wide min/max   (sorter.cpp:29-140)
//   radix passes     phase B-D bucket mapping fused into pass 0 of a stable LSD radix sort over
//                    key' = range-1-bucket of the REVERSED list, which reproduces the reference's output
//                    order exactly: buckets far->near, equal buckets in reverse input order
//                    (sorter.cpp:142-167).
// Arithmetic that decides the order is spelled with non-contracting intrinsics (__fmul_rn ...) so hipcc's
// default fp-contract=fast cannot fuse it; this file is also built with -ffp-contract=off.
#include <math.h>
#include <stdlib.h>

#include "radix.hpp"

// WASM (emscripten) float->int: NaN / out of range -> INT32_MIN (same rule as oracle/sort_oracle.c)
__host__ __device__ static inline int32_t trunc_f64_i32(double v) {
    if (!(v > -2147483649.0 && v < 2147483648.0)) return INT32_MIN;
    return (int32_t)v;
}

constexpr uint32_t MODE_INT = 1, MODE_DYNAMIC = 2, MODE_PRECOMPUTED = 4;
#ifndef GS_KEY_GRID_MULT
#define GS_KEY_GRID_MULT 2             // workgroups of k_depth_key<VEC4> per CU
#endif

struct SceneRows {                 // per-scene clip-z row (mvp * transform)[2], as int x1000 and as float
    int32_t im[GS_MAX_SCENES][4];
    float fm[GS_MAX_SCENES][4];
};

struct KeyParams {
    const uint32_t *cx, *cy, *cz, *cw;       // SoA planes (int32 or float bit patterns): streaming (identity list) path
    const uint4* aos;                        // the same centres as uploaded (x,y,z,w): ONE 16-byte gather per index
    const uint32_t* scene_idx;
    const uint32_t* idx_in;                  // nullable: identity
    const uint32_t* precomputed;             // int32 or float bit patterns
    const SceneRows* rows;
    int32_t* keys_out;
    SortFrame* frame;                        // min / max / clamp counter of THIS sort (pre-initialised, see below)
    SortFrame* next_frame;                   // the other buffer: reset here for the next sort (no separate init kernel)
    uint32_t* digit_total;                   // radix digit totals of this sort's passes: zeroed here, before any histogram
    uint32_t sort_start, render_count, mode;
    uint32_t last_splat;                     // uploaded - 1: list entries are clamped to it (the reference reads whatever
                                             // WASM memory a stale index points at; here that would be a GPU page fault)
    int32_t im0, im1, im2;                   // static path: (int)(mvp[k]*1000.0), k = 2,6,10
    float fm0, fm1, fm2;                     // static float path: mvp[2], mvp[6], mvp[10]
    float mvp[16];                           // frustum-cull variant only: the whole modelViewProj, fp32 column-major
    unsigned long long* keep;                // cull variants only: 1 bit per list position
    const uint32_t* count_dev;               // visibility-cull variant: the list length lives on the device (SortFrame::kept)
    uint32_t ext_minmax;                     // ... and min / max (over EVERY splat) were taken by k_minmax_count: leave them alone
};

// The per-scene rows of a dynamic sort travel as a kernel ARGUMENT (1 KB of kernarg, captured when the launch is enqueued),
// so the host copy may die with the caller's stack frame and the sort stays asynchronous.
__global__ __launch_bounds__(256) void k_store_scene_rows(SceneRows rows, SceneRows* __restrict__ dst) {
    const uint32_t t = threadIdx.x;
    if (t < GS_MAX_SCENES * 4) {
        dst->im[t >> 2][t & 3] = rows.im[t >> 2][t & 3];
        dst->fm[t >> 2][t & 3] = rows.fm[t >> 2][t & 3];
    }
}

__global__ __launch_bounds__(256) void k_aos4_to_soa(const uint4* __restrict__ aos, uint32_t count, uint32_t from,
                                                     uint32_t* __restrict__ x, uint32_t* __restrict__ y,
                                                     uint32_t* __restrict__ z, uint32_t* __restrict__ w) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
        const uint4 v = aos[i];
        x[from + i] = v.x;
        y[from + i] = v.y;
        z[from + i] = v.z;
        if (w) w[from + i] = v.w;
    }
}

__device__ __forceinline__ int32_t depth_key_one(const KeyParams& p, uint32_t g) {
    if (p.mode & MODE_PRECOMPUTED) {
        const uint32_t raw = p.precomputed[g];
        if (p.mode & MODE_INT) return (int32_t)raw;                                        // sorter.cpp:31-38
        return trunc_f64_i32((double)__uint_as_float(raw) * 4096.0);                       // :79-86
    }
    const uint4 c = p.aos[g];                                                              // gathered by list index
    if (p.mode & MODE_INT) {
        const uint32_t x = c.x, y = c.y, z = c.z;                                          // wrap-around int32
        if (p.mode & MODE_DYNAMIC) {                                                       // :41-62
            const int32_t* r = p.rows->im[p.scene_idx[g]];
            return (int32_t)(x * (uint32_t)r[0] + y * (uint32_t)r[1] + z * (uint32_t)r[2] + c.w * (uint32_t)r[3]);
        }
        return (int32_t)(x * (uint32_t)p.im0 + y * (uint32_t)p.im1 + z * (uint32_t)p.im2); // :63-75, w lane unused
    }
    const float x = __uint_as_float(c.x), y = __uint_as_float(c.y), z = __uint_as_float(c.z);
    float s;
    if (p.mode & MODE_DYNAMIC) {                                                           // :110-126
        const float* r = p.rows->fm[p.scene_idx[g]];
        s = __fmul_rn(r[0], x);
        s = __fadd_rn(s, __fmul_rn(r[1], y));
        s = __fadd_rn(s, __fmul_rn(r[2], z));
        s = __fadd_rn(s, __fmul_rn(r[3], __uint_as_float(c.w)));
    } else {                                                                               // :128-138
        s = __fmul_rn(p.fm0, x);
        s = __fadd_rn(s, __fmul_rn(p.fm1, y));
        s = __fadd_rn(s, __fmul_rn(p.fm2, z));
    }
    return trunc_f64_i32((double)s * 4096.0);
}

// Phase A: keys for list positions [sort_start, render_count) + device-wide min / max.
// VEC4: identity index list + static integer mode (the Viewer's default, cull off): each lane reads 4 consecutive
// splats per plane as one 16-byte load and writes 4 keys as one 16-byte store.
template <bool VEC4>
__global__ __launch_bounds__(256) void k_depth_key(KeyParams p) {
    __shared__ int32_t s_lo[4], s_hi[4];
    int32_t lo = 2147483640, hi = -2147483640;
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (p.ext_minmax) {
        // visibility-culled sort: k_minmax_count did the housekeeping and took min / max over EVERY splat; this launch only
        // keys the compacted list, whose length lives on the device
        const uint32_t R = *p.count_dev;
        for (uint32_t i = t; i < R; i += stride) p.keys_out[i] = depth_key_one(p, min(p.idx_in[i], p.last_splat));
        return;
    }
    // housekeeping folded into the first kernel of a sort (two launches and their boundaries saved per sort)
    for (uint32_t w = t; w < (uint32_t)RADIX_TOTAL_WORDS; w += stride) p.digit_total[w] = 0u;
    if (t < SORT_SHARDS) {
        p.next_frame->key_min[t] = 2147483640;   // sorter.cpp:25
        p.next_frame->key_max[t] = -2147483640;  // sorter.cpp:24
    }
    if (t == 0) {
        p.next_frame->clamped = 0;
        p.next_frame->kept = 0;
    }
    if (VEC4) {
        // positions [a4*4, b4*4) are handled as vectors, the ragged head/tail as scalars by the first lanes
        const uint32_t a4 = (p.sort_start + 3u) / 4u, b4 = p.render_count / 4u;
        const uint4* x4 = reinterpret_cast<const uint4*>(p.cx);
        const uint4* y4 = reinterpret_cast<const uint4*>(p.cy);
        const uint4* z4 = reinterpret_cast<const uint4*>(p.cz);
        int4* o4 = reinterpret_cast<int4*>(p.keys_out);
        const uint32_t m0 = (uint32_t)p.im0, m1 = (uint32_t)p.im1, m2 = (uint32_t)p.im2;
        for (uint32_t v = a4 + t; v < b4; v += stride) {
            const uint4 x = x4[v], y = y4[v], z = z4[v];
            int4 k;
            k.x = (int32_t)(x.x * m0 + y.x * m1 + z.x * m2);
            k.y = (int32_t)(x.y * m0 + y.y * m1 + z.y * m2);
            k.z = (int32_t)(x.z * m0 + y.z * m1 + z.z * m2);
            k.w = (int32_t)(x.w * m0 + y.w * m1 + z.w * m2);
            o4[v] = k;
            lo = min(min(lo, k.x), min(min(k.y, k.z), k.w));
            hi = max(max(hi, k.x), max(max(k.y, k.z), k.w));
        }
        if (a4 <= b4) {
            const uint32_t head_end = min(a4 * 4u, p.render_count), tail_begin = max(b4 * 4u, head_end);
            for (uint32_t i = p.sort_start + t; i < head_end; i += stride) {
                const int32_t k = depth_key_one(p, i);
                p.keys_out[i] = k; lo = min(lo, k); hi = max(hi, k);
            }
            for (uint32_t i = tail_begin + t; i < p.render_count; i += stride) {
                const int32_t k = depth_key_one(p, i);
                p.keys_out[i] = k; lo = min(lo, k); hi = max(hi, k);
            }
        } else {                                           // fewer than one aligned vector: all scalar
            for (uint32_t i = p.sort_start + t; i < p.render_count; i += stride) {
                const int32_t k = depth_key_one(p, i);
                p.keys_out[i] = k; lo = min(lo, k); hi = max(hi, k);
            }
        }
    } else {
        // the list's length may live on the device (an asynchronous gs_tree_gather): the sorted result then has that length,
        // published where the consumers of a culled sort look for it
        const uint32_t R = p.count_dev ? min(*p.count_dev, p.render_count) : p.render_count;
        if (p.count_dev && t == 0) p.frame->kept = R;
        for (uint32_t i = p.sort_start + t; i < R; i += stride) {
            const uint32_t g = p.idx_in ? min(p.idx_in[i], p.last_splat) : i;
            const int32_t k = depth_key_one(p, g);
            p.keys_out[i] = k;
            lo = min(lo, k);
            hi = max(hi, k);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = min(lo, __shfl_xor(lo, o, 64));
        hi = max(hi, __shfl_xor(hi, o, 64));
    }
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        s_lo[wave] = lo;
        s_hi[wave] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        lo = min(min(s_lo[0], s_lo[1]), min(s_lo[2], s_lo[3]));
        hi = max(max(s_hi[0], s_hi[1]), max(s_hi[2], s_hi[3]));
        // one device-scope word retires ~88 atomics per microsecond: 2 x 512 workgroups on one pair of words were ~10 us of
        // this kernel's tail.  SORT_SHARDS pairs, reduced by whoever reads them (SortFrame::lo() / hi()).
        atomicMin(&p.frame->key_min[blockIdx.x % SORT_SHARDS], lo);
        atomicMax(&p.frame->key_max[blockIdx.x % SORT_SHARDS], hi);
    }
}

// Phase A with the per-splat frustum cull (gs_sorter_set_frustum_cull).  Same keys, same min / max over EVERY list
// position (so the bucket of a kept splat is the one the full sort gives it), plus one keep bit per position:
//     q = mvp * (x, y, z, 1)    fp32, ((m0*x + m4*y) + m8*z) + m12 per row, no contraction
//     drop  <=>  |q.x| > 1.25*q.w + 0.01  or  |q.y| > 1.25*q.w + 0.01  or  q.z < -(1.01*q.w + 0.01)  or  q.z > 1.01*q.w + 0.01
// with (x, y, z) = the sorter's own centres as floats (integer mode: (float)int * 0.001f).  The vertex stage drops a
// splat on |clip.xy| > 1.2 w, clip.z < -1.2 w or ndc.z outside [-1, 1] (SplatMaterial.js:160-164 and the GL clip of a
// quad that sits at its centre's depth), so for the same camera the kept set is a superset of what can reach the
// frame, and the frame is bit-identical to the one the full sort produces.  Restated in oracle.frustum_keep.
__device__ __forceinline__ bool frustum_keep_one(const float* m, float x, float y, float z) {
    float q[4];
#pragma unroll
    for (int r = 0; r < 4; r++)
        q[r] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(m[r], x), __fmul_rn(m[4 + r], y)), __fmul_rn(m[8 + r], z)), m[12 + r]);
    const float lim_xy = __fadd_rn(__fmul_rn(1.25f, q[3]), 0.01f);
    const float lim_z = __fadd_rn(__fmul_rn(1.01f, q[3]), 0.01f);
    return !(fabsf(q[0]) > lim_xy || fabsf(q[1]) > lim_xy || q[2] < -lim_z || q[2] > lim_z);
}

__device__ __forceinline__ bool frustum_keep_int(const float* m, uint32_t x, uint32_t y, uint32_t z) {
    return frustum_keep_one(m, __fmul_rn((float)(int32_t)x, 0.001f), __fmul_rn((float)(int32_t)y, 0.001f),
                            __fmul_rn((float)(int32_t)z, 0.001f));
}

// VEC4 (identity list, static integer mode): lane l of a wave owns positions 4*(v0 + l) .. +3 of a 256-position window,
// read as three 16-byte plane loads; the 4 keep bits of 16 neighbouring lanes are OR-combined into one mask word.
// Otherwise: 4 positions per lane, 64 apart, so 4 index loads and then 4 centre gathers are in flight per lane and
// every ballot is one mask word.
template <bool VEC4>
__global__ __launch_bounds__(256) void k_depth_key_cull(KeyParams p) {
    __shared__ int32_t s_lo[4], s_hi[4];
    __shared__ uint32_t s_kept[4];
    int32_t lo = 2147483640, hi = -2147483640;
    uint32_t kept = 0;
    const uint32_t stride = gridDim.x * blockDim.x;               // a multiple of 64
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t w = t; w < (uint32_t)RADIX_TOTAL_WORDS; w += stride) p.digit_total[w] = 0u;
    if (t < SORT_SHARDS) {
        p.next_frame->key_min[t] = 2147483640;
        p.next_frame->key_max[t] = -2147483640;
    }
    if (t == 0) {
        p.next_frame->clamped = 0;
        p.next_frame->kept = 0;
    }
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // sort_start == 0 in this variant; the length of a gathered list may live on the device (asynchronous gs_tree_gather)
    const uint32_t R = (!VEC4 && p.count_dev) ? min(*p.count_dev, p.render_count) : p.render_count;
    if (VEC4) {
        const uint32_t nvec = (R + 3u) / 4u, full = R / 4u, padded = (nvec + 63u) & ~63u;
        const uint4* x4 = reinterpret_cast<const uint4*>(p.cx);
        const uint4* y4 = reinterpret_cast<const uint4*>(p.cy);
        const uint4* z4 = reinterpret_cast<const uint4*>(p.cz);
        int4* o4 = reinterpret_cast<int4*>(p.keys_out);
        const uint32_t m0 = (uint32_t)p.im0, m1 = (uint32_t)p.im1, m2 = (uint32_t)p.im2;
        for (uint32_t v = t; v < padded; v += stride) {
            uint32_t nib = 0;
            if (v < full) {
                const uint4 x = x4[v], y = y4[v], z = z4[v];
                int4 k;
                k.x = (int32_t)(x.x * m0 + y.x * m1 + z.x * m2);
                k.y = (int32_t)(x.y * m0 + y.y * m1 + z.y * m2);
                k.z = (int32_t)(x.z * m0 + y.z * m1 + z.z * m2);
                k.w = (int32_t)(x.w * m0 + y.w * m1 + z.w * m2);
                o4[v] = k;
                lo = min(min(lo, k.x), min(min(k.y, k.z), k.w));
                hi = max(max(hi, k.x), max(max(k.y, k.z), k.w));
                nib = (frustum_keep_int(p.mvp, x.x, y.x, z.x) ? 1u : 0u) | (frustum_keep_int(p.mvp, x.y, y.y, z.y) ? 2u : 0u) |
                      (frustum_keep_int(p.mvp, x.z, y.z, z.z) ? 4u : 0u) | (frustum_keep_int(p.mvp, x.w, y.w, z.w) ? 8u : 0u);
            } else if (v < nvec) {                                 // the ragged last vector
                for (uint32_t c = 0; c < 4u; c++) {
                    const uint32_t i = 4u * v + c;
                    if (i < R) {
                        const uint32_t x = p.cx[i], y = p.cy[i], z = p.cz[i];
                        const int32_t k = (int32_t)(x * m0 + y * m1 + z * m2);
                        p.keys_out[i] = k; lo = min(lo, k); hi = max(hi, k);
                        nib |= frustum_keep_int(p.mvp, x, y, z) ? (1u << c) : 0u;
                    }
                }
            }
            kept += (uint32_t)__popc(nib);
            unsigned long long word = (unsigned long long)nib << (4u * (lane & 15u));
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) word |= __shfl_xor(word, o, 64);
            if ((lane & 15u) == 0u) p.keep[v >> 4] = word;          // positions 4v .. 4v+63
        }
    } else {
        const uint32_t padded = (R + 255u) & ~255u;
        for (uint32_t base = (t >> 6) * 256u; base < padded; base += (stride >> 6) * 256u) {
            uint32_t g[4];
            bool in[4], keep[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t i = base + 64u * k + lane;
                in[k] = i < R;
                g[k] = in[k] ? (p.idx_in ? min(p.idx_in[i], p.last_splat) : i) : 0u;
            }
            uint4 c[4];
#pragma unroll
            for (int k = 0; k < 4; k++) c[k] = in[k] ? p.aos[g[k]] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                keep[k] = false;
                if (in[k]) {
                    const uint32_t i = base + 64u * k + lane;
                    int32_t key;
                    float x, y, z;
                    if (p.mode & MODE_INT) {
                        key = (int32_t)(c[k].x * (uint32_t)p.im0 + c[k].y * (uint32_t)p.im1 + c[k].z * (uint32_t)p.im2);
                        x = __fmul_rn((float)(int32_t)c[k].x, 0.001f); y = __fmul_rn((float)(int32_t)c[k].y, 0.001f);
                        z = __fmul_rn((float)(int32_t)c[k].z, 0.001f);
                    } else {
                        x = __uint_as_float(c[k].x); y = __uint_as_float(c[k].y); z = __uint_as_float(c[k].z);
                        float s = __fmul_rn(p.fm0, x);
                        s = __fadd_rn(s, __fmul_rn(p.fm1, y));
                        s = __fadd_rn(s, __fmul_rn(p.fm2, z));
                        key = trunc_f64_i32((double)s * 4096.0);
                    }
                    p.keys_out[i] = key;
                    lo = min(lo, key);
                    hi = max(hi, key);
                    keep[k] = frustum_keep_one(p.mvp, x, y, z);
                }
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const unsigned long long word = __ballot(keep[k]);
                if (lane == 0 && base + 64u * k < padded) {
                    p.keep[(base >> 6) + k] = word;
                    kept += (uint32_t)__popcll(word);
                }
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = min(lo, __shfl_xor(lo, o, 64));
        hi = max(hi, __shfl_xor(hi, o, 64));
        kept += __shfl_xor(kept, o, 64);
    }
    if (lane == 0) {
        s_lo[wave] = lo;
        s_hi[wave] = hi;
        s_kept[wave] = kept;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        lo = min(min(s_lo[0], s_lo[1]), min(s_lo[2], s_lo[3]));
        hi = max(max(s_hi[0], s_hi[1]), max(s_hi[2], s_hi[3]));
        atomicMin(&p.frame->key_min[blockIdx.x % SORT_SHARDS], lo);
        atomicMax(&p.frame->key_max[blockIdx.x % SORT_SHARDS], hi);
        // (the number of kept positions is published by pass 0 of the radix sort - the sum of its digit totals - instead of by
        // one atomicAdd per workgroup on a single word here: radix.hpp, count_out)
    }
}

// ---------------------------------------------------------------------------------------------------
// Octree-gathered lists (gs_tree_gather -> gs_sorter_sort_gathered): the copy and phase A in one kernel.
// gs_tree_gather only PLANS (per kept leaf: where its index list goes); a full sort of a static scene then runs this kernel
// instead of k_tree_copy + k_depth_key(_cull): one wave per leaf streams the leaf's indexes, its centres and its payloads from
// arrays kept in the TREE'S leaf-major order (leaf_centers / leaf_pos: built once per (tree, centres, bound mesh) by
// k_tree_leaf_cache), and writes the list, its payloads, its keys, min / max and - with the per-splat frustum cull - the keep
// bits.  Rounds 2-3 keyed the gathered list through 16-byte centre gathers at random (86 us for 4.4 M entries: MI355X retires
// ~55 G random accesses per second whatever their size, tools/probes/gather_rate.hip) and translated the payloads through
// another random gather in pass 0 of the radix sort.
struct TreeCopyParams {
    TreeGatherView v;
    const uint4* leaf_centers;
    const uint32_t* leaf_pos;
    uint32_t* idx_out;
    uint32_t* pay_out;
};

__global__ __launch_bounds__(256) void k_tree_leaf_cache(const uint32_t* __restrict__ leaf_indexes, uint32_t n, const uint4* __restrict__ caos,
                                                         const uint32_t* __restrict__ map, uint32_t last_splat,
                                                         uint4* __restrict__ leaf_centers, uint32_t* __restrict__ leaf_pos) {
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
        const uint32_t o = min(leaf_indexes[j], last_splat);
        leaf_centers[j] = caos[o];
        leaf_pos[j] = map ? map[o] : o;
    }
}

template <bool CULL>
__global__ __launch_bounds__(256) void k_tree_copy_keys(TreeCopyParams c, KeyParams p) {
    __shared__ int32_t s_lo[4], s_hi[4];
    __shared__ uint32_t s_kept[4];
    const uint32_t stride = gridDim.x * blockDim.x, gt = blockIdx.x * blockDim.x + threadIdx.x;
    // the housekeeping of a sort's first kernel (see k_depth_key)
    for (uint32_t w = gt; w < (uint32_t)RADIX_TOTAL_WORDS; w += stride) p.digit_total[w] = 0u;
    if (gt < SORT_SHARDS) {
        p.next_frame->key_min[gt] = 2147483640;
        p.next_frame->key_max[gt] = -2147483640;
    }
    if (gt == 0) {
        p.next_frame->clamped = 0;
        p.next_frame->kept = 0;
        if (!CULL) p.frame->kept = c.v.totals[0];            // the list's length, where the consumers of a culled sort look for it
    }
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    int32_t lo = 2147483640, hi = -2147483640;
    uint32_t kept = 0;
    // A wave's work is a sequence of ROUNDS: 256 consecutive entries (four 64-splat slices, twelve loads that leave together) of
    // one kept leaf's list - a leaf holds ~200 splats, so most leaves are one round.  Round 6: the loads of round k + 1 are issued
    // BEFORE round k is keyed and stored (two register sets, no copies: the loop is unrolled by two), also across leaves; before,
    // a wave's memory system idled between the stores of one leaf and the loads of the next (the same finding as k_cull_front,
    // load -> compute -> store in step leaves the memory idle between bursts).  The loads are unconditional - a lane
    // past the end of its leaf, or a wave past its last round, reads entry 0 of the round's leaf / of the arrays - because a load
    // inside a branch costs a full s_waitcnt at the join, taken or not.
    struct Round { uint32_t off, n, src, t0, valid; };
    struct Loads { uint32_t idx[4], pos[4]; uint4 ce[4]; };
    const uint32_t wstride = gridDim.x * 4u;
    uint32_t leaf = blockIdx.x * 4u + wave;                  // the next leaf whose header words are in m_*
    uint32_t m_off = 0xFFFFFFFFu, m_n = 0, m_src = 0;
    if (leaf < c.v.leaves) { m_off = c.v.leaf_offset[leaf]; m_n = c.v.leaf_count[leaf]; m_src = c.v.leaf_begin[leaf]; }
    auto advance = [&](Round& r) {                           // (wave-uniform)
        if (r.valid && r.t0 + 256u < r.n) { r.t0 += 256u; return; }
        r.valid = 0u;
        while (leaf < c.v.leaves) {
            const uint32_t off = m_off, n = m_n, src = m_src;
            leaf += wstride;
            if (leaf < c.v.leaves) { m_off = c.v.leaf_offset[leaf]; m_n = c.v.leaf_count[leaf]; m_src = c.v.leaf_begin[leaf]; }
            if (off != 0xFFFFFFFFu && n) { r.off = off; r.n = n; r.src = src; r.t0 = 0u; r.valid = 1u; return; }   // (else: a culled leaf)
        }
    };
    auto load = [&](const Round& r, Loads& L) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t t = r.t0 + 64u * k + lane;
            const uint32_t j = r.valid ? r.src + (t < r.n ? t : 0u) : 0u;
            L.idx[k] = c.v.leaf_indexes[j];
            L.pos[k] = c.leaf_pos[j];
            L.ce[k] = c.leaf_centers[j];
        }
    };
    unsigned long long carry = 0ull;                         // (lane 0) keep bits of this leaf that belong to the next round's first word
    auto process = [&](const Round& r, const Loads& L) {
        const uint32_t off = r.off, n = r.n, t0 = r.t0;
        if (t0 == 0u) carry = 0ull;
        unsigned long long bits[4] = {0ull, 0ull, 0ull, 0ull};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t t = t0 + 64u * k + lane;
            const bool in = t < n;
            int32_t key;
            float x, y, z;
            if (p.mode & MODE_INT) {
                key = (int32_t)(L.ce[k].x * (uint32_t)p.im0 + L.ce[k].y * (uint32_t)p.im1 + L.ce[k].z * (uint32_t)p.im2);
                x = __fmul_rn((float)(int32_t)L.ce[k].x, 0.001f); y = __fmul_rn((float)(int32_t)L.ce[k].y, 0.001f);
                z = __fmul_rn((float)(int32_t)L.ce[k].z, 0.001f);
            } else {
                x = __uint_as_float(L.ce[k].x); y = __uint_as_float(L.ce[k].y); z = __uint_as_float(L.ce[k].z);
                float s = __fmul_rn(p.fm0, x);
                s = __fadd_rn(s, __fmul_rn(p.fm1, y));
                s = __fadd_rn(s, __fmul_rn(p.fm2, z));
                key = trunc_f64_i32((double)s * 4096.0);
            }
            if (in) {
                c.idx_out[off + t] = L.idx[k];
                c.pay_out[off + t] = L.pos[k];
                p.keys_out[off + t] = key;
                lo = min(lo, key);
                hi = max(hi, key);
            }
            if (CULL) {
                bits[k] = __ballot(in && frustum_keep_one(p.mvp, x, y, z));
                kept += (uint32_t)__popcll(bits[k]);         // (the same value in every lane)
            }
        }
        if (CULL && lane == 0u) {
            // The round's 256 list positions start at `first`: four words of the (zeroed) keep mask and a carry into the
            // fifth, which the next round of this leaf completes.  A word that lies wholly inside this leaf's range is ours
            // alone - a plain store; the words the leaf shares with its neighbours in the list take an atomicOr (two per
            // slice, 136 k atomics per gather, were ~15 us of this kernel).
            const uint32_t first = off + t0, sh = first & 63u, w0 = first >> 6;
            unsigned long long W[4];
            W[0] = carry | (bits[0] << sh);
#pragma unroll
            for (int j = 1; j < 4; j++) W[j] = (bits[j] << sh) | (sh ? bits[j - 1] >> (64u - sh) : 0ull);
            carry = sh ? bits[3] >> (64u - sh) : 0ull;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t wbeg = (w0 + (uint32_t)j) * 64u;            // first list position of the word
                if (wbeg >= off && wbeg + 64u <= off + n) p.keep[w0 + j] = W[j];
                else if (W[j]) atomicOr(&p.keep[w0 + j], W[j]);
            }
            if (t0 + 256u >= n && carry) atomicOr(&p.keep[w0 + 4u], carry);   // the leaf's last round: its tail word
        }
    };
    Round ra = {0u, 0u, 0u, 0u, 0u}, rb;
    Loads A, B;
    advance(ra);
    load(ra, A);
    while (ra.valid) {
        rb = ra; advance(rb);
        load(rb, B);                                         // in flight while A is keyed and stored
        process(ra, A);
        if (!rb.valid) break;
        ra = rb; advance(ra);
        load(ra, A);
        process(rb, B);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = min(lo, __shfl_xor(lo, o, 64));
        hi = max(hi, __shfl_xor(hi, o, 64));
    }
    if (lane == 0) {
        s_lo[wave] = lo;
        s_hi[wave] = hi;
        s_kept[wave] = kept;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        lo = min(min(s_lo[0], s_lo[1]), min(s_lo[2], s_lo[3]));
        hi = max(max(s_hi[0], s_hi[1]), max(s_hi[2], s_hi[3]));
        if (lo <= hi) {
            atomicMin(&p.frame->key_min[blockIdx.x % SORT_SHARDS], lo);
            atomicMax(&p.frame->key_max[blockIdx.x % SORT_SHARDS], hi);
        }
        // (CULL: the kept count is published by pass 0 of the radix sort, radix.hpp count_out)
    }
}

// ---------------------------------------------------------------------------------------------------
// Visibility-culled sort (gs_sorter_set_visibility_cull): compact, then sort the survivors.
// The bound mesh's vertex stage left one bit per ORIGINAL splat index (gs_mesh::vis_orig) for this camera and strip.  The
// reference's buckets depend on min / max over every sorted position, so every splat is still keyed once - but nothing is
// stored for the ones that draw nothing:
//   k_minmax_count   one streaming pass over the centres (12 bytes per splat): min / max, and the number of set mask bits of
//                    every workgroup's contiguous chunk of positions;
//   k_mask_compact   mask -> ascending list of the surviving splat indexes (a subsequence of the identity list, so the stable
//                    sort keeps the reference's tie order), its length -> SortFrame::kept;
//   then the ordinary index-list sort runs on that list with the min / max above (k_depth_key ext_minmax, DepthLoader n_dev).
// A rank of a multi-GPU draw therefore streams 12 bytes per splat and radix-sorts only what its strip draws.
constexpr uint32_t VC_THREADS = 256, VC_SPAN = 1024;       // positions per workgroup iteration (4 per lane; 32 mask words)

__device__ __forceinline__ int32_t depth_key_planes(const KeyParams& p, uint32_t i) {
    if (p.mode & MODE_INT) return (int32_t)(p.cx[i] * (uint32_t)p.im0 + p.cy[i] * (uint32_t)p.im1 + p.cz[i] * (uint32_t)p.im2);
    float s = __fmul_rn(p.fm0, __uint_as_float(p.cx[i]));
    s = __fadd_rn(s, __fmul_rn(p.fm1, __uint_as_float(p.cy[i])));
    s = __fadd_rn(s, __fmul_rn(p.fm2, __uint_as_float(p.cz[i])));
    return trunc_f64_i32((double)s * 4096.0);
}

__global__ __launch_bounds__(VC_THREADS) void k_minmax_count(KeyParams p, const uint32_t* __restrict__ mask, uint32_t chunk_len,
                                                             uint32_t* __restrict__ chunk_counts) {
    __shared__ int32_t s_lo[4], s_hi[4];
    __shared__ uint32_t s_cnt[4];
    const uint32_t stride = gridDim.x * blockDim.x, t = blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t w = t; w < (uint32_t)RADIX_TOTAL_WORDS; w += stride) p.digit_total[w] = 0u;
    if (t < SORT_SHARDS) {
        p.next_frame->key_min[t] = 2147483640;
        p.next_frame->key_max[t] = -2147483640;
    }
    if (t == 0) {
        p.next_frame->clamped = 0;
        p.next_frame->kept = 0;
    }
    const uint32_t N = p.render_count;
    const uint32_t begin = min(blockIdx.x * chunk_len, N), end = min(begin + chunk_len, N);     // chunk_len % VC_SPAN == 0
    int32_t lo = 2147483640, hi = -2147483640;
    uint32_t cnt = 0;
    const uint32_t m0 = (uint32_t)p.im0, m1 = (uint32_t)p.im1, m2 = (uint32_t)p.im2;
    for (uint32_t base = begin; base < end; base += VC_SPAN) {
        const uint32_t i0 = base + 4u * threadIdx.x;
        if ((p.mode & MODE_INT) && i0 + 4u <= end) {               // 16-byte plane loads
            const uint4 x = reinterpret_cast<const uint4*>(p.cx)[i0 >> 2], y = reinterpret_cast<const uint4*>(p.cy)[i0 >> 2],
                        z = reinterpret_cast<const uint4*>(p.cz)[i0 >> 2];
            const int32_t k0 = (int32_t)(x.x * m0 + y.x * m1 + z.x * m2), k1 = (int32_t)(x.y * m0 + y.y * m1 + z.y * m2);
            const int32_t k2 = (int32_t)(x.z * m0 + y.z * m1 + z.z * m2), k3 = (int32_t)(x.w * m0 + y.w * m1 + z.w * m2);
            lo = min(min(lo, k0), min(min(k1, k2), k3));
            hi = max(max(hi, k0), max(max(k1, k2), k3));
        } else {
            for (uint32_t i = i0; i < min(i0 + 4u, end); i++) {
                const int32_t k = depth_key_planes(p, i);
                lo = min(lo, k);
                hi = max(hi, k);
            }
        }
        const uint32_t w = (base >> 5) + threadIdx.x;              // the 32 mask words of this span
        // (a sort over fewer splats than the vertex stage looked at leaves set bits beyond `end` in the last word)
        if (threadIdx.x < VC_SPAN / 32u && (w << 5) < end)
            cnt += (uint32_t)__popc(mask[w] & (end - (w << 5) >= 32u ? 0xFFFFFFFFu : (1u << (end - (w << 5))) - 1u));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = min(lo, __shfl_xor(lo, o, 64));
        hi = max(hi, __shfl_xor(hi, o, 64));
        cnt += __shfl_xor(cnt, o, 64);
    }
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (lane == 0u) { s_lo[wave] = lo; s_hi[wave] = hi; s_cnt[wave] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicMin(&p.frame->key_min[blockIdx.x % SORT_SHARDS], min(min(s_lo[0], s_lo[1]), min(s_lo[2], s_lo[3])));
        atomicMax(&p.frame->key_max[blockIdx.x % SORT_SHARDS], max(max(s_hi[0], s_hi[1]), max(s_hi[2], s_hi[3])));
        chunk_counts[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    }
}

// The mask is CONSUMED: every word is copied to the sorter's own buffer (gs_sorter_debug_read) and zeroed, which is the state the
// next gs_mesh_project expects (its survivors set bits with atomicOr) - a 725 KB memset per frame less on every rank.
__global__ __launch_bounds__(VC_THREADS) void k_mask_compact(uint32_t* __restrict__ mask, uint32_t* __restrict__ mask_copy,
                                                             const uint32_t* __restrict__ chunk_counts,
                                                             uint32_t N, uint32_t chunk_len, uint32_t* __restrict__ idx_out,
                                                             SortFrame* __restrict__ frame) {
    __shared__ uint32_t s_tmp[4], s_before[4], s_all[4];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t before = 0, all = 0;                                   // survivors of the chunks before this one / of all chunks
    for (uint32_t c = threadIdx.x; c < gridDim.x; c += VC_THREADS) {
        const uint32_t v = chunk_counts[c];
        all += v;
        before += c < blockIdx.x ? v : 0u;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        before += __shfl_xor(before, o, 64);
        all += __shfl_xor(all, o, 64);
    }
    if (lane == 0u) { s_before[wave] = before; s_all[wave] = all; }
    __syncthreads();
    uint32_t out = s_before[0] + s_before[1] + s_before[2] + s_before[3];
    if (blockIdx.x == 0 && threadIdx.x == 0) frame->kept = s_all[0] + s_all[1] + s_all[2] + s_all[3];
    const uint32_t begin = min(blockIdx.x * chunk_len, N), end = min(begin + chunk_len, N);
    for (uint32_t base = begin; base < end; base += 32u * VC_THREADS) {        // one mask word (32 positions) per thread
        const uint32_t first = base + 32u * threadIdx.x;
        uint32_t w = 0u;
        if (first < end) {
            w = mask[first >> 5];
            mask_copy[first >> 5] = w;
            if (w) mask[first >> 5] = 0u;
            if (end - first < 32u) w &= (1u << (end - first)) - 1u;        // positions beyond this sort's list
        }
        uint32_t total;
        uint32_t o = out + block_excl_scan<4>((uint32_t)__popc(w), s_tmp, &total);
        while (w) {
            idx_out[o++] = first + (uint32_t)__builtin_ctz(w);
            w &= w - 1u;
        }
        out += total;
    }
}

// Round 5: the same front end as TWO kernels that never gather.  Round 4's chain keyed the survivors through one 16-byte centre
// gather each (k_depth_key over the compacted list: 1.44 M gathers at the machine's ~55 G/s = 26 us for a full frame) and
// translated their payloads through another gather in pass 0; but the survivors are a SUBSEQUENCE of the identity list, so the
// pass that streams every centre for min / max can key them on the way:
//   k_mask_count   survivors per chunk of positions, from the mask alone (1 bit per splat: 0.7 MB) + the sort's housekeeping;
//   k_cull_front   one streaming pass over the centres (12 bytes per splat): min / max over EVERY splat, and for the survivors of
//                  the chunk - whose first output slot is the sum of the earlier chunks' counts - key and payload (the bound
//                  mesh's position, read for survivors only) written to the compacted list.  The mask is consumed as before.
// No look-back, no spinning: the offsets come from the first kernel's counts.
// A workgroup owns a contiguous run of positions (chunk_len % VC_TURN == 0) and walks it in turns of VC_TURN: wave w of the workgroup
// takes VC_UNROLL consecutive spans of the turn, a lane four consecutive positions of a span (one 16-byte load per
// plane), so the workgroup streams 8 KB per plane per turn and the chip ~2000 streams in all.  A wave's first output slot in a
// turn = the workgroup's running offset + the survivors of the lower waves in this turn (one barrier per turn, counts double
// buffered); inside a wave the prefix over the 64 lanes is a DPP scan.  (Turn = 2048 positions with VC_UNROLL = 2.)
// (History, r05c-h: cutting the list per workgroup with a block scan every 1024 positions cost a rank's strip 22 barriers per chunk
// for nothing; one contiguous run per WAVE needs no barrier at all but makes 8192 concurrent DRAM streams - 36-42 us for the 93 MB;
// spans handled one by one put a store round trip between them - predicated stores cannot be counted, the compiler drains the queue -
// and a run-time `map ? load : index` put every payload load in a branch of its own with a full wait behind it.)
constexpr uint32_t VC_WAVE_SPAN = 256;                     // positions per wave and span (4 per lane; 8 mask words)
constexpr uint32_t VC_UNROLL = 2;                          // spans per wave and turn (two turns' loads in flight: 2 x 8 x 16 bytes per lane)
constexpr uint32_t VC_TURN = VC_UNROLL * VC_WAVE_SPAN * (VC_THREADS / 64u);   // positions per workgroup and turn (4096)

__global__ __launch_bounds__(VC_THREADS) void k_mask_count(KeyParams p, const uint32_t* __restrict__ mask, uint32_t chunk_len,
                                                           uint32_t* __restrict__ chunk_counts) {
    __shared__ uint32_t s_cnt[4];
    const uint32_t stride = gridDim.x * blockDim.x, t = blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t w = t; w < (uint32_t)RADIX_TOTAL_WORDS; w += stride) p.digit_total[w] = 0u;
    if (t < SORT_SHARDS) {
        p.next_frame->key_min[t] = 2147483640;
        p.next_frame->key_max[t] = -2147483640;
    }
    if (t == 0) {
        p.next_frame->clamped = 0;
        p.next_frame->kept = 0;
    }
    const uint32_t N = p.render_count;
    const uint32_t begin = min(blockIdx.x * chunk_len, N), end = min(begin + chunk_len, N);
    uint32_t cnt = 0;
    for (uint32_t first = begin + 32u * threadIdx.x; first < end; first += 32u * VC_THREADS) {
        uint32_t w = mask[first >> 5];
        if (end - first < 32u) w &= (1u << (end - first)) - 1u;            // positions beyond this sort's list
        cnt += (uint32_t)__popc(w);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if ((threadIdx.x & 63u) == 0u) s_cnt[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) chunk_counts[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

// k_mask_count for a projection whose by-original-index mask was NOT written by the vertex stage (round 6).  gs_mesh_project used
// to set one bit per survivor with an atomicOr at inv_perm[position]: 1.44 M atomics scattered over a 0.7 MB table cost k_project
// 25 us of a full C3 frame (47 -> 72 us in the kernel table of the visibility-culled frame, profiles/r06n_vis_n1_kstats.txt).
// The sorter can build the same mask from what it streams anyway: position i of the identity list is the splat at storage
// position map[i], whose bit is in the vertex stage's own storage-order mask (vis_mask, 1 bit per position, 0.7 MB: it stays in
// L2) - and three of four splats sit in a storage block with no survivor at all (block_any, kept as bits in LDS as the binner
// does), so only a quarter of the positions gather anything.  64 positions per wave and step (ballot -> one 8-byte word), four
// steps in flight.  Writes the mask words of its chunk (so k_cull_front / k_mask_compact find what they always found) and the
// chunk's survivor count; full-frame projections only - a strip keeps a few per cent and pays few atomics.
constexpr uint32_t VC_ANY_WORDS = 2048;                    // block_any as bits in LDS: 65536 storage blocks = 16.7 M splats
__global__ __launch_bounds__(VC_THREADS) void k_mask_derive_count(KeyParams p, const uint32_t* __restrict__ map,
                                                                  const unsigned long long* __restrict__ vis_mask,
                                                                  const uint8_t* __restrict__ block_any, uint32_t blocks, uint32_t chunk_len,
                                                                  uint32_t subs, unsigned long long* __restrict__ mask64,
                                                                  uint32_t* __restrict__ chunk_counts) {
    __shared__ uint32_t s_cnt[4];
    __shared__ uint32_t s_any[VC_ANY_WORDS];
    const uint32_t stride = gridDim.x * blockDim.x, t = blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t w = t; w < (uint32_t)RADIX_TOTAL_WORDS; w += stride) p.digit_total[w] = 0u;      // (the sort's housekeeping: k_mask_count's)
    if (t < SORT_SHARDS) {
        p.next_frame->key_min[t] = 2147483640;
        p.next_frame->key_max[t] = -2147483640;
    }
    if (t == 0) {
        p.next_frame->clamped = 0;
        p.next_frame->kept = 0;
    }
    const bool coarse = blocks <= VC_ANY_WORDS * 32u;
    if (coarse) {
        for (uint32_t w = threadIdx.x; w < (blocks + 31u) / 32u; w += VC_THREADS) {
            const uint4* src = reinterpret_cast<const uint4*>(block_any + 32u * w);               // (the buffer is padded to 64 bytes)
            const uint4 a = src[0], b = src[1];
            const uint32_t v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            uint32_t bits = 0;
#pragma unroll
            for (int k = 0; k < 8; k++)
                bits |= (((v[k] & 0xFFu) ? 1u : 0u) | ((v[k] & 0xFF00u) ? 2u : 0u) | ((v[k] & 0xFF0000u) ? 4u : 0u) |
                         ((v[k] & 0xFF000000u) ? 8u : 0u)) << (4 * k);
            s_any[w] = bits;
        }
    }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t N = p.render_count;
    // `subs` workgroups share a chunk of the front end (which runs two workgroups per CU; this kernel is a chain of two dependent
    // round trips per round - the map, then the gather - and wants every SIMD full of waves): piece blockIdx.x % subs of chunk
    // blockIdx.x / subs, counts per piece.  A wave takes DERIVE_STEPS x 64 consecutive positions per round, and the NEXT round's
    // map words are fetched before this round's gathers are used.  (One workgroup per chunk, 4 steps, no prefetch: 25 us for the
    // C3 frame's 5.8 M positions; 8 steps + prefetch: 20.9 us; r06p kernel tables.)
    const uint32_t chunk = blockIdx.x / subs, piece = blockIdx.x % subs, piece_len = chunk_len / subs;     // chunk_len % (4096 * subs) == 0
    const uint32_t cend = min(min(chunk * chunk_len, N) + chunk_len, N);
    const uint32_t begin = min(chunk * chunk_len + piece * piece_len, cend), end = min(begin + piece_len, cend);
    uint32_t cnt = 0;
    constexpr uint32_t DERIVE_STEPS = 8, DERIVE_ROUND = DERIVE_STEPS * 64u * (VC_THREADS / 64u);
    static_assert(VC_TURN % DERIVE_ROUND == 0, "a chunk is a whole number of rounds");
    uint32_t nxt[DERIVE_STEPS];
    auto fetch = [&](uint32_t base) {
#pragma unroll
        for (uint32_t k = 0; k < DERIVE_STEPS; k++) nxt[k] = map[min(base + 64u * k + lane, N - 1u)];     // (unconditional, clamped)
    };
    if (begin < end) fetch(begin + DERIVE_STEPS * 64u * wave);
    for (uint32_t base = begin + DERIVE_STEPS * 64u * wave; base < end; base += DERIVE_ROUND) {
        uint32_t pos[DERIVE_STEPS];
        bool live[DERIVE_STEPS];
#pragma unroll
        for (uint32_t k = 0; k < DERIVE_STEPS; k++) {
            pos[k] = nxt[k];
            live[k] = base + 64u * k + lane < end;
        }
        fetch(min(base + DERIVE_ROUND, N - 1u));
#pragma unroll
        for (uint32_t k = 0; k < DERIVE_STEPS; k++)
            live[k] = live[k] && (coarse ? ((s_any[pos[k] >> 13] >> ((pos[k] >> 8) & 31u)) & 1u) != 0u : block_any[pos[k] >> 8] != 0);
        unsigned long long w[DERIVE_STEPS];
#pragma unroll
        for (uint32_t k = 0; k < DERIVE_STEPS; k++) w[k] = vis_mask[live[k] ? pos[k] >> 6 : 0u];         // (unconditional: no wait at a join)
#pragma unroll
        for (uint32_t k = 0; k < DERIVE_STEPS; k++) {
            const unsigned long long word = __ballot(live[k] && ((w[k] >> (pos[k] & 63u)) & 1ull));
            cnt += (uint32_t)__popcll(word);                                                      // (the same value in every lane)
            if (lane == 0u && base + 64u * k < end) mask64[(base + 64u * k) >> 6] = word;
        }
    }
    if (lane == 0u) s_cnt[wave] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) chunk_counts[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

// MAP: payloads come from the bound mesh's position table (a compile-time switch, see above)
template <bool MAP>
__global__ __launch_bounds__(VC_THREADS) void k_cull_front(KeyParams p, uint32_t* __restrict__ mask, uint32_t* __restrict__ mask_copy,
                                                           const uint32_t* __restrict__ chunk_counts, uint32_t chunk_len,
                                                           const uint32_t* __restrict__ map, uint32_t* __restrict__ pay_out, uint32_t subs) {
    __shared__ uint32_t s_before[4], s_all[4], s_turn[2][4];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    static_assert(VC_THREADS == 256, "four waves per workgroup");
    uint32_t before = 0, all = 0;                                   // survivors of the chunks before this one / of all chunks
    // (the counts arrive in `subs` consecutive pieces per chunk: k_mask_count writes one, k_mask_derive_count one per sub-workgroup)
    const uint32_t pieces = gridDim.x * subs, mine = blockIdx.x * subs;
    for (uint32_t c0 = 0; c0 < pieces; c0 += 4u * VC_THREADS) {     // (four loads in flight per lane)
        uint32_t x[4];
#pragma unroll
        for (uint32_t k = 0; k < 4u; k++) x[k] = chunk_counts[min(c0 + VC_THREADS * k + threadIdx.x, pieces - 1u)];
#pragma unroll
        for (uint32_t k = 0; k < 4u; k++) {
            const uint32_t c = c0 + VC_THREADS * k + threadIdx.x;
            all += c < pieces ? x[k] : 0u;
            before += c < mine ? x[k] : 0u;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        before += __shfl_xor(before, o, 64);
        all += __shfl_xor(all, o, 64);
    }
    if (lane == 0u) { s_before[wave] = before; s_all[wave] = all; }
    __syncthreads();
    uint32_t out = s_before[0] + s_before[1] + s_before[2] + s_before[3];          // the workgroup's running output slot
    if (blockIdx.x == 0u && threadIdx.x == 0u) p.frame->kept = s_all[0] + s_all[1] + s_all[2] + s_all[3];
    const uint32_t N = p.render_count;
    const uint32_t begin = min(blockIdx.x * chunk_len, N), end = min(begin + chunk_len, N);       // chunk_len % VC_TURN == 0
    int32_t lo = 2147483640, hi = -2147483640;
    const uint32_t m0 = (uint32_t)p.im0, m1 = (uint32_t)p.im1, m2 = (uint32_t)p.im2;
    const uint4* __restrict__ x4 = reinterpret_cast<const uint4*>(p.cx);
    const uint4* __restrict__ y4 = reinterpret_cast<const uint4*>(p.cy);
    const uint4* __restrict__ z4 = reinterpret_cast<const uint4*>(p.cz);
    int32_t* __restrict__ keys_out = p.keys_out;
    const uint4* __restrict__ map4 = reinterpret_cast<const uint4*>(map);
    // The loads of turn t + 1 are issued BEFORE turn t is keyed and stored (two register sets): all workgroups run in step, so a
    // turn that loads, then computes, then stores left the memory system idle between the bursts - 36-44 us for 93 MB (r05h/i).
    struct Turn { uint4 X[VC_UNROLL], Y[VC_UNROLL], Z[VC_UNROLL], M[VC_UNROLL]; uint32_t raw[VC_UNROLL]; };
    auto fetch = [&](uint32_t base, Turn& t) {
        const uint32_t wbase = base + wave * (VC_UNROLL * VC_WAVE_SPAN);    // this wave's spans of the turn
#pragma unroll
        for (uint32_t u = 0; u < VC_UNROLL; u++) {
            const uint32_t i0 = wbase + u * VC_WAVE_SPAN + 4u * lane;      // this lane's four positions of span u
            // (unconditional loads; the planes and the mesh's position table are padded by 16 bytes, so the vector that holds the
            // list's last position is readable whatever N is)
            const uint32_t ic = i0 < N ? i0 : 0u;
            t.X[u] = x4[ic >> 2]; t.Y[u] = y4[ic >> 2]; t.Z[u] = z4[ic >> 2];
            t.M[u] = MAP ? map4[ic >> 2] : make_uint4(ic, ic + 1u, ic + 2u, ic + 3u);   // the payloads travel with the centres
            t.raw[u] = i0 < end ? mask[i0 >> 5] : 0u;                      // the mask word of the four positions (eight lanes share one)
        }
    };
    Turn cur, nxt;
    if (begin < end) fetch(begin, cur);
    uint32_t turn = 0;
    for (uint32_t base = begin; base < end; base += VC_TURN, turn++) {
        const uint32_t wbase = base + wave * (VC_UNROLL * VC_WAVE_SPAN);
        fetch(min(base + VC_TURN, end - 1u), nxt);                          // (past the end: a clamped re-read, never used)
        // phase 1, registers only: keys, survivor nibbles and slots (relative to the wave's first) of all the spans of the turn ...
        int32_t K[VC_UNROLL][4];
        uint32_t nibs[VC_UNROLL], first[VC_UNROLL], mine_total = 0;
#pragma unroll
        for (uint32_t u = 0; u < VC_UNROLL; u++) {
            const uint32_t i0 = wbase + u * VC_WAVE_SPAN + 4u * lane;
            int32_t k[4] = {0, 0, 0, 0};
            uint32_t word = cur.raw[u];
            if (i0 < end && end - (i0 & ~31u) < 32u) word &= (1u << (end - (i0 & ~31u))) - 1u;    // positions beyond this sort's list
            {   // keys from the vectors (no memory access in this phase: a load in a branch here puts a full wait at its join)
                const uint32_t xs[4] = {cur.X[u].x, cur.X[u].y, cur.X[u].z, cur.X[u].w}, ys[4] = {cur.Y[u].x, cur.Y[u].y, cur.Y[u].z, cur.Y[u].w},
                               zs[4] = {cur.Z[u].x, cur.Z[u].y, cur.Z[u].z, cur.Z[u].w};
#pragma unroll
                for (uint32_t c = 0; c < 4u; c++) {
                    if (p.mode & MODE_INT) {
                        k[c] = (int32_t)(xs[c] * m0 + ys[c] * m1 + zs[c] * m2);
                    } else {
                        float sm = __fmul_rn(p.fm0, __uint_as_float(xs[c]));
                        sm = __fadd_rn(sm, __fmul_rn(p.fm1, __uint_as_float(ys[c])));
                        sm = __fadd_rn(sm, __fmul_rn(p.fm2, __uint_as_float(zs[c])));
                        k[c] = trunc_f64_i32((double)sm * 4096.0);
                    }
                    if (i0 + c < end) {                                    // (min / max over the list's positions only)
                        lo = min(lo, k[c]);
                        hi = max(hi, k[c]);
                    }
                }
            }
            const uint32_t nib = (word >> (i0 & 31u)) & 15u, mine = (uint32_t)__popc(nib);
            const uint32_t incl = wave_incl_scan_dpp(mine);
            nibs[u] = nib;
            first[u] = mine_total + incl - mine;
            mine_total += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
#pragma unroll
            for (uint32_t c = 0; c < 4u; c++) K[u][c] = k[c];
        }
        // the wave's first slot of the turn: the survivors of the lower waves (counts double buffered: one barrier per turn)
        if (lane == 0u) s_turn[turn & 1u][wave] = mine_total;
        __syncthreads();
        uint32_t wave_out = out, turn_total = 0;
#pragma unroll
        for (uint32_t w = 0; w < 4u; w++) {
            const uint32_t c = s_turn[turn & 1u][w];
            wave_out += w < wave ? c : 0u;
            turn_total += c;
        }
        out += turn_total;
        // ... phase 2, every store of the turn behind them
#pragma unroll
        for (uint32_t u = 0; u < VC_UNROLL; u++) {
            const uint32_t i0 = wbase + u * VC_WAVE_SPAN + 4u * lane;
            // the mask is consumed: copied for gs_sorter_debug_read, then zeroed (no other wave touches these words, and this
            // wave has read them all)
            if (i0 < end && (i0 & 31u) == 0u) {
                mask_copy[i0 >> 5] = cur.raw[u];
                if (cur.raw[u]) mask[i0 >> 5] = 0u;
            }
            uint32_t o = wave_out + first[u];
            const uint32_t pm[4] = {cur.M[u].x, cur.M[u].y, cur.M[u].z, cur.M[u].w};
#pragma unroll
            for (uint32_t c = 0; c < 4u; c++)
                if ((nibs[u] >> c) & 1u) {                                 // (stores only: see phase 1)
                    keys_out[o] = K[u][c];
                    pay_out[o] = pm[c];
                    o++;
                }
        }
        cur = nxt;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = min(lo, __shfl_xor(lo, o, 64));
        hi = max(hi, __shfl_xor(hi, o, 64));
    }
    if (lane == 0u && lo <= hi) {
        atomicMin(&p.frame->key_min[(blockIdx.x * 4u + wave) % SORT_SHARDS], lo);
        atomicMax(&p.frame->key_max[(blockIdx.x * 4u + wave) % SORT_SHARDS], hi);
    }
}

// Phase B as a radix loader.  Logical element j <-> list position i = R-1-j (reverse traversal makes the
// stable ascending sort of key' = range-1-bucket equal to the reference's descending, tie-reversed order).
// IDX: the list is an index (or ready-made payload) array, not the identity - a compile-time switch: as a run-time `idx ? load : i`
// every index load sat in a branch of its own with a full memory wait behind it, eight serialised round trips per tile in every
// sort of a gathered / compacted list (r05k ISA of the pass-0 scatter).
template <bool CULL, bool IDX>
struct DepthLoaderT {
    const int32_t* __restrict__ keys;
    const unsigned long long* __restrict__ keep;   // CULL: 1 bit per list position (k_depth_key_cull)
    const uint32_t* __restrict__ idx;      // nullable: identity
    const uint32_t* __restrict__ map;      // nullable: payload = map[splat index] (a bound mesh's internal position)
    SortFrame* frame;
    const uint32_t* n_dev;                 // nullable: the list length lives on the device (visibility-culled sort)
    uint32_t sort_start, render_count, range;
    uint32_t last_splat;                   // list entries are clamped to it
    uint32_t count_clamps;                 // only the histogram launch counts, so each element counts once
    int32_t lo;
    float range_map;

    __device__ __forceinline__ void prepare() {
        if (n_dev) render_count = *n_dev;                       // sort_start is 0 in that variant
        // the sort's min / max live in SORT_SHARDS words each: lane l reads shard l, five xor-shuffles reduce them
        // (every thread reading all 64 words cost the histogram kernel 5 us)
        const uint32_t l = threadIdx.x & (SORT_SHARDS - 1u);
        lo = frame->key_min[l];
        int32_t hi = frame->key_max[l];
#pragma unroll
        for (int o = SORT_SHARDS / 2; o > 0; o >>= 1) {
            lo = min(lo, __shfl_xor(lo, o, 64));
            hi = max(hi, __shfl_xor(hi, o, 64));
        }
        // sorter.cpp:142-143: (float)(range-1) / ((float)max - (float)min), fp32, correctly rounded
        range_map = __fdiv_rn((float)(range - 1), __fsub_rn((float)hi, (float)lo));
        // wave-uniform by construction: keep both in scalar registers
        lo = __builtin_amdgcn_readfirstlane(lo);
        range_map = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(range_map)));
    }
    __device__ __forceinline__ uint32_t count() const { return render_count - sort_start; }
    __device__ __forceinline__ uint32_t bucket(uint32_t i) const { return bucket_of<false>(keys[i]); }
    // sorter.cpp:146: (int)((float)(mapped - min) * rangeMap): int32 wrap, one fp32 multiply, truncation.  Branch-free (selects):
    // the scatter decodes eight of these per lane between two waits.  COUNT: the histogram launch counts the clamped ones.
    template <bool COUNT>
    __device__ __forceinline__ uint32_t bucket_of(int32_t depth) const {
        const int32_t diff = (int32_t)((uint32_t)depth - (uint32_t)lo);
        const float f = __fmul_rn((float)diff, range_map);
        const bool fits = f >= -2147483648.0f && f < 2147483648.0f;   // else NaN (hi==lo) / overflow: WASM -> bucket 0
        const int32_t b = fits ? (int32_t)f : 0;
        const bool under = b < 0, over = !under && (uint32_t)b >= range;
        if (COUNT && count_clamps && (under || over)) atomicAdd(&frame->clamped, 1u);
        return under ? 0u : (over ? range - 1u : (uint32_t)b);
    }
    __device__ __forceinline__ uint32_t key(uint32_t j) const { return (range - 1) - bucket(render_count - 1 - j); }
    __device__ __forceinline__ uint32_t payload(uint32_t i) const {
        const uint32_t o = IDX ? min(ld32(idx, i), last_splat) : i;
        return map ? ld32(map, o) : o;
    }
    __device__ __forceinline__ uint32_t val(uint32_t j) const { return payload(render_count - 1 - j); }
    // pre + fetch = the memory reads of element j (the index list first, then everything that depends on it, so that a tile's
    // loads leave as two batches), decode = the arithmetic on them
    struct Raw { int32_t depth; uint32_t payload; };
    static __device__ __forceinline__ int prof_slot(int) { return 0; }       // GS_RADIX_PROFILE
    __device__ __forceinline__ uint32_t pre(uint32_t j) const {
        const uint32_t i = render_count - 1 - j;
        return IDX ? min(ld32(idx, i), last_splat) : i;
    }
    __device__ __forceinline__ Raw fetch(uint32_t j, uint32_t o) const {
        Raw r;
        r.depth = ld32(keys, render_count - 1 - j);
        r.payload = map ? ld32(map, o) : o;
        return r;
    }
    __device__ __forceinline__ void decode(const Raw& r, uint32_t& k, uint32_t& v) const {
        k = (range - 1) - bucket_of<false>(r.depth);
        v = r.payload;
    }
    // the histogram's view of element j
    typedef int32_t HRaw;
    __device__ __forceinline__ HRaw hist_fetch(uint32_t j) const { return ld32(keys, render_count - 1 - j); }
    __device__ __forceinline__ uint32_t hist_key(HRaw depth) const { return (range - 1) - bucket_of<true>(depth); }
    __device__ __forceinline__ bool valid(uint32_t j) const {
        if (!CULL) return true;
        const uint32_t i = render_count - 1 - j;
        return (ld32(keep, i >> 6) >> (i & 63u)) & 1ull;
    }
};
typedef DepthLoaderT<false, false> DepthLoader;          // (the identity list without a cull: what k_debug_buckets walks)

// A pass that follows a PACKING pass (radix.hpp, PACK_OUT) reads ONE word per element: what is left of the key above a value of
// `val_bits` bits; its digit is the low byte of that remainder (shift 0), its histogram is ArrayLoader<uint32_t>'s over the same
// words with shift = val_bits.
struct PackedLoader {
    const uint32_t* __restrict__ words;
    const uint32_t* __restrict__ n_dev;   // device-resident count (nullable)
    uint32_t n_host, val_bits;
    __device__ __forceinline__ void prepare() {}
    __device__ __forceinline__ uint32_t count() const { return n_dev ? *n_dev : n_host; }
    struct Raw { uint32_t w; };
    __device__ __forceinline__ uint32_t pre(uint32_t) const { return 0u; }
    __device__ __forceinline__ Raw fetch(uint32_t j, uint32_t) const { return Raw{ld32(words, j)}; }
    __device__ __forceinline__ void decode(const Raw& r, uint32_t& k, uint32_t& v) const {
        k = r.w >> val_bits;
        v = r.w & ((1u << val_bits) - 1u);
    }
    __device__ __forceinline__ bool valid(uint32_t) const { return true; }
    static __device__ __forceinline__ int prof_slot(int) { return 1; }       // GS_RADIX_PROFILE
};

__global__ void k_copy_head(const uint32_t* __restrict__ idx, const uint32_t* __restrict__ map, uint32_t* __restrict__ out,
                            uint32_t n, uint32_t last_splat) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t o = idx ? min(idx[i], last_splat) : i;
        out[i] = map ? map[o] : o;
    }
}

// (positions beyond a culled / gathered result's length hold stale words: clamped, so a caller that reads too far gets
// garbage indexes, never a fault)
__global__ void k_unmap(const uint32_t* __restrict__ in, const uint32_t* __restrict__ unmap, uint32_t* __restrict__ out,
                        uint32_t n, uint32_t last) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = unmap[min(in[i], last)];
}

// Phase A + the histogram of pass 0 in ONE launch ($GSPLAT_KEY_HIST_FUSED=1, opt-in).  A key's bucket needs the
// exact min / max over ALL keys (sorter.cpp:142-146), which exist only when the last workgroup has keyed its splats - so the launch
// is two phases around a barrier ACROSS THE GRID: one workgroup per chunk of the pass-0 scatter (radix.hpp: <= CHUNK_TILES tiles),
// every thread keeps its <= 12 keys in registers, the grid meets on one counter (no reset: it only ever grows, every launch waits for
// `base + gridDim.x`), then each workgroup reads the final min / max and histograms its own keys - the 24 MB re-read of the keys and
// the kernel boundary of k_radix_hist are what this saves, the barrier is what it costs.  Needs the whole grid resident at once (the
// host launches it only for <= 2 workgroups of 1024 threads per CU); the wait is bounded all the same: a workgroup that runs out of
// patience raises `fail` (the sort reports GS_ERR_HIP at the next statistics read) instead of hanging the device.
// Identity list, static integer mode, sort_start = 0, render_count a multiple of 4 (the 16-byte plane loads).
#ifndef KEY_HIST_SLEEP
#define KEY_HIST_SLEEP 4
#endif
struct KeyHistSync {
    uint32_t* arrive;          // grows by gridDim.x per launch
    uint32_t* fail;
    uint32_t base;             // *arrive before this launch
};
__global__ __launch_bounds__(HIST_THREADS) void k_depth_key_hist(KeyParams p, DepthLoader ld, int shift, uint32_t* __restrict__ block_hist,
                                                                  uint32_t* __restrict__ digit_total, KeyHistSync sync) {
    __shared__ uint32_t s_hist[4][RADIX_BINS];
    __shared__ int32_t s_lo[HIST_THREADS / 64], s_hi[HIST_THREADS / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t stride = gridDim.x * blockDim.x, t = blockIdx.x * blockDim.x + tid;
    // (k_depth_key's housekeeping.  The group rows are added to BEHIND the barrier by workgroups of other XCDs: zeroed with
    // agent-scope stores, which go through to the memory side - a plain store would sit dirty in this XCD's L2 until a release
    // wrote the whole L2 back, and 472 such releases made this kernel 80 us long)
    // (only pass 0's rows - the first RADIX_MAX_GROUPS x RADIX_BINS words - are touched inside this launch; the later passes' rows are
    // read by later kernels and take plain stores)
    for (uint32_t w = t; w < (uint32_t)RADIX_TOTAL_WORDS; w += stride) {
        if (w < (uint32_t)(RADIX_MAX_GROUPS * RADIX_BINS)) __hip_atomic_store(&p.digit_total[w], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else p.digit_total[w] = 0u;
    }
    if (t < SORT_SHARDS) {
        p.next_frame->key_min[t] = 2147483640;
        p.next_frame->key_max[t] = -2147483640;
    }
    if (t == 0) {
        p.next_frame->clamped = 0;
        p.next_frame->kept = 0;
    }
    for (uint32_t k = tid; k < 4u * RADIX_BINS; k += HIST_THREADS) (&s_hist[0][0])[k] = 0;
    const uint32_t R = p.render_count;
    const RadixChunk ch = radix_chunk(R);
    // logical element j of the radix passes is list position R - 1 - j: this chunk's elements [j0, j1) are positions (R - j1, R - j0]
    const uint32_t j0 = min(ch.tile_begin * (uint32_t)RADIX_TILE, R), j1 = min(ch.tile_end * (uint32_t)RADIX_TILE, R);
    const uint32_t v0 = (R - j1) / 4u, v1 = (R - j0) / 4u;                  // vectors of four positions (R, j0, j1 are multiples of 4)
    constexpr uint32_t KV = (uint32_t)CHUNK_TILES * RADIX_TILE / (HIST_THREADS * 4u);
    static_assert(KV * HIST_THREADS * 4u == (uint32_t)CHUNK_TILES * RADIX_TILE, "whole vectors per thread");
    const uint4* x4 = reinterpret_cast<const uint4*>(p.cx);
    const uint4* y4 = reinterpret_cast<const uint4*>(p.cy);
    const uint4* z4 = reinterpret_cast<const uint4*>(p.cz);
    int4* o4 = reinterpret_cast<int4*>(p.keys_out);
    const uint32_t m0 = (uint32_t)p.im0, m1 = (uint32_t)p.im1, m2 = (uint32_t)p.im2;
    int4 key[KV];
    int32_t lo = 2147483640, hi = -2147483640;
    uint4 x[KV], y[KV], z[KV];
#pragma unroll
    for (uint32_t k = 0; k < KV; k++) {
        const uint32_t v = min(v0 + k * HIST_THREADS + tid, v1 ? v1 - 1u : 0u);     // (unconditional loads, clamped)
        x[k] = x4[v]; y[k] = y4[v]; z[k] = z4[v];
    }
#pragma unroll
    for (uint32_t k = 0; k < KV; k++) {
        const uint32_t v = v0 + k * HIST_THREADS + tid;
        key[k].x = (int32_t)(x[k].x * m0 + y[k].x * m1 + z[k].x * m2);
        key[k].y = (int32_t)(x[k].y * m0 + y[k].y * m1 + z[k].y * m2);
        key[k].z = (int32_t)(x[k].z * m0 + y[k].z * m1 + z[k].z * m2);
        key[k].w = (int32_t)(x[k].w * m0 + y[k].w * m1 + z[k].w * m2);
        if (v < v1) {
            o4[v] = key[k];
            lo = min(min(lo, key[k].x), min(min(key[k].y, key[k].z), key[k].w));
            hi = max(max(hi, key[k].x), max(max(key[k].y, key[k].z), key[k].w));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = min(lo, __shfl_xor(lo, o, 64));
        hi = max(hi, __shfl_xor(hi, o, 64));
    }
    if (lane == 0) { s_lo[wave] = lo; s_hi[wave] = hi; }
    __syncthreads();
    if (tid == 0) {
        for (uint32_t w = 1; w < HIST_THREADS / 64; w++) { lo = min(lo, s_lo[w]); hi = max(hi, s_hi[w]); }
        if (v1 > v0) {
            atomicMin(&p.frame->key_min[blockIdx.x % SORT_SHARDS], lo);
            atomicMax(&p.frame->key_max[blockIdx.x % SORT_SHARDS], hi);
        }
        // The barrier across the grid.  What the other workgroups need from this one are agent-scope atomics and stores (min / max,
        // the zeroed group rows): performed at the memory side, so "complete" is "visible" - the arrival only has to be issued after
        // they are complete (a workgroup-scope release = s_waitcnt), not after an L2 write-back.  The keys and the histogram rows are
        // for the NEXT kernel.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __hip_atomic_fetch_add(sync.arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t spins = 0;
        // (relaxed polls: an acquiring load per poll invalidates the caches the workgroups that are still keying read through)
        while (__hip_atomic_load(sync.arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - sync.base < gridDim.x) {
            __builtin_amdgcn_s_sleep(KEY_HIST_SLEEP);
            if (++spins > 4000000u) { atomicAdd(sync.fail, 1u); break; }            // (seconds: the grid was not resident at once)
        }
        // (what is read behind the barrier is read with agent-scope atomic loads and updated with agent-scope atomics: no acquire)
    }
    __syncthreads();
    // phase 2: the final range (fresh loads: the words were written by other workgroups' atomics)
    {
        const uint32_t l = tid & (SORT_SHARDS - 1u);
        int32_t glo = __hip_atomic_load(&ld.frame->key_min[l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int32_t ghi = __hip_atomic_load(&ld.frame->key_max[l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int o = SORT_SHARDS / 2; o > 0; o >>= 1) {
            glo = min(glo, __shfl_xor(glo, o, 64));
            ghi = max(ghi, __shfl_xor(ghi, o, 64));
        }
        ld.range_map = __fdiv_rn((float)(ld.range - 1), __fsub_rn((float)ghi, (float)glo));
        ld.lo = __builtin_amdgcn_readfirstlane(glo);
        ld.range_map = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(ld.range_map)));
    }
    uint32_t* hist = s_hist[wave & 3u];
#pragma unroll
    for (uint32_t k = 0; k < KV; k++) {
        if (v0 + k * HIST_THREADS + tid < v1) {
            atomicAdd(&hist[(ld.hist_key(key[k].x) >> shift) & 255u], 1u);
            atomicAdd(&hist[(ld.hist_key(key[k].y) >> shift) & 255u], 1u);
            atomicAdd(&hist[(ld.hist_key(key[k].z) >> shift) & 255u], 1u);
            atomicAdd(&hist[(ld.hist_key(key[k].w) >> shift) & 255u], 1u);
        }
    }
    __syncthreads();
    if (tid >= RADIX_BINS || ch.tile_begin >= ch.tile_end) return;
    const uint32_t total = s_hist[0][tid] + s_hist[1][tid] + s_hist[2][tid] + s_hist[3][tid];
    block_hist[ch.id * RADIX_BINS + tid] = total;
    if (total) atomicAdd(&digit_total[(ch.id / RADIX_GROUP) * RADIX_BINS + tid], total);
}

__global__ void k_debug_buckets(DepthLoader ld, int32_t* out) {
    ld.prepare();
    for (uint32_t i = ld.sort_start + blockIdx.x * blockDim.x + threadIdx.x; i < ld.render_count;
         i += gridDim.x * blockDim.x)
        out[i] = (int32_t)ld.bucket(i);
}

// Pass 0 of a depth sort for one (CULL, IDX) combination of the loader: packed + chunk-staged, packed, or key + value output.
struct Pass0Args {
    DepthLoader base;                   // the fields every variant shares (copied member by member below)
    const unsigned long long* keep;
    uint32_t Rs, val_bits;
    int shift;
    bool pack, chunked, wide;
    void* kbuf0;
    uint32_t* vo;
    uint32_t* kept_out;
    bool skip_hist;                     // k_depth_key_hist already left pass 0's histogram rows
};
template <bool CULL, bool IDX>
static int depth_pass0(const RadixExec& ex, const Pass0Args& a) {
    typedef DepthLoaderT<CULL, IDX> L;
    L d = {};
    d.keys = a.base.keys; d.keep = a.keep; d.idx = a.base.idx; d.map = a.base.map; d.frame = a.base.frame; d.n_dev = a.base.n_dev;
    d.sort_start = a.base.sort_start; d.render_count = a.base.render_count; d.range = a.base.range; d.last_splat = a.base.last_splat;
    L h = d;                            // only the histogram launch counts clamped buckets (once per element)
    h.count_clamps = 1;
    if (a.pack && a.chunked) return radix_pass_chunk<L, L, true>(ex, h, a.shift, d, a.Rs, a.shift, 0, a.vo, a.val_bits, a.kept_out, a.skip_hist);
    if (a.pack) return radix_pass_ex<L, L, uint8_t, false, false, true>(ex, h, a.shift, d, a.Rs, a.shift, 0, (uint8_t*)nullptr, a.vo, nullptr, 0u, a.val_bits, a.kept_out);
    if (a.wide) return radix_pass_ex<L, L, uint32_t, true, false, false>(ex, h, a.shift, d, a.Rs, a.shift, 0, (uint32_t*)a.kbuf0, a.vo, nullptr, 0u, 0u, a.kept_out);
    return radix_pass_ex<L, L, uint16_t, true, false, false>(ex, h, a.shift, d, a.Rs, a.shift, 0, (uint16_t*)a.kbuf0, a.vo, nullptr, 0u, 0u, a.kept_out);
}

static inline uint32_t grid_for(uint32_t n, uint32_t per_block, uint32_t cap) {
    uint32_t g = (n + per_block - 1) / per_block;
    if (g < 1) g = 1;
    return g > cap ? cap : g;
}

#ifdef GS_RADIX_PROFILE
extern "C" int gs_debug_radix_prof(void* dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_radix_prof), sizeof(g_radix_prof), 0, hipMemcpyDeviceToHost);
}
#endif

extern "C" {

int gs_sorter_create(gs_context* ctx, uint32_t max_splat_count, uint32_t flags, uint32_t precision_bits,
                     gs_sorter** out) {
    GS_REQUIRE(ctx && out, "ctx / out == NULL");
    *out = nullptr;
    GS_REQUIRE(max_splat_count > 0, "max_splat_count == 0");
    GS_REQUIRE((flags & ~(GS_SORT_INTEGER | GS_SORT_DYNAMIC)) == 0, "unknown sorter flags");
    const uint32_t max_bits = (flags & GS_SORT_INTEGER) ? 20u : 24u;   // src/Viewer.js:208-210
    GS_REQUIRE(precision_bits >= 10 && precision_bits <= max_bits, "precision outside the Viewer's clamp (10..20 int, 10..24 float)");
    ScopedDevice sd(ctx->device);
    gs_sorter* s = new (std::nothrow) gs_sorter();
    if (!s) return GS_ERR_NOMEM;
    s->ctx = ctx;
    s->max_count = max_splat_count;
    s->flags = flags;
    s->precision = precision_bits;
    const size_t n = max_splat_count, b4 = n * 4;
    int st = GS_OK;
    auto A = [&](DevBuf& b, size_t bytes) { if (st == GS_OK) st = b.alloc(bytes); };
    A(s->cx, b4 + 16); A(s->cy, b4 + 16); A(s->cz, b4 + 16); A(s->caos, n * 16);   // (+16: k_cull_front reads whole 16-byte vectors)
    if (flags & GS_SORT_DYNAMIC) { A(s->cw, b4); A(s->scene_idx, b4); A(s->scene_rows, sizeof(SceneRows)); }
    A(s->keys, b4); A(s->keyA, b4); A(s->keyB, b4); A(s->valA, b4); A(s->valB, b4); A(s->sorted, b4);
    A(s->frame, 2 * sizeof(SortFrame));
    if (st == GS_OK) st = s->radix.init();
    if (st == GS_OK && (hipEventCreate(&s->ev0) != hipSuccess || hipEventCreate(&s->ev1) != hipSuccess ||
                        hipEventCreateWithFlags(&s->ev_consumed, hipEventDisableTiming) != hipSuccess)) {
        gs_set_error("hipEventCreate failed");
        st = GS_ERR_HIP;
    }
    if (st == GS_OK) {
        SortFrame init[2];
        for (SortFrame& f : init) {
            for (uint32_t k = 0; k < SORT_SHARDS; k++) { f.key_min[k] = 2147483640; f.key_max[k] = -2147483640; }
            f.clamped = f.kept = 0;
        }
        if (hipMemcpy(s->frame.p, init, sizeof(init), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemset(s->radix.digit_total.p, 0, sizeof(uint32_t) * RADIX_TOTAL_WORDS) != hipSuccess) {
            gs_set_error("initialising the sorter scratch failed");
            st = GS_ERR_HIP;
        }
    }
    if (st == GS_OK) {
        if (ctx->serial) {
            s->stream = ctx->stream;
        } else if (hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) == hipSuccess) {
            s->own_stream = true;
        } else {
            gs_set_error("hipStreamCreate(sorter) failed");
            st = GS_ERR_HIP;
        }
    }
    if (st != GS_OK) {
        gs_sorter_destroy(s);
        return st;
    }
    *out = s;
    return GS_OK;
}

void gs_sorter_destroy(gs_sorter* s) {
    if (!s) return;
    if (s->pending_tree) gs_tree_forget_sorter(s->pending_tree, s);
    ScopedDevice sd(s->ctx->device);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    (void)hipStreamSynchronize(s->ctx->stream);        // a draw may still be reading `sorted`
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    if (s->ev_consumed) (void)hipEventDestroy(s->ev_consumed);
    if (s->own_stream) (void)hipStreamDestroy(s->stream);
    delete s;
}

static void sorter_tell_mesh(gs_sorter* s);   // (defined with gs_sorter_bind_mesh)
int gs_sorter_upload_centers(gs_sorter* s, uint32_t from, uint32_t count, const void* centers_aos4,
                             const uint32_t* scene_indexes) {
    GS_REQUIRE(s && centers_aos4, "sorter / centers == NULL");
    GS_REQUIRE((uint64_t)from + count <= s->max_count, "range exceeds max_splat_count");
    GS_REQUIRE(!(s->flags & GS_SORT_DYNAMIC) || scene_indexes, "dynamic sorter needs scene_indexes");
    if (count == 0) return GS_OK;
    ScopedDevice sd(s->ctx->device);
    hipStream_t st = s->stream;
    uint4* aos = s->caos.as<uint4>() + from;               // kept: index-list sorts gather 16 bytes per splat from it
    GS_HIP(hipMemcpyAsync(aos, centers_aos4, (size_t)count * 16, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_aos4_to_soa, dim3(grid_for(count, 256, 4096)), dim3(256), 0, st, aos, count,
                       from, s->cx.as<uint32_t>(), s->cy.as<uint32_t>(), s->cz.as<uint32_t>(),
                       (s->flags & GS_SORT_DYNAMIC) ? s->cw.as<uint32_t>() : nullptr);
    GS_HIP(hipGetLastError());
    if (s->flags & GS_SORT_DYNAMIC)
        GS_HIP(hipMemcpyAsync(s->scene_idx.as<uint32_t>() + from, scene_indexes, (size_t)count * 4, hipMemcpyHostToDevice, st));
    GS_HIP(hipStreamSynchronize(st));   // staging and the caller's buffers are reusable on return
    if (from + count > s->uploaded) s->uploaded = from + count;   // uploadedSplatCount, SortWorker.js:97
    s->centers_version++;
    sorter_tell_mesh(s);                                   // (more centres than the bound mesh holds: no position map, no derived mask)
    return GS_OK;
}

static int sorter_collect_stats(gs_sorter* s, gs_sort_stats* stats) {
    SortFrame f;
    GS_HIP(hipMemcpyAsync(&f, s->frame.as<SortFrame>() + s->frame_index, sizeof(f), hipMemcpyDeviceToHost, s->stream));
    GS_HIP(hipStreamSynchronize(s->stream));
    if (s->key_sync_used) {                        // k_depth_key_hist: a workgroup gave up waiting for the rest of its grid
        uint32_t w[2] = {0u, 0u};
        GS_HIP(hipMemcpyAsync(w, s->key_sync.p, sizeof(w), hipMemcpyDeviceToHost, s->stream));
        GS_HIP(hipStreamSynchronize(s->stream));
        if (w[1]) {
            GS_HIP(hipMemsetAsync((char*)s->key_sync.p + 4, 0, 4, s->stream));       // (reported once)
            gs_set_error("the key kernel's barrier across its grid timed out (k_depth_key_hist): the sorted list is not valid");
            return GS_ERR_HIP;
        }
    }
    float ms = 0.f;
    if (s->timed_sort) GS_HIP(hipEventElapsedTime(&ms, s->ev0, s->ev1));
    int32_t key_lo = f.lo(), key_hi = f.hi();
    if (s->last_sort == 0) {                       // nothing was keyed: the frame belongs to an earlier sort
        key_lo = 2147483640;
        key_hi = -2147483640;
        f.clamped = 0;
    }
    stats->device_ms = ms;
    stats->key_min = key_lo;
    stats->key_max = key_hi;
    stats->clamped = f.clamped;
    stats->passes = s->last_passes;
    stats->result_count = s->last_culled ? f.kept : s->last_render;
    return f.clamped ? GS_WARN_KEY_CLAMPED : GS_OK;
}

static int sorter_sort_impl(gs_sorter* s, const float* mvp, const uint32_t* indexes_to_sort, bool device_list,
                            uint32_t sort_count, uint32_t render_count, const void* precomputed, const float* transforms,
                            uint32_t* sorted_out, gs_sort_stats* stats, const uint32_t* list_count_dev = nullptr) {
    GS_REQUIRE(s && mvp, "sorter / mvp == NULL");
    // SortWorker.js:100-101 clamps both counts to the uploaded splat count
    if (render_count > s->uploaded) render_count = s->uploaded;
    if (sort_count > s->uploaded) sort_count = s->uploaded;
    GS_REQUIRE(sort_count <= render_count, "splatSortCount > splatRenderCount");
    const bool dynamic = (s->flags & GS_SORT_DYNAMIC) != 0;
    GS_REQUIRE(!dynamic || transforms, "dynamic sorter needs transforms");
    const bool vis_cull = s->visibility_cull && !list_count_dev;   // (sorts the identity list: not combinable with a gathered one)
    const bool cull = s->frustum_cull && !vis_cull;        // the frustum cull's keep-mask path; the visibility cull compacts instead
    // list_count_dev: the list's real length lives on the device (asynchronous gs_tree_gather); render_count bounds it
    GS_REQUIRE(!list_count_dev || (device_list && sort_count == render_count && !dynamic && !precomputed),
               "a gathered list whose length lives on the device needs a full sort of a static scene without precomputed distances");
    GS_REQUIRE(!(cull || vis_cull) || (sort_count == render_count && !dynamic && !precomputed),
               "a per-splat cull needs a full sort (splatSortCount == splatRenderCount) of a static scene without precomputed distances");
    gs_context* ctx = s->ctx;
    ScopedDevice sd(ctx->device);
    hipStream_t st = s->stream;
    const uint32_t R = render_count, Rs = sort_count, sort_start = R - Rs;
    const RadixExec ex = {st, &s->radix, ctx->lds_atomic_lane_order};
    // a bound mesh stores its splats in its own (Morton) order: hand it positions in that order
    if (s->bound_mesh) {                           // the mesh may have been destroyed since the bind
        bool alive = false;
        for (gs_mesh* m : ctx->live_meshes) alive = alive || (m == s->bound_mesh);
        if (!alive) s->bound_mesh = nullptr;
    }
    const uint32_t* map = s->bound_mesh ? gs_mesh_payload_map(s->bound_mesh, s->uploaded) : nullptr;
    if (vis_cull) {
        GS_REQUIRE(s->bound_mesh && s->bound_mesh->projection_pending && s->bound_mesh->uploaded >= s->uploaded,
                   "the visibility cull needs gs_sorter_bind_mesh and a gs_mesh_project of this frame's camera before the sort");
        GS_REQUIRE(!indexes_to_sort && !device_list, "the visibility cull sorts the identity list (pass indexes_to_sort = NULL)");
        if (s->bound_mesh->ev_p1 && ctx->aux != st) GS_HIP(hipStreamWaitEvent(st, s->bound_mesh->ev_p1, 0));   // the mask it reads
    }
    const uint32_t* unmap = map ? gs_mesh_payload_unmap(s->bound_mesh) : nullptr;
    if (ctx->fork_join && st != ctx->stream) {
        // serial frames on a multi-stream context: this sort starts when everything the context's stream holds now (the previous
        // frame's draw) has finished, then runs beside this frame's vertex stage
        GS_HIP(hipEventRecord(ctx->ev_fork, ctx->stream));
        GS_HIP(hipStreamWaitEvent(st, ctx->ev_fork, 0));
    }
    if (s->consumer_pending) {           // a draw enqueued on ctx->stream still reads the previous result
        GS_HIP(hipStreamWaitEvent(st, s->ev_consumed, 0));
        s->consumer_pending = false;
    }

    const uint32_t* idx_dev = nullptr;
    bool fused_tree = false;                               // the gathered list is still to be copied, and this sort does it itself
    bool vis_front = false;                                // the visibility cull's front end wrote keys and payloads of the survivors itself
    if (device_list) {
        idx_dev = s->idx_in.as<uint32_t>();                // written by gs_tree_gather on this stream ...
        if (s->pending_tree) {                             // ... or only planned by it (tree.hip): the copy is ours
            static const bool no_fuse = getenv("GSPLAT_TREE_NO_FUSE") != nullptr;
            fused_tree = !no_fuse && Rs == R && Rs > 0 && !dynamic && !precomputed && !vis_cull && (!cull || s->pending_keep_zeroed);
            if (!fused_tree) {
                GS_TRY(gs_tree_copy_plain(s->pending_tree, s->idx_in.as<uint32_t>(), st));
                gs_tree_forget_sorter(s->pending_tree, s);
            }
        }
    } else {
        // A sort that does not consume the gathered list may destroy what a gather left for a later gs_sorter_sort_gathered
        // (ADVICE r04): a planned gather's fused copy ORs its keep bits into a mask the GATHER zeroed - a frustum-culled sort in
        // between rewrites that mask, so the copy falls back to the plain one + the ordinary key kernel; a list that was already
        // copied into idx_in is gone once a host list or a compaction overwrites it (the compaction: below, where the front end
        // of a visibility-culled sort is chosen - ADVICE r05: the streaming one writes pay_in and leaves idx_in alone).
        if (s->pending_tree) {
            if (cull) s->pending_keep_zeroed = false;
        } else if (indexes_to_sort) {
            s->has_gathered = false;
        }
    }
    if (!device_list && indexes_to_sort) {
        GS_TRY(s->idx_in.ensure((size_t)s->max_count * 4));
        if (R) GS_HIP(hipMemcpyAsync(s->idx_in.p, indexes_to_sort, (size_t)R * 4, hipMemcpyHostToDevice, st));
        idx_dev = s->idx_in.as<uint32_t>();
    }
    KeyParams kp = {};
    kp.mode = ((s->flags & GS_SORT_INTEGER) ? MODE_INT : 0) | (dynamic ? MODE_DYNAMIC : 0) | (precomputed ? MODE_PRECOMPUTED : 0);
    if (precomputed) {
        GS_TRY(s->precomputed.ensure((size_t)s->max_count * 4));
        GS_HIP(hipMemcpyAsync(s->precomputed.p, precomputed, (size_t)s->uploaded * 4, hipMemcpyHostToDevice, st));
        kp.precomputed = s->precomputed.as<uint32_t>();
    }
    if (dynamic && !precomputed) {
        // computeMatMul4x4ThirdRow (sorter.cpp:11-15), fp32, left-to-right, no contraction; then x1000 in double
        SceneRows rows;
        for (uint32_t sc = 0; sc < GS_MAX_SCENES; sc++) {
            const float* b = transforms + 16 * sc;
            for (int c = 0; c < 4; c++) {
                volatile float acc = mvp[2] * b[4 * c + 0];
                acc = acc + mvp[6] * b[4 * c + 1];
                acc = acc + mvp[10] * b[4 * c + 2];
                acc = acc + mvp[14] * b[4 * c + 3];
                rows.fm[sc][c] = acc;
                rows.im[sc][c] = trunc_f64_i32((double)rows.fm[sc][c] * 1000.0);
            }
        }
        k_store_scene_rows<<<1, 256, 0, st>>>(rows, s->scene_rows.as<SceneRows>());
        GS_HIP(hipGetLastError());
        kp.rows = s->scene_rows.as<SceneRows>();
    }
    kp.cx = s->cx.as<uint32_t>(); kp.cy = s->cy.as<uint32_t>(); kp.cz = s->cz.as<uint32_t>();
    kp.cw = s->cw.as<uint32_t>(); kp.scene_idx = s->scene_idx.as<uint32_t>();
    kp.aos = s->caos.as<uint4>();
    kp.idx_in = idx_dev;
    kp.keys_out = s->keys.as<int32_t>();
    if (Rs > 0) s->frame_index ^= 1u;          // this sort's frame was reset by the previous sort's k_depth_key (or at create)
    kp.frame = s->frame.as<SortFrame>() + s->frame_index;
    kp.next_frame = s->frame.as<SortFrame>() + (s->frame_index ^ 1u);
    kp.digit_total = s->radix.digit_total.as<uint32_t>();
    kp.sort_start = sort_start;
    kp.render_count = R;
    kp.last_splat = s->uploaded ? s->uploaded - 1u : 0u;
    kp.im0 = trunc_f64_i32((double)mvp[2] * 1000.0);     // sorter.cpp:64
    kp.im1 = trunc_f64_i32((double)mvp[6] * 1000.0);
    kp.im2 = trunc_f64_i32((double)mvp[10] * 1000.0);
    kp.fm0 = mvp[2]; kp.fm1 = mvp[6]; kp.fm2 = mvp[10];

    if (cull) {
        GS_TRY(s->keep_mask.ensure((((size_t)s->max_count + 63) / 64 + 8) * 8));   // the key kernel writes whole 256-position windows
        kp.keep = s->keep_mask.as<unsigned long long>();
        memcpy(kp.mvp, mvp, sizeof(kp.mvp));
    }
    if (list_count_dev) kp.count_dev = list_count_dev;

    s->timed_sort = stats != nullptr || ctx->stage_events;
    if (s->timed_sort) GS_HIP(hipEventRecord(s->ev0, st));
    uint32_t passes = 0;
    if (Rs > 0) {
        const bool vec4 = (kp.mode == MODE_INT) && !idx_dev;
        // $GSPLAT_KEY_HIST_FUSED=1: keys + pass 0's histogram in one launch behind a barrier across the grid (k_depth_key_hist) - only for
        // the plain full sort of the identity list whose pass 0 is the packed, chunk-staged one, and only when the whole grid is
        // resident at once (<= 2 workgroups of 1024 threads per CU)
        uint32_t kh_grid = 0;
        bool key_hist_fused = false;
        if (getenv("GSPLAT_KEY_HIST_FUSED") && vec4 && !fused_tree && !vis_cull && !cull && sort_start == 0u && Rs == R && (R & 3u) == 0u &&
            !list_count_dev && !getenv("GSPLAT_NO_SORT_PACK") && !getenv("GSPLAT_NO_SORT_CHUNK") && s->precision > 8u) {
            uint32_t mp = kp.last_splat;
            if (map && s->bound_mesh->uploaded && s->bound_mesh->uploaded - 1u > mp) mp = s->bound_mesh->uploaded - 1u;
            uint32_t vb = 1;
            while (vb < 32u && (mp >> vb)) vb++;
            kh_grid = radix_chunk_grid_for(Rs);
            key_hist_fused = vb <= 24u && (s->precision - 8u) + vb <= 32u && kh_grid != 0u && kh_grid <= 2u * (uint32_t)ctx->cu_count;
            if (key_hist_fused) {
                if (!s->key_sync.p) {
                    GS_TRY(s->key_sync.alloc(64));
                    GS_HIP(hipMemsetAsync(s->key_sync.p, 0, 64, st));
                    s->key_sync_base = 0u;
                }
            }
        }
        if (fused_tree) {
            // copy + keys (+ keep bits) of the planned gather in one kernel over leaf-major copies of the centres and payloads
            TreeCopyParams cp;
            gs_tree_view(s->pending_tree, &cp.v);
            const void* mesh_key = map ? (const void*)s->bound_mesh : nullptr;
            const uint32_t layout = map ? s->bound_mesh->layout_version : 0u;
            GS_TRY(s->pay_in.ensure((size_t)s->max_count * 4));
            if (s->leaf_cache_tree != cp.v.tree_uid || s->leaf_cache_centers != s->centers_version || s->leaf_cache_mesh != mesh_key ||
                s->leaf_cache_layout != layout || s->leaf_cache_uploaded != s->uploaded || !s->leaf_centers.p) {
                GS_TRY(s->leaf_centers.ensure((size_t)cp.v.tree_splats * 16 + 16));
                GS_TRY(s->leaf_pos.ensure((size_t)cp.v.tree_splats * 4 + 16));
                if (cp.v.tree_splats)
                    hipLaunchKernelGGL(k_tree_leaf_cache, dim3(grid_for(cp.v.tree_splats, 1024, 4096)), dim3(256), 0, st, cp.v.leaf_indexes,
                                       cp.v.tree_splats, s->caos.as<uint4>(), map, kp.last_splat, s->leaf_centers.as<uint4>(),
                                       s->leaf_pos.as<uint32_t>());
                s->leaf_cache_tree = cp.v.tree_uid; s->leaf_cache_centers = s->centers_version; s->leaf_cache_mesh = mesh_key;
                s->leaf_cache_layout = layout; s->leaf_cache_uploaded = s->uploaded;
            }
            cp.leaf_centers = s->leaf_centers.as<uint4>();
            cp.leaf_pos = s->leaf_pos.as<uint32_t>();
            cp.idx_out = s->idx_in.as<uint32_t>();
            cp.pay_out = s->pay_in.as<uint32_t>();
            // (a workgroup ends with one atomic on the sort's kept counter, and same-address atomics retire at ~12 ns each:
            // 6.5 k workgroups queue ~80 us of them behind 30 us of memory traffic - eight workgroups per CU walk the leaves)
            uint32_t cgrid = (cp.v.leaves + 3u) / 4u;
            cgrid = cgrid < 1u ? 1u : (cgrid > (uint32_t)ctx->cu_count * 8u ? (uint32_t)ctx->cu_count * 8u : cgrid);
            if (cull) hipLaunchKernelGGL(k_tree_copy_keys<true>, dim3(cgrid), dim3(256), 0, st, cp, kp);
            else hipLaunchKernelGGL(k_tree_copy_keys<false>, dim3(cgrid), dim3(256), 0, st, cp, kp);
            gs_tree_forget_sorter(s->pending_tree, s);
        } else if (vis_cull) {
            // compact, then sort the survivors (see k_minmax_count): the list is what the bound mesh's vertex stage kept
            const uint32_t spans = (R + VC_SPAN - 1u) / VC_SPAN;
            const uint32_t grid = spans < (uint32_t)ctx->cu_count * 2u ? spans : (uint32_t)ctx->cu_count * 2u;
            const uint32_t chunk_len = ((spans + grid - 1u) / grid) * VC_SPAN;
            GS_TRY(s->chunk_counts.ensure((size_t)grid * 4));
            GS_TRY(s->idx_in.ensure((size_t)s->max_count * 4));
            GS_TRY(s->mask_copy.ensure(((size_t)s->max_count + 31) / 32 * 4 + 64));
            uint32_t* mask = s->bound_mesh->vis_orig.as<uint32_t>();
            // The projection left the by-original-index mask to us (gs_mesh_project of a full frame for a mesh whose bound sorter
            // culls by visibility): k_mask_derive_count builds it from the storage-order mask through the payload map.
            const bool lazy = s->bound_mesh->vis_orig_lazy;
            GS_REQUIRE(!lazy || map, "the pending projection expects the sorter to derive its visibility mask, but the sorter has no position map of the mesh");
            const uint32_t mblocks = (s->bound_mesh->uploaded + 255u) >> 8;
            constexpr uint32_t DERIVE_SUBS = 4;
            auto derive = [&](uint32_t dgrid, uint32_t dlen) {
                hipLaunchKernelGGL(k_mask_derive_count, dim3(dgrid * DERIVE_SUBS), dim3(VC_THREADS), 0, st, kp, map,
                                   s->bound_mesh->vis_mask.as<unsigned long long>(), s->bound_mesh->block_any.as<uint8_t>(), mblocks, dlen,
                                   DERIVE_SUBS, reinterpret_cast<unsigned long long*>(mask), s->chunk_counts.as<uint32_t>());
            };
            // Two front ends, chosen by what the vertex stage drew.  A full frame keeps a large part of the scene (C3: a quarter):
            // the streaming front end keys the survivors on the way (N = 1 frame 0.285 -> 0.270 ms, r05k).  A rank's strip keeps
            // a few per cent: round 4's chain - a 12-byte-per-splat min / max pass, a compaction of the mask, and centre gathers
            // for the few survivors - is the cheaper one there (C3 rank 4 of 8: 0.116 vs 0.126 ms, same box), because the streaming
            // kernel also reads the payload table (+4 bytes per splat) and runs at 2.8 TB/s where the pure reduction reaches 4.
            // $GSPLAT_VIS_FRONT = stream | compact forces one (A/B and tests).
            static const char* force = getenv("GSPLAT_VIS_FRONT");
            // (projected_cam is the camera of the gs_mesh_project this sort consumes: the entry checks refused the call unless
            // the bound mesh's projection_pending is set)
            const gs_camera& pc = s->bound_mesh->projected_cam;
            const uint32_t rows_total = (pc.height + GS_TILE - 1u) / GS_TILE;
            const bool strip = !(pc.tile_row_begin == 0u && (pc.tile_row_end == 0u || pc.tile_row_end >= rows_total));
            vis_front = force ? force[0] == 's' : !strip;
            if (vis_front) {
                // one contiguous run of positions per workgroup, walked in turns of 4096 (two workgroups per CU)
                // (a chunk is a whole number of turns; with the mask derived here, of DERIVE_SUBS x turns: one piece per sub-workgroup)
                const uint32_t tlen = lazy ? VC_TURN * DERIVE_SUBS : VC_TURN;
                const uint32_t turns = (R + tlen - 1u) / tlen;
                const uint32_t wgrid = turns < (uint32_t)ctx->cu_count * 2u ? turns : (uint32_t)ctx->cu_count * 2u;
                const uint32_t wlen = ((turns + wgrid - 1u) / wgrid) * tlen;
                GS_TRY(s->chunk_counts.ensure((size_t)wgrid * DERIVE_SUBS * 4));
                GS_TRY(s->pay_in.ensure((size_t)s->max_count * 4));
                if (lazy) derive(wgrid, wlen);
                else hipLaunchKernelGGL(k_mask_count, dim3(wgrid), dim3(VC_THREADS), 0, st, kp, mask, wlen, s->chunk_counts.as<uint32_t>());
                if (map) hipLaunchKernelGGL(k_cull_front<true>, dim3(wgrid), dim3(VC_THREADS), 0, st, kp, mask, s->mask_copy.as<uint32_t>(),
                                            s->chunk_counts.as<uint32_t>(), wlen, map, s->pay_in.as<uint32_t>(), lazy ? DERIVE_SUBS : 1u);
                else hipLaunchKernelGGL(k_cull_front<false>, dim3(wgrid), dim3(VC_THREADS), 0, st, kp, mask, s->mask_copy.as<uint32_t>(),
                                        s->chunk_counts.as<uint32_t>(), wlen, map, s->pay_in.as<uint32_t>(), 1u);
            } else {
                if (!s->pending_tree) s->has_gathered = false;     // k_mask_compact overwrites a gathered list that was already copied into idx_in
                if (lazy) {                                        // (a forced compact front end: the mask first, its counts are redone below)
                    const uint32_t tlen = VC_TURN * DERIVE_SUBS, turns = (R + tlen - 1u) / tlen;
                    const uint32_t dgrid = turns < (uint32_t)ctx->cu_count * 2u ? turns : (uint32_t)ctx->cu_count * 2u;
                    GS_TRY(s->chunk_counts.ensure((size_t)std::max(dgrid * DERIVE_SUBS, grid) * 4));
                    derive(dgrid, ((turns + dgrid - 1u) / dgrid) * tlen);
                }
                hipLaunchKernelGGL(k_minmax_count, dim3(grid), dim3(VC_THREADS), 0, st, kp, mask, chunk_len, s->chunk_counts.as<uint32_t>());
                hipLaunchKernelGGL(k_mask_compact, dim3(grid), dim3(VC_THREADS), 0, st, mask, s->mask_copy.as<uint32_t>(),
                                   s->chunk_counts.as<uint32_t>(), R, chunk_len, s->idx_in.as<uint32_t>(), kp.frame);
            }
            // the mask is all zero again if this sort covered every splat the vertex stage looked at
            if (R >= s->bound_mesh->vis_orig_count) s->bound_mesh->vis_orig_dirty = false;
            idx_dev = s->idx_in.as<uint32_t>();
            kp.idx_in = idx_dev;
            kp.count_dev = &kp.frame->kept;
            kp.ext_minmax = 1u;
            if (!vis_front) hipLaunchKernelGGL(k_depth_key<false>, dim3(grid_for(Rs, 256 * 4, (uint32_t)ctx->cu_count * 2)), dim3(256), 0, st, kp);
        } else if (cull && vec4)
            hipLaunchKernelGGL(k_depth_key_cull<true>, dim3(grid_for(Rs, 256 * 16, (uint32_t)ctx->cu_count * 2)), dim3(256), 0, st, kp);
        else if (cull)
            hipLaunchKernelGGL(k_depth_key_cull<false>, dim3(grid_for(Rs, 256 * 4, (uint32_t)ctx->cu_count * 4)), dim3(256), 0, st, kp);
        else if (vec4 && key_hist_fused) {
            DepthLoader hl = {};
            hl.keys = s->keys.as<int32_t>(); hl.frame = kp.frame; hl.sort_start = 0u; hl.render_count = R; hl.range = 1u << s->precision;
            hl.last_splat = kp.last_splat; hl.count_clamps = 1u;
            KeyHistSync sync = {s->key_sync.as<uint32_t>(), s->key_sync.as<uint32_t>() + 1, s->key_sync_base};
            s->key_sync_base += kh_grid;
            s->key_sync_used = true;
            hipLaunchKernelGGL(k_depth_key_hist, dim3(kh_grid), dim3(HIST_THREADS), 0, st, kp, hl, 0, s->radix.block_hist.as<uint32_t>(),
                               s->radix.digit_total.as<uint32_t>(), sync);
        } else if (vec4)
            hipLaunchKernelGGL(k_depth_key<true>, dim3(grid_for(Rs, 256 * 16, (uint32_t)ctx->cu_count * GS_KEY_GRID_MULT)), dim3(256), 0, st, kp);
        else
            hipLaunchKernelGGL(k_depth_key<false>, dim3(grid_for(Rs, 256 * 4, (uint32_t)ctx->cu_count * 2)), dim3(256), 0, st, kp);
        GS_HIP(hipGetLastError());
        DepthLoader dl = {};
        dl.keys = s->keys.as<int32_t>();
        dl.idx = (fused_tree || vis_front) ? s->pay_in.as<uint32_t>() : idx_dev;     // (the fused copy / the cull's front end wrote the payloads themselves)
        dl.map = (fused_tree || vis_front) ? nullptr : map;
        dl.frame = kp.frame;
        dl.sort_start = sort_start;
        dl.render_count = R;
        dl.range = 1u << s->precision;
        // The payload's range, not the list's: with a bound mesh a payload is perm[o], a position inside the mesh's slotted range,
        // which reaches mesh.uploaded - 1 even when fewer centres have reached this sorter (gs_mesh_payload_map allows
        // sorter.uploaded <= mesh.uploaded).  Sized from last_splat alone, the payload's top bits OR-ed into the key.
        uint32_t max_payload = kp.last_splat;
        if (map && s->bound_mesh->uploaded && s->bound_mesh->uploaded - 1u > max_payload) max_payload = s->bound_mesh->uploaded - 1u;
        // (the loader clamps what it reads from `idx`: splat indexes normally, ready-made payloads after a fused copy / front end)
        dl.last_splat = (fused_tree || vis_front) ? max_payload : kp.last_splat;
        dl.n_dev = vis_cull ? &kp.frame->kept : list_count_dev;
        passes = (s->precision + 7) / 8;
        uint32_t* out_tail = s->sorted.as<uint32_t>() + sort_start;
        void* kbuf[2] = {s->keyA.p, s->keyB.p};
        uint32_t* vbuf[2] = {s->valA.as<uint32_t>(), s->valB.as<uint32_t>()};
        // What travels between the passes: after pass p the key still holds precision - 8 (p + 1) bits.  When those fit above the
        // payload in one 32-bit word (payloads are splat positions < uploaded: 23 bits at 5.8 M splats, 24 at 16 M, so the default
        // 16-bit buckets always do below 2^24 splats) the pass writes that word instead of a key array and a value array
        // (radix.hpp PACK_OUT); otherwise 16- or 32-bit keys + values as before, and a later pass packs as soon as it can.
        uint32_t val_bits = 1;
        while (val_bits < 32u && (max_payload >> val_bits)) val_bits++;
        static const bool no_pack = getenv("GSPLAT_NO_SORT_PACK") != nullptr;       // A/B and tests: the unpacked path
        static const bool no_chunk = getenv("GSPLAT_NO_SORT_CHUNK") != nullptr;     // ... the tile-at-a-time scatter for packed passes
        // packed passes stage a whole chunk in LDS (radix.hpp) when the list is short enough for <= CHUNK_TILES tiles per
        // workgroup and the payload leaves 8 bits for the digit beside it in the last pass's staging word
        const bool chunked = !no_chunk && val_bits <= 24u && radix_chunk_grid_for(Rs) != 0u;
        const bool wide = s->precision > 16;
        bool in_packed = false;                                                       // the previous pass packed
        for (uint32_t p = 0; p < passes; p++) {
            const bool last = (p + 1 == passes);
            const int left = (int)s->precision - 8 * (int)(p + 1);                    // key bits after this pass
            const bool pack = !last && !no_pack && (uint32_t)left + val_bits <= 32u;
            const int shift = in_packed ? 0 : 8 * (int)p;
            uint32_t* vo = last ? out_tail : vbuf[p & 1];
            // after a culling pass 0 the element count is the device-resident kept count
            const uint32_t* n_dev = (cull || vis_cull || list_count_dev) ? &kp.frame->kept : nullptr;
            if (p == 0) {
                Pass0Args a = {};
                a.base = dl; a.keep = cull ? kp.keep : nullptr; a.Rs = Rs; a.val_bits = val_bits; a.shift = shift;
                a.pack = pack; a.chunked = chunked; a.wide = wide; a.kbuf0 = kbuf[0]; a.vo = vo;
                a.kept_out = cull ? &kp.frame->kept : nullptr;         // a culling pass 0 compacts: it publishes the result's length
                a.skip_hist = key_hist_fused;
                const bool has_idx = dl.idx != nullptr;
                if (cull && has_idx) GS_TRY((depth_pass0<true, true>(ex, a)));
                else if (cull) GS_TRY((depth_pass0<true, false>(ex, a)));
                else if (has_idx) GS_TRY((depth_pass0<false, true>(ex, a)));
                else GS_TRY((depth_pass0<false, false>(ex, a)));
            } else if (in_packed) {
                PackedLoader pl = {vbuf[(p - 1) & 1], n_dev, Rs, val_bits};
                ArrayLoader<uint32_t> ph = {vbuf[(p - 1) & 1], nullptr, n_dev, Rs};
                if (chunked && last) GS_TRY((radix_pass_chunk<ArrayLoader<uint32_t>, PackedLoader, false>(ex, ph, (int)val_bits, pl, Rs, 0, (int)p, vo, 0u)));
                else if (chunked) GS_TRY((radix_pass_chunk<ArrayLoader<uint32_t>, PackedLoader, true>(ex, ph, (int)val_bits, pl, Rs, 0, (int)p, vo, val_bits)));
                else if (last) GS_TRY((radix_pass_ex<ArrayLoader<uint32_t>, PackedLoader, uint8_t, false, false, false>(ex, ph, (int)val_bits, pl, Rs, 0, (int)p, (uint8_t*)nullptr, vo, nullptr, 0u, 0u)));
                else GS_TRY((radix_pass_ex<ArrayLoader<uint32_t>, PackedLoader, uint8_t, false, false, true>(ex, ph, (int)val_bits, pl, Rs, 0, (int)p, (uint8_t*)nullptr, vo, nullptr, 0u, val_bits)));
            } else if (wide) {
                ArrayLoader<uint32_t> al = {(const uint32_t*)kbuf[(p - 1) & 1], vbuf[(p - 1) & 1], n_dev, Rs};
                if (last) GS_TRY((radix_pass<ArrayLoader<uint32_t>, uint32_t, false>(ex, al, al, Rs, shift, (int)p, (uint32_t*)nullptr, vo)));
                else if (pack) GS_TRY((radix_pass_ex<ArrayLoader<uint32_t>, ArrayLoader<uint32_t>, uint8_t, false, false, true>(ex, al, shift, al, Rs, shift, (int)p, (uint8_t*)nullptr, vo, nullptr, 0u, val_bits)));
                else GS_TRY((radix_pass<ArrayLoader<uint32_t>, uint32_t, true>(ex, al, al, Rs, shift, (int)p, (uint32_t*)kbuf[p & 1], vo)));
            } else {
                ArrayLoader<uint16_t> al = {(const uint16_t*)kbuf[(p - 1) & 1], vbuf[(p - 1) & 1], n_dev, Rs};
                if (last) GS_TRY((radix_pass<ArrayLoader<uint16_t>, uint16_t, false>(ex, al, al, Rs, shift, (int)p, (uint16_t*)nullptr, vo)));
                else GS_TRY((radix_pass<ArrayLoader<uint16_t>, uint16_t, true>(ex, al, al, Rs, shift, (int)p, (uint16_t*)kbuf[p & 1], vo)));
            }
            in_packed = in_packed || pack;
        }
    }
    if (sort_start > 0) {
        hipLaunchKernelGGL(k_copy_head, dim3(grid_for(sort_start, 1024, 2048)), dim3(256), 0, st, idx_dev, map,
                           s->sorted.as<uint32_t>(), sort_start, s->uploaded - 1u);
    }
    GS_HIP(hipGetLastError());
    if (s->timed_sort || st != ctx->stream) GS_HIP(hipEventRecord(s->ev1, st));   // a draw on another stream waits for it
    s->last_render = R;
    s->last_sort = Rs;
    s->last_passes = passes;
    s->last_identity = (idx_dev == nullptr);
    s->last_culled = (cull || vis_cull || list_count_dev) && Rs > 0;   // the result's length lives in result_frame->kept
    s->last_vis_culled = vis_cull && Rs > 0;
    s->result_frame = kp.frame;
    s->result_mesh = map ? s->bound_mesh : nullptr;
    s->result_unmap = unmap;
    // (the clamp of the host-visible un-mapping: payloads of a bound mesh are positions in ITS storage, < mesh.uploaded)
    s->result_payload_max = (map && s->bound_mesh->uploaded && s->bound_mesh->uploaded - 1u > kp.last_splat) ? s->bound_mesh->uploaded - 1u : kp.last_splat;
    s->has_result = true;

    int status = GS_OK;
    uint32_t out_count = R;
    if (sorted_out && s->last_culled) {            // the result's length is only known on the device
        SortFrame f;
        GS_HIP(hipMemcpyAsync(&f, kp.frame, sizeof(f), hipMemcpyDeviceToHost, st));
        GS_HIP(hipStreamSynchronize(st));
        out_count = f.kept;
    }
    if (sorted_out && out_count) {
        const void* src = s->sorted.p;
        if (unmap) {                               // the host always sees the caller's splat indexes
            GS_TRY(s->debug.ensure((size_t)s->max_count * 4));
            hipLaunchKernelGGL(k_unmap, dim3(grid_for(out_count, 1024, 2048)), dim3(256), 0, st, s->sorted.as<uint32_t>(), unmap,
                               s->debug.as<uint32_t>(), out_count, s->result_payload_max);
            GS_HIP(hipGetLastError());
            src = s->debug.p;
        }
        GS_HIP(hipMemcpyAsync(sorted_out, src, (size_t)out_count * 4, hipMemcpyDeviceToHost, st));
        GS_HIP(hipStreamSynchronize(st));
    }
    if (stats) status = sorter_collect_stats(s, stats);
    return status;
}

int gs_sorter_sort(gs_sorter* s, const float* mvp, const uint32_t* indexes_to_sort, uint32_t sort_count,
                   uint32_t render_count, const void* precomputed, const float* transforms, uint32_t* sorted_out,
                   gs_sort_stats* stats) {
    return sorter_sort_impl(s, mvp, indexes_to_sort, false, sort_count, render_count, precomputed, transforms, sorted_out,
                            stats);
}

int gs_sorter_sort_gathered(gs_sorter* s, const float* mvp, uint32_t sort_count, const void* precomputed,
                            const float* transforms, uint32_t* sorted_out, gs_sort_stats* stats) {
    GS_REQUIRE(s != nullptr, "sorter == NULL");
    GS_REQUIRE(s->has_gathered, "no gs_tree_gather has filled this sorter's index list");
    if (s->gathered_on_device) {                                  // s->gathered = the tree's splat count, an upper bound
        GS_REQUIRE(sort_count >= s->gathered, "after an asynchronous gs_tree_gather only the whole list can be sorted: the "
                                              "partial-sort schedule needs splatRenderCount on the host (pass render_count)");
        return sorter_sort_impl(s, mvp, nullptr, true, s->gathered, s->gathered, precomputed, transforms, sorted_out, stats,
                                s->gathered_dev.as<uint32_t>());
    }
    if (sort_count > s->gathered) sort_count = s->gathered;       // Math.min(queuedSorts.shift(), splatRenderCount)
    return sorter_sort_impl(s, mvp, nullptr, true, sort_count, s->gathered, precomputed, transforms, sorted_out, stats);
}

// A bound sorter that culls by visibility and holds the mesh's position map derives the by-original-index mask itself
// (k_mask_derive_count): the mesh's next full-frame gs_mesh_project may then skip its per-survivor atomics.
static void sorter_tell_mesh(gs_sorter* s) {
    gs_mesh* m = s->bound_mesh;
    if (!m) return;
    bool alive = false;
    for (gs_mesh* l : s->ctx->live_meshes) alive = alive || l == m;
    if (alive) m->derive_orig_mask = s->visibility_cull && gs_mesh_payload_map(m, s->uploaded) != nullptr;
}

int gs_sorter_bind_mesh(gs_sorter* s, gs_mesh* m) {
    GS_REQUIRE(s != nullptr, "sorter == NULL");
    GS_REQUIRE(!m || m->ctx == s->ctx, "mesh lives on another context");
    if (s->bound_mesh && s->bound_mesh != m) {
        for (gs_mesh* l : s->ctx->live_meshes) if (l == s->bound_mesh) l->derive_orig_mask = false;
    }
    s->bound_mesh = m;
    sorter_tell_mesh(s);
    return GS_OK;
}

int gs_sorter_set_frustum_cull(gs_sorter* s, int enable) {
    GS_REQUIRE(s != nullptr, "sorter == NULL");
    GS_REQUIRE(!enable || !(s->flags & GS_SORT_DYNAMIC), "the per-splat frustum cull is not available to a dynamic-mode sorter");
    s->frustum_cull = enable != 0;
    return GS_OK;
}

int gs_sorter_set_visibility_cull(gs_sorter* s, int enable) {
    GS_REQUIRE(s != nullptr, "sorter == NULL");
    GS_REQUIRE(!enable || !(s->flags & GS_SORT_DYNAMIC), "the visibility cull is not available to a dynamic-mode sorter");
    s->visibility_cull = enable != 0;
    sorter_tell_mesh(s);
    return GS_OK;
}

int gs_sorter_last_stats(gs_sorter* s, gs_sort_stats* stats) {
    GS_REQUIRE(s && stats, "sorter / stats == NULL");
    GS_REQUIRE(s->has_result, "no sort has run");
    ScopedDevice sd(s->ctx->device);
    return sorter_collect_stats(s, stats);
}

int gs_sorter_debug_read(gs_sorter* s, int what, void* dst, uint32_t count) {
    GS_REQUIRE(s && dst, "sorter / dst == NULL");
    GS_REQUIRE(s->has_result, "no sort has run");
    GS_REQUIRE(count <= s->last_render, "count exceeds the last render_count");
    ScopedDevice sd(s->ctx->device);
    hipStream_t st = s->stream;
    const void* src = nullptr;
    if (what == 0) src = s->keys.p;
    else if (what == 2) {
        src = s->sorted.p;
        bool alive = false;
        for (gs_mesh* m : s->ctx->live_meshes) alive = alive || (m == s->result_mesh);
        if (alive && s->result_unmap && count) {
            GS_TRY(s->debug.ensure((size_t)s->max_count * 4));
            hipLaunchKernelGGL(k_unmap, dim3(grid_for(count, 1024, 2048)), dim3(256), 0, st, s->sorted.as<uint32_t>(),
                               s->result_unmap, s->debug.as<uint32_t>(), count, s->result_payload_max);
            GS_HIP(hipGetLastError());
            src = s->debug.p;
        }
    }
    else if (what == 1) {
        GS_TRY(s->debug.ensure((size_t)s->max_count * 4));
        DepthLoader dl = {};
        dl.keys = s->keys.as<int32_t>();
        dl.frame = s->frame.as<SortFrame>() + s->frame_index;
        dl.sort_start = s->last_render - s->last_sort;
        dl.render_count = s->last_render;
        dl.range = 1u << s->precision;
        GS_HIP(hipMemsetAsync(s->debug.p, 0, (size_t)s->last_render * 4, st));
        if (s->last_sort)
            hipLaunchKernelGGL(k_debug_buckets, dim3(grid_for(s->last_sort, 1024, 2048)), dim3(256), 0, st, dl, s->debug.as<int32_t>());
        GS_HIP(hipGetLastError());
        src = s->debug.p;
    } else if (what == 3) {
        GS_REQUIRE(s->last_culled, "the last sort did not cull");
        if (s->last_vis_culled) {                  // the bound mesh's per-splat mask (original splat numbering), as consumed
            GS_REQUIRE((size_t)count * 4 <= s->mask_copy.bytes, "count exceeds the mask length");
            src = s->mask_copy.p;
        } else {
            GS_REQUIRE((size_t)count * 4 <= s->keep_mask.bytes, "count exceeds the mask length");
            src = s->keep_mask.p;
        }
    } else {
        GS_REQUIRE(false, "unknown debug selector");
    }
    if (count) GS_HIP(hipMemcpyAsync(dst, src, (size_t)count * 4, hipMemcpyDeviceToHost, st));
    GS_HIP(hipStreamSynchronize(st));
    return GS_OK;
}

}  // extern "C"
