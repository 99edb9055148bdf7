// mesh.hip — the RENDER SEAM object: device-resident splat data + one draw =
//   k_project (project.hip) -> binning + tile sort (tile_bin.hip) -> k_tile_blend (tile_blend.hip).
// Replaces SplatMesh's data textures / uniforms / instanced draw
// (/root/reference/src/splatmesh/SplatMesh.js:637-898, 1228-1280; src/Viewer.js:1616).
#include <math.h>
#include <stdlib.h>

#include <algorithm>
#include <utility>

#include "radix.hpp"

void gs_set_error(const char* fmt, ...);

// Storage order.  Splats of one upload are re-ordered along a 30-bit Morton curve of their centres:
// `perm[original] = internal`.  Neighbours in memory are then neighbours in space, so whole waves of k_project fail
// the frustum test together and never fetch their covariance / SH bytes.  Every index that crosses the C ABI stays
// an ORIGINAL splat index; the binner translates once per list position.
#define DST(i) (perm ? perm[from + (i)] : from + (i))

__global__ __launch_bounds__(256) void k_morton_keys(const float* __restrict__ c3, uint32_t count, float3 mn, float3 inv_extent,
                                                     uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
        auto q = [](float v, float lo, float inv) {
            const float t = (v - lo) * inv * 1023.0f;
            return (uint32_t)(t < 0.0f ? 0.0f : (t > 1023.0f ? 1023.0f : t));        // NaN -> 0
        };
        auto spread = [](uint32_t v) {                                               // 10 bits -> every third bit
            v = (v | (v << 16)) & 0x030000FFu;
            v = (v | (v << 8)) & 0x0300F00Fu;
            v = (v | (v << 4)) & 0x030C30C3u;
            v = (v | (v << 2)) & 0x09249249u;
            return v;
        };
        const uint32_t x = q(c3[3 * (size_t)i], mn.x, inv_extent.x), y = q(c3[3 * (size_t)i + 1], mn.y, inv_extent.y),
                       z = q(c3[3 * (size_t)i + 2], mn.z, inv_extent.z);
        keys[i] = spread(x) | (spread(y) << 1) | (spread(z) << 2);
        vals[i] = i;
    }
}

__global__ __launch_bounds__(256) void k_perm_from_sorted(const uint32_t* __restrict__ sorted_local, uint32_t count, uint32_t from,
                                                          uint32_t* __restrict__ perm, uint32_t* __restrict__ inv_perm) {
    for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < count; p += gridDim.x * blockDim.x) {
        perm[from + sorted_local[p]] = from + p;
        inv_perm[from + p] = from + sorted_local[p];
    }
}

const uint32_t* gs_mesh_payload_map(gs_mesh* m, uint32_t splats) {
    return (m->reorder && splats <= m->uploaded) ? m->perm.as<uint32_t>() : nullptr;
}
const uint32_t* gs_mesh_payload_unmap(gs_mesh* m) { return m->reorder ? m->inv_perm.as<uint32_t>() : nullptr; }

__global__ __launch_bounds__(256) void k_scatter_u32(const uint32_t* __restrict__ src, uint32_t count, uint32_t from,
                                                     const uint32_t* __restrict__ perm, uint32_t* __restrict__ dst) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) dst[DST(i)] = src[i];
}

// AoS upload formats -> SoA planes -----------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_split_centers(const float* __restrict__ c3, uint32_t count, uint32_t from,
                                                       const uint32_t* __restrict__ perm, float* __restrict__ x,
                                                       float* __restrict__ y, float* __restrict__ z) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
        const uint32_t d = DST(i);
        x[d] = c3[3 * (size_t)i];
        y[d] = c3[3 * (size_t)i + 1];
        z[d] = c3[3 * (size_t)i + 2];
    }
}

// Gershgorin: the spectral radius of the symmetric [[c0 c1 c2] [c1 c3 c4] [c2 c4 c5]] is at most its largest absolute row sum
// (true for ANY values, PSD or not; NaN stays NaN and the pre-test that uses the bound then keeps the splat)
__device__ __forceinline__ float cov_spectral_bound(float c0, float c1, float c2, float c3, float c4, float c5) {
    const float r0 = fabsf(c0) + fabsf(c1) + fabsf(c2), r1 = fabsf(c1) + fabsf(c3) + fabsf(c4), r2 = fabsf(c2) + fabsf(c4) + fabsf(c5);
    const float m = fmaxf(r0, fmaxf(r1, r2));
    return (r0 != r0 || r1 != r1 || r2 != r2) ? NAN : m * 1.00001f;
}

__global__ __launch_bounds__(256) void k_split_cov_f32(const float* __restrict__ c6, uint32_t count, uint32_t from,
                                                       const uint32_t* __restrict__ perm, float4* __restrict__ a,
                                                       float2* __restrict__ b, float* __restrict__ bound) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
        const float* s = c6 + 6 * (size_t)i;
        const uint32_t d = DST(i);
        a[d] = make_float4(s[0], s[1], s[2], s[3]);
        b[d] = make_float2(s[4], s[5]);
        bound[d] = cov_spectral_bound(s[0], s[1], s[2], s[3], s[4], s[5]);
    }
}

__global__ __launch_bounds__(256) void k_split_cov_f16(const uint16_t* __restrict__ c6, uint32_t count, uint32_t from,
                                                       const uint32_t* __restrict__ perm, uint2* __restrict__ a,
                                                       uint32_t* __restrict__ b, float* __restrict__ bound) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
        const uint16_t* s = c6 + 6 * (size_t)i;
        const uint32_t d = DST(i);
        a[d] = make_uint2((uint32_t)s[0] | ((uint32_t)s[1] << 16), (uint32_t)s[2] | ((uint32_t)s[3] << 16));
        b[d] = (uint32_t)s[4] | ((uint32_t)s[5] << 16);
        float c[6];                                        // the values the shader will read: the halfs, widened
#pragma unroll
        for (int k = 0; k < 6; k++) c[k] = (float)__builtin_bit_cast(_Float16, s[k]);
        bound[d] = cov_spectral_bound(c[0], c[1], c[2], c[3], c[4], c[5]);
    }
}

// coefficient-major RGB triples (9 or 24 halfs per splat) -> 16-byte planes
__global__ __launch_bounds__(256) void k_split_sh(const uint16_t* __restrict__ sh, uint32_t count, uint32_t from,
                                                  const uint32_t* __restrict__ perm, uint32_t ncoef, uint4* __restrict__ p0,
                                                  void* __restrict__ p1, uint4* __restrict__ p2) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
        const uint16_t* s = sh + (size_t)ncoef * i;
        auto pk = [&](int k) { return (uint32_t)s[k] | ((uint32_t)s[k + 1] << 16); };
        const uint32_t d = DST(i);
        p0[d] = make_uint4(pk(0), pk(2), pk(4), pk(6));
        if (ncoef == 9) {
            reinterpret_cast<uint32_t*>(p1)[d] = (uint32_t)s[8];
        } else {
            reinterpret_cast<uint4*>(p1)[d] = make_uint4(pk(8), pk(10), pk(12), pk(14));
            p2[d] = make_uint4(pk(16), pk(18), pk(20), pk(22));
        }
    }
}

// test hook: undo k_project's per-block compaction -> one record / rect per splat in storage order (zeros when culled)
__global__ __launch_bounds__(256) void k_debug_expand(const unsigned long long* __restrict__ vis_mask, const uint2* __restrict__ vis32, uint32_t count,
                                                      const uint32_t* __restrict__ perm, const uint4* __restrict__ recs,
                                                      const uint2* __restrict__ rects, uint4* __restrict__ out_recs,
                                                      uint2* __restrict__ out_rects, unsigned long long* __restrict__ out_mask) {
    const uint32_t orig = blockIdx.x * 256u + threadIdx.x;
    const uint32_t i = orig < count ? (perm ? perm[orig] : orig) : 0u;     // internal position of this original splat
    const unsigned long long* mw = vis_mask + ((i >> 8) << 2);
    const uint32_t w = (i >> 6) & 3u, bit = i & 63u;
    const uint2 t = vis32[i >> 5];                                          // {mask of these 32 splats, slot of their first survivor}
    const uint32_t slot = t.y + (uint32_t)__popc(t.x & ((1u << (i & 31u)) - 1u));
    const bool vis = orig < count && ((mw[w] >> bit) & 1ull);
    if (out_mask) {                                                         // the mask in ORIGINAL splat order
        const unsigned long long b = __ballot(vis);
        if ((threadIdx.x & 63u) == 0u) out_mask[orig >> 6] = b;
    }
    if (orig >= count) return;
    const uint4 z = make_uint4(0, 0, 0, 0);
    if (out_recs) {
        out_recs[2 * (size_t)orig] = vis ? recs[2 * (size_t)slot] : z;
        out_recs[2 * (size_t)orig + 1] = vis ? recs[2 * (size_t)slot + 1] : z;
    }
    if (out_rects) out_rects[orig] = vis ? rects[slot] : make_uint2(0xFFFFu, 0u);
}

// 8-bit SH: 9 or 24 bytes per splat -> one 16-byte plane (+ one 8-byte plane for degree 2)
__global__ __launch_bounds__(256) void k_split_sh_u8(const uint8_t* __restrict__ sh, uint32_t count, uint32_t from,
                                                     const uint32_t* __restrict__ perm, uint32_t ncoef, uint4* __restrict__ p0,
                                                     uint2* __restrict__ p1) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
        const uint8_t* s = sh + (size_t)ncoef * i;
        auto pk = [&](uint32_t k) {
            uint32_t v = 0;
            for (uint32_t t = 0; t < 4; t++)
                if (k + t < ncoef) v |= (uint32_t)s[k + t] << (8 * t);
            return v;
        };
        const uint32_t d = DST(i);
        p0[d] = make_uint4(pk(0), pk(4), pk(8), pk(12));
        if (ncoef > 16) p1[d] = make_uint2(pk(16), pk(20));
    }
}

// Per storage block of 256 splats: the box of its centres and its largest covariance bound (NaN centres / bounds are left out
// of the box: such splats draw nothing; a NaN covariance bound keeps the block's strip test from ever culling it).
__global__ __launch_bounds__(256) void k_block_boxes(const float* __restrict__ px, const float* __restrict__ py, const float* __restrict__ pz,
                                                     const float* __restrict__ cov_bound, uint32_t n, uint32_t first_block,
                                                     float* __restrict__ boxes) {
    __shared__ float s_red[4][7];
    const uint32_t block = first_block + blockIdx.x, i = block * 256u + threadIdx.x;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY}, cb = 0.0f;
    bool cb_nan = false;
    if (i < n) {
        const float c[3] = {px[i], py[i], pz[i]};
#pragma unroll
        for (int k = 0; k < 3; k++) { lo[k] = fminf(lo[k], c[k]); hi[k] = fmaxf(hi[k], c[k]); }
        const float b = cov_bound[i];
        cb_nan = b != b;
        cb = cb_nan ? 0.0f : b;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int k = 0; k < 3; k++) { lo[k] = fminf(lo[k], __shfl_xor(lo[k], o, 64)); hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], o, 64)); }
        cb = fmaxf(cb, __shfl_xor(cb, o, 64));
    }
    const bool any_nan = __ballot(cb_nan) != 0ull;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (lane == 0u) {
        for (int k = 0; k < 3; k++) { s_red[wave][k] = lo[k]; s_red[wave][3 + k] = hi[k]; }
        s_red[wave][6] = any_nan ? NAN : cb;
    }
    __syncthreads();
    if (threadIdx.x < 8u) {
        const uint32_t k = threadIdx.x;
        float v = 0.0f;
        if (k < 3u) v = fminf(fminf(s_red[0][k], s_red[1][k]), fminf(s_red[2][k], s_red[3][k]));
        else if (k < 6u) v = fmaxf(fmaxf(s_red[0][k], s_red[1][k]), fmaxf(s_red[2][k], s_red[3][k]));
        else if (k == 6u) {
            v = fmaxf(fmaxf(s_red[0][6], s_red[1][6]), fmaxf(s_red[2][6], s_red[3][6]));
            if (s_red[0][6] != s_red[0][6] || s_red[1][6] != s_red[1][6] || s_red[2][6] != s_red[2][6] || s_red[3][6] != s_red[3][6]) v = NAN;
        }
        boxes[8u * (size_t)block + k] = v;
    }
}

__global__ __launch_bounds__(256) void k_iota2(uint32_t* __restrict__ a, uint32_t* __restrict__ b, uint32_t n) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) a[i] = b[i] = i;
}

static inline uint32_t up_grid(uint32_t n) {
    uint32_t g = (n + 255u) / 256u;
    return g < 1 ? 1 : (g > 4096u ? 4096u : g);
}

static int mesh_alloc_entries(gs_mesh* m, uint32_t capacity) {
    // keys are sized for 32-bit tile ids so the same buffers serve > 65536-tile strips
    GS_TRY(m->ekeyA.alloc((size_t)capacity * 4));
    GS_TRY(m->ekeyB.alloc((size_t)capacity * 4));
    GS_TRY(m->evalA.alloc((size_t)capacity * 4));
    GS_TRY(m->evalB.alloc((size_t)capacity * 4));
    m->entry_capacity = capacity;
    return GS_OK;
}

extern "C" {

int gs_mesh_create(gs_context* ctx, uint32_t max_splat_count, uint32_t sh_degree, uint32_t flags, gs_mesh** out) {
    GS_REQUIRE(ctx && out, "ctx / out == NULL");
    *out = nullptr;
    GS_REQUIRE(max_splat_count > 0, "max_splat_count == 0");
    GS_REQUIRE(max_splat_count <= (1u << 28), "max_splat_count > 2^28");
    GS_REQUIRE(sh_degree <= 2, "sh_degree > 2 (the reference renders degrees 0..2, src/Viewer.js:154)");
    GS_REQUIRE((flags & ~(GS_MESH_COV_HALF | GS_MESH_SH_U8 | GS_MESH_KEEP_ORDER)) == 0, "unknown mesh flags");
    ScopedDevice sd(ctx->device);
    gs_mesh* m = new (std::nothrow) gs_mesh();
    if (!m) return GS_ERR_NOMEM;
    m->ctx = ctx;
    m->max_count = max_splat_count;
    m->sh_degree = sh_degree;
    m->flags = flags;
    const size_t n = max_splat_count;
    const bool half = (flags & GS_MESH_COV_HALF) != 0;
    int st = GS_OK;
    auto A = [&](DevBuf& b, size_t bytes) { if (st == GS_OK) st = b.alloc(bytes); };
    A(m->px, n * 4); A(m->py, n * 4); A(m->pz, n * 4);
    A(m->covA, n * (half ? 8 : 16)); A(m->covB, n * (half ? 4 : 8));
    A(m->rgba, n * 4);
    if (flags & GS_MESH_SH_U8) {
        if (sh_degree >= 1) A(m->sh0, n * 16);
        if (sh_degree >= 2) A(m->sh1, n * 8);
    } else {
        if (sh_degree >= 1) { A(m->sh0, n * 16); A(m->sh1, n * (sh_degree == 2 ? 16 : 4)); }
        if (sh_degree >= 2) A(m->sh2, n * 16);
    }
    A(m->scene_dev, sizeof(gs_scene_params));
    A(m->cov_bound, n * 4);
    A(m->block_box, ((n + 255) / 256) * 32 + 32);
    m->reorder = !(flags & GS_MESH_KEEP_ORDER) && !getenv("GSPLAT_NO_REORDER");
    m->no_block_cull = getenv("GSPLAT_NO_BLOCK_CULL") != nullptr;
    m->no_block_list = getenv("GSPLAT_NO_BLOCK_LIST") != nullptr;
    m->block_test_always = getenv("GSPLAT_BLOCK_TEST_ALWAYS") != nullptr;
    m->no_deep = getenv("GSPLAT_NO_DEEP") != nullptr;
    if (const char* ls = getenv("GSPLAT_LIST_SHIFT"))
        if (ls[0] >= '1' && ls[0] <= '6' && ls[1] == '\0') m->forced_list_shift = ls[0] - '0';
    if (m->reorder) { A(m->perm, n * 4 + 16); A(m->inv_perm, n * 4); }   // (+16: the sorter's k_cull_front reads perm as 16-byte vectors)
    A(m->recs, n * sizeof(SplatRec)); A(m->rects, n * 8); A(m->rect_q, n * 8 + 2048); A(m->cidx, n * 4 + 1024); A(m->coff, n * 4 + 1024);
    A(m->vis_mask, ((n + 255) / 256) * 32 + 32);   // whole 256-splat blocks: 4 words each
    A(m->vis32, ((n + 255) / 256) * 64 + 64);
    A(m->prect, ((n + 255) / 256) * 256 * 8 + 64);
    A(m->block_any, ((n + 255) / 256 + 127) & ~(size_t)63);
    // (a context with one stream runs the frames' kernels in order: a second set would never be written early)
    m->two_sets = ctx->aux != ctx->stream && !getenv("GSPLAT_ONE_RECORD_SET");
    if (m->two_sets) {
        A(m->alt.recs, n * sizeof(SplatRec)); A(m->alt.rects, n * 8);
        A(m->alt.vis_mask, ((n + 255) / 256) * 32 + 32);
        A(m->alt.vis32, ((n + 255) / 256) * 64 + 64);
        A(m->alt.prect, ((n + 255) / 256) * 256 * 8 + 64);
        A(m->alt.block_any, ((n + 255) / 256 + 127) & ~(size_t)63);
    }
    A(m->bin_sums, 4 * 3 * 2048 + 64);               // uint32 [3][BIN_MAX_BLOCKS] + the batches-per-workgroup of the last count
    A(m->frame, sizeof(RenderFrame));
    if (st == GS_OK) st = m->radix.init();
    if (st == GS_OK) {
        // Never-uploaded splats are what the reference's zero-filled data textures hold (SplatMesh.js:686-697: new
        // Uint32Array / Float32Array): centre 0, covariance 0, colour 0 with alpha 0 - they draw nothing.  gs_mesh_upload
        // accepts any range, so a first upload of [100, 200) leaves [0, 100) in this state, `uploaded` = 200, and a draw
        // (or a stale index list) may name them: the planes must be defined and the storage permutation a bijection from
        // the start (identity until an upload assigns Morton slots).
        hipError_t e = hipSuccess;
        auto Z = [&](DevBuf& b) { if (e == hipSuccess && b.p) e = hipMemsetAsync(b.p, 0, b.bytes, ctx->stream); };
        Z(m->px); Z(m->py); Z(m->pz); Z(m->covA); Z(m->covB); Z(m->cov_bound); Z(m->rgba); Z(m->sh0); Z(m->sh1); Z(m->sh2);
        Z(m->block_box);                                      // boxes of never-uploaded blocks: the origin (their zero centres)
        if (e == hipSuccess && m->reorder) {
            hipLaunchKernelGGL(k_iota2, dim3(up_grid(max_splat_count)), dim3(256), 0, ctx->stream, m->perm.as<uint32_t>(),
                               m->inv_perm.as<uint32_t>(), max_splat_count);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            gs_set_error("initialising the mesh planes failed: %s", hipGetErrorString(e));
            st = GS_ERR_HIP;
        }
    }
    if (st == GS_OK) {
        // first guess: 8 entries per splat, at least 1M; grown on overflow
        uint64_t cap = (uint64_t)n * 8;
        if (cap < (1u << 20)) cap = 1u << 20;
        if (cap > 0x7FFFFFFFull) cap = 0x7FFFFFFFull;
        st = mesh_alloc_entries(m, (uint32_t)cap);
    }
    for (int i = 0; i < 6 && st == GS_OK; i++)
        if (hipEventCreate(&m->ev[i]) != hipSuccess) {
            gs_set_error("hipEventCreate failed");
            st = GS_ERR_HIP;
        }
    if (st == GS_OK && (hipEventCreateWithFlags(&m->ev_done, hipEventDisableTiming) != hipSuccess ||
                        (m->two_sets && hipEventCreateWithFlags(&m->alt.ev_done, hipEventDisableTiming) != hipSuccess))) {
        gs_set_error("hipEventCreate failed");
        st = GS_ERR_HIP;
    }
    for (int i = 0; i < gs_mesh::TIMING_RING && st == GS_OK; i++)
        if (hipEventCreate(&m->ring0[i]) != hipSuccess || hipEventCreate(&m->ring1[i]) != hipSuccess) {
            gs_set_error("hipEventCreate failed");
            st = GS_ERR_HIP;
        }
    if (st == GS_OK) {
        if (hipHostMalloc((void**)&m->mirror_host, 64, hipHostMallocMapped) != hipSuccess ||
            hipHostGetDevicePointer((void**)&m->mirror_dev, m->mirror_host, 0) != hipSuccess) {
            gs_set_error("hipHostMalloc(mapped) failed");
            st = GS_ERR_HIP;
        } else {
            memset(m->mirror_host, 0, 64);
        }
    }
    if (st != GS_OK) {
        gs_mesh_destroy(m);
        return st;
    }
    ctx->live_meshes.push_back(m);
    *out = m;
    return GS_OK;
}

void gs_mesh_destroy(gs_mesh* m) {
    if (!m) return;
    for (size_t i = 0; i < m->ctx->live_meshes.size(); i++)
        if (m->ctx->live_meshes[i] == m) {
            m->ctx->live_meshes.erase(m->ctx->live_meshes.begin() + i);
            break;
        }
    ScopedDevice sd(m->ctx->device);
    (void)hipStreamSynchronize(m->ctx->stream);
    if (m->ctx->aux != m->ctx->stream) (void)hipStreamSynchronize(m->ctx->aux);
    for (int i = 0; i < 6; i++)
        if (m->ev[i]) (void)hipEventDestroy(m->ev[i]);
    for (int i = 0; i < gs_mesh::TIMING_RING; i++) {
        if (m->ring0[i]) (void)hipEventDestroy(m->ring0[i]);
        if (m->ring1[i]) (void)hipEventDestroy(m->ring1[i]);
    }
    if (m->ev_done) (void)hipEventDestroy(m->ev_done);
    if (m->alt.ev_done) (void)hipEventDestroy(m->alt.ev_done);
    if (m->mirror_host) (void)hipHostFree(m->mirror_host);
    delete m;
}

// true when every splat of [from, from + count) already owns a storage slot (gs_mesh_upload has seen it)
static bool mesh_range_slotted(const gs_mesh* m, uint32_t from, uint32_t count) {
    if (!m->reorder || count == 0) return true;
    for (const auto& r : m->slotted)
        if (r.first <= from && from + count <= r.second) return true;      // ranges are merged, so one must cover it
    return false;
}

// One segment of an upload.  fresh: these splats have no storage slots yet - they get the Morton order of the segment;
// otherwise they keep the slots an earlier upload gave them (perm stays a bijection, planes uploaded through the other
// entry points and a bound sorter's resident result stay valid).
static int mesh_upload_segment(gs_mesh* m, uint32_t from, uint32_t count, const float* centers, const float* cov_f32,
                               const uint16_t* cov_f16, const uint8_t* rgba, const uint16_t* sh_f16, bool fresh) {
    const bool half = (m->flags & GS_MESH_COV_HALF) != 0;
    const bool sh_u8 = (m->flags & GS_MESH_SH_U8) != 0;
    hipStream_t st = m->ctx->stream;
    const uint32_t ncoef = (m->sh_degree == 0 || sh_u8) ? 0 : (m->sh_degree == 1 ? 9 : 24);
    const size_t b_c = (size_t)count * 12, b_cov = (size_t)count * (half ? 12 : 24), b_sh = (size_t)count * ncoef * 2;
    size_t off_cov = (b_c + 255) & ~(size_t)255, off_sh = (off_cov + b_cov + 255) & ~(size_t)255;
    GS_TRY(m->staging.ensure(off_sh + b_sh + 512 + (size_t)count * 4));
    char* stg = m->staging.as<char>();
    GS_HIP(hipMemcpyAsync(stg, centers, b_c, hipMemcpyHostToDevice, st));
    GS_HIP(hipMemcpyAsync(stg + off_cov, half ? (const void*)cov_f16 : (const void*)cov_f32, b_cov, hipMemcpyHostToDevice, st));
    if (ncoef) GS_HIP(hipMemcpyAsync(stg + off_sh, sh_f16, b_sh, hipMemcpyHostToDevice, st));
    const dim3 g(up_grid(count)), b(256);
    const uint32_t* perm = m->reorder ? m->perm.as<uint32_t>() : nullptr;
    if (fresh) m->layout_version++;
    if (m->reorder && fresh) {
        // Morton order of this segment: bounds on the host (the centres are host memory anyway), 30-bit codes, 4 stable
        // radix passes with the entry ping-pong buffers as scratch, then perm[original] = internal
        float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (uint32_t i = 0; i < count; i++)
            for (int k = 0; k < 3; k++) {
                const float v = centers[3 * (size_t)i + k];
                if (v < mn[k]) mn[k] = v;
                if (v > mx[k]) mx[k] = v;
            }
        float3 lo = make_float3(mn[0], mn[1], mn[2]), inv;
        inv.x = mx[0] > mn[0] ? 1.0f / (mx[0] - mn[0]) : 0.0f;
        inv.y = mx[1] > mn[1] ? 1.0f / (mx[1] - mn[1]) : 0.0f;
        inv.z = mx[2] > mn[2] ? 1.0f / (mx[2] - mn[2]) : 0.0f;
        if (count > m->entry_capacity) GS_TRY(mesh_alloc_entries(m, count));
        uint32_t* kbuf[2] = {m->ekeyA.as<uint32_t>(), m->ekeyB.as<uint32_t>()};
        uint32_t* vbuf[2] = {m->evalA.as<uint32_t>(), m->evalB.as<uint32_t>()};
        hipLaunchKernelGGL(k_morton_keys, g, b, 0, st, (const float*)stg, count, lo, inv, kbuf[0], vbuf[0]);
        GS_HIP(hipMemsetAsync(m->radix.digit_total.p, 0, sizeof(uint32_t) * RADIX_TOTAL_WORDS, st));
        const RadixExec ex = {st, &m->radix, m->ctx->lds_atomic_lane_order};
        for (int pass = 0; pass < 4; pass++) {
            ArrayLoader<uint32_t> al = {kbuf[pass & 1], vbuf[pass & 1], nullptr, count};
            GS_TRY((radix_pass<ArrayLoader<uint32_t>, uint32_t, true>(ex, al, al, count, 8 * pass, pass, kbuf[(pass + 1) & 1],
                                                                     vbuf[(pass + 1) & 1])));
        }
        hipLaunchKernelGGL(k_perm_from_sorted, g, b, 0, st, vbuf[0], count, from, m->perm.as<uint32_t>(), m->inv_perm.as<uint32_t>());
    }
    const size_t off_rgba = (off_sh + b_sh + 255) & ~(size_t)255;
    GS_HIP(hipMemcpyAsync(stg + off_rgba, rgba, (size_t)count * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_scatter_u32, g, b, 0, st, (const uint32_t*)(stg + off_rgba), count, from, perm, m->rgba.as<uint32_t>());
    hipLaunchKernelGGL(k_split_centers, g, b, 0, st, (const float*)stg, count, from, perm, m->px.as<float>(), m->py.as<float>(),
                       m->pz.as<float>());
    if (half)
        hipLaunchKernelGGL(k_split_cov_f16, g, b, 0, st, (const uint16_t*)(stg + off_cov), count, from, perm, m->covA.as<uint2>(),
                           m->covB.as<uint32_t>(), m->cov_bound.as<float>());
    else
        hipLaunchKernelGGL(k_split_cov_f32, g, b, 0, st, (const float*)(stg + off_cov), count, from, perm, m->covA.as<float4>(),
                           m->covB.as<float2>(), m->cov_bound.as<float>());
    if (ncoef)
        hipLaunchKernelGGL(k_split_sh, g, b, 0, st, (const uint16_t*)(stg + off_sh), count, from, perm, ncoef, m->sh0.as<uint4>(),
                           m->sh1.p, m->sh2.as<uint4>());
    {   // The boxes of the storage blocks this segment touches.  A fresh segment owns the slots [from, from + count); splats
        // uploaded before keep THEIR slots, scattered over the Morton run(s) of the earlier upload(s) - anywhere inside the merged
        // slotted range that contains the segment (segments are cut at the borders of those ranges): all of its blocks are redone.
        // (only [from, from + count) was redone, so a block kept the box of the data it held before the edit and
        // k_project's block cull could drop splats that had moved into view.)
        uint32_t lo = from, hi = from + count;
        if (!fresh && m->reorder)
            for (const auto& r : m->slotted)
                if (r.first <= from && from + count <= r.second) { lo = r.first; hi = r.second; }
        const uint32_t b0 = lo / 256u, b1 = (hi + 255u) / 256u;
        hipLaunchKernelGGL(k_block_boxes, dim3(b1 - b0), dim3(256), 0, st, m->px.as<float>(), m->py.as<float>(), m->pz.as<float>(),
                           m->cov_bound.as<float>(), m->max_count, b0, m->block_box.as<float>());
    }
    GS_HIP(hipGetLastError());
    GS_HIP(hipStreamSynchronize(st));
    return GS_OK;
}

// SplatMesh.updateDataTexturesFromBaseData(fromSplat, toSplat) accepts any range, any number of times
// (/root/reference/src/splatmesh/SplatMesh.js:900-1062).  Splats seen for the first time are stored along the Morton curve
// of their own contiguous run; splats uploaded before keep their slots and only have their data replaced.
int gs_mesh_upload(gs_mesh* m, uint32_t from, uint32_t count, const float* centers, const float* cov_f32,
                   const uint16_t* cov_f16, const uint8_t* rgba, const uint16_t* sh_f16) {
    GS_REQUIRE(m && centers && rgba, "mesh / centers / rgba == NULL");
    GS_REQUIRE((uint64_t)from + count <= m->max_count, "range exceeds max_splat_count");
    const bool half = (m->flags & GS_MESH_COV_HALF) != 0;
    GS_REQUIRE(half ? (cov_f16 && !cov_f32) : (cov_f32 && !cov_f16), "covariance format does not match the mesh (GS_MESH_COV_HALF)");
    const bool sh_u8 = (m->flags & GS_MESH_SH_U8) != 0;
    GS_REQUIRE(m->sh_degree == 0 || sh_u8 || sh_f16, "mesh stores spherical harmonics but sh_f16 == NULL");
    GS_REQUIRE(!(sh_u8 && sh_f16), "GS_MESH_SH_U8 mesh: upload SH with gs_mesh_upload_sh_u8");
    if (count == 0) return GS_OK;
    m->projection_pending = false;                        // the scene changed under a pending gs_mesh_project
    ScopedDevice sd(m->ctx->device);
    // earlier draws may still read the planes / the permutation on either stream
    GS_HIP(hipStreamSynchronize(m->ctx->stream));
    if (m->ctx->aux != m->ctx->stream) GS_HIP(hipStreamSynchronize(m->ctx->aux));
    const uint32_t ncoef = (m->sh_degree == 0 || sh_u8) ? 0 : (m->sh_degree == 1 ? 9 : 24);
    const uint32_t end = from + count;
    {   // where the scene is and how large (mean and RMS radius of the centres; a heuristic's inputs: re-uploads count twice)
        double sx = 0.0, sy = 0.0, sz = 0.0, sq = 0.0;
        uint64_t n = 0;
        for (uint32_t i = 0; i < count; i++) {
            const float x = centers[3 * (size_t)i], y = centers[3 * (size_t)i + 1], z = centers[3 * (size_t)i + 2];
            const float r2 = x * x + y * y + z * z;
            if (!(r2 < 1e30f)) continue;                      // NaN / infinite centres draw nothing
            sx += x; sy += y; sz += z; sq += r2; n++;
        }
        m->centre_sum[0] += sx; m->centre_sum[1] += sy; m->centre_sum[2] += sz; m->centre_sq += sq; m->centre_n += n;
    }
    // cut [from, end) at the borders of the ranges that already own storage slots (disjoint, sorted by begin)
    std::vector<std::pair<uint32_t, bool>> cuts;          // (segment begin, fresh)
    uint32_t pos = from;
    for (const auto& r : m->slotted) {
        if (r.second <= pos) continue;
        if (r.first >= end) break;
        if (r.first > pos) { cuts.push_back({pos, true}); pos = r.first; }
        cuts.push_back({pos, false});
        pos = r.second < end ? r.second : end;
        if (pos == end) break;
    }
    if (pos < end) cuts.push_back({pos, true});
    for (size_t k = 0; k < cuts.size(); k++) {
        const uint32_t b = cuts[k].first, e = k + 1 < cuts.size() ? cuts[k + 1].first : end, o = b - from;
        GS_TRY(mesh_upload_segment(m, b, e - b, centers + 3 * (size_t)o, cov_f32 ? cov_f32 + 6 * (size_t)o : nullptr,
                                   cov_f16 ? cov_f16 + 6 * (size_t)o : nullptr, rgba + 4 * (size_t)o,
                                   sh_f16 ? sh_f16 + (size_t)ncoef * o : nullptr, cuts[k].second || !m->reorder));
    }
    // merge [from, end) into the slotted ranges
    std::vector<std::pair<uint32_t, uint32_t>> merged;
    std::pair<uint32_t, uint32_t> cur(from, end);
    for (const auto& r : m->slotted) {
        if (r.second < cur.first || r.first > cur.second) merged.push_back(r);
        else { cur.first = r.first < cur.first ? r.first : cur.first; cur.second = r.second > cur.second ? r.second : cur.second; }
    }
    merged.push_back(cur);
    std::sort(merged.begin(), merged.end());
    m->slotted.swap(merged);
    if (end > m->uploaded) m->uploaded = end;
    return GS_OK;
}

int gs_mesh_upload_sh_u8(gs_mesh* m, uint32_t from, uint32_t count, const uint8_t* sh_u8) {
    GS_REQUIRE(m && sh_u8, "mesh / sh_u8 == NULL");
    GS_REQUIRE((m->flags & GS_MESH_SH_U8) && m->sh_degree >= 1, "mesh was not created with GS_MESH_SH_U8 and sh_degree >= 1");
    GS_REQUIRE((uint64_t)from + count <= m->max_count, "range exceeds max_splat_count");
    GS_REQUIRE(mesh_range_slotted(m, from, count), "upload the splats with gs_mesh_upload before their 8-bit SH");
    if (count == 0) return GS_OK;
    m->projection_pending = false;                        // the scene changed under a pending gs_mesh_project
    ScopedDevice sd(m->ctx->device);
    hipStream_t st = m->ctx->stream;
    const uint32_t ncoef = m->sh_degree == 1 ? 9 : 24;
    GS_TRY(m->staging.ensure((size_t)count * ncoef + 256));
    GS_HIP(hipMemcpyAsync(m->staging.p, sh_u8, (size_t)count * ncoef, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_split_sh_u8, dim3(up_grid(count)), dim3(256), 0, st, m->staging.as<uint8_t>(), count, from,
                       m->reorder ? m->perm.as<uint32_t>() : nullptr, ncoef, m->sh0.as<uint4>(), m->sh1.as<uint2>());
    GS_HIP(hipGetLastError());
    GS_HIP(hipStreamSynchronize(st));
    return GS_OK;
}

int gs_mesh_upload_scene_indexes(gs_mesh* m, uint32_t from, uint32_t count, const uint32_t* scene_indexes) {
    GS_REQUIRE(m && scene_indexes, "mesh / scene_indexes == NULL");
    GS_REQUIRE((uint64_t)from + count <= m->max_count, "range exceeds max_splat_count");
    GS_REQUIRE(mesh_range_slotted(m, from, count), "upload the splats with gs_mesh_upload before their scene indexes");
    m->projection_pending = false;                        // the scene changed under a pending gs_mesh_project
    ScopedDevice sd(m->ctx->device);
    hipStream_t st = m->ctx->stream;
    if (!m->scene_idx.p) {
        GS_TRY(m->scene_idx.alloc((size_t)m->max_count * 4));
        GS_HIP(hipMemsetAsync(m->scene_idx.p, 0, (size_t)m->max_count * 4, st));
    }
    for (uint32_t i = 0; i < count; i++) GS_REQUIRE(scene_indexes[i] < GS_MAX_SCENES, "scene index >= GS_MAX_SCENES");
    if (count) {
        GS_TRY(m->staging.ensure((size_t)count * 4));
        GS_HIP(hipMemcpyAsync(m->staging.p, scene_indexes, (size_t)count * 4, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_scatter_u32, dim3(up_grid(count)), dim3(256), 0, st, m->staging.as<uint32_t>(), count, from,
                           m->reorder ? m->perm.as<uint32_t>() : nullptr, m->scene_idx.as<uint32_t>());
        GS_HIP(hipGetLastError());
    }
    GS_HIP(hipStreamSynchronize(st));
    return GS_OK;
}

int gs_mesh_set_scenes(gs_mesh* m, const gs_scene_params* params) {
    GS_REQUIRE(m && params, "mesh / params == NULL");
    GS_REQUIRE(params->scene_count >= 1 && params->scene_count <= GS_MAX_SCENES, "scene_count outside 1..GS_MAX_SCENES");
    m->projection_pending = false;                        // the scene changed under a pending gs_mesh_project
    ScopedDevice sd(m->ctx->device);
    hipStream_t st = m->ctx->stream;
    // the vertex stage of an earlier draw may still read the previous values on ctx->aux
    if (m->ctx->aux != st) GS_HIP(hipStreamSynchronize(m->ctx->aux));
    GS_HIP(hipMemcpyAsync(m->scene_dev.p, params, sizeof(*params), hipMemcpyHostToDevice, st));
    GS_HIP(hipStreamSynchronize(st));
    m->scene_count = params->scene_count;
    m->has_scenes = true;
    return GS_OK;
}

// List-bin size of the following draws, from what a draw saw: the 16-px tiles its visible splats touch.  Large lists only pay when
// splats are large enough to share them (a scene of tiny splats under 128-px lists makes every 32-px bin scan 16 bins' worth of
// entries: the capture-like C3S stand-in draws in 4.5 ms instead of 1.2).  Fed by gs_mesh_last_stats / a draw with statistics AND by
// the words every draw leaves in mapped host memory (mesh_read_view_share), so that a host that never asks for statistics - a
// viewer's render loop - gets the same bins a few frames later (profiles/r06w_motion_ab.txt: the flaw this closed).
// (with 8 % of hysteresis around each threshold: a camera that sits on one does not flip the size - and with it the entry
// statistics and the frame time - from draw to draw)
static void mesh_adapt_list_bins(gs_mesh* m, uint64_t tiles16, uint32_t visible) {
    if (visible == 0u) return;
    const float tiles = (float)tiles16, vis = (float)visible;
    static const uint32_t shifts[4] = {GS_LIST_SHIFT_SMALL, GS_LIST_SHIFT_LARGE, GS_LIST_SHIFT_BIG, GS_LIST_SHIFT_HUGE};
    static const float thr[3] = {GS_LIST_TILES_PER_SPLAT, GS_LIST_TILES_PER_SPLAT_BIG, GS_LIST_TILES_PER_SPLAT_HUGE};
    uint32_t level = 0;
    while (level < 3u && shifts[level] != m->list_shift) level++;
    while (level < 3u && tiles >= 1.08f * thr[level] * vis) level++;
    while (level > 0u && tiles < 0.92f * thr[level - 1u] * vis) level--;
    m->list_shift = shifts[level];
}

static int mesh_collect_stats(gs_mesh* m, gs_render_stats* stats) {
    hipStream_t st = m->ctx->stream;
    RenderFrame f;
    GS_HIP(hipMemcpyAsync(&f, m->frame.p, sizeof(f), hipMemcpyDeviceToHost, st));
    GS_HIP(hipStreamSynchronize(st));
    float t[5] = {0, 0, 0, 0, 0};
    if (m->timed_draw) {
        for (int i = 1; i < 5; i++) GS_HIP(hipEventElapsedTime(&t[i], m->ev[i], m->ev[i + 1]));
    }
    // on ctx->aux: may overlap a sort and the tail of the previous draw
    if (m->timed_draw && m->timed_project) GS_HIP(hipEventElapsedTime(&t[0], m->ev_p0, m->ev_p1));
    m->last.project_ms = t[0];
    m->last.bin_ms = t[1];
    m->last.tile_sort_ms = t[2];
    m->last.blend_ms = t[3] + t[4];
    float total = 0;
    if (m->timed_draw) GS_HIP(hipEventElapsedTime(&total, m->ev[0], m->ev[5]));
    m->last.device_ms = total;
    m->last.visible_splats = f.visible;
    if (m->last_pp.row_begin == 0u && m->last_pp.row_end >= m->last_pp.tiles_y) {     // (a strip's count says nothing about the scene)
        m->measured_visible = f.visible;
        m->measured_count = m->last_pp.count;
    }
    m->last.tile_entries = ((uint64_t)f.entries_hi << 32) | f.entries_lo;
    m->last.tiles16 = ((uint64_t)f.tiles16_hi << 32) | f.tiles16_lo;
    m->last.entry_capacity = m->entry_capacity;
    m->last.list_bin_px = GS_TILE << m->drawn_list_shift;
    m->last.flags = 0;
    if (m->deep_flags.p) {                                 // the chunked composite's per-draw words (tile_blend.hip)
        uint32_t w[8];
        GS_HIP(hipMemcpyAsync(w, m->deep_flags.p, sizeof(w), hipMemcpyDeviceToHost, st));
        GS_HIP(hipStreamSynchronize(st));
        if (w[GS_FLAG_POOL_OVER]) m->last.flags |= GS_DRAW_POOL_EXHAUSTED;
    }
    if (m->bin_scan_ready) {                               // k_bin_fused: a poll of the scan across its grid ran out of patience
        uint32_t fail = 0;
        GS_HIP(hipMemcpyAsync(&fail, (const char*)m->bin_scan.p + ((size_t)3 * 2048 + 3 * 64) * 8, 4, hipMemcpyDeviceToHost, st));
        GS_HIP(hipStreamSynchronize(st));
        if (fail) {
            // (reported once: the word is cleared so that the draws that follow are judged on their own)
            GS_HIP(hipMemsetAsync((char*)m->bin_scan.p + ((size_t)3 * 2048 + 3 * 64) * 8, 0, 4, st));
            gs_set_error("the binner's cross-workgroup scan timed out (k_bin_fused): the frame is incomplete");
            return GS_ERR_HIP;
        }
    }
    m->last.entries_scanned = 0;
    m->last.splats_walked = 0;
    m->last.halves_evaluated = 0;
    if (m->blend_bins) {                                   // per-workgroup counters of the blend, summed here
        const size_t nb = m->blend_bins;
        std::vector<uint32_t> bs(3 * nb);                  // uint2 [nb] {staged, halves} | uint32 [nb] pairs
        GS_HIP(hipMemcpyAsync(bs.data(), m->blend_stats.p, nb * 12, hipMemcpyDeviceToHost, st));
        GS_HIP(hipStreamSynchronize(st));
        for (size_t b = 0; b < nb; b++) {
            m->last.entries_scanned += bs[2 * b];
            m->last.halves_evaluated += bs[2 * b + 1];
            m->last.splats_walked += bs[2 * nb + b];
        }
    }
    mesh_adapt_list_bins(m, m->last.tiles16, m->last.visible_splats);
    if (stats) *stats = m->last;
    return f.overflow ? 1 : 0;
}

// The vertex stage of a draw.  It only depends on the scene and the camera, so it runs on ctx->aux next to whatever the
// caller-visible stream and the sorter's stream are doing; it may start once the previous draw has consumed the records /
// rects / mask it is about to overwrite.
static void swap_buf(DevBuf& a, DevBuf& b) { std::swap(a.p, b.p); std::swap(a.bytes, b.bytes); }
static void mesh_swap_sets(gs_mesh* m) {
    gs_mesh::ProjSet& o = m->alt;
    swap_buf(m->recs, o.recs); swap_buf(m->zrec, o.zrec); swap_buf(m->rects, o.rects); swap_buf(m->vis_mask, o.vis_mask); swap_buf(m->block_any, o.block_any);
    swap_buf(m->vis32, o.vis32); swap_buf(m->prect, o.prect); swap_buf(m->vis_orig, o.vis_orig);
    std::swap(m->ev_done, o.ev_done); std::swap(m->set_drawn, o.drawn);
    std::swap(m->vis_orig_dirty, o.vis_orig_dirty); std::swap(m->vis_orig_count, o.vis_orig_count); std::swap(m->vis_orig_lazy, o.vis_orig_lazy);
}

static int mesh_project(gs_mesh* m, const ProjectParams& pp, bool orig_mask, bool timed) {
    gs_context* ctx = m->ctx;
    hipStream_t st = ctx->stream, aux = ctx->aux;
    if (timed) GS_HIP(hipEventRecord(m->ev[0], st));
    // the set this vertex stage writes: the one the draw BEFORE the last read (two sets), else the last draw's own
    if (m->two_sets) mesh_swap_sets(m);
    if (aux != st && m->set_drawn) GS_HIP(hipStreamWaitEvent(aux, m->ev_done, 0));
    const bool sample = timed || aux != st || (ctx->kernel_sample && m->project_serial++ % ctx->kernel_sample == 0);
    if (sample) {   // next slot of the timing ring; a slot about to be reused is harvested first (skipped if still in flight)
        const uint32_t slot = m->ring_next++ % gs_mesh::TIMING_RING;
        if (m->ring_used[slot]) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, m->ring0[slot], m->ring1[slot]) == hipSuccess) {
                //  a timed draw brackets the whole vertex stage, a sampled one k_project alone: two clocks
                if (m->ring_whole[slot]) { m->stage_sum_ms += ms; m->stage_launches++; }
                else { m->proj_sum_ms += ms; m->proj_launches++; }
            } else {
                (void)hipGetLastError();
            }
        }
        m->ring_used[slot] = true;
        m->ring_whole[slot] = timed;
        m->ev_p0 = m->ring0[slot];
        m->ev_p1 = m->ring1[slot];
    }
    m->timed_project = sample;
    GS_TRY(gs_launch_project(m, pp, orig_mask, sample ? m->ev_p0 : nullptr, sample ? m->ev_p1 : nullptr, timed));
    return GS_OK;
}

static int mesh_draw_once(gs_mesh* m, const ProjectParams& pp, const uint32_t* order_dev, gs_sorter* sorter, uint32_t R,
                          uint8_t* out_dev, bool projected, bool timed) {
    gs_context* ctx = m->ctx;
    hipStream_t st = ctx->stream, aux = ctx->aux;
    timed = timed || ctx->stage_events;
    m->timed_draw = timed;
    const uint32_t tiles = pp.lists_x * (pp.list_row_end - pp.list_row_begin);  // one entry list per list bin
    GS_TRY(m->tile_ranges.ensure((size_t)tiles * 8 + 16));
    if (!projected) GS_TRY(mesh_project(m, pp, false, timed));   // else gs_mesh_project already ran it for this camera
    else if (timed) GS_HIP(hipEventRecord(m->ev[0], st));
    // join: projection and (if a sorter feeds this draw) the sort result
    if (aux != st) GS_HIP(hipStreamWaitEvent(st, m->ev_p1, 0));
    if (sorter && sorter->stream != st) GS_HIP(hipStreamWaitEvent(st, sorter->ev1, 0));
    if (timed) GS_HIP(hipEventRecord(m->ev[1], st));
    // the index list is in the caller's splat numbering unless it comes from a sorter bound to this mesh
    m->translate = m->reorder && !(sorter && sorter->result_mesh == m);
    GS_TRY(gs_launch_binning(m, pp, order_dev, sorter, R));   // records ev[2] between emit and the tile sort
    if (timed) GS_HIP(hipEventRecord(m->ev[3], st));
    GS_TRY(gs_launch_blend(m, pp, out_dev));
    m->stats_pp = pp;                                          // whose view the per-bin blend statistics now describe
    m->stats_pp_valid = true;
    m->drawn_dest_depth = m->dest_depth; m->drawn_dest_rgba = m->dest_rgba;      // the destination this draw saw (gs_mesh_debug_rop8)
    m->drawn_dest_w = m->dest_w; m->drawn_dest_h = m->dest_h; m->drawn_dest_flags = m->dest_flags;
    if (timed) {
        GS_HIP(hipEventRecord(m->ev[4], st));
        GS_HIP(hipEventRecord(m->ev[5], st));
    }
    if (aux != st) GS_HIP(hipEventRecord(m->ev_done, st));     // only another stream ever waits for it
    m->set_drawn = true;
    m->has_draw = true;
    return GS_OK;
}

// gs_camera -> the kernels' parameter block (shared by gs_mesh_project and gs_mesh_render)
static int mesh_params(gs_mesh* m, const gs_camera* cam, ProjectParams& pp) {
    GS_REQUIRE(cam->width > 0 && cam->height > 0 && cam->width <= 4096u * GS_TILE && cam->height <= 4096u * GS_TILE,
               "viewport size (at most 65536 x 65536 px: tile coordinates travel as 12-bit fields)");
    GS_REQUIRE(cam->sh_degree <= 2, "sphericalHarmonicsDegree > 2");
    memset(&pp, 0, sizeof(pp));
    memcpy(pp.view, cam->view, sizeof(pp.view));
    memcpy(pp.proj, cam->proj, sizeof(pp.proj));
    memcpy(pp.cam_pos, cam->cam_pos, sizeof(pp.cam_pos));
    pp.focal_x = cam->focal[0];
    pp.focal_y = cam->focal[1];
    pp.width = (float)cam->width;
    pp.height = (float)cam->height;
    pp.splat_scale = cam->splat_scale;
    pp.kernel2d = cam->kernel2d;
    pp.max_splat_px = cam->max_splat_px;
    pp.inv_focal_adj = cam->inv_focal_adj;
    pp.sh_degree = cam->sh_degree < m->sh_degree ? cam->sh_degree : m->sh_degree;
    pp.sh_stored = m->sh_degree;
    pp.cov_half = (m->flags & GS_MESH_COV_HALF) ? 1u : 0u;
    pp.flags = cam->flags;
    pp.sh_u8 = (m->flags & GS_MESH_SH_U8) ? 1u : 0u;
    pp.scene_count = m->has_scenes ? m->scene_count : 1u;
    pp.ortho_zoom = cam->ortho_zoom;
    pp.fade_start = cam->fade_start_radius;
    memcpy(pp.scene_center, cam->scene_center, sizeof(pp.scene_center));
    memcpy(pp.view_matrix, cam->view_matrix, sizeof(pp.view_matrix));
    for (int r = 0; r < 3; r++) {                           // row r of mat3(view), column-major storage
        const double a = cam->view[r], b = cam->view[4 + r], c = cam->view[8 + r];
        pp.mv_row_norm[r] = (float)(sqrt(a * a + b * b + c * c) * (1.0 + 1e-6));
    }
    const bool needs_scenes = (cam->flags & (GS_CAM_SCENE_EFFECTS | GS_CAM_DYNAMIC)) || (pp.sh_u8 && pp.sh_degree >= 1);
    GS_REQUIRE(!needs_scenes || m->has_scenes, "this draw needs per-scene uniforms: call gs_mesh_set_scenes first");
    GS_REQUIRE(pp.scene_count <= 1 || m->scene_idx.p, "several scenes but no scene indexes uploaded");
    pp.tiles_x = (cam->width + GS_TILE - 1) / GS_TILE;
    pp.tiles_y = (cam->height + GS_TILE - 1) / GS_TILE;
    pp.row_begin = cam->tile_row_begin;
    pp.row_end = cam->tile_row_end;
    if (pp.row_begin == 0 && pp.row_end == 0) pp.row_end = pp.tiles_y;
    GS_REQUIRE(pp.row_begin <= pp.row_end && pp.row_end <= pp.tiles_y, "tile row range outside the viewport");
    pp.count = m->uploaded;
    // whole-block tests assume one modelView for every splat of a block: off under per-scene transforms (and GSPLAT_NO_BLOCK_CULL)
    pp.block_cull = (!(cam->flags & GS_CAM_DYNAMIC) && !m->no_block_cull) ? 1u : 0u;
    GS_REQUIRE(m->max_count <= GS_ENT_SLOT_MASK, "the mesh holds more than 2^28 - 1 splats");

    // pixel rows covered by this rank's strip, and the 32-px bins that cover them
    const uint32_t y0 = pp.row_begin * GS_TILE;
    const uint32_t y1 = pp.row_end * GS_TILE < cam->height ? pp.row_end * GS_TILE : cam->height;
    pp.y0 = y0;
    pp.y1 = y1 > y0 ? y1 : y0;
    pp.bins_x = (cam->width + GS_BIN - 1) / GS_BIN;
    pp.bin_row_begin = y0 / GS_BIN;
    pp.bin_row_end = pp.y1 > y0 ? (pp.y1 + GS_BIN - 1) / GS_BIN : pp.bin_row_begin;
    pp.list_shift = m->forced_list_shift >= 0 ? (uint32_t)m->forced_list_shift : m->list_shift;
    const uint32_t list_px = GS_TILE << pp.list_shift;
    pp.lists_x = (cam->width + list_px - 1) / list_px;
    pp.list_row_begin = y0 / list_px;
    pp.list_row_end = pp.y1 > y0 ? (pp.y1 + list_px - 1) / list_px : pp.list_row_begin;
    // the destination the splats are tested against and blended over (gs_mesh_set_destination)
    GS_REQUIRE(!(m->dest_depth || m->dest_rgba) || (m->dest_w == cam->width && m->dest_h == cam->height),
               "the destination set with gs_mesh_set_destination does not have this camera's viewport size");
    pp.depth_mode = m->dest_depth ? ((m->dest_flags & GS_DEST_DEPTH_UNORM24) ? 2u : 1u) : 0u;
    return GS_OK;
}

// An asynchronous draw (no stats, no host output) cannot know whether its entry buffer overflowed; the frame it produced is
// then missing its farthest list entries.  Every draw leaves its verdict in mapped host memory (k_bin_emit), and the next
// draw reads it here - no synchronisation when everything fitted.  After an overflow the buffers are grown to what that
// draw needed (this waits for the stream once) and the caller is told with GS_WARN_FRAME_TRUNCATED.
// The share of the scene in view, from the last FULL-frame draw whose verdict has arrived (a 16-byte store of k_bin_emit's into the
// mapped words: {serial, visible, 16-px tiles}), and the size of the list bins from the last draw of any kind; hints for the vertex
// stage's launch shape (project.hip) and the binner's geometry: frames do not depend on either.
static void mesh_read_view_share(gs_mesh* m) {
    volatile uint32_t* mir = m->mirror_host;
    const uint32_t vs = mir[8], vv = mir[9], t_lo = mir[10], t_hi = mir[11];
    if (vs == 0u || mir[8] != vs) return;                   // nothing yet / a newer draw is writing: look again next time
    if (vs == m->full_serial[vs & 7u]) {
        m->measured_visible = vv;
        m->measured_count = m->full_count[vs & 7u];
    }
    if (vs != m->adapted_serial) {                          // (any draw, strips included: tiles per visible splat is a ratio)
        m->adapted_serial = vs;
        if (!getenv("GSPLAT_NO_ASYNC_LIST_BINS")) {
            const uint32_t before = m->list_shift;
            const uint64_t tiles16 = ((uint64_t)t_hi << 32) | t_lo;
            mesh_adapt_list_bins(m, tiles16, vv);
            // smaller list bins mean more entries - at most one per 16-px tile touched: room for them before the first draw that
            // uses the new size, instead of one truncated frame and a heal after it (mesh_heal_overflow grows the buffers)
            if (m->list_shift < before) {
                const uint64_t want = tiles16 + tiles16 / 8 + 1024;
                m->grow_entries_to = (uint32_t)(want > 0x7FFFFFFFull ? 0x7FFFFFFFull : want);
            }
        }
    }
}

static int mesh_heal_overflow(gs_mesh* m, bool* healed) {
    *healed = false;
    volatile uint32_t* mir = m->mirror_host;
    mesh_read_view_share(m);
    if (m->grow_entries_to > m->entry_capacity) {           // the list bins just became smaller (mesh_read_view_share)
        GS_HIP(hipStreamSynchronize(m->ctx->stream));       // draws in flight still use the old buffers
        GS_TRY(mesh_alloc_entries(m, m->grow_entries_to));
    }
    m->grow_entries_to = 0;
    const uint32_t serial = mir[0];
    if (serial == m->healed_serial || !mir[1]) return GS_OK;
    const uint64_t need = ((uint64_t)mir[3] << 32) | mir[2];
    if (mir[0] != serial) return GS_OK;                     // a newer draw is writing: look again next time
    m->healed_serial = serial;
    if (need <= m->entry_capacity) return GS_OK;            // already grown (by a synchronous draw or gs_mesh_last_stats)
    uint64_t want = need + need / 8 + 1024;
    if (want > 0x7FFFFFFFull) want = 0x7FFFFFFFull;
    GS_HIP(hipStreamSynchronize(m->ctx->stream));           // draws in flight still use the old buffers
    GS_TRY(mesh_alloc_entries(m, (uint32_t)want));
    m->truncated_draws++;
    *healed = true;
    return GS_OK;
}

int gs_mesh_project(gs_mesh* m, const gs_camera* cam) {
    GS_REQUIRE(m && cam, "mesh / camera == NULL");
    ProjectParams pp;
    GS_TRY(mesh_params(m, cam, pp));
    ScopedDevice sd(m->ctx->device);
    // + the per-splat mask a visibility-culled sort reads: written here by one atomic per survivor - or, for a full frame whose
    // consumer is a bound sorter that holds the mesh's position map, left to that sorter (k_mask_derive_count: 25 us of a C3 frame)
    static const bool no_lazy = getenv("GSPLAT_NO_LAZY_MASK") != nullptr;      // (A/B and tests)
    const bool lazy = m->derive_orig_mask && m->reorder && !no_lazy && pp.row_begin == 0u && pp.row_end >= pp.tiles_y && pp.count == m->uploaded;
    GS_TRY(mesh_project(m, pp, !lazy, m->ctx->stage_events));
    m->vis_orig_lazy = lazy;
    if (lazy) {
        GS_TRY(m->vis_orig.ensure(((size_t)m->max_count + 63) / 64 * 8 + 64));
        m->vis_orig_dirty = true;                              // (whatever it holds beyond the words the sorter will write)
        m->vis_orig_count = pp.count;
    }
    m->projection_pending = true;
    m->projected_cam = *cam;
    m->projected_depth_mode = pp.depth_mode;
    return GS_OK;
}

int gs_mesh_render(gs_mesh* m, const gs_camera* cam, const uint32_t* sorted_host, gs_sorter* sorter,
                   uint32_t render_count, uint8_t* rgba_out_host, void* rgba_out_dev, gs_render_stats* stats) {
    GS_REQUIRE(m && cam, "mesh / camera == NULL");
    GS_REQUIRE(render_count <= m->uploaded, "render_count exceeds the uploaded splat count");
    GS_REQUIRE(!(sorted_host && sorter), "pass either host indexes or a sorter, not both");
    GS_REQUIRE(!sorter || (sorter->ctx == m->ctx && sorter->has_result && sorter->last_render >= render_count),
               "sorter has no device-resident result covering render_count (or lives on another context)");
    gs_context* ctx = m->ctx;
    ScopedDevice sd(ctx->device);
    hipStream_t st = ctx->stream;

    ProjectParams pp;
    GS_TRY(mesh_params(m, cam, pp));
    bool healed = false;
    GS_TRY(mesh_heal_overflow(m, &healed));
    // a gs_mesh_project of exactly this camera is consumed by exactly one draw (the vertex stage runs once per frame)
    bool projected = m->projection_pending && memcmp(cam, &m->projected_cam, sizeof(*cam)) == 0 && m->projected_depth_mode == pp.depth_mode;
    m->projection_pending = false;
    const uint32_t y0 = pp.y0, y1 = pp.y1;
    m->drawn_list_shift = pp.list_shift;
    m->last_pp = pp;
    const size_t out_bytes = (size_t)(y1 > y0 ? y1 - y0 : 0) * cam->width * 4;

    const uint32_t* order_dev = nullptr;
    if (sorted_host) {
        GS_TRY(m->order.ensure((size_t)m->max_count * 4));
        if (render_count) GS_HIP(hipMemcpyAsync(m->order.p, sorted_host, (size_t)render_count * 4, hipMemcpyHostToDevice, st));
        order_dev = m->order.as<uint32_t>();
    } else if (sorter) {
        order_dev = sorter->sorted.as<uint32_t>();
    }
    uint8_t* out_dev = reinterpret_cast<uint8_t*>(rgba_out_dev);
    if (!out_dev) {
        GS_TRY(m->fb.ensure(out_bytes + 16));
        out_dev = m->fb.as<uint8_t>();
    }

    m->last = gs_render_stats();
    GS_TRY(mesh_draw_once(m, pp, order_dev, sorter, render_count, out_dev, projected, stats != nullptr));
    m->has_draw = true;
    m->last_count = pp.count;

    int status = healed ? GS_WARN_FRAME_TRUNCATED : GS_OK;
    const bool need_sync = rgba_out_host || stats;
    if (need_sync) {
        // overflow check: the only host<->device round trip of a draw, and only when the caller syncs anyway
        int ov = mesh_collect_stats(m, nullptr);
        if (ov < 0) return ov;
        int guard = 0;
        while (ov == 1 && guard++ < 4) {
            uint64_t want = m->last.tile_entries + m->last.tile_entries / 8 + 1024;
            if (want > 0x7FFFFFFFull) {
                gs_set_error("tile entries (%llu) exceed the 2^31 limit", (unsigned long long)m->last.tile_entries);
                return GS_ERR_CAPACITY;
            }
            GS_TRY(mesh_alloc_entries(m, (uint32_t)want));
            GS_TRY(mesh_draw_once(m, pp, order_dev, sorter, render_count, out_dev, true, stats != nullptr));   // records still valid
            ov = mesh_collect_stats(m, nullptr);
            if (ov < 0) return ov;
            m->last.overflowed = 1;
        }
        if (ov == 1) {
            gs_set_error("tile entry buffer still overflowing after regrowth");
            return GS_ERR_CAPACITY;
        }
        if (rgba_out_host && out_bytes) {
            GS_HIP(hipMemcpyAsync(rgba_out_host, out_dev, out_bytes, hipMemcpyDeviceToHost, st));
            GS_HIP(hipStreamSynchronize(st));
        }
        if (stats) *stats = m->last;
    }
    return status;
}

int gs_mesh_debug_set_entry_capacity(gs_mesh* m, uint32_t capacity) {
    GS_REQUIRE(m != nullptr && capacity >= 1024u && capacity <= 0x7FFFFFFFu, "mesh == NULL or capacity outside [1024, 2^31)");
    ScopedDevice sd(m->ctx->device);
    GS_HIP(hipStreamSynchronize(m->ctx->stream));
    return mesh_alloc_entries(m, capacity);
}

int gs_mesh_last_stats(gs_mesh* m, gs_render_stats* stats) {
    GS_REQUIRE(m && stats, "mesh / stats == NULL");
    GS_REQUIRE(m->has_draw, "no draw has run");
    ScopedDevice sd(m->ctx->device);
    const uint32_t was_overflowed = m->last.overflowed;
    int ov = mesh_collect_stats(m, stats);
    if (ov < 0) return ov;
    stats->overflowed = was_overflowed;
    if (ov == 1) {
        // an asynchronous draw overflowed its entry buffer: grow now so the next draw fits, and tell the caller
        uint64_t want = m->last.tile_entries + m->last.tile_entries / 8 + 1024;
        if (want > 0x7FFFFFFFull) want = 0x7FFFFFFFull;
        GS_TRY(mesh_alloc_entries(m, (uint32_t)want));
        gs_set_error("tile entry buffer overflowed (%llu entries); capacity grown, redraw the frame",
                     (unsigned long long)m->last.tile_entries);
        return GS_ERR_CAPACITY;
    }
    return GS_OK;
}

int gs_mesh_kernel_time(gs_mesh* m, int which, int reset, double* sum_ms, uint32_t* launches) {
    GS_REQUIRE(m && sum_ms && launches, "mesh / outputs == NULL");
    GS_REQUIRE(which == 0 || which == 1, "unknown selector (0 = k_project alone, 1 = the whole vertex stage of timed draws)");
    ScopedDevice sd(m->ctx->device);
    GS_HIP(hipStreamSynchronize(m->ctx->stream));
    if (m->ctx->aux != m->ctx->stream) GS_HIP(hipStreamSynchronize(m->ctx->aux));
    for (int i = 0; i < gs_mesh::TIMING_RING; i++) {
        if (!m->ring_used[i]) continue;
        float ms = 0.f;
        GS_HIP(hipEventElapsedTime(&ms, m->ring0[i], m->ring1[i]));
        if (m->ring_whole[i]) { m->stage_sum_ms += ms; m->stage_launches++; }
        else { m->proj_sum_ms += ms; m->proj_launches++; }
        m->ring_used[i] = false;
    }
    *sum_ms = which == 0 ? m->proj_sum_ms : m->stage_sum_ms;
    *launches = which == 0 ? m->proj_launches : m->stage_launches;
    if (reset) {
        m->proj_sum_ms = m->stage_sum_ms = 0.0;
        m->proj_launches = m->stage_launches = 0;
        m->project_serial = 0;                              // the next launch is a measured one
    }
    return GS_OK;
}

int gs_mesh_set_draw_mode(gs_mesh* m, uint32_t mode) {
    GS_REQUIRE(m, "mesh == NULL");
    GS_REQUIRE(mode == GS_DRAW_FP32 || mode == GS_DRAW_ROP8 || mode == GS_DRAW_ROP8_FULL, "unknown draw mode");
    m->draw_mode = mode;
    return GS_OK;
}

int gs_mesh_set_deep_pass(gs_mesh* m, int enabled) {
    GS_REQUIRE(m, "mesh == NULL");
    m->no_deep = !enabled;
    return GS_OK;
}

int gs_mesh_debug_read(gs_mesh* m, int what, void* dst, uint32_t count) {
    GS_REQUIRE(m && dst, "mesh / dst == NULL");
    GS_REQUIRE(m->has_draw && (what >= 2 || count <= m->last_count), "no draw / count too large");
    ScopedDevice sd(m->ctx->device);
    hipStream_t st = m->ctx->stream;
    if (what == 0 || what == 1 || what == 3) {
        if (count == 0) return GS_OK;
        const uint32_t splats = what == 3 ? (count * 64u < m->last_count ? count * 64u : m->last_count) : count;
        const size_t bytes = what == 3 ? (size_t)count * 8 : (size_t)count * (what == 0 ? sizeof(SplatRec) : 8);
        GS_REQUIRE(what != 3 || (size_t)count * 8 <= m->vis_mask.bytes, "count exceeds the mask length");
        GS_TRY(m->staging.ensure(bytes + 64));
        if (what == 3) GS_HIP(hipMemsetAsync(m->staging.p, 0, bytes, st));
        hipLaunchKernelGGL(k_debug_expand, dim3((splats + 255u) / 256u), dim3(256), 0, st, m->vis_mask.as<unsigned long long>(), m->vis32.as<uint2>(), splats,
                           m->reorder ? m->perm.as<uint32_t>() : nullptr, m->recs.as<uint4>(), m->rects.as<uint2>(),
                           what == 0 ? m->staging.as<uint4>() : nullptr, what == 1 ? m->staging.as<uint2>() : nullptr,
                           what == 3 ? m->staging.as<unsigned long long>() : nullptr);
        GS_HIP(hipGetLastError());
        GS_HIP(hipMemcpyAsync(dst, m->staging.p, bytes, hipMemcpyDeviceToHost, st));
    } else if (what == 2) {   // [begin,end) of every tile of the last draw's strip; count = number of tiles
        GS_REQUIRE((size_t)count * 8 <= m->tile_ranges.bytes, "count exceeds the tile count of the last draw");
        GS_HIP(hipMemcpyAsync(dst, m->tile_ranges.p, (size_t)count * 8, hipMemcpyDeviceToHost, st));
    } else if (what == 4) {   // per 32-px blend bin of the last draw: {entries staged, (splat, tile) pairs walked}
        GS_REQUIRE(count <= m->blend_bins, "count exceeds the blend bins of the last draw");
        if (count) GS_HIP(hipMemcpyAsync(dst, m->blend_stats.p, (size_t)count * 8, hipMemcpyDeviceToHost, st));
    } else if (what == 5) {   // the last draw's deep pass: {bins it drew, bins over the threshold, chunk partials the per-bin kernel closed,
                              // pool exhausted, then the bin numbers}; count = words (4 .. 4 + GS_DEEP_MAX_BINS)
        GS_REQUIRE(m->deep_flags.p && count >= 4 && count <= 4u + GS_DEEP_MAX_BINS, "no draw yet / count outside 4 .. 4 + GS_DEEP_MAX_BINS");
        GS_HIP(hipMemcpyAsync(dst, m->deep_flags.as<uint32_t>(), 16, hipMemcpyDeviceToHost, st));
        if (count > 4) GS_HIP(hipMemcpyAsync(static_cast<uint32_t*>(dst) + 4, m->deep_flags.as<uint32_t>() + GS_FLAG_LIST, (size_t)(count - 4) * 4, hipMemcpyDeviceToHost, st));
    } else if (what == 6) {   // host state: {visible splats, splats projected} of the last full-frame draw whose verdict has arrived,
                              // and where the last vertex stage ran its block test (1 separate kernel, 0 per workgroup, 2 nowhere)
        GS_REQUIRE(count == 3, "count == 3");
        GS_HIP(hipStreamSynchronize(st));
        mesh_read_view_share(m);                           // (the mapped words the last draw left)
        uint32_t* w = static_cast<uint32_t*>(dst);
        w[0] = m->measured_visible; w[1] = m->measured_count; w[2] = m->last_project_mode;
        return GS_OK;
    } else GS_REQUIRE(false, "unknown debug selector");
    GS_HIP(hipStreamSynchronize(st));
    return GS_OK;
}

int gs_mesh_set_destination(gs_mesh* m, const gs_destination* dest) {
    GS_REQUIRE(m != nullptr, "mesh == NULL");
    ScopedDevice sd(m->ctx->device);
    hipStream_t st = m->ctx->stream;
    // draws in flight still read the previous destination (and, on a context with streams of its own, a pending vertex stage
    // was run for the previous depth mode)
    GS_HIP(hipStreamSynchronize(st));
    if (m->ctx->aux != st) GS_HIP(hipStreamSynchronize(m->ctx->aux));
    m->dest_depth = nullptr;
    m->dest_rgba = nullptr;
    m->dest_w = m->dest_h = m->dest_flags = 0;
    if (!dest) return GS_OK;
    GS_REQUIRE(!(dest->depth_host && dest->depth_dev), "pass the destination depth on the host OR on the device, not both");
    GS_REQUIRE(!(dest->rgba_host && dest->rgba_dev), "pass the destination colour on the host OR on the device, not both");
    GS_REQUIRE((dest->flags & ~GS_DEST_DEPTH_UNORM24) == 0, "unknown destination flags");
    const bool any = dest->depth_host || dest->depth_dev || dest->rgba_host || dest->rgba_dev;
    if (!any) return GS_OK;
    GS_REQUIRE(dest->width > 0 && dest->height > 0 && dest->width <= 4096u * GS_TILE && dest->height <= 4096u * GS_TILE, "destination size");
    const size_t px = (size_t)dest->width * dest->height;
    //  staged in locals and committed together: a refused call - an allocation or a copy that fails half way - leaves
    // the mesh WITHOUT a destination, as the header promises, not with a depth pointer and a size of 0
    const float* new_depth = nullptr;
    const uint32_t* new_rgba = nullptr;
    if (dest->depth_host) {
        GS_TRY(m->dest_depth_own.ensure(px * 4));
        GS_HIP(hipMemcpyAsync(m->dest_depth_own.p, dest->depth_host, px * 4, hipMemcpyHostToDevice, st));
        new_depth = m->dest_depth_own.as<float>();
    } else if (dest->depth_dev) {
        new_depth = reinterpret_cast<const float*>(dest->depth_dev);
    }
    if (dest->rgba_host) {
        GS_TRY(m->dest_rgba_own.ensure(px * 4));
        GS_HIP(hipMemcpyAsync(m->dest_rgba_own.p, dest->rgba_host, px * 4, hipMemcpyHostToDevice, st));
        new_rgba = m->dest_rgba_own.as<uint32_t>();
    } else if (dest->rgba_dev) {
        new_rgba = reinterpret_cast<const uint32_t*>(dest->rgba_dev);
    }
    GS_HIP(hipStreamSynchronize(st));                      // the host buffers are reusable on return
    m->dest_depth = new_depth;
    m->dest_rgba = new_rgba;
    m->dest_w = dest->width;
    m->dest_h = dest->height;
    m->dest_flags = dest->flags;
    return GS_OK;
}

int gs_mesh_debug_rop8(gs_mesh* m, uint32_t x0, uint32_t y0, uint32_t width, uint32_t height, uint8_t* rgba_out_host) {
    GS_REQUIRE(m && rgba_out_host, "mesh / out == NULL");
    GS_REQUIRE(m->has_draw, "no draw yet");
    //  on a context with two sets of vertex-stage outputs a gs_mesh_project for the NEXT frame has swapped the sets:
    // the last draw's lists would be composited against the other set's records
    GS_REQUIRE(!m->projection_pending, "a gs_mesh_project is pending: the records of the last draw are no longer current (draw first)");
    GS_REQUIRE(width > 0 && height > 0 && (uint64_t)width * height <= 65536u, "the window holds 1 .. 65536 pixels");
    const ProjectParams& pp = m->last_pp;
    GS_REQUIRE(x0 + width <= (uint32_t)pp.width && y0 >= pp.y0 && y0 + height <= pp.y1, "the window leaves the rows the last draw covered");
    //  the walk reads the mesh's CURRENT destination with the LAST draw's width / height / depth mode: a destination
    // set (or resized, or cleared) since that draw would be indexed out of bounds - and is not what the drawn frame saw
    GS_REQUIRE(m->drawn_dest_depth == m->dest_depth && m->drawn_dest_rgba == m->dest_rgba && m->drawn_dest_w == m->dest_w &&
               m->drawn_dest_h == m->dest_h && m->drawn_dest_flags == m->dest_flags,
               "the destination changed since the last draw (gs_mesh_set_destination): draw again before gs_mesh_debug_rop8");
    ScopedDevice sd(m->ctx->device);
    hipStream_t st = m->ctx->stream;
    const size_t bytes = (size_t)width * height * 4;
    GS_TRY(m->staging.ensure(bytes + 64));
    GS_TRY(gs_launch_rop8_window(m, pp, x0, y0, width, height, m->staging.as<uint32_t>()));
    GS_HIP(hipMemcpyAsync(rgba_out_host, m->staging.p, bytes, hipMemcpyDeviceToHost, st));
    GS_HIP(hipStreamSynchronize(st));
    return GS_OK;
}

}  // extern "C"
