// tile_bin.hip — order-preserving duplication of splats into 16x16-pixel screen tiles.
//
// The reference has no binning: it draws one instanced quad per splat, back to front, and lets the ROPs blend
// (/root/reference/src/splatmesh/SplatGeometry.js:11-37, SplatMaterial3D.js:65-75, src/Viewer.js:1616).  A
// tile rasteriser needs each tile's splats as a list in that same draw order, so:
//   k_bin_count   walk the sorted index list NEAR->FAR (q = R-1-p), gather each splat's tile rect (8 B),
//                 write it back in traversal order and reduce the per-workgroup entry counts
//   k_bin_scan    exclusive scan of <= 1024 workgroup sums; publishes D and min(D, capacity)
//   k_bin_emit    expands every splat into (tile id, splat index) entries, cooperatively: a 256-splat batch's
//                 entries are numbered by an LDS prefix sum and each lane finds its owner by binary search, so
//                 a splat covering 4000 tiles costs the same per entry as one covering 4 (wave64-coalesced
//                 stores, no per-thread serial loops)
//   tile sort     stable LSD radix passes on the tile id (radix.hpp) - stability keeps near->far order per tile
//   k_tile_ranges [begin,end) of every tile in the sorted entry array
// Entry count D only ever lives on the device; all downstream grids are sized for the capacity and read D there.
#include "radix.hpp"

constexpr int BIN_THREADS = 256;
constexpr int BIN_MAX_BLOCKS = 1024;

__device__ __forceinline__ uint32_t rect_tiles(uint2 r) {
    const uint32_t x0 = r.x & 0xFFFFu, y0 = r.x >> 16, x1 = r.y & 0xFFFFu, y1 = r.y >> 16;
    return (x1 >= x0 && y1 >= y0) ? (x1 - x0 + 1u) * (y1 - y0 + 1u) : 0u;
}

struct BinChunk {
    uint32_t begin, end;   // batch indices (256 splats per batch)
};
__device__ __forceinline__ BinChunk bin_chunk(uint32_t n) {
    const uint32_t batches = (n + BIN_THREADS - 1) / BIN_THREADS;
    const uint32_t per = (batches + gridDim.x - 1) / gridDim.x;
    BinChunk c;
    c.begin = min(blockIdx.x * per, batches);
    c.end = min(c.begin + per, batches);
    return c;
}

__global__ void k_render_frame_init(RenderFrame* f, uint32_t* digit_total, uint2* tile_ranges, uint32_t tiles) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0) {
        f->visible = 0; f->entries_lo = 0; f->entries_hi = 0; f->overflow = 0; f->entry_count = 0;
        f->pad[0] = f->pad[1] = f->pad[2] = 0;
    }
    if (t < RADIX_MAX_PASSES * RADIX_BINS) digit_total[t] = 0;
    for (uint32_t i = t; i < tiles; i += gridDim.x * blockDim.x) tile_ranges[i] = make_uint2(0u, 0u);
}

__global__ __launch_bounds__(BIN_THREADS) void k_bin_count(const uint32_t* __restrict__ order, uint32_t R,
                                                           const uint2* __restrict__ rects, uint2* __restrict__ rect_q,
                                                           uint32_t* __restrict__ block_sums) {
    __shared__ uint32_t s_tmp[4];
    const BinChunk ch = bin_chunk(R);
    uint32_t sum = 0, vis = 0;
    for (uint32_t b = ch.begin; b < ch.end; b++) {
        const uint32_t q = b * BIN_THREADS + threadIdx.x;
        if (q < R) {
            const uint32_t p = R - 1u - q;                     // draw order is back-to-front; we go front-to-back
            const uint32_t idx = order ? order[p] : p;
            const uint2 r = rects[idx];
            rect_q[q] = r;
            const uint32_t n = rect_tiles(r);
            sum += n;
            vis += n ? 1u : 0u;
        }
    }
    uint32_t total = 0, vtotal = 0;
    (void)block_excl_scan_256(sum, s_tmp, &total);
    (void)block_excl_scan_256(vis, s_tmp, &vtotal);
    if (threadIdx.x == 0) {
        block_sums[blockIdx.x] = total;
        block_sums[BIN_MAX_BLOCKS + blockIdx.x] = vtotal;      // second half: splats with >= 1 tile entry
    }
}

// one workgroup of 1024 threads: exclusive scan of the workgroup sums, 64-bit total
__global__ __launch_bounds__(1024) void k_bin_scan(uint32_t* __restrict__ block_sums, uint32_t grid, uint32_t capacity,
                                                   RenderFrame* frame) {
    __shared__ unsigned long long s_wave[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long v = tid < grid ? block_sums[tid] : 0ull;
    uint32_t vis = tid < grid ? block_sums[BIN_MAX_BLOCKS + tid] : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) vis += __shfl_xor(vis, o, 64);
    __shared__ uint32_t s_vis[16];
    if (lane == 0) s_vis[wave] = vis;
    unsigned long long incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long t = __shfl_up(incl, o, 64);
        if ((int)lane >= o) incl += t;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    unsigned long long base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) {
        if ((uint32_t)w < wave) base += s_wave[w];
        total += s_wave[w];
    }
    const unsigned long long excl = base + incl - v;
    if (tid < grid) block_sums[tid] = excl > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)excl;
    if (tid == 0) {
        uint32_t vsum = 0;
        for (int w = 0; w < 16; w++) vsum += s_vis[w];
        frame->visible = vsum;
        frame->entries_lo = (uint32_t)total;
        frame->entries_hi = (uint32_t)(total >> 32);
        frame->overflow = total > capacity ? 1u : 0u;
        frame->entry_count = total > capacity ? capacity : (uint32_t)total;
    }
}

template <class KeyT>
__global__ __launch_bounds__(BIN_THREADS) void k_bin_emit(const uint32_t* __restrict__ order, uint32_t R,
                                                          const uint2* __restrict__ rect_q,
                                                          const uint32_t* __restrict__ block_offsets, uint32_t capacity,
                                                          uint32_t tiles_x, uint32_t row_begin, KeyT* __restrict__ keys_out,
                                                          uint32_t* __restrict__ vals_out) {
    __shared__ uint32_t s_prefix[BIN_THREADS + 1];
    __shared__ uint2 s_rect[BIN_THREADS];
    __shared__ uint32_t s_idx[BIN_THREADS];
    __shared__ uint32_t s_tmp[4];
    const BinChunk ch = bin_chunk(R);
    const uint32_t tid = threadIdx.x;
    uint32_t base = block_offsets[blockIdx.x];
    for (uint32_t b = ch.begin; b < ch.end; b++) {
        const uint32_t q = b * BIN_THREADS + tid;
        uint2 r = make_uint2(0xFFFFu, 0u);
        uint32_t idx = 0;
        if (q < R) {
            const uint32_t p = R - 1u - q;
            idx = order ? order[p] : p;
            r = rect_q[q];
        }
        const uint32_t n = rect_tiles(r);
        uint32_t total = 0;
        const uint32_t excl = block_excl_scan_256(n, s_tmp, &total);
        s_prefix[tid] = excl;
        s_rect[tid] = r;
        s_idx[tid] = idx;
        if (tid == 0) s_prefix[BIN_THREADS] = total;
        __syncthreads();
        for (uint32_t e = tid; e < total; e += BIN_THREADS) {
            // owner = LARGEST j with s_prefix[j] <= e: empty splats share their prefix with a successor and lose
            uint32_t lo = 0, hi = BIN_THREADS;
#pragma unroll
            for (int it = 0; it < 8; it++) {
                const uint32_t mid = (lo + hi) >> 1;
                if (s_prefix[mid] <= e) lo = mid; else hi = mid;
            }
            const uint2 rr = s_rect[lo];
            const uint32_t x0 = rr.x & 0xFFFFu, y0 = rr.x >> 16, w = (rr.y & 0xFFFFu) - x0 + 1u;
            const uint32_t k = e - s_prefix[lo];
            const uint32_t dy = k / w, dx = k - dy * w;
            const uint32_t g = base + e;
            if (g < capacity && g >= base) {
                keys_out[g] = (KeyT)((y0 + dy - row_begin) * tiles_x + x0 + dx);
                vals_out[g] = s_idx[lo];
            }
        }
        base += total;
        __syncthreads();
    }
}

template <class KeyT>
__global__ __launch_bounds__(256) void k_tile_ranges(const KeyT* __restrict__ keys, const RenderFrame* frame,
                                                     uint2* __restrict__ ranges) {
    const uint32_t n = frame->entry_count;
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
        const uint32_t k = keys[e];
        if (e == 0 || (uint32_t)keys[e - 1] != k) ranges[k].x = e;
        if (e + 1 == n || (uint32_t)keys[e + 1] != k) ranges[k].y = e + 1;
    }
}

template <class KeyT>
static int binning_typed(gs_mesh* m, const ProjectParams& pp, const uint32_t* order_dev, uint32_t R, uint32_t tiles) {
    gs_context* ctx = m->ctx;
    hipStream_t st = ctx->stream;
    RenderFrame* frame = m->frame.as<RenderFrame>();
    uint32_t grid = (R + BIN_THREADS - 1) / BIN_THREADS;
    if (grid < 1) grid = 1;
    if (grid > (uint32_t)BIN_MAX_BLOCKS) grid = BIN_MAX_BLOCKS;
    const uint32_t cap = m->entry_capacity;
    hipLaunchKernelGGL(k_bin_count, dim3(grid), dim3(BIN_THREADS), 0, st, order_dev, R, m->rects.as<uint2>(),
                       m->rect_q.as<uint2>(), m->bin_sums.as<uint32_t>());
    hipLaunchKernelGGL(k_bin_scan, dim3(1), dim3(1024), 0, st, m->bin_sums.as<uint32_t>(), grid, cap, frame);
    hipLaunchKernelGGL((k_bin_emit<KeyT>), dim3(grid), dim3(BIN_THREADS), 0, st, order_dev, R, m->rect_q.as<uint2>(),
                       m->bin_sums.as<uint32_t>(), cap, pp.tiles_x, pp.row_begin, m->ekeyA.as<KeyT>(), m->evalA.as<uint32_t>());
    GS_HIP(hipGetLastError());
    GS_HIP(hipEventRecord(m->ev[2], st));

    uint32_t bits = 1;
    while ((1ull << bits) < tiles) bits++;
    const uint32_t passes = (bits + 7) / 8;
    KeyT* kbuf[2] = {m->ekeyA.as<KeyT>(), m->ekeyB.as<KeyT>()};
    uint32_t* vbuf[2] = {m->evalA.as<uint32_t>(), m->evalB.as<uint32_t>()};
    for (uint32_t p = 0; p < passes; p++) {
        ArrayLoader<KeyT> al = {kbuf[p & 1], vbuf[p & 1], &frame->entry_count, 0u};
        GS_TRY((radix_pass<ArrayLoader<KeyT>, KeyT, true>(ctx, al, al, cap, 8 * (int)p, (int)p, kbuf[(p + 1) & 1], vbuf[(p + 1) & 1])));
    }
    // after `passes` swaps the sorted entries sit in buffer (passes & 1)
    hipLaunchKernelGGL((k_tile_ranges<KeyT>), dim3(2048), dim3(256), 0, st, kbuf[passes & 1], frame, m->tile_ranges.as<uint2>());
    GS_HIP(hipGetLastError());
    m->sorted_buf = (passes & 1);            // which ping-pong buffer holds the tile-sorted entries
    return GS_OK;
}

int gs_launch_frame_init(gs_mesh* m, uint32_t tiles) {
    hipLaunchKernelGGL(k_render_frame_init, dim3(64), dim3(256), 0, m->ctx->stream, m->frame.as<RenderFrame>(),
                       m->ctx->radix.digit_total.as<uint32_t>(), m->tile_ranges.as<uint2>(), tiles);
    GS_HIP(hipGetLastError());
    return GS_OK;
}

int gs_launch_binning(gs_mesh* m, const ProjectParams& pp, const uint32_t* order_dev, uint32_t render_count) {
    const uint32_t tiles = pp.tiles_x * (pp.row_end - pp.row_begin);
    if (tiles <= 65536u) return binning_typed<uint16_t>(m, pp, order_dev, render_count, tiles);
    return binning_typed<uint32_t>(m, pp, order_dev, render_count, tiles);
}
