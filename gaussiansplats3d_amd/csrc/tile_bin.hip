// tile_bin.hip — order-preserving duplication of splats into the entry lists of the screen's list bins.
//
// The reference has no binning: it draws one instanced quad per splat, back to front, and lets the ROPs blend
// (/root/reference/src/splatmesh/SplatGeometry.js:11-37, SplatMaterial3D.js:65-75, src/Viewer.js:1616).  A
// tile rasteriser needs each tile's splats as a list in that same draw order, so:
//   k_bin_count   walk the sorted index list NEAR->FAR (q = R-1-p).  One flag per 256-splat storage block (block_any, kept as a
//                 bitmap in LDS) filters the list before anything is gathered; a position of a live block gathers ONE 8-byte
//                 word written by k_project - {visible, record slot in the block, tile rect}.  Survivors are compacted
//                 (wave64 ballot + popcount prefix) into the workgroup's own slice of a (slot, rect, offset) list, and
//                 entry counts are reduced.  The kernel runs at the machine's random-gather rate.
//   k_bin_emit    splat-centric expansion into (list bin, record slot) pairs, one lane per surviving splat, runs
//                 written at the offsets the count pass fixed; each workgroup scans the binning workgroups' counts itself
//                 (no scan kernel) and takes batches of 256 splats round-robin, so the near end of the list - whose splats
//                 touch 30x the mean number of lists - is spread over the whole grid.
//   entry sort    stable LSD radix passes on the list-bin id (radix.hpp): stability keeps near->far order per list;
//                 one pass for <= 256 lists (1080p at 128-px lists), and the pass publishes every list's [begin,end)
// Entry count D only ever lives on the device; downstream grids are sized for the capacity and read D there.
#include <stdlib.h>

#include "radix.hpp"

constexpr int BIN_THREADS = 256;
#ifndef BIN_PER_LANE
#define BIN_PER_LANE 4
#endif
#ifndef BIN_MAX_BLOCKS_CFG
#define BIN_MAX_BLOCKS_CFG 2048
#endif
constexpr int BIN_MAX_BLOCKS = BIN_MAX_BLOCKS_CFG;        // 8 workgroups of 256 per CU: full wave occupancy
constexpr uint32_t ANY_WORDS = 2048;                      // coarse visibility bits kept in LDS: 65536 blocks = 16.7 M splats

// vertex-stage rects are in 16-px tiles; the entry lists are per list bin of (16 << list_shift) px
__device__ __forceinline__ uint32_t rect_tiles(uint2 r) {
    const uint32_t x0 = r.x & 0xFFFFu, y0 = r.x >> 16, x1 = r.y & 0xFFFFu, y1 = r.y >> 16;
    return (x1 >= x0 && y1 >= y0) ? (x1 - x0 + 1u) * (y1 - y0 + 1u) : 0u;
}
// 16-px tile rect -> list-bin rect: per-field shift of (x | y << 16) with the bits that cross the field boundary masked
__device__ __forceinline__ uint2 rect_to_bins(uint2 r, uint32_t list_shift) {
    const uint32_t field = 0xFFFFu >> list_shift, mask = field | (field << 16);
    return make_uint2((r.x >> list_shift) & mask, (r.y >> list_shift) & mask);
}

struct BinChunk {
    uint32_t begin, end;   // batch indices (256 list positions per batch)
    uint32_t id;           // chunk index: workgroups of one XCD walk neighbouring chunks of the sorted list (radix.hpp, xcd_chunk)
};
__device__ __forceinline__ BinChunk bin_chunk(uint32_t n) {
    const uint32_t batches = (n + BIN_THREADS - 1) / BIN_THREADS;
    const uint32_t per = (batches + gridDim.x - 1) / gridDim.x;
    BinChunk c;
    c.id = xcd_chunk(blockIdx.x, gridDim.x);
    c.begin = min(c.id * per, batches);
    c.end = min(c.begin + per, batches);
    return c;
}

// block_sums layout: [0,BIN_MAX_BLOCKS) entries per workgroup | [BIN_MAX_BLOCKS, 2*BIN_MAX_BLOCKS) compacted (visible) splats per
// workgroup | [2*BIN_MAX_BLOCKS, 3*BIN_MAX_BLOCKS) 16-px tiles touched (statistics)
// Each lane owns 4 consecutive list positions per iteration, so 4 index loads, then 4 mask look-ups, then up to 4
// rect gathers are in flight together (the kernel is a chain of dependent memory round trips), and one packed
// 64-bit block scan per 1024 positions yields both the compaction slot and the entry offset.
#ifdef GS_BIN_PROFILE
// tools/bin_profile.py: per workgroup of k_bin_count [0] / k_bin_emit [1]: {start, after the prologue, end} of the 100 MHz
// clock and the workgroup's output count
__device__ unsigned long long g_bin_prof[2 * 2048 * 4];
extern "C" int gs_debug_bin_prof(void* dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_bin_prof), sizeof(unsigned long long) * 2 * 2048 * 4, 0, hipMemcpyDeviceToHost);
}
#define BIN_PROF(k, slot, v) do { if (threadIdx.x == 0 && blockIdx.x < 2048u) g_bin_prof[((k) * 2048 + blockIdx.x) * 4 + (slot)] = (v); } while (0)
#else
#define BIN_PROF(k, slot, v) do { } while (0)
#endif

// The count pass over one workgroup's slice [pos_begin, pos_end) of the near->far walk: filter, one gather per position of a live
// block, compaction into the slice of (slot, rect, first entry) lists.  Shared by k_bin_count and k_bin_fused.
struct SliceCount {
    uint32_t entries, splats, t16;       // entries emitted by the slice | compacted (visible) splats | this LANE's 16-px tiles (statistics)
};
__device__ __forceinline__ SliceCount bin_count_slice(const uint32_t* __restrict__ order, uint32_t R, const uint32_t* __restrict__ perm,
                                                      const uint2* __restrict__ prect, uint32_t* __restrict__ cidx, uint2* __restrict__ crect,
                                                      uint32_t* __restrict__ coff, uint32_t list_shift, uint32_t splat_count,
                                                      const uint8_t* __restrict__ block_any, bool coarse, const uint32_t* s_any,
                                                      unsigned long long* s_w, uint32_t pos_begin, uint32_t pos_end) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t sum = 0;                                      // entries emitted so far by this workgroup
    uint32_t out = pos_begin;                              // next slot of this workgroup's compacted slice
    uint32_t t16 = 0;                                      // 16-px tiles touched by this lane's splats (statistics only)
    for (uint32_t pos0 = pos_begin; pos0 < pos_end; pos0 += BIN_PER_LANE * BIN_THREADS) {
        const uint32_t q0 = pos0 + (uint32_t)BIN_PER_LANE * threadIdx.x;
        uint32_t idx[BIN_PER_LANE];
        bool keep[BIN_PER_LANE];
        uint2 r[BIN_PER_LANE];
#pragma unroll
        for (int k = 0; k < BIN_PER_LANE; k++) {
            const uint32_t q = q0 + k;
            keep[k] = q < pos_end;
            const uint32_t p = R - 1u - min(q, R - 1u);    // draw order is back-to-front; we go front-to-back
            idx[k] = order ? order[p] : p;
            // a stale or wrong caller list must not fault the GPU (WebGL's texelFetch out of range is harmless too):
            // entries beyond the uploaded splats draw nothing
            keep[k] = keep[k] && idx[k] < splat_count;
            idx[k] = min(idx[k], splat_count - 1u);
        }
        if (perm) {                                        // caller's splat index -> internal (Morton) position
#pragma unroll
            for (int k = 0; k < BIN_PER_LANE; k++) idx[k] = perm[idx[k]];
        }
        // k_project compacts survivors inside their 256-splat block and leaves, per position of a live block, one 8-byte word
        // {visible, slot in the block, tile rect}: a single gather gives the filter, the slot and the rect
        if (coarse) {
#pragma unroll
            for (int k = 0; k < BIN_PER_LANE; k++) keep[k] = keep[k] && ((s_any[idx[k] >> 13] >> ((idx[k] >> 8) & 31u)) & 1u);
        } else if (block_any) {                            // more blocks than the LDS bitmap holds: the flag byte itself
#pragma unroll
            for (int k = 0; k < BIN_PER_LANE; k++) keep[k] = keep[k] && block_any[idx[k] >> 8] != 0;
        }
        uint32_t slot[BIN_PER_LANE];
#pragma unroll
        for (int k = 0; k < BIN_PER_LANE; k++) {
            const uint2 w = keep[k] ? prect[idx[k]] : make_uint2(0u, 0u);
            keep[k] = keep[k] && ((w.y >> 24) & 1u);
            slot[k] = (idx[k] & ~255u) + (w.x >> 24);
            r[k] = keep[k] ? make_uint2((w.x & 0xFFFu) | (((w.x >> 12) & 0xFFFu) << 16), (w.y & 0xFFFu) | (((w.y >> 12) & 0xFFFu) << 16))
                           : make_uint2(0xFFFFu, 0u);
        }
#pragma unroll
        for (int k = 0; k < BIN_PER_LANE; k++) t16 += rect_tiles(r[k]);
        uint32_t n[BIN_PER_LANE], cnt = 0, ent = 0;
#pragma unroll
        for (int k = 0; k < BIN_PER_LANE; k++) {
            n[k] = keep[k] ? rect_tiles(rect_to_bins(r[k], list_shift)) : 0u;   // entries = list bins touched
            cnt += keep[k] ? 1u : 0u;
            ent += n[k];
        }
        // packed inclusive scan: high word = compacted splats, low word = tile entries (both < 2^31 per workgroup)
        const unsigned long long mine = ((unsigned long long)cnt << 32) | ent;
        unsigned long long incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long t = __shfl_up(incl, o, 64);
            if ((int)lane >= o) incl += t;
        }
        if (lane == 63) s_w[wave] = incl;
        __syncthreads();
        unsigned long long base = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const unsigned long long c = s_w[w];
            base += ((uint32_t)w < wave) ? c : 0ull;
            total += c;
        }
        const unsigned long long excl = base + incl - mine;
        uint32_t o = out + (uint32_t)(excl >> 32);
        uint32_t e = sum + (uint32_t)excl;
#pragma unroll
        for (int k = 0; k < BIN_PER_LANE; k++) {
            if (keep[k]) {
                cidx[o] = slot[k];                         // what the blend gathers records by
                crect[o] = r[k];
                coff[o] = e;                               // first entry slot of this splat, relative to the workgroup
                o++;
                e += n[k];
            }
        }
        out += (uint32_t)(total >> 32);
        sum += (uint32_t)total;
        __syncthreads();
    }
    SliceCount sc;
    sc.entries = sum; sc.splats = out - pos_begin; sc.t16 = t16;
    return sc;
}

// (waves_per_eu 8: at 62 VGPRs the compiler reports 8 waves per SIMD, yet only 7 workgroups per CU became resident and the last
// 256 of the 2048 started 10 us late - r02v timeline; with the attribute all start together: span 35.4 -> 33.0 us.
// Dealing spans of 1024 positions round-robin instead of one contiguous chunk per workgroup, with a scan kernel between count
// and emit, was tried as well: the entries per workgroup even out (max 1841 vs 3681) but the body time does not - 24.6 us max
// either way, it is a chain of loaded memory round trips - and the extra kernel makes the C3 frame 3 us slower.)
__global__ __launch_bounds__(BIN_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_bin_count(const uint32_t* __restrict__ order, uint32_t R_host,
                                                           const uint32_t* __restrict__ R_dev /* nullable */,
                                                           const uint32_t* __restrict__ perm,
                                                           const uint2* __restrict__ prect,
                                                           uint32_t* __restrict__ cidx,
                                                           uint2* __restrict__ crect, uint32_t* __restrict__ coff,
                                                           uint32_t* __restrict__ block_sums,
                                                           uint32_t* __restrict__ digit_total,
                                                           uint2* __restrict__ tile_ranges, uint32_t tiles, uint32_t list_shift,
                                                           uint32_t splat_count, const uint8_t* __restrict__ block_any,
                                                           uint32_t* __restrict__ deep_flags, uint32_t blend_bins) {
    __shared__ unsigned long long s_w[4];
    // Coarse visibility, one bit per 256-splat storage block, in LDS.  75 % of a scene's splats draw nothing and, stored along
    // a Morton curve, mostly whole blocks of them; an LDS bit test spares those list positions the 8-byte L2 gather of their
    // visibility word (5.8 M random L2 transactions per frame were this kernel's real cost: making the rects dense did nothing)
    __shared__ uint32_t s_any[ANY_WORDS];
    BIN_PROF(0, 0, wall_clock64());
    const uint32_t blocks = (splat_count + 255u) >> 8;
    const bool coarse = block_any != nullptr && blocks <= ANY_WORDS * 32u;
    if (coarse) {
        for (uint32_t w = threadIdx.x; w < (blocks + 31u) / 32u; w += BIN_THREADS) {
            const uint4* src = reinterpret_cast<const uint4*>(block_any + 32u * w);      // the buffer is padded to 64 bytes
            const uint4 a = src[0], b = src[1];
            const uint32_t v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            uint32_t bits = 0;
#pragma unroll
            for (int k = 0; k < 8; k++)                                                   // 4 flag bytes per word -> 4 bits
                bits |= (((v[k] & 0xFFu) ? 1u : 0u) | ((v[k] & 0xFF00u) ? 2u : 0u) | ((v[k] & 0xFF0000u) ? 4u : 0u) |
                         ((v[k] & 0xFF000000u) ? 8u : 0u)) << (4 * k);
            s_any[w] = bits;
        }
        __syncthreads();
    }
    // The draw's housekeeping (no separate init kernel; these tables are idle now): zero the group rows of every entry-sort
    // pass, reset the bin ranges.
    {
        const uint32_t t = blockIdx.x * BIN_THREADS + threadIdx.x, stride = gridDim.x * BIN_THREADS;
        for (uint32_t w = t; w < (uint32_t)RADIX_TOTAL_WORDS; w += stride) digit_total[w] = 0u;
        for (uint32_t w = t; w < tiles; w += stride) tile_ranges[w] = make_uint2(0xFFFFFFFFu, 0u);
        // the chunked composite's per-draw words: no deep bins (k_bin_emit names them), an empty partial pool
        if (t < GS_FLAG_LIST) deep_flags[t] = 0u;
        for (uint32_t w = t; w < blend_bins; w += stride) deep_flags[GS_FLAG_OF + w] = GS_DEEP_NONE;
    }
    // the grid is sized for the host's count; a list whose real length only exists on the device (frustum-culled sort)
    // is spread over the same grid, and k_bin_emit learns the batches per workgroup from block_sums[3*BIN_MAX_BLOCKS]
    const uint32_t R = R_dev ? min(*R_dev, R_host) : R_host;
    const BinChunk ch = bin_chunk(R);
    BIN_PROF(0, 1, wall_clock64());
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const uint32_t batches = (R + BIN_THREADS - 1) / BIN_THREADS;
        block_sums[3 * BIN_MAX_BLOCKS] = (batches + gridDim.x - 1) / gridDim.x;      // = bin_chunk()'s `per`
    }
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t pos_begin = ch.begin * BIN_THREADS, pos_end = min(ch.end * BIN_THREADS, R);
    const SliceCount sc = bin_count_slice(order, R, perm, prect, cidx, crect, coff, list_shift, splat_count, block_any, coarse, s_any, s_w, pos_begin, pos_end);
    uint32_t t16 = sc.t16;
    const uint32_t sum = sc.entries, out = pos_begin + sc.splats;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t16 += __shfl_xor(t16, o, 64);
    if (lane == 0) s_w[wave] = t16;                        // the loop's trailing barrier makes s_w reusable
    __syncthreads();
    if (threadIdx.x == 0) {
        block_sums[ch.id] = sum;
        block_sums[BIN_MAX_BLOCKS + ch.id] = out - pos_begin;
        block_sums[2 * BIN_MAX_BLOCKS + ch.id] = (uint32_t)(s_w[0] + s_w[1] + s_w[2] + s_w[3]);
    }
    BIN_PROF(0, 2, wall_clock64());
    BIN_PROF(0, 3, (unsigned long long)sum);
}

// One batch of <= 256 compacted splats of a slice -> (list bin, record slot) entries from slot `base` on: a lane writes the first
// EMIT_OWN entries of its own splat walking the rect row-major, what is left of the few splats that cover more list bins is written
// by the whole wave.  src: this thread's position in the compacted lists; live: it holds a splat.  Returns the lane's entry count.
constexpr uint32_t EMIT_OWN = 16;      // entries a lane writes for its own splat; longer runs are shared by the wave
template <class KeyT>
__device__ __forceinline__ uint32_t bin_emit_batch(const uint32_t* __restrict__ cidx, const uint2* __restrict__ crect, const uint32_t* __restrict__ coff,
                                                   uint32_t src, bool live, uint32_t base, uint32_t limit, uint32_t tiles_x, uint32_t row_begin,
                                                   uint32_t list_shift, KeyT* __restrict__ keys_out, uint32_t* __restrict__ vals_out) {
    const uint32_t lane = threadIdx.x & 63u;
    auto put = [&](uint32_t e, uint32_t key, uint32_t idx) {
        if (e >= limit) return;                              // dropped by an overflowing draw (it is redone)
        keys_out[e] = (KeyT)key;
        vals_out[e] = idx;
    };
    uint2 r = make_uint2(0u, 0u);
    uint32_t idx = 0, e0 = 0, n = 0;
    if (live) {
        r = rect_to_bins(crect[src], list_shift);                    // the list bins the splat touches
        idx = cidx[src];
        e0 = base + coff[src];
        n = rect_tiles(r);
    }
    const uint32_t x0 = r.x & 0xFFFFu, y0 = r.x >> 16, w = (r.y & 0xFFFFu) - x0 + 1u;
    for (uint32_t k = 0, xx = x0, yy = y0; k < min(n, EMIT_OWN); k++) {     // row-major walk of the rect, no k / w, k % w
        put(e0 + k, (yy - row_begin) * tiles_x + xx, idx);
        if (++xx == x0 + w) { xx = x0; yy++; }
    }
    // the few near splats that cover many lists: all 64 lanes write one splat's remaining entries together
    unsigned long long big = __ballot(n > EMIT_OWN);
    while (big) {
        const int sl = __builtin_ctzll(big);
        big &= big - 1ull;
        const uint32_t bn = __shfl(n, sl, 64), be0 = __shfl(e0, sl, 64), bidx = __shfl(idx, sl, 64);
        const uint32_t bx0 = __shfl(x0, sl, 64), by0 = __shfl(y0, sl, 64), bw = __shfl(w, sl, 64);
        const float inv_w = 1.0f / (float)bw;
        for (uint32_t k = EMIT_OWN + lane; k < bn; k += 64u) {
            uint32_t q = (uint32_t)((float)k * inv_w);               // k / bw: k < 2^24, so the estimate is off by <= 1
            int32_t rem = (int32_t)(k - q * bw);
            if (rem < 0) { q--; rem += (int32_t)bw; }
            else if (rem >= (int32_t)bw) { q++; rem -= (int32_t)bw; }
            put(be0 + k, (by0 + q - row_begin) * tiles_x + bx0 + (uint32_t)rem, bidx);
        }
    }
    return n;
}

// The previous draw's per-bin statistics describe ITS view.  When the camera has moved since, the host says how far the scene's centre
// of mass moved on screen, in bins (binning_typed): the schedule reads the statistics of the bin a bin's content came FROM - a camera
// that turns about its own position shifts the whole picture, and with it the costly bins and the deep pass's members, by that much.
struct StatShift {
    int32_t sx, sy;            // this draw's bin (bx, by) shows what the previous draw's bin (bx - sx, by - sy) showed
    uint32_t bins_x;
};
// The blend's schedule and the deep pass's members, one workgroup of BIN_THREADS (see where it is called: k_bin_emit / k_bin_fused)
__device__ __forceinline__ void blend_schedule_job(const uint2* __restrict__ prev_blend_stats, uint32_t blend_bins, uint32_t* __restrict__ blend_order,
                                                   uint32_t deep, uint32_t* __restrict__ deep_flags, uint32_t* __restrict__ blend_stats_w,
                                                   uint32_t deep_min, uint32_t deep_factor, volatile uint32_t* __restrict__ mirror, StatShift sh) {
    __shared__ uint32_t s_cost[RADIX_BINS], s_tmp2[4];
    // the chunked composite's per-draw words: no deep bins yet, an empty partial pool (k_bin_count resets them as well; in the fused
    // launch this workgroup runs BESIDE the counting workgroups, so the reset has to be its own)
    if (threadIdx.x < GS_FLAG_LIST) deep_flags[threadIdx.x] = 0u;
    for (uint32_t w = threadIdx.x; w < blend_bins; w += BIN_THREADS) deep_flags[GS_FLAG_OF + w] = GS_DEEP_NONE;
    __threadfence_block();
    __syncthreads();
    // three sweeps over the statistics, 8 loads in flight per lane (registers for all 8192 / 256 values would set the
    // whole kernel's VGPR allocation and cost every emitting workgroup its occupancy)
    constexpr uint32_t SWEEP = 8;
    auto cost_of = [&](uint32_t i) -> uint32_t {
        if (sh.sx == 0 && sh.sy == 0) return prev_blend_stats[i].y;
        const uint32_t by = i / sh.bins_x, bx = i - by * sh.bins_x;
        const int32_t fx = (int32_t)bx - sh.sx, fy = (int32_t)by - sh.sy;
        const uint32_t from = (uint32_t)fy * sh.bins_x + (uint32_t)fx;
        return (fx >= 0 && fy >= 0 && (uint32_t)fx < sh.bins_x && from < blend_bins) ? prev_blend_stats[from].y : 0u;
    };
    auto sweep = [&](auto&& use) {
        for (uint32_t base = 0; base < blend_bins; base += SWEEP * BIN_THREADS) {
            uint32_t c[SWEEP];
#pragma unroll
            for (uint32_t k = 0; k < SWEEP; k++) {
                const uint32_t i = base + threadIdx.x + k * BIN_THREADS;
                c[k] = i < blend_bins ? cost_of(i) : 0u;
            }
#pragma unroll
            for (uint32_t k = 0; k < SWEEP; k++) {
                const uint32_t i = base + threadIdx.x + k * BIN_THREADS;
                if (i < blend_bins) use(i, c[k]);
            }
        }
    };
    uint32_t sum = 0;
    sweep([&](uint32_t, uint32_t c) { sum += c; });
    uint32_t total_walked;
    (void)block_excl_scan_256(sum, s_tmp2, &total_walked);
    // 8-bit cost key scaled to the scene: the mean bin lands near 48 whatever the scene walks per bin
    uint32_t shift = 0;
    while (((total_walked / blend_bins) >> shift) > 48u) shift++;
    s_cost[threadIdx.x] = 0u;
    __syncthreads();
    sweep([&](uint32_t, uint32_t c) { atomicAdd(&s_cost[255u - min(c >> shift, 255u)], 1u); });
    __syncthreads();
    const uint32_t cnt_d = s_cost[threadIdx.x];
    const uint32_t start = block_excl_scan_256(cnt_d, s_tmp2, nullptr);
    s_cost[threadIdx.x] = start;
    __syncthreads();
    sweep([&](uint32_t i, uint32_t c) { blend_order[atomicAdd(&s_cost[255u - min(c >> shift, 255u)], 1u)] = i; });
    // The deep pass (tile_blend.hip): among the GS_DEEP_MAX_BINS costliest bins (the head of the order just written), the ones
    // whose previous draw walked >= deep_min (splat, quadrant) pairs are composited by one wave per (quadrant, chunk) + a fold
    // instead of by one workgroup - same pixels either way, so this is scheduling only.  Their count goes to the host (the NEXT
    // draw launches the deep pass when it is non-zero); they are only named when THIS draw runs the pass.  Their statistics
    // are then summed atomically by many waves: zeroed here.
    // (a bin qualifies when it walked >= deep_min pairs AND >= deep_factor x the mean bin: the pass costs three launches, and
    // only a tail that is long against the body of the frame pays for them - C3T's costliest bins are 3 x its mean and gain
    // nothing, C3S's are 18 x; .y = half quadrants evaluated = 2 per pair, total_walked is their sum)
    // Once the pass runs, more bins in it cost next to nothing, and every bin left behind is a workgroup that walks alone at the
    // end of the launch: membership starts at 3/4 of the mean (and never below deep_min).
    const uint32_t mean_halves = total_walked / max(blend_bins, 1u);
    const uint32_t deep_trigger = max(2u * deep_min, deep_factor * mean_halves), deep_thr = max(2u * deep_min, mean_halves - mean_halves / 4u);
    __shared__ uint32_t s_trigger;
    if (threadIdx.x == 0) s_trigger = 0u;
    __shared__ uint32_t s_deep_n, s_deep_cost;
    if (threadIdx.x == 0) { s_deep_n = 0u; s_deep_cost = 0u; }
    __threadfence_block();
    __syncthreads();                                   // blend_order complete (and visible to this workgroup)
    for (uint32_t p = threadIdx.x; p < min(blend_bins, GS_DEEP_MAX_BINS); p += BIN_THREADS) {
        const uint32_t i = __hip_atomic_load(&blend_order[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint32_t cost = i < blend_bins ? cost_of(i) : 0u;
        if (cost >= deep_trigger) s_trigger = 1u;
        if (cost >= deep_thr) {
            const uint32_t k = atomicAdd(&s_deep_n, 1u);
            atomicAdd(&s_deep_cost, cost >> 4);            // (the members' share of the frame's walk: sixteenths, < 2^32 for any frame)
            if (deep) {
                deep_flags[GS_FLAG_LIST + k] = i;
                deep_flags[GS_FLAG_OF + i] = k;
                blend_stats_w[2u * i] = 0u; blend_stats_w[2u * i + 1u] = 0u; blend_stats_w[2u * blend_bins + i] = 0u;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // (a draw whose tail no longer reaches the trigger still runs the pass it was launched with - on the bins of the
        // membership rule - and tells the next one to stop)
        deep_flags[GS_FLAG_CAND] = s_trigger ? s_deep_n : 0u;
        if (deep) deep_flags[GS_FLAG_COUNT] = s_deep_n;
        if (mirror) {
            mirror[4] = s_trigger ? s_deep_n : 0u;
            // ... and what share of the previous draw's walk those bins were, in 1 / 1024: how many of the blend's workgroups the pass
            // should get (tile_blend.hip, gs_launch_blend)
            mirror[5] = (uint32_t)(((unsigned long long)s_deep_cost << 14) / (unsigned long long)max(total_walked, 1u));
        }
    }
}

// k_bin_emit: splat-centric expansion of the compacted list into (list bin, record slot) pairs.
// Work unit = one batch of 256 compacted splats of one binning workgroup's slice; the units are dealt round-robin to the
// workgroups.  (One workgroup per binning workgroup, the r01 shape, lasted as long as its heaviest slice: the slices that hold
// the nearest splats carry 30x the mean entry count - r02n timeline: mean workgroup 8 us, last one 27 us, and the same 27 us
// for a strip-sharded rank that emits an eighth of the entries.)  A lane writes the first OWN entries of its splat itself,
// walking the rect row-major; what is left of the few splats that cover more list bins is written by the whole wave.
// block_sums: [0,BIN_MAX_BLOCKS) entries of every binning workgroup | [BIN_MAX_BLOCKS,..) its compacted splat count |
// [2*BIN_MAX_BLOCKS,..) its 16-px tiles | [3*BIN_MAX_BLOCKS] batches per binning workgroup.
// Every workgroup scans the binning workgroups' sums itself (8 KB of hot L2 lines: cheaper than a one-workgroup scan kernel
// and its two kernel boundaries); workgroup 0 publishes the RenderFrame scalars that the following kernels read.
template <class KeyT>
__global__ __launch_bounds__(BIN_THREADS) void k_bin_emit(RenderFrame* __restrict__ frame, uint32_t capacity,
                                                          const uint32_t* __restrict__ cidx,
                                                          const uint2* __restrict__ crect, const uint32_t* __restrict__ coff,
                                                          const uint32_t* __restrict__ block_sums, uint32_t bin_grid,
                                                          uint32_t tiles_x /* list bins per row */,
                                                          uint32_t row_begin /* first list-bin row */, KeyT* __restrict__ keys_out,
                                                          uint32_t* __restrict__ vals_out, uint32_t list_shift,
                                                          volatile uint32_t* __restrict__ mirror, uint32_t serial,
                                                          const uint2* __restrict__ prev_blend_stats, uint32_t blend_bins,
                                                          uint32_t* __restrict__ blend_order, uint32_t deep,
                                                          uint32_t* __restrict__ deep_flags, uint32_t* __restrict__ blend_stats_w,
                                                          uint32_t deep_min, uint32_t deep_factor, StatShift sh) {
    __shared__ __attribute__((aligned(16))) uint32_t s_eoff[BIN_MAX_BLOCKS];   // entries of the binning workgroups before b (saturating)
    __shared__ __attribute__((aligned(16))) uint32_t s_cnt[BIN_MAX_BLOCKS];    // compacted splats of binning workgroup b
    __shared__ unsigned long long s_wsum[4], s_t16[4];
    __shared__ uint32_t s_vis[4];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    BIN_PROF(1, 0, wall_clock64());
    BIN_PROF(1, 2, 0ull);
    // Blend schedule of THIS draw, by one extra workgroup (the first to be dispatched): the 32-px bins in descending order of
    // what they cost in the PREVIOUS draw ((splat, tile) pairs walked, k_tile_blend's per-bin statistics).  Every bin of a 1080p
    // frame is resident at once and a SIMD's time is the sum of its waves' walks, so the kernel used to last ~2x its mean bin
    // (r02g counters: VALU busy 39 % of the launch, ~all of the time while 8 waves are resident); heavy bins first + fewer
    // resident workgroups lets the cheap ones backfill.  A counting sort of <= 8 k keys held in registers; the order of equal
    // keys is irrelevant (pixels do not depend on which workgroup draws a bin).
    const uint32_t first_wg = blend_order ? 1u : 0u;
    if (blend_order && blockIdx.x == 0) {
        blend_schedule_job(prev_blend_stats, blend_bins, blend_order, deep, deep_flags, blend_stats_w, deep_min, deep_factor, mirror, sh);
        BIN_PROF(1, 1, wall_clock64());
        BIN_PROF(1, 2, wall_clock64());
        return;
    }
    const uint32_t wg = blockIdx.x - first_wg, wgs = gridDim.x - first_wg;
    // exclusive scan of the binning workgroups' entry sums: thread t owns the PER_T consecutive sums [t * PER_T, ..), read
    // with 16-byte loads (block_sums rows are 16-byte aligned)
    constexpr uint32_t PER_T = BIN_MAX_BLOCKS / BIN_THREADS;
    static_assert(PER_T % 4 == 0, "16-byte loads of the workgroup sums");
    uint32_t mine[PER_T];
    unsigned long long own = 0;
#pragma unroll
    for (uint32_t q = 0; q < PER_T / 4; q++) {
        const uint32_t i0 = threadIdx.x * PER_T + 4 * q;
        const uint4 e = reinterpret_cast<const uint4*>(block_sums)[i0 / 4];
        uint4 c = reinterpret_cast<const uint4*>(block_sums + BIN_MAX_BLOCKS)[i0 / 4];
        mine[4 * q + 0] = i0 + 0 < bin_grid ? e.x : 0u;
        mine[4 * q + 1] = i0 + 1 < bin_grid ? e.y : 0u;
        mine[4 * q + 2] = i0 + 2 < bin_grid ? e.z : 0u;
        mine[4 * q + 3] = i0 + 3 < bin_grid ? e.w : 0u;
        c.x = i0 + 0 < bin_grid ? c.x : 0u;
        c.y = i0 + 1 < bin_grid ? c.y : 0u;
        c.z = i0 + 2 < bin_grid ? c.z : 0u;
        c.w = i0 + 3 < bin_grid ? c.w : 0u;
        reinterpret_cast<uint4*>(s_cnt)[i0 / 4] = c;
    }
#pragma unroll
    for (uint32_t k = 0; k < PER_T; k++) own += mine[k];
    unsigned long long incl = own;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long t = __shfl_up(incl, o, 64);
        if ((int)lane >= o) incl += t;
    }
    if (lane == 63u) s_wsum[wave] = incl;
    __syncthreads();
    unsigned long long run = incl - own, D64 = 0;
#pragma unroll
    for (uint32_t w = 0; w < 4; w++) {
        const unsigned long long c = s_wsum[w];
        run += w < wave ? c : 0ull;
        D64 += c;
    }
#pragma unroll
    for (uint32_t q = 0; q < PER_T / 4; q++) {
        uint32_t v[4];
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) {
            v[k] = run > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)run;
            run += mine[4 * q + k];
        }
        reinterpret_cast<uint4*>(s_eoff)[(threadIdx.x * PER_T + 4 * q) / 4] = make_uint4(v[0], v[1], v[2], v[3]);
    }
    const uint32_t D = D64 > capacity ? capacity : (uint32_t)D64;
    if (wg == 0) {                                           // the frame's scalars (read by the sort passes and the host)
        unsigned long long t16 = 0;
        uint32_t vis = 0;
        for (uint32_t i = threadIdx.x; i < bin_grid; i += BIN_THREADS) {
            vis += block_sums[BIN_MAX_BLOCKS + i];
            t16 += block_sums[2 * BIN_MAX_BLOCKS + i];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            vis += __shfl_xor(vis, o, 64);
            t16 += __shfl_xor(t16, o, 64);
        }
        if (lane == 0u) { s_t16[wave] = t16; s_vis[wave] = vis; }
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned long long tsum = s_t16[0] + s_t16[1] + s_t16[2] + s_t16[3];
            frame->tiles16_lo = (uint32_t)tsum;
            frame->tiles16_hi = (uint32_t)(tsum >> 32);
            frame->visible = s_vis[0] + s_vis[1] + s_vis[2] + s_vis[3];
            frame->entries_lo = (uint32_t)D64;
            frame->entries_hi = (uint32_t)(D64 >> 32);
            frame->overflow = D64 > capacity ? 1u : 0u;
            frame->entry_count = D;
            frame->pad = 0;
            // host-visible copy of the overflow verdict (mapped pinned memory): an asynchronous draw that ran out of entry
            // slots is noticed by the NEXT gs_mesh_render without a synchronisation (mesh.hip, mesh_heal_overflow).
            // One 16-byte store = one PCIe write: the four words land together, no system-scope fence between them
            if (mirror) {
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 v = {serial, D64 > capacity ? 1u : 0u, (uint32_t)D64, (uint32_t)(D64 >> 32)};
                *reinterpret_cast<volatile u32x4*>(mirror) = v;
                // ... and, as a second 16-byte store, {serial, visible splats, 16-px tiles they touch}: what tells the following draws how
                // much of the scene is in view (project.hip: where the block test runs) and how large its splats are on screen (the size
                // of the list bins, mesh.hip: mesh_adapt_list_bins) without anybody asking for statistics
                const u32x4 sv = {serial, s_vis[0] + s_vis[1] + s_vis[2] + s_vis[3], (uint32_t)tsum, (uint32_t)(tsum >> 32)};
                *reinterpret_cast<volatile u32x4*>(mirror + 8) = sv;
            }
        }
    }
    __syncthreads();                                         // s_eoff complete
    BIN_PROF(1, 1, wall_clock64());
    const uint32_t per = block_sums[3 * BIN_MAX_BLOCKS];     // batches per binning workgroup (k_bin_count)
    const uint32_t units = bin_grid * per;
    uint32_t emitted = 0;
    for (uint32_t u = wg; u < units; u += wgs) {
        const uint32_t b = u / per, jb = (u - b * per) * BIN_THREADS;    // batch jb / 256 of binning workgroup b (uniform)
        const uint32_t cnt = s_cnt[b], boff = s_eoff[b];
        if (jb >= cnt || boff >= D) continue;
        const uint32_t src = b * per * BIN_THREADS + jb + threadIdx.x;   // the workgroup's slice of the compacted list
        emitted += bin_emit_batch<KeyT>(cidx, crect, coff, src, jb + threadIdx.x < cnt, boff, D, tiles_x, row_begin, list_shift, keys_out, vals_out);
    }
    BIN_PROF(1, 2, wall_clock64());
    BIN_PROF(1, 3, (unsigned long long)emitted);
    (void)emitted;
}

// ---------------------------------------------------------------------------------------------------------------------------
// k_bin_fused: count + emit in ONE launch (round 6).
// k_bin_count fixes where every splat's entries go and k_bin_emit, a kernel boundary later, re-reads the compacted lists from a
// grid that first scans all 2048 slice sums again.  Here the workgroup that counted a slice also emits it, as soon as it knows how
// many entries the slices in front of it hold - a scan ACROSS the running grid:
//   * every workgroup publishes {entries, visible splats, 16-px tiles} of its slice as 8-byte granules {tag, value} (one relaxed
//     agent-scope store each: the value and its "ready" flag travel together, no fence; tag = the draw's serial, so the rows are
//     never reset);
//   * two levels: the last slice of a GROUP of 32 sums its group and publishes the group's row; a slice needs its <= 31
//     predecessors inside the group and the <= 63 group rows in front = two rounds of polls by one wave (a flat look-back over
//     2048 slices that all finish counting at the same moment would walk ~1000 granules per slice);
//   * the LAST slice knows the frame's totals and writes the RenderFrame scalars and the host mirror.
// Who waits for whom: slice c only ever waits for slices < c.  Slice ids are a function of blockIdx (xcd_chunk: monotone within
// an XCD), every XCD starts its workgroups in blockIdx order, so the smallest unfinished slice is always running or the next one
// its XCD starts: no cycle, whatever the residency (and all 2048 are resident at once on a whole MI355X: 8 per CU).  Every poll is
// bounded all the same: a poll that runs out of patience raises `fail` (the draw reports GS_ERR_HIP on the next statistics read)
// instead of hanging the device.
struct BinScan {
    unsigned long long* chunk_rows;     // [3][BIN_MAX_BLOCKS]
    unsigned long long* group_rows;     // [3][BIN_MAX_BLOCKS / BIN_SCAN_GROUP]
    uint32_t* fail;
    uint32_t tag;
};
#ifndef BIN_SCAN_SLEEP
#define BIN_SCAN_SLEEP 100              // units of 64 cycles between two polls of a wave (4 / 32 / 100: C3 bin stage 0.0642 / 0.0610 / 0.0598 ms)
#endif
constexpr uint32_t BIN_SCAN_GROUP = 32, BIN_SCAN_GROUPS = BIN_MAX_BLOCKS / BIN_SCAN_GROUP;
static_assert(BIN_SCAN_GROUP <= 64 && BIN_SCAN_GROUPS <= 64, "one lane per predecessor / per group row");
__device__ __forceinline__ void bin_scan_put(unsigned long long* p, uint32_t value, uint32_t tag) {
    __hip_atomic_store(p, ((unsigned long long)tag << 32) | value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t bin_scan_poll(const unsigned long long* p, uint32_t tag, uint32_t* fail) {
    for (uint32_t spin = 0; spin < (1u << 21); spin++) {                  // ~ 1 s with the sleeps: a wrong frame, never a hung GPU
        const unsigned long long g = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)(g >> 32) == tag) return (uint32_t)g;
        __builtin_amdgcn_s_sleep(BIN_SCAN_SLEEP);
    }
    *fail = 1u;
    return 0u;
}
__device__ __forceinline__ unsigned long long wave_sum64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <class KeyT>
__global__ __launch_bounds__(BIN_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_bin_fused(
    const uint32_t* __restrict__ order, uint32_t R_host, const uint32_t* __restrict__ R_dev /* nullable */, const uint32_t* __restrict__ perm,
    const uint2* __restrict__ prect, uint32_t* __restrict__ cidx, uint2* __restrict__ crect, uint32_t* __restrict__ coff,
    uint32_t* __restrict__ digit_total, uint2* __restrict__ tile_ranges, uint32_t tiles, uint32_t list_shift, uint32_t splat_count,
    const uint8_t* __restrict__ block_any, uint32_t* __restrict__ deep_flags, uint32_t blend_bins,
    // (k_bin_emit's)
    RenderFrame* __restrict__ frame, uint32_t capacity, uint32_t tiles_x /* list bins per row */, uint32_t row_begin /* first list-bin row */,
    KeyT* __restrict__ keys_out, uint32_t* __restrict__ vals_out, volatile uint32_t* __restrict__ mirror, uint32_t serial,
    const uint2* __restrict__ prev_blend_stats, uint32_t* __restrict__ blend_order, uint32_t deep, uint32_t* __restrict__ blend_stats_w,
    uint32_t deep_min, uint32_t deep_factor, BinScan scan, StatShift sh) {
    __shared__ unsigned long long s_w[4];
    __shared__ uint32_t s_any[ANY_WORDS];
    __shared__ uint32_t s_base;
    // (+ one workgroup, the first to be dispatched, for the blend's schedule and the deep pass's members: k_bin_emit's)
    const uint32_t first_wg = blend_order ? 1u : 0u;
    if (blend_order && blockIdx.x == 0) {
        blend_schedule_job(prev_blend_stats, blend_bins, blend_order, deep, deep_flags, blend_stats_w, deep_min, deep_factor, mirror, sh);
        return;
    }
    const uint32_t wg = blockIdx.x - first_wg, G = gridDim.x - first_wg;
    const uint32_t blocks = (splat_count + 255u) >> 8;
    const bool coarse = block_any != nullptr && blocks <= ANY_WORDS * 32u;
    if (coarse) {                                                          // (k_bin_count's LDS bitmap of live storage blocks)
        for (uint32_t w = threadIdx.x; w < (blocks + 31u) / 32u; w += BIN_THREADS) {
            const uint4* src = reinterpret_cast<const uint4*>(block_any + 32u * w);
            const uint4 a = src[0], b = src[1];
            const uint32_t v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            uint32_t bits = 0;
#pragma unroll
            for (int k = 0; k < 8; k++)
                bits |= (((v[k] & 0xFFu) ? 1u : 0u) | ((v[k] & 0xFF00u) ? 2u : 0u) | ((v[k] & 0xFF0000u) ? 4u : 0u) |
                         ((v[k] & 0xFF000000u) ? 8u : 0u)) << (4 * k);
            s_any[w] = bits;
        }
        __syncthreads();
    }
    {   // the draw's housekeeping (k_bin_count's)
        const uint32_t t = wg * BIN_THREADS + threadIdx.x, stride = G * BIN_THREADS;
        for (uint32_t w = t; w < (uint32_t)RADIX_TOTAL_WORDS; w += stride) digit_total[w] = 0u;
        for (uint32_t w = t; w < tiles; w += stride) tile_ranges[w] = make_uint2(0xFFFFFFFFu, 0u);
        if (!first_wg) {                                                   // (a schedule workgroup resets - and then fills - these itself)
            if (t < GS_FLAG_LIST) deep_flags[t] = 0u;
            for (uint32_t w = t; w < blend_bins; w += stride) deep_flags[GS_FLAG_OF + w] = GS_DEEP_NONE;
        }
    }
    const uint32_t R = R_dev ? min(*R_dev, R_host) : R_host;
    const uint32_t batches = (R + BIN_THREADS - 1) / BIN_THREADS, per = (batches + G - 1) / G;
    const uint32_t c = xcd_chunk(wg, G);
    const uint32_t b0 = min(c * per, batches), b1 = min(b0 + per, batches);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t pos_begin = b0 * BIN_THREADS, pos_end = min(b1 * BIN_THREADS, R);
    const SliceCount sc = bin_count_slice(order, R, perm, prect, cidx, crect, coff, list_shift, splat_count, block_any, coarse, s_any, s_w, pos_begin, pos_end);
    uint32_t t16 = sc.t16;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t16 += __shfl_xor(t16, o, 64);
    if (lane == 0) s_w[wave] = t16;
    __syncthreads();
    if (wave == 0u) {
        const uint32_t E = sc.entries, V = sc.splats, T = (uint32_t)(s_w[0] + s_w[1] + s_w[2] + s_w[3]);
        if (lane < 3u) bin_scan_put(scan.chunk_rows + (size_t)lane * BIN_MAX_BLOCKS + c, lane == 0u ? E : lane == 1u ? V : T, scan.tag);
        const uint32_t g = c / BIN_SCAN_GROUP, j = c % BIN_SCAN_GROUP;
        const bool closer = j == BIN_SCAN_GROUP - 1u || c == G - 1u;
        // entries in front of this slice: the slices of its group before it + the groups before its group
        const unsigned long long in_e = wave_sum64(lane < j ? bin_scan_poll(scan.chunk_rows + g * BIN_SCAN_GROUP + lane, scan.tag, scan.fail) : 0u);
        if (closer) {
            const unsigned long long ge = in_e + E;
            if (lane == 0u) bin_scan_put(scan.group_rows + g, ge > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)ge, scan.tag);
        }
        const unsigned long long front = in_e + wave_sum64(lane < g ? bin_scan_poll(scan.group_rows + lane, scan.tag, scan.fail) : 0u);
        if (lane == 0u) s_base = front > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)front;
        if (closer) {                                                      // the group's statistics rows; the last slice: the frame's totals
            const unsigned long long gv = V + wave_sum64(lane < j ? bin_scan_poll(scan.chunk_rows + BIN_MAX_BLOCKS + g * BIN_SCAN_GROUP + lane, scan.tag, scan.fail) : 0u);
            const unsigned long long gt = T + wave_sum64(lane < j ? bin_scan_poll(scan.chunk_rows + 2u * BIN_MAX_BLOCKS + g * BIN_SCAN_GROUP + lane, scan.tag, scan.fail) : 0u);
            if (c != G - 1u) {
                if (lane == 0u) {
                    bin_scan_put(scan.group_rows + BIN_SCAN_GROUPS + g, (uint32_t)gv, scan.tag);
                    bin_scan_put(scan.group_rows + 2u * BIN_SCAN_GROUPS + g, gt > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)gt, scan.tag);
                }
            } else {
                const unsigned long long vis64 = gv + wave_sum64(lane < g ? bin_scan_poll(scan.group_rows + BIN_SCAN_GROUPS + lane, scan.tag, scan.fail) : 0u);
                const unsigned long long tsum = gt + wave_sum64(lane < g ? bin_scan_poll(scan.group_rows + 2u * BIN_SCAN_GROUPS + lane, scan.tag, scan.fail) : 0u);
                const unsigned long long D64 = front + E;
                if (lane == 0u) {
                    const uint32_t vis = (uint32_t)vis64;
                    frame->tiles16_lo = (uint32_t)tsum;
                    frame->tiles16_hi = (uint32_t)(tsum >> 32);
                    frame->visible = vis;
                    frame->entries_lo = (uint32_t)D64;
                    frame->entries_hi = (uint32_t)(D64 >> 32);
                    frame->overflow = D64 > capacity ? 1u : 0u;
                    frame->entry_count = D64 > capacity ? capacity : (uint32_t)D64;
                    frame->pad = 0;
                    if (mirror) {                                           // (see k_bin_emit: two 16-byte stores to mapped host memory)
                        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                        const u32x4 v = {serial, D64 > capacity ? 1u : 0u, (uint32_t)D64, (uint32_t)(D64 >> 32)};
                        *reinterpret_cast<volatile u32x4*>(mirror) = v;
                        const u32x4 sv = {serial, vis, (uint32_t)tsum, (uint32_t)(tsum >> 32)};
                        *reinterpret_cast<volatile u32x4*>(mirror + 8) = sv;
                    }
                }
            }
        }
    }
    __syncthreads();                                                        // s_base; this workgroup's compacted lists (written above)
    const uint32_t base = s_base;
    if (base >= capacity) return;
    for (uint32_t jb = 0; jb < sc.splats; jb += BIN_THREADS)
        (void)bin_emit_batch<KeyT>(cidx, crect, coff, pos_begin + jb + threadIdx.x, jb + threadIdx.x < sc.splats, base, capacity, tiles_x, row_begin,
                                   list_shift, keys_out, vals_out);
}

template <class KeyT>
static int binning_typed(gs_mesh* m, const ProjectParams& pp, const uint32_t* order_dev, gs_sorter* sorter, uint32_t R, uint32_t tiles /* sort keys */) {
    gs_context* ctx = m->ctx;
    hipStream_t st = ctx->stream;
    const RadixExec ex = {st, &m->radix, ctx->lds_atomic_lane_order};
    RenderFrame* frame = m->frame.as<RenderFrame>();
    uint32_t grid = (R + BIN_THREADS - 1) / BIN_THREADS;
    if (grid < 1) grid = 1;
    if (grid > (uint32_t)BIN_MAX_BLOCKS) grid = BIN_MAX_BLOCKS;
    const uint32_t cap = m->entry_capacity;
    const uint32_t* R_dev = (sorter && sorter->last_culled) ? &sorter->result_frame->kept : nullptr;
    const uint32_t blend_bins = pp.bins_x * (pp.bin_row_end - pp.bin_row_begin);
    // the chunked composite's per-draw words and the pool the per-bin kernel closes chunks into (reset by k_bin_count)
    GS_TRY(m->deep_flags.ensure(((size_t)GS_FLAG_OF + blend_bins) * 4 + 64));
    GS_TRY(m->chunk_pool.ensure((size_t)GS_POOL_SLOTS * 256 * sizeof(float4)));
    GS_TRY(m->blend_stats.ensure((size_t)blend_bins * 12));
    if (sorter && sorter->stream != st) {      // the sorter's private stream may overwrite `sorted` from here on
        GS_HIP(hipEventRecord(sorter->ev_consumed, st));
        sorter->consumer_pending = true;
    }
    // the previous draw's per-bin blend statistics order this draw's blend workgroups, if it drew the same bins
    // (only while the bins outnumber the resident workgroups by a small factor: an 8K frame's 32 k bins balance themselves by
    // backfilling, and ordering them in one workgroup would cost more than it gives)
    // deep pass: a bin qualifies when its previous draw walked >= deep_min (splat, quadrant) pairs and >= deep_factor x the mean bin
    // (4096 pairs.  C3S frame ms at 2048 / 3072 / 4096 / 6144 / 8192: 1.336 / 1.338 / 1.309 / 1.296 / 1.500 - k_deep_scan costs
    // 0.17 ms for 512 bins and is bound by its gathers, so: fewer, deeper bins; profiles/r03zz_kstats_C3S.txt)
    static const uint32_t deep_min_cfg = getenv("GSPLAT_DEEP_MIN") ? (uint32_t)atoi(getenv("GSPLAT_DEEP_MIN")) : 4u * GS_CHUNK;
    static const uint32_t deep_factor = getenv("GSPLAT_DEEP_FACTOR") ? (uint32_t)atoi(getenv("GSPLAT_DEEP_FACTOR")) : 3u;
    // (a GS_DRAW_ROP8 draw has no deep pass - rounding after every splat does not split into chunks - and must not raise its trigger)
    const uint32_t deep_min = m->draw_mode == GS_DRAW_FP32 ? deep_min_cfg : 0x7FFFFFFFu;
    const bool order_ok = blend_bins > 0 && blend_bins <= 8192u && m->blend_bins == blend_bins && m->blend_row_begin == pp.bin_row_begin &&
                          m->blend_width == (uint32_t)pp.width && m->blend_stats_mode == m->draw_mode && !getenv("GSPLAT_NO_BLEND_ORDER");
    // (the same draw mode: a GS_DRAW_ROP8 draw walks its lists to the saturation depth twice or to their ends - its per-bin counters
    // say nothing about what an fp32 draw costs, and the other way round)
    if (order_ok) GS_TRY(m->blend_order.ensure((size_t)blend_bins * 4));
    // ... and they ORDER this draw's blend workgroups only when that draw had THIS draw's view.  Heaviest-first from statistics of
    // the same view is worth 5-8 % of a frame (C3 demo pose 0.261 -> 0.248 ms, the orbit's poses held fixed 0.350 -> 0.322); from
    // a view 6 degrees away it is worth less than the workgroup that computes it costs - a moving camera draws its frames faster
    // in plain row-major order (C3 orbit 0.359 -> 0.349 ms per frame, C2 0.264 -> 0.255; profiles/r06g_orbit_gate.txt).  Two
    // stateless orders were built and measured as well, both no better than row-major: by the length of the list a bin scans (a
    // long list is a dense region that saturates at once: C2 fixed pose 0.327 vs 0.318 ms without any order, r06c) and a stride
    // permutation that makes the late starters a uniform sample of the screen (r06e: 0.3624 vs 0.3625).  The deep pass's
    // membership (which executor composites a bin, not when) keeps using the statistics of whatever view came before.
    const ProjectParams& lp = m->stats_pp;
    const bool same_frame = m->stats_pp_valid && memcmp(lp.proj, pp.proj, sizeof(pp.proj)) == 0 && lp.width == pp.width && lp.height == pp.height &&
                            lp.count == pp.count && lp.list_shift == pp.list_shift;
    bool same_view = same_frame && memcmp(lp.view, pp.view, sizeof(pp.view)) == 0;
    // ... or a view close to it.  Two things are taken from the two cameras (second half of round 6; tools/motion_ab.py):
    //  * how far the scene's centre of mass moved on screen, in whole bins: the schedule reads the statistics of the bin a bin's
    //    content came FROM (StatShift).  A camera turning about its own position shifts the whole picture - and the costly bins, and
    //    the deep pass's members - by that much: capture-like C3S turning 2 / 4 / 8 degrees per frame 2.14 / 2.77 / 3.78 -> 1.16 / 1.39 /
    //    2.06 ms per frame, C2 with the previous order at 4 / 8 degrees 0.3175 / 0.3139 -> 0.2913 / 0.3001 (row-major: 0.303)
    //    (profiles/r06zz_stat_shift_ab.txt; $GSPLAT_NO_STAT_SHIFT);
    //  * what that shift does NOT describe - parallax: how far points at half and at twice the centre's distance, on the previous
    //    camera's ray through the centre, end up from the centre in the new picture, in screen heights.  Orbiting the scene at 0.25 ...
    //    6 degrees per frame the previous frame's order beats row-major up to 2 degrees (C2 7-11 %, C3 1-2 % of the frame), is level
    //    at 3 and loses 2.5-3 % at 6 (profiles/r06w_motion_ab.txt): the order is taken below a parallax of 0.045 screen heights
    //    ($GSPLAT_ORDER_MOTION; 0 = the same view only).  Turning in place has none, at any speed.
    StatShift stat_shift = {0, 0, pp.bins_x};
    if (same_frame && !same_view && pp.block_cull && m->centre_n > 0) {
        const float limit = getenv("GSPLAT_ORDER_MOTION") ? (float)atof(getenv("GSPLAT_ORDER_MOTION")) : 0.045f;
        const double inv_n = 1.0 / (double)m->centre_n;
        const float c[3] = {(float)(m->centre_sum[0] * inv_n), (float)(m->centre_sum[1] * inv_n), (float)(m->centre_sum[2] * inv_n)};
        auto window = [&](const ProjectParams& q, const float* p, float* xy) {       // (column-major matrices, window y up: project.hip)
            float v[4], o[4];
            for (int r = 0; r < 4; r++) v[r] = q.view[r] * p[0] + q.view[4 + r] * p[1] + q.view[8 + r] * p[2] + q.view[12 + r];
            for (int r = 0; r < 4; r++) o[r] = q.proj[r] * v[0] + q.proj[4 + r] * v[1] + q.proj[8 + r] * v[2] + q.proj[12 + r] * v[3];
            if (!(o[3] > 1e-6f)) return false;
            xy[0] = (o[0] / o[3] * 0.5f + 0.5f) * q.width;
            xy[1] = (o[1] / o[3] * 0.5f + 0.5f) * q.height;
            return true;
        };
        float was[2], is[2];
        if (window(lp, c, was) && window(pp, c, is)) {
            const float dx = (is[0] - was[0]) / (float)GS_BIN, dy = (is[1] - was[1]) / (float)GS_BIN;
            const bool shift_ok = fabsf(dx) < 4096.0f && fabsf(dy) < 4096.0f && !getenv("GSPLAT_NO_STAT_SHIFT");   // (false for NaN)
            if (shift_ok) {
                stat_shift.sx = (int32_t)lrintf(dx);
                stat_shift.sy = (int32_t)lrintf(dy);
            }
            // the previous camera's position in the splats' own space: -A^-1 b of its modelView = [A | b]
            const float* V = lp.view;
            const float a00 = V[0], a10 = V[1], a20 = V[2], a01 = V[4], a11 = V[5], a21 = V[6], a02 = V[8], a12 = V[9], a22 = V[10];
            const float c00 = a11 * a22 - a12 * a21, c01 = a02 * a21 - a01 * a22, c02 = a01 * a12 - a02 * a11;
            const float det = a00 * c00 + a10 * c01 + a20 * c02;
            const float b0 = V[12], b1 = V[13], b2 = V[14], id = 1.0f / det;
            const float eye[3] = {-(c00 * b0 + c01 * b1 + c02 * b2) * id,
                                  -((a12 * a20 - a10 * a22) * b0 + (a00 * a22 - a02 * a20) * b1 + (a02 * a10 - a00 * a12) * b2) * id,
                                  -((a10 * a21 - a11 * a20) * b0 + (a01 * a20 - a00 * a21) * b1 + (a00 * a11 - a01 * a10) * b2) * id};
            float parallax = 0.0f;
            bool seen = fabsf(det) > 1e-20f;
            for (int k = 0; k < 2 && seen; k++) {
                const float t = k ? 2.0f : 0.5f;
                const float p[3] = {eye[0] + t * (c[0] - eye[0]), eye[1] + t * (c[1] - eye[1]), eye[2] + t * (c[2] - eye[2])};
                float at[2];
                seen = window(pp, p, at);
                if (seen) parallax = fmaxf(parallax, sqrtf((at[0] - is[0]) * (at[0] - is[0]) + (at[1] - is[1]) * (at[1] - is[1])));
            }
            // (without the shift the whole motion counts, not only the parallax)
            const float moved = (parallax + (shift_ok ? 0.0f : sqrtf(dx * dx + dy * dy) * (float)GS_BIN)) / fmaxf(pp.height, 1.0f);
            same_view = seen && moved <= limit;            // (NaN compares false)
        }
    }
    const bool stale_order = getenv("GSPLAT_BLEND_ORDER_STALE") != nullptr;   // (A/B: rounds 2-5 - order from whatever draw came before)
    // The deep pass runs when the last draw whose verdict has arrived (mapped host word, no synchronisation) left bins over the
    // threshold - the decision only moves work between executors, the pixels do not depend on it (tile_blend.hip)
    m->deep_pass = order_ok && m->draw_mode == GS_DRAW_FP32 && !m->no_deep && m->mirror_host && ((volatile uint32_t*)m->mirror_host)[4] > 0u;
    if (m->deep_pass) {
        GS_TRY(m->deep_ent.ensure((size_t)GS_DEEP_MAX_BINS * GS_DEEP_LIST_CAP * 4));
        GS_TRY(m->deep_cnt.ensure((size_t)GS_DEEP_MAX_BINS * GS_DEEP_RANGES * 4 * 4));
        GS_TRY(m->deep_partial.ensure((size_t)GS_DEEP_UNITS * 256 * sizeof(float4)));
        GS_TRY(m->deep_work.ensure((size_t)GS_DEEP_UNITS * 4));
    }
    // $GSPLAT_BIN_FUSED=1: count + emit in one launch behind a scan across the running grid (k_bin_fused).  Built, bit-identical,
    // and SLOWER than the two kernels on every configuration, so it is not the default: bin stage C3 0.0513 -> 0.0598-0.0642 ms,
    // C2 0.037 -> 0.047, C3S 0.084 -> 0.100, C4 0.419 -> 0.425-0.431 (profiles/r06i_ab_fused.txt, r06j_ab_fused_sleep.txt; longer
    // back-off between polls recovers 4 of the 12 us).  Every slice's emit has to wait for the slowest slice's count - the scan
    // is a grid-wide barrier in disguise - so the launch is count + barrier + emit like the two kernels, minus one kernel boundary,
    // plus two hops of polling under the count's own memory traffic, and the emit loses k_bin_emit's round-robin deal of batches.
    const char* fused_env = getenv("GSPLAT_BIN_FUSED");
    const bool fused = fused_env && fused_env[0] == '1';
    // (+ one workgroup that orders the blend's bins and names the deep pass's members)
    // (that workgroup also raises the deep pass's trigger, so under a camera that keeps moving it still runs when the pass is on,
    // for scenes of tiny splats - the ones that grow deep bins - and every 8th draw otherwise)
    const bool order_wg = order_ok && (same_view || stale_order || m->deep_pass || pp.list_shift == GS_LIST_SHIFT_SMALL || (m->draw_serial & 7u) == 7u);
    ++m->draw_serial;
    if (m->draw_serial == 0u) m->draw_serial = 1u;       // (the scan's granules are tagged with the serial: 0 is "never written")
    if (fused) {
        GS_TRY(m->bin_scan.ensure(((size_t)3 * BIN_MAX_BLOCKS + 3 * BIN_SCAN_GROUPS) * 8 + 64));
        if (!m->bin_scan_ready) {                        // once: no granule carries a tag
            GS_HIP(hipMemsetAsync(m->bin_scan.p, 0, m->bin_scan.bytes, st));
            m->bin_scan_ready = true;
        }
        BinScan scan;
        scan.chunk_rows = m->bin_scan.as<unsigned long long>();
        scan.group_rows = scan.chunk_rows + 3 * BIN_MAX_BLOCKS;
        scan.fail = reinterpret_cast<uint32_t*>(scan.group_rows + 3 * BIN_SCAN_GROUPS);
        scan.tag = m->draw_serial;
        hipLaunchKernelGGL((k_bin_fused<KeyT>), dim3(grid + (order_wg ? 1u : 0u)), dim3(BIN_THREADS), 0, st, order_dev, R, R_dev,
                           m->translate ? m->perm.as<uint32_t>() : nullptr, m->prect.as<uint2>(), m->cidx.as<uint32_t>(), m->rect_q.as<uint2>(),
                           m->coff.as<uint32_t>(), m->radix.digit_total.as<uint32_t>(), m->tile_ranges.as<uint2>(), tiles, pp.list_shift, pp.count,
                           m->block_any.as<uint8_t>(), m->deep_flags.as<uint32_t>(), blend_bins, frame, cap, pp.lists_x, pp.list_row_begin,
                           m->ekeyA.as<KeyT>(), m->evalA.as<uint32_t>(), m->mirror_dev, m->draw_serial,
                           order_wg ? m->blend_stats.as<uint2>() : nullptr, order_wg ? m->blend_order.as<uint32_t>() : nullptr,
                           m->deep_pass ? 1u : 0u, m->blend_stats.as<uint32_t>(), deep_min, deep_factor, scan, stat_shift);
    } else {
        hipLaunchKernelGGL(k_bin_count, dim3(grid), dim3(BIN_THREADS), 0, st, order_dev, R, R_dev,
                           m->translate ? m->perm.as<uint32_t>() : nullptr, m->prect.as<uint2>(), m->cidx.as<uint32_t>(), m->rect_q.as<uint2>(), m->coff.as<uint32_t>(),
                           m->bin_sums.as<uint32_t>(), m->radix.digit_total.as<uint32_t>(),
                           m->tile_ranges.as<uint2>(), tiles, pp.list_shift, pp.count,
                           m->block_any.as<uint8_t>(), m->deep_flags.as<uint32_t>(), blend_bins);
        hipLaunchKernelGGL((k_bin_emit<KeyT>), dim3(grid + (order_wg ? 1u : 0u)), dim3(BIN_THREADS), 0, st, frame, cap, m->cidx.as<uint32_t>(),
                           m->rect_q.as<uint2>(), m->coff.as<uint32_t>(), m->bin_sums.as<uint32_t>(), grid, pp.lists_x, pp.list_row_begin,
                           m->ekeyA.as<KeyT>(), m->evalA.as<uint32_t>(), pp.list_shift, m->mirror_dev, m->draw_serial,
                           order_wg ? m->blend_stats.as<uint2>() : nullptr, blend_bins, order_wg ? m->blend_order.as<uint32_t>() : nullptr,
                           m->deep_pass ? 1u : 0u, m->deep_flags.as<uint32_t>(), m->blend_stats.as<uint32_t>(), deep_min, deep_factor, stat_shift);
    }
    // ... and whenever the deep pass runs: its frames have the long tail that an order - even one from a view several degrees away -
    // and the pass's workgroups behind the costliest bins (tile_blend.hip) shorten.  Capture-like C3S, orbit at 3 / 6 / 12 degrees per
    // frame: 1.24 / 1.356 / 1.54 -> 1.18 / 1.31 / 1.51 ms, turning 4 degrees per frame 2.84 -> 2.77, never slower
    // (profiles/r06zz_deep_keeps_order_ab.txt; $GSPLAT_DEEP_ROW_MAJOR_ON_MOTION=1 is the earlier rule).
    m->blend_order_valid = order_ok && (same_view || stale_order || (m->deep_pass && !getenv("GSPLAT_DEEP_ROW_MAJOR_ON_MOTION")));
    GS_HIP(hipGetLastError());
    if (pp.row_begin == 0u && pp.row_end >= pp.tiles_y) {     // (a strip's visible count says nothing about the scene: mesh_heal_overflow)
        m->full_serial[m->draw_serial & 7u] = m->draw_serial;
        m->full_count[m->draw_serial & 7u] = pp.count;
    }
    if (m->timed_draw) GS_HIP(hipEventRecord(m->ev[2], st));

    uint32_t bits = 1;
    while ((1ull << bits) < tiles) bits++;
    const uint32_t passes = (bits + 7) / 8;
    KeyT* kbuf[2] = {m->ekeyA.as<KeyT>(), m->ekeyB.as<KeyT>()};
    uint32_t* vbuf[2] = {m->evalA.as<uint32_t>(), m->evalB.as<uint32_t>()};
    for (uint32_t p = 0; p < passes; p++) {
        ArrayLoader<KeyT> al = {kbuf[p & 1], vbuf[p & 1], &frame->entry_count, 0u};
        if (passes == 1)                           // <= 256 lists: the digit is the key, ranges come from the digit totals
            GS_TRY((radix_pass<ArrayLoader<KeyT>, KeyT, false, false>(ex, al, al, cap, 0, 0, (KeyT*)nullptr, vbuf[1],
                                                                      m->tile_ranges.as<uint2>(), tiles)));
        else if (p + 1 == passes)
            GS_TRY((radix_pass<ArrayLoader<KeyT>, KeyT, false, true>(ex, al, al, cap, 8 * (int)p, (int)p, (KeyT*)nullptr,
                                                                     vbuf[(p + 1) & 1], m->tile_ranges.as<uint2>())));
        else
            GS_TRY((radix_pass<ArrayLoader<KeyT>, KeyT, true, false>(ex, al, al, cap, 8 * (int)p, (int)p, kbuf[(p + 1) & 1],
                                                                     vbuf[(p + 1) & 1])));
    }
    m->sorted_buf = (passes & 1);            // which ping-pong buffer holds the tile-sorted entries
    return GS_OK;
}

int gs_launch_binning(gs_mesh* m, const ProjectParams& pp, const uint32_t* order_dev, gs_sorter* sorter, uint32_t render_count) {
    const uint32_t tiles = pp.lists_x * (pp.list_row_end - pp.list_row_begin);   // one list per sort key
    if (tiles <= 65536u && !m->ctx->wide_entry_keys) return binning_typed<uint16_t>(m, pp, order_dev, sorter, render_count, tiles);
    return binning_typed<uint32_t>(m, pp, order_dev, sorter, render_count, tiles);
}
