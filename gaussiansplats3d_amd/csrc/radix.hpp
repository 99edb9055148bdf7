// radix.hpp — stable LSD radix sort passes (8-bit digits) for gfx950, wave64.
//
// Used twice per frame:
//   * the depth sort that replaces the reference's counting sort (/root/reference/src/worker/sorter.cpp:142-167):
//     16..24-bit bucket keys, 2..3 passes, payload = global splat index;
//   * the tile-entry sort of the rasteriser (key = tile id, payload = splat index), which must be STABLE so
//     that each tile's list keeps the depth order established by the first sort.
//
// One pass = k_radix_hist -> k_radix_scatter.  The histogram kernel leaves a two-level table: one 1 KiB row of digit
// counts per workgroup plus one row per GROUP of 32 workgroups (atomics, 32 adders per word); every scatter workgroup
// sums the <= 16 group rows and the <= 31 rows of its own group that precede it (coalesced 1 KiB rows) and scans the
// 256 digit totals itself, so no separate scan kernel (10.7 us and two kernel boundaries per pass on MI355X) sits
// between the two.  Deterministic: integer sums only, no inter-workgroup spinning, no dependence on dispatch order
// (cdna_hip_programming.md §6 G16).  A fixed grid of <= RADIX_MAX_BLOCKS = 512 workgroups walks contiguous runs of
// 4096-key tiles (C3's depth sort: 472 workgroups x 3 tiles), so the table stays <= 0.5 MB whatever N is, and N may
// live in device memory (the tile-entry count and the length of a frustum-culled list only exist on the device).
//
// Ranking inside a tile is wave64-native.  Fast path (ATOMIC_RANK): one `ds_add_rtn_u32` on a per-wave LDS
// histogram per key.  gfx950's LDS serves the lanes of ONE wave instruction that hit the same address in ascending
// lane order (tools/probes/lds_atomic_order.hip: 0 mismatches in 3.3e9 checks; re-verified by a self-test at
// gs_context_create), so the returned value IS the stable rank of the key among equal digits seen so far by its
// wave.  Portable path (self-test failed): 8 ballots build the "same digit" lane mask, rank = popcount of the
// lower lanes + the per-wave counter.  Keys are then reordered through LDS so that every digit's run leaves the
// CU as contiguous stores.
#pragma once
#include "gs_internal.hpp"

// ---------------------------------------------------------------------------------------------------
// small wave / block helpers
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, uint32_t lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t t = __shfl_up(v, o, 64);
        if ((int)lane >= o) v += t;
    }
    return v;
}

// exclusive scan of one value per thread over a WAVES*64-thread block; returns exclusive prefix, *total = block sum
template <int WAVES>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* s_tmp /*>=WAVES*/, uint32_t* total) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t incl = wave_incl_scan(v, lane);
    if (lane == 63) s_tmp[wave] = incl;
    __syncthreads();
    uint32_t wave_base = 0, sum = 0;
#pragma unroll
    for (int w = 0; w < WAVES; w++) {
        const uint32_t c = s_tmp[w];
        wave_base += ((uint32_t)w < wave) ? c : 0u;
        sum += c;
    }
    if (total) *total = sum;
    __syncthreads();
    return wave_base + incl - v;
}
__device__ __forceinline__ uint32_t block_excl_scan_256(uint32_t v, uint32_t* s_tmp /*>=4*/, uint32_t* total) {
    return block_excl_scan<4>(v, s_tmp, total);
}

// ---------------------------------------------------------------------------------------------------
// loaders: logical element j in [0, count()) -> (key, payload)
// ---------------------------------------------------------------------------------------------------
template <class KeyT>
struct ArrayLoader {
    const KeyT* __restrict__ keys;
    const uint32_t* __restrict__ vals;
    const uint32_t* __restrict__ n_dev;   // device-resident count (nullable)
    uint32_t n_host;
    __device__ __forceinline__ void prepare() {}
    __device__ __forceinline__ uint32_t count() const { return n_dev ? *n_dev : n_host; }
    __device__ __forceinline__ uint32_t key(uint32_t j) const { return (uint32_t)keys[j]; }
    __device__ __forceinline__ uint32_t val(uint32_t j) const { return vals[j]; }
    // fetch = the memory reads of element j, decode = the arithmetic on them (none here)
    struct Raw { uint32_t key, val; };
    __device__ __forceinline__ Raw fetch(uint32_t j) const { return Raw{(uint32_t)keys[j], vals[j]}; }
    __device__ __forceinline__ void decode(const Raw& r, uint32_t& k, uint32_t& v) const { k = r.key; v = r.val; }
    __device__ __forceinline__ bool valid(uint32_t) const { return true; }
    // the histogram's view of element j: the memory read, then the arithmetic on it
    typedef uint32_t HRaw;
    __device__ __forceinline__ HRaw hist_fetch(uint32_t j) const { return (uint32_t)keys[j]; }
    __device__ __forceinline__ uint32_t hist_key(HRaw r) const { return r; }
    __device__ __forceinline__ void note_clamp(bool) const {}
    static __device__ __forceinline__ int prof_slot(int shift) { return shift == 8 && sizeof(KeyT) == 2 ? 1 : -1; }   // GS_RADIX_PROFILE
};

// ---------------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------------
// XCD-aware chunk order for walks over the depth-sorted list (the binner, tile_bin.hip).  Workgroup b runs on XCD b % 8
// (observed dispatch rule, used for speed only) and each XCD has an L2 of its own.  Giving one XCD RUNS of neighbouring
// chunks makes the 8-byte rect gathers of Morton neighbours (similar depth, so close in the sorted list) find the line a
// neighbour fetched into that L2.  Runs, not one contiguous eighth per XCD: the near end of the list holds most of the
// visible splats, and an XCD that owned it alone finished last (C3 binning 74 -> 92 us; with runs it keeps the balance).
// The radix passes gain nothing from either order (measured) and keep chunk = workgroup.
#ifndef GS_XCD_RUN
#define GS_XCD_RUN 16
#endif
__device__ __forceinline__ uint32_t xcd_chunk(uint32_t b, uint32_t grid) {
#if GS_XCD_RUN
    if (grid % (8u * GS_XCD_RUN)) return b;                       // only grids that split evenly (the binner's 2048)
    const uint32_t x = b % 8u, k = b / 8u;                       // the k-th workgroup of XCD x
    return ((k / GS_XCD_RUN) * 8u + x) * GS_XCD_RUN + k % GS_XCD_RUN;
#else
    (void)grid;
    return b;
#endif
}

struct RadixChunk {
    uint32_t n, tile_begin, tile_end, id;
};
__device__ __forceinline__ RadixChunk radix_chunk(uint32_t n) {
    const uint32_t tiles = (n + RADIX_TILE - 1) / RADIX_TILE;
    const uint32_t per = (tiles + gridDim.x - 1) / gridDim.x;
    RadixChunk c;
    c.n = n;
    c.id = blockIdx.x;
    c.tile_begin = min(c.id * per, tiles);
    c.tile_end = min(c.tile_begin + per, tiles);
    return c;
}

// The histogram kernel runs 1024-thread workgroups: its grid is the scatter's (one table row per workgroup, <= 512 of
// them), and a gather-type loader (the depth keys) needs more waves in flight than 4 per workgroup to hide its latency
// (r01e: 20.5 us with 256 threads).
constexpr int HIST_THREADS = 1024;
constexpr int HIST_ITEMS = RADIX_TILE / HIST_THREADS;

// The histogram of a workgroup's tiles, HIST_GROUP tiles at a time: all their loads are issued before the first LDS atomic,
// so the workgroup waits for one memory round trip per group instead of one per tile.  Measured r03 (same box, ab_libs): C4 sort
// 0.240 -> 0.231 ms, entry sort 0.245 -> 0.241; C3 / C5 within noise (their histograms already overlapped).  Order is irrelevant
// for a histogram.
constexpr uint32_t HIST_GROUP = 4;

// full tiles of an array of keys: 16-byte loads
template <class KeyT>
__device__ __forceinline__ void hist_tiles(const ArrayLoader<KeyT>& ld, uint32_t tile, uint32_t tiles, uint32_t n, int shift, uint32_t* hist) {
    constexpr uint32_t LOADS = RADIX_TILE * (uint32_t)sizeof(KeyT) / 16u;        // 16-byte loads per tile
    constexpr uint32_t PER = (LOADS + HIST_THREADS - 1) / HIST_THREADS;          // per thread and tile (1 for 16- and 32-bit keys)
    uint4 v[HIST_GROUP][PER];
    bool full[HIST_GROUP];
#pragma unroll
    for (uint32_t g = 0; g < HIST_GROUP; g++) {
        const uint32_t base = (tile + g) * RADIX_TILE;                           // a multiple of 4096 keys
        full[g] = g < tiles && base + RADIX_TILE <= n;
        const uint4* src = reinterpret_cast<const uint4*>(ld.keys + base);
#pragma unroll
        for (uint32_t k = 0; k < PER; k++) {
            const uint32_t l = k * HIST_THREADS + threadIdx.x;
            v[g][k] = (full[g] && l < LOADS) ? src[l] : make_uint4(0u, 0u, 0u, 0u);
        }
    }
#pragma unroll
    for (uint32_t g = 0; g < HIST_GROUP; g++) {
        if (full[g]) {
#pragma unroll
            for (uint32_t k = 0; k < PER; k++) {
                if (k * HIST_THREADS + threadIdx.x >= LOADS) continue;
                const uint32_t w[4] = {v[g][k].x, v[g][k].y, v[g][k].z, v[g][k].w};
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    if (sizeof(KeyT) == 2) {
                        atomicAdd(&hist[((w[c] & 0xFFFFu) >> shift) & 255u], 1u);
                        atomicAdd(&hist[((w[c] >> 16) >> shift) & 255u], 1u);
                    } else {
                        atomicAdd(&hist[(w[c] >> shift) & 255u], 1u);
                    }
                }
            }
        } else if (g < tiles) {                                                   // the ragged last tile
            const uint32_t base = (tile + g) * RADIX_TILE;
#pragma unroll
            for (int r = 0; r < HIST_ITEMS; r++) {
                const uint32_t j = base + r * HIST_THREADS + threadIdx.x;
                if (j < n) atomicAdd(&hist[((uint32_t)ld.keys[j] >> shift) & 255u], 1u);
            }
        }
    }
}
// any other loader (the depth keys: one 4-byte read per element, a keep bit for the culled variants)
template <class Loader>
__device__ __forceinline__ void hist_tiles(const Loader& ld, uint32_t tile, uint32_t tiles, uint32_t n, int shift, uint32_t* hist) {
    typename Loader::HRaw raw[HIST_GROUP][HIST_ITEMS];
    bool ok[HIST_GROUP][HIST_ITEMS];
#pragma unroll
    for (uint32_t g = 0; g < HIST_GROUP; g++)
#pragma unroll
        for (int r = 0; r < HIST_ITEMS; r++) {
            const uint32_t j = (tile + g) * RADIX_TILE + r * HIST_THREADS + threadIdx.x;
            ok[g][r] = g < tiles && j < n && ld.valid(j);
            raw[g][r] = ok[g][r] ? ld.hist_fetch(j) : typename Loader::HRaw();
        }
#pragma unroll
    for (uint32_t g = 0; g < HIST_GROUP; g++)
#pragma unroll
        for (int r = 0; r < HIST_ITEMS; r++)
            if (ok[g][r]) atomicAdd(&hist[(ld.hist_key(raw[g][r]) >> shift) & 255u], 1u);
}

template <class Loader>
__global__ __launch_bounds__(HIST_THREADS) void k_radix_hist(Loader ld, int shift, uint32_t* __restrict__ block_hist,
                                                             uint32_t* __restrict__ digit_total) {
    __shared__ uint32_t s_hist[4][RADIX_BINS];               // waves w, w+4, w+8, w+12 share one
    ld.prepare();
    const RadixChunk ch = radix_chunk(ld.count());
    const uint32_t tid = threadIdx.x, wave = (tid >> 6) & 3u;
    (&s_hist[0][0])[tid] = 0;
    __syncthreads();
    for (uint32_t tile = ch.tile_begin; tile < ch.tile_end; tile += HIST_GROUP)
        hist_tiles(ld, tile, min(HIST_GROUP, ch.tile_end - tile), ch.n, shift, s_hist[wave]);
    __syncthreads();
    if (tid >= RADIX_BINS) return;
    const uint32_t total = s_hist[0][tid] + s_hist[1][tid] + s_hist[2][tid] + s_hist[3][tid];
    block_hist[ch.id * RADIX_BINS + tid] = total;                // one coalesced 1 KiB row per workgroup
    // Group rows.  Same-address atomics serialise in the fabric (~12 ns each, MI355X_MICROARCH.md row "fanin"): with 32
    // adders per word they stay < 1 us, where one word per digit for the whole grid cost 10-15 us per pass (A/B r01b).
    if (total) atomicAdd(&digit_total[(ch.id / RADIX_GROUP) * RADIX_BINS + tid], total);
}


// WRITE_KEYS: also emit the keys (needed by every pass but the last).
// RANGES: this is the last pass of a multi-pass tile sort - publish each key's [begin,end) in the sorted output.  Inside a
// workgroup tile equal keys are contiguous (the earlier passes ordered the lower digits, this pass is stable), so a
// run boundary costs one atomicMin/atomicMax pair; `ranges` must be pre-set to (0xFFFFFFFF, 0).
//
// Geometry: 512 threads = 8 waves x 8 keys per lane over a 4096-key tile, <= 64 VGPRs and ~35 KB LDS (16-bit keys
// are staged as 16 bits).  The kernel is a chain of load -> rank -> barrier -> reorder -> barrier -> store phases per
// tile; with the grid capped at 512 a CU holds two workgroups (16 waves) whose phases overlap each other.
// 1024-thread workgroups sort faster in isolation (0.173 vs 0.186 ms for C3) but pipeline worse against the draw of the
// previous frame (0.483 vs 0.468 ms per frame, same-box A/B r01e): 512 it is.
#ifndef SCATTER_THREADS_CFG
#define SCATTER_THREADS_CFG 512
#endif
constexpr int SCATTER_THREADS = SCATTER_THREADS_CFG;
constexpr int SCATTER_WAVES = SCATTER_THREADS / 64;
constexpr int SCATTER_ITEMS = RADIX_TILE / SCATTER_THREADS;
constexpr int SCATTER_PARTS = SCATTER_THREADS / RADIX_BINS;     // threads per digit in the offset prologue
static_assert(SCATTER_THREADS % RADIX_BINS == 0 && 2 * SCATTER_PARTS <= SCATTER_WAVES, "offset prologue layout");

#ifdef GS_RADIX_PROFILE
// tools/radix_profile.py: per scatter workgroup (slot = shift / 8 of the depth sort: 0 / 1) the 100 MHz clock at kernel start,
// after the offset prologue, and for its FIRST tile: data decoded, ranked, offsets scanned, reordered in LDS, stored; then the
// end of the kernel and the number of tiles
static __device__ unsigned long long g_radix_prof[2 * 512 * 10];      // one copy per translation unit; sorter.hip's is read
#define RADIX_PROF(slot, v) do { const int ps_ = Loader::prof_slot(shift); if (threadIdx.x == 0 && blockIdx.x < 512u && ps_ >= 0) g_radix_prof[(ps_ * 512 + blockIdx.x) * 10 + (slot)] = (v); } while (0)
#define RADIX_PROF_WAIT() __builtin_amdgcn_s_waitcnt(0)
#else
#define RADIX_PROF(slot, v) do { } while (0)
#define RADIX_PROF_WAIT() do { } while (0)
#endif

template <class Loader, class KeyOutT, bool WRITE_KEYS, bool RANGES, bool ATOMIC_RANK>
__global__ __launch_bounds__(SCATTER_THREADS, 4) void k_radix_scatter(Loader ld, int shift,
                                                                      const uint32_t* __restrict__ block_hist,
                                                                      const uint32_t* __restrict__ group_hist,
                                                                      KeyOutT* __restrict__ keys_out,
                                                                      uint32_t* __restrict__ vals_out, uint2* ranges,
                                                                      uint32_t direct_ranges) {
    __shared__ KeyOutT s_keys[RADIX_TILE];                  // staged in the output key width (u16 or u32)
    __shared__ uint32_t s_vals[RADIX_TILE];
    __shared__ uint32_t s_wave[SCATTER_WAVES][RADIX_BINS];  // per-wave digit counts, then per-wave exclusive offsets
    __shared__ uint32_t s_base[RADIX_BINS];                 // running global offset of each digit for this workgroup
    __shared__ uint32_t s_local[RADIX_BINS];                // first staging slot of each digit in the current tile
    __shared__ uint32_t s_total[RADIX_BINS];
    __shared__ uint32_t s_tmp[SCATTER_WAVES];

    RADIX_PROF(0, wall_clock64());
    ld.prepare();
    const RadixChunk ch = radix_chunk(ld.count());
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    volatile uint32_t* my_hist = s_wave[wave];

    {   // this workgroup's first output slot per digit = keys with a smaller digit + keys of this digit in earlier
        // workgroups; two threads per digit split the rows, every load is independent of the others
        const uint32_t d = tid & 255u, half = tid >> 8;      // `half` = which of the SCATTER_PARTS row subsets
        const uint32_t g = ch.id / RADIX_GROUP, groups = (gridDim.x + RADIX_GROUP - 1) / RADIX_GROUP;
        // fixed trip counts, fully unrolled and predicated: every load of the prologue is in flight at once (one memory
        // round trip instead of a chain of batches: isolated C3 sort 0.182 -> 0.176 ms)
        constexpr uint32_t GROUP_LOADS = (RADIX_MAX_GROUPS + SCATTER_PARTS - 1) / SCATTER_PARTS;
        constexpr uint32_t ROW_LOADS = (RADIX_GROUP + SCATTER_PARTS - 1) / SCATTER_PARTS;
        uint32_t gv[GROUP_LOADS], rv[ROW_LOADS];
#pragma unroll
        for (uint32_t k = 0; k < GROUP_LOADS; k++) {
            const uint32_t r = half + k * (uint32_t)SCATTER_PARTS;
            gv[k] = r < groups ? group_hist[r * RADIX_BINS + d] : 0u;
        }
#pragma unroll
        for (uint32_t k = 0; k < ROW_LOADS; k++) {
            const uint32_t r = g * RADIX_GROUP + half + k * (uint32_t)SCATTER_PARTS;
            rv[k] = r < ch.id ? block_hist[r * RADIX_BINS + d] : 0u;
        }
        uint32_t before = 0, all = 0;
#pragma unroll
        for (uint32_t k = 0; k < GROUP_LOADS; k++) {
            all += gv[k];
            before += (half + k * (uint32_t)SCATTER_PARTS) < g ? gv[k] : 0u;
        }
#pragma unroll
        for (uint32_t k = 0; k < ROW_LOADS; k++) before += rv[k];
        uint32_t* s_before = &s_wave[0][0];                  // [PARTS][256], free until the tile loop zeroes it
        uint32_t* s_all = &s_wave[SCATTER_PARTS][0];
        s_before[tid] = before;
        s_all[tid] = all;
        __syncthreads();
        uint32_t tot = 0, mine = 0;
        if (tid < RADIX_BINS) {
#pragma unroll
            for (int q = 0; q < SCATTER_PARTS; q++) {
                tot += s_all[q * RADIX_BINS + tid];
                mine += s_before[q * RADIX_BINS + tid];
            }
        }
        const uint32_t smaller = block_excl_scan<SCATTER_WAVES>(tot, s_tmp, nullptr);
        if (tid < RADIX_BINS) s_base[tid] = smaller + mine;
        // A single-pass sort's digit IS the key, so key d ends up in [smaller, smaller + tot): workgroup 0 publishes the
        // ranges of the first `direct_ranges` keys with plain stores.  (The RANGES path's atomicMin / atomicMax pairs all
        // land on ~9 cache lines when there are only 135 keys: 100 us for the 1080p entry sort instead of 30.)
        if (direct_ranges && blockIdx.x == 0 && tid < direct_ranges)
            ranges[tid] = tot ? make_uint2(smaller, smaller + tot) : make_uint2(0xFFFFFFFFu, 0u);
        __syncthreads();                                     // s_wave is rewritten below
    }

    RADIX_PROF(1, wall_clock64());
    for (uint32_t tile = ch.tile_begin; tile < ch.tile_end; tile++) {
        const bool prof_tile = tile == ch.tile_begin;
        (void)prof_tile;
        uint32_t key[SCATTER_ITEMS], val[SCATTER_ITEMS], rank[SCATTER_ITEMS];
        bool ok[SCATTER_ITEMS];
        typename Loader::Raw raw[SCATTER_ITEMS];
        // wave-striped load: the stable order inside a tile is (wave, r, lane).  All memory reads first, then the arithmetic
        // on them.  (Issuing the next tile's reads before ranking this one changed nothing: 0.100 vs 0.100 ms per C3 sort.)
        const uint32_t wbase = tile * RADIX_TILE + wave * (64 * SCATTER_ITEMS) + lane;
#pragma unroll
        for (int r = 0; r < SCATTER_ITEMS; r++) {
            const uint32_t j = wbase + r * 64;
            ok[r] = j < ch.n && ld.valid(j);      // a loader that drops elements turns the pass into a stable compaction
            if (ok[r]) raw[r] = ld.fetch(j);
        }
#pragma unroll
        for (int r = 0; r < SCATTER_ITEMS; r++) {
            key[r] = 0xFFFFFFFFu;
            val[r] = 0u;
            if (ok[r]) ld.decode(raw[r], key[r], val[r]);
        }
        RADIX_PROF_WAIT();
        if (prof_tile) RADIX_PROF(2, wall_clock64());
#pragma unroll
        for (int k = 0; k < SCATTER_WAVES * RADIX_BINS / SCATTER_THREADS; k++) (&s_wave[0][0])[k * SCATTER_THREADS + tid] = 0;
        __syncthreads();

        if (ATOMIC_RANK) {
#pragma unroll
            for (int r = 0; r < SCATTER_ITEMS; r++) {
                const uint32_t digit = (key[r] >> shift) & 255u;
                // lanes of this instruction that share `digit` are served in ascending lane order (see header)
                if (ok[r]) rank[r] = atomicAdd(&s_wave[wave][digit], 1u);
            }
        } else {
#pragma unroll
            for (int r = 0; r < SCATTER_ITEMS; r++) {
                const uint32_t digit = (key[r] >> shift) & 255u;
                uint64_t same = __ballot(ok[r]);
#pragma unroll
                for (int b = 0; b < 8; b++) {
                    const uint64_t vote = __ballot(ok[r] && ((digit >> b) & 1u));
                    same &= ((digit >> b) & 1u) ? vote : ~vote;
                }
                if (ok[r]) {
                    const uint32_t prior = my_hist[digit];
                    rank[r] = prior + __popcll(same & lt_mask);
                    if ((same >> lane) == 1ull) my_hist[digit] = prior + __popcll(same);   // highest lane of the group
                }
            }
        }
        __syncthreads();
        if (prof_tile) RADIX_PROF(3, wall_clock64());

        uint32_t tile_count = 0;                    // keys staged by this tile (= its length unless the loader drops some)
        {   // thread t < 256 owns digit t: wave-exclusive offsets and the tile-local digit base
            uint32_t tot = 0;
            if (tid < RADIX_BINS) {
#pragma unroll
                for (int w = 0; w < SCATTER_WAVES; w++) {
                    const uint32_t c = s_wave[w][tid];
                    s_wave[w][tid] = tot;
                    tot += c;
                }
                s_total[tid] = tot;
            }
            const uint32_t excl = block_excl_scan<SCATTER_WAVES>(tot, s_tmp, &tile_count);   // contains the barriers
            if (tid < RADIX_BINS) s_local[tid] = excl;
        }
        __syncthreads();
        if (prof_tile) RADIX_PROF(4, wall_clock64());
#pragma unroll
        for (int r = 0; r < SCATTER_ITEMS; r++) {
            if (ok[r]) {
                const uint32_t digit = (key[r] >> shift) & 255u;
                const uint32_t pos = s_local[digit] + s_wave[wave][digit] + rank[r];
                s_keys[pos] = (KeyOutT)key[r];
                s_vals[pos] = val[r];
            }
        }
        __syncthreads();
        if (prof_tile) RADIX_PROF(5, wall_clock64());
#pragma unroll
        for (int k = 0; k < SCATTER_ITEMS; k++) {
            const uint32_t e = k * SCATTER_THREADS + tid;
            if (e < tile_count) {
                const uint32_t kk = s_keys[e];
                const uint32_t digit = (kk >> shift) & 255u;
                const uint32_t g = s_base[digit] + (e - s_local[digit]);
                if (WRITE_KEYS) keys_out[g] = (KeyOutT)kk;
                vals_out[g] = s_vals[e];
                if (RANGES) {
                    if (e == 0 || (uint32_t)s_keys[e - 1] != kk) atomicMin(&ranges[kk].x, g);
                    if (e + 1 == tile_count || (uint32_t)s_keys[e + 1] != kk) atomicMax(&ranges[kk].y, g + 1u);
                }
            }
        }
        RADIX_PROF_WAIT();
        __syncthreads();
        if (prof_tile) RADIX_PROF(6, wall_clock64());
        if (tid < RADIX_BINS) s_base[tid] += s_total[tid];
        // the next iteration's first barrier orders this update before its use
    }
    RADIX_PROF(7, wall_clock64());
    RADIX_PROF(8, (unsigned long long)(ch.tile_end - ch.tile_begin));
}

// ---------------------------------------------------------------------------------------------------
// host launcher for one pass
// ---------------------------------------------------------------------------------------------------
inline uint32_t radix_grid_for(uint32_t n_upper) {
    uint32_t tiles = (n_upper + RADIX_TILE - 1) / RADIX_TILE;
    if (tiles < 1) tiles = 1;
    if (tiles <= (uint32_t)RADIX_MAX_BLOCKS) return tiles;
    const uint32_t per = (tiles + RADIX_MAX_BLOCKS - 1) / RADIX_MAX_BLOCKS;
    return (tiles + per - 1) / per;
}

// n_upper: host-side upper bound of the element count (sizes the grid); pass_slot picks the zeroed
// digit_total row (the caller zeroes RadixScratch::digit_total once per frame).
template <class Loader, class KeyOutT, bool WRITE_KEYS, bool RANGES = false>
int radix_pass(const RadixExec& ex, const Loader& ld_hist, const Loader& ld, uint32_t n_upper, int shift, int pass_slot,
               KeyOutT* keys_out, uint32_t* vals_out, uint2* ranges = nullptr, uint32_t direct_ranges = 0) {
    const uint32_t grid = radix_grid_for(n_upper);
    uint32_t* bh = ex.scratch->block_hist.as<uint32_t>();
    uint32_t* dt = ex.scratch->digit_total.as<uint32_t>() + pass_slot * RADIX_MAX_GROUPS * RADIX_BINS;
    hipLaunchKernelGGL((k_radix_hist<Loader>), dim3(grid), dim3(HIST_THREADS), 0, ex.stream, ld_hist, shift, bh, dt);
    if (ex.atomic_rank)
        hipLaunchKernelGGL((k_radix_scatter<Loader, KeyOutT, WRITE_KEYS, RANGES, true>), dim3(grid), dim3(SCATTER_THREADS), 0,
                           ex.stream, ld, shift, bh, dt, keys_out, vals_out, ranges, direct_ranges);
    else
        hipLaunchKernelGGL((k_radix_scatter<Loader, KeyOutT, WRITE_KEYS, RANGES, false>), dim3(grid), dim3(SCATTER_THREADS), 0,
                           ex.stream, ld, shift, bh, dt, keys_out, vals_out, ranges, direct_ranges);
    GS_HIP(hipGetLastError());
    return GS_OK;
}
