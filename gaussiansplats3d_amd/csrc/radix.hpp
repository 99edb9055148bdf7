// radix.hpp — stable LSD radix sort passes (8-bit digits) for gfx950, wave64.
//
// Used twice per frame:
//   * the depth sort that replaces the reference's counting sort (/root/reference/src/worker/sorter.cpp:142-167):
//     16..24-bit bucket keys, 2..3 passes, payload = global splat index;
//   * the tile-entry sort of the rasteriser (key = tile id, payload = splat index), which must be STABLE so
//     that each tile's list keeps the depth order established by the first sort.
//
// One pass = k_radix_hist -> k_radix_scatter (tile at a time) or k_radix_scatter_chunk (the depth sort's packed passes: a
// workgroup's whole chunk staged in LDS, one store sweep).  The histogram kernel leaves a two-level table: one 1 KiB row of
// digit counts per workgroup plus one row per GROUP of 32 workgroups (atomics, 32 adders per word); every scatter workgroup
// sums the group rows and the <= 31 rows of its own group that precede it (coalesced 1 KiB rows, batches of 8 loads) and scans
// the 256 digit totals itself, so no separate scan kernel (10.7 us and two kernel boundaries per pass on MI355X) sits
// between the two.  Deterministic: integer sums only, no inter-workgroup spinning, no dependence on dispatch order
// (cdna_hip_programming.md §6 G16).  A workgroup walks a contiguous run of 4096-key tiles (C3's depth sort: 472 workgroups x 3
// tiles), the table has <= RADIX_MAX_BLOCKS rows whatever N is, and N may live in device memory (the tile-entry count and
// the length of a frustum-culled list only exist on the device).
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

// The same scan on the DPP network (no LDS round trips): Hillis-Steele inside each row of 16 lanes (row_shr 1, 2, 4, 8: a lane
// whose source falls outside its row adds 0), then lane 15 of row r into row r + 1 (rows 1 and 3), then lane 31 into rows 2 and 3.
__device__ __forceinline__ uint32_t wave_incl_scan_dpp(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
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

// Element `index` of an array that is < 4 GiB long (every array a radix pass touches: gs_sorter_create / the mesh bound their
// element counts): the byte offset is formed in 32 bits, so the access is `global_load v, v_offset, s[base:base+1]` - ONE address
// VGPR per access instead of a 64-bit pair (the scatter kernel holds 16 loads in flight per lane; with 64-bit addresses it
// spilled 44 dwords per lane at 64 VGPRs).
template <class T>
__device__ __forceinline__ T ld32(const T* base, uint32_t index) {
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + index * (uint32_t)sizeof(T));
}
template <class T>
__device__ __forceinline__ void st32(T* base, uint32_t index, T v) {
    *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + index * (uint32_t)sizeof(T)) = v;
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
    // pre + fetch = the memory reads of element j (two dependent stages), decode = the arithmetic on them (none here)
    struct Raw { uint32_t key, val; };
    __device__ __forceinline__ uint32_t pre(uint32_t) const { return 0u; }       // (a loader with an index list reads it here)
    __device__ __forceinline__ Raw fetch(uint32_t j, uint32_t) const { return Raw{(uint32_t)ld32(keys, j), ld32(vals, j)}; }
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
// Chunk = a contiguous run of tiles; chunk c's row of the offset table is row c.  The grid is sized for the host's upper bound
// of the list, the real length may live on the device: only the first `active` chunks hold tiles.  RADIX_XCD_CONTIG: the
// workgroups of one XCD (b % 8, the observed dispatch rule: speed only) take CONTIGUOUS chunks of the ACTIVE ones - worth
// 1.5-2.5 us on a full C3 / C2 sort - while a short list (a rank's strip, a culled list) still spreads over all eight XCDs
// (mapping over the whole grid instead put a 68-tile list on two XCDs: a rank's frame 0.227 -> 0.241 ms, r04).
#ifndef RADIX_XCD_CONTIG
#define RADIX_XCD_CONTIG 1
#endif
__device__ __forceinline__ RadixChunk radix_chunk(uint32_t n) {
    const uint32_t tiles = (n + RADIX_TILE - 1) / RADIX_TILE;
    const uint32_t per = (tiles + gridDim.x - 1) / gridDim.x;
    const uint32_t active = per ? (tiles + per - 1) / per : 0u;          // chunks that hold tiles (<= gridDim.x)
    RadixChunk c;
    c.n = n;
#if RADIX_XCD_CONTIG
    const uint32_t x = blockIdx.x % 8u, k = blockIdx.x / 8u, q = active / 8u, rem = active % 8u;
    c.id = k < q + (x < rem ? 1u : 0u) ? x * q + min(x, rem) + k : 0xFFFFFFFFu;   // XCD x owns q (+1 for x < rem) chunks
#else
    c.id = blockIdx.x < active ? blockIdx.x : 0xFFFFFFFFu;
#endif
    if (c.id == 0xFFFFFFFFu) {               // no tiles for this workgroup (its id only names a row it never touches)
        c.id = 0;
        c.tile_begin = c.tile_end = tiles;
    } else {
        c.tile_begin = min(c.id * per, tiles);
        c.tile_end = min(c.tile_begin + per, tiles);
    }
    return c;
}

// The histogram kernel runs 1024-thread workgroups: its grid is the scatter's (one table row per workgroup, <= 512 of
// them), and a gather-type loader (the depth keys) needs more waves in flight than 4 per workgroup to hide its latency
// (r01e: 20.5 us with 256 threads).
#ifndef HIST_THREADS_CFG
#define HIST_THREADS_CFG 1024
#endif
constexpr int HIST_THREADS = HIST_THREADS_CFG;
constexpr int HIST_ITEMS = RADIX_TILE / HIST_THREADS;

// The histogram of a workgroup's tiles, HIST_GROUP tiles at a time: all their loads are issued before the first LDS atomic,
// so the workgroup waits for one memory round trip per group instead of one per tile.  Measured r03 (same box, ab_libs): C4 sort
// 0.240 -> 0.231 ms, entry sort 0.245 -> 0.241; C3 / C5 within noise (their histograms already overlapped).  Order is irrelevant
// for a histogram.
#ifndef HIST_GROUP_CFG
#define HIST_GROUP_CFG 4
#endif
constexpr uint32_t HIST_GROUP = HIST_GROUP_CFG;

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
    // (unconditional loads of clamped positions: one batch, no exec-masked branch per load; n > 0 when a workgroup has tiles)
#pragma unroll
    for (uint32_t g = 0; g < HIST_GROUP; g++)
#pragma unroll
        for (int r = 0; r < HIST_ITEMS; r++) {
            const uint32_t j = (tile + g) * RADIX_TILE + r * HIST_THREADS + threadIdx.x;
            const uint32_t jc = j < n ? j : n - 1u;
            ok[g][r] = g < tiles && j < n && ld.valid(jc);
            raw[g][r] = ld.hist_fetch(jc);
        }
#pragma unroll
    for (uint32_t g = 0; g < HIST_GROUP; g++)
#pragma unroll
        for (int r = 0; r < HIST_ITEMS; r++) {
            if (ok[g][r]) atomicAdd(&hist[(ld.hist_key(raw[g][r]) >> shift) & 255u], 1u);     // (hist_key may count a clamp: valid elements only)
        }
}

template <class Loader>
__global__ __launch_bounds__(HIST_THREADS) void k_radix_hist(Loader ld, int shift, uint32_t* __restrict__ block_hist,
                                                             uint32_t* __restrict__ digit_total) {
    __shared__ uint32_t s_hist[4][RADIX_BINS];               // waves w, w+4, w+8, w+12 share one
    ld.prepare();
    const RadixChunk ch = radix_chunk(ld.count());
    const uint32_t tid = threadIdx.x, wave = (tid >> 6) & 3u;
    for (uint32_t k = tid; k < 4u * RADIX_BINS; k += HIST_THREADS) (&s_hist[0][0])[k] = 0;
    __syncthreads();
    for (uint32_t tile = ch.tile_begin; tile < ch.tile_end; tile += HIST_GROUP)
        hist_tiles(ld, tile, min(HIST_GROUP, ch.tile_end - tile), ch.n, shift, s_hist[wave]);
    __syncthreads();
    if (tid >= RADIX_BINS || ch.tile_begin >= ch.tile_end) return;      // (a workgroup without tiles has no row)
    const uint32_t total = s_hist[0][tid] + s_hist[1][tid] + s_hist[2][tid] + s_hist[3][tid];
    block_hist[ch.id * RADIX_BINS + tid] = total;                // one coalesced 1 KiB row per workgroup
    // Group rows.  Same-address atomics serialise in the fabric (~12 ns each, MI355X_MICROARCH.md row "fanin"): with 32
    // adders per word they stay < 1 us, where one word per digit for the whole grid cost 10-15 us per pass (A/B r01b).
    if (total) atomicAdd(&digit_total[(ch.id / RADIX_GROUP) * RADIX_BINS + tid], total);
}


// One scatter pass.  Template switches:
//   WRITE_KEYS  also emit the keys (every pass but the last of a key + value sort).
//   RANGES      the last pass of a multi-pass tile sort: publish each key's [begin,end) in the sorted output.  Inside a
//               workgroup tile equal keys are contiguous (the earlier passes ordered the lower digits, this pass is stable),
//               so a run boundary costs one atomicMin/atomicMax pair; `ranges` must be pre-set to (0xFFFFFFFF, 0).
//   PACK_OUT    key and value leave as ONE 32-bit word, (key >> (shift + 8)) << val_bits | value: what is left of the key
//               after this pass's digit, above a value of val_bits bits (the depth sort: 16-bit buckets over < 2^24 splat
//               positions -> 4 bytes per element instead of 2 + 4, and digit runs of whole words).  The digit cannot be
//               recovered from that word, so the staging area holds it as a byte (KeyOutT = uint8_t).
//
// Geometry: 512 threads = 8 waves x 8 keys per lane over a 4096-key tile on a grid of <= RADIX_TILE_GRID = 512 workgroups
// (fewer, longer workgroups win: gs_internal.hpp).  A tile is a chain of load -> rank -> offsets -> reorder -> store phases;
// round 4 cut it to three barriers: ranks are taken on a per-wave histogram that only its own wave zeroes and increments (no
// barrier between the two), ONE wave turns all counts into offsets (four digits per lane, 16-byte LDS accesses, the scan over
// the 256 digits on the DPP network), the offsets are folded into the per-wave table (one LDS look-up per key in the reorder),
// a full tile is stored without predicates (all LDS reads, then all stores), and the next tile's loads are issued before the
// stores of this one (the first tile's before the offset prologue's), as branch-free batches with 32-bit offsets.
// Measured (isolated C3 depth sort, same box, profiles/r04*): r03 0.0865 ms -> 0.0717 ms with this kernel for both passes ->
// 0.0667 ms with the chunk-staged kernel below for the packed passes (which the depth sort uses whenever it can; this kernel
// remains for key + value sorts - the tile-entry sort, the octree and Morton sorts - and for lists of more than 25 M keys).
// 64 VGPRs (4 workgroups per CU) was tried and is slower (spills; and more than two workgroups per CU do not help: r04a
// 1024-workgroup grids 0.0795-0.0822 ms against 0.0727-0.0748 at 512).
#ifndef SCATTER_THREADS_CFG
#define SCATTER_THREADS_CFG 512
#endif
#ifndef SCATTER_OCC_CFG
#define SCATTER_OCC_CFG 5              // waves per SIMD the register allocation must allow (5 <=> 96 VGPRs <=> 2 workgroups of 512 per CU)
#endif
constexpr int SCATTER_THREADS = SCATTER_THREADS_CFG;
constexpr int SCATTER_WAVES = SCATTER_THREADS / 64;
constexpr int SCATTER_ITEMS = RADIX_TILE / SCATTER_THREADS;
constexpr int SCATTER_PARTS = SCATTER_THREADS / RADIX_BINS;     // threads per digit in the offset prologue
static_assert(SCATTER_THREADS % RADIX_BINS == 0 && 2 * SCATTER_PARTS <= SCATTER_WAVES, "offset prologue layout");
static_assert(SCATTER_ITEMS * SCATTER_THREADS == RADIX_TILE && SCATTER_ITEMS <= 32, "tile layout");

#ifdef GS_RADIX_PROFILE
// tools/radix_profile.py: per scatter workgroup (slot = shift / 8 of the depth sort: 0 / 1) the 100 MHz clock at kernel start,
// after the offset prologue, and for its FIRST tile: data decoded, ranked, offsets scanned, reordered in LDS, stored; then the
// end of the kernel and the number of tiles
static __device__ unsigned long long g_radix_prof[2 * 1024 * 10];      // one copy per translation unit; sorter.hip's is read
#define RADIX_PROF(slot, v) do { const int ps_ = Loader::prof_slot(shift); if (threadIdx.x == 0 && blockIdx.x < 1024u && ps_ >= 0) g_radix_prof[(ps_ * 1024 + blockIdx.x) * 10 + (slot)] = (v); } while (0)
#define RADIX_PROF_WAIT() __builtin_amdgcn_s_waitcnt(0)
#else
#define RADIX_PROF(slot, v) do { } while (0)
#define RADIX_PROF_WAIT() do { } while (0)
#endif

template <class Loader, class KeyOutT, bool WRITE_KEYS, bool RANGES, bool ATOMIC_RANK, bool PACK_OUT>
__global__ __launch_bounds__(SCATTER_THREADS, SCATTER_OCC_CFG) void k_radix_scatter(Loader ld, int shift,
                                                                      const uint32_t* __restrict__ block_hist,
                                                                      const uint32_t* __restrict__ group_hist,
                                                                      KeyOutT* __restrict__ keys_out,
                                                                      uint32_t* __restrict__ vals_out, uint2* ranges,
                                                                      uint32_t direct_ranges, uint32_t val_bits,
                                                                      uint32_t* __restrict__ count_out) {
    static_assert(!PACK_OUT || (sizeof(KeyOutT) == 1 && !WRITE_KEYS && !RANGES), "a packing pass stages the digit as a byte");
    __shared__ KeyOutT s_keys[RADIX_TILE];                  // staged in the output key width (PACK_OUT: the digit)
    __shared__ uint32_t s_vals[RADIX_TILE];                 // the value (PACK_OUT: the packed word)
    __shared__ __attribute__((aligned(16))) uint32_t s_wave[SCATTER_WAVES][RADIX_BINS];  // per-wave digit counts, then: first staging slot of (wave, digit)
    __shared__ __attribute__((aligned(16))) uint32_t s_gbase[RADIX_BINS];   // global slot of a digit's first staged key minus its staging slot
    __shared__ __attribute__((aligned(16))) uint32_t s_base[RADIX_BINS];    // next global slot of each digit for this workgroup
    __shared__ uint32_t s_tmp[SCATTER_WAVES];

    RADIX_PROF(0, wall_clock64());
    ld.prepare();
    const RadixChunk ch = radix_chunk(ld.count());
    // A grid is sized for the host's upper bound of the list; when the real length lives on the device (a culled or gathered
    // list, a rank's strip: 0.27 M of 5.8 M) most workgroups have no tile - they leave before the offset prologue (workgroup 0
    // stays: it publishes the ranges / the kept count even of an empty list)
    if (ch.tile_begin >= ch.tile_end && blockIdx.x != 0) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    volatile uint32_t* my_hist = s_wave[wave];
    const int dshift = PACK_OUT ? 0 : shift;                // the staged key of a packing pass IS the digit

    uint32_t okm = 0;                                       // bit r: element r of this lane exists (and survives the loader)
    typename Loader::Raw raw[SCATTER_ITEMS];
    // wave-striped load: the stable order inside a tile is (wave, r, lane).  All memory reads first, then the arithmetic.
    // Branch-free: an element beyond the list re-reads the last one (its okm bit stays clear), so the eight loads of a lane are
    // one batch behind one wait instead of eight exec-masked branches.
    auto fetch_tile = [&](uint32_t tile) {
        const uint32_t wbase = tile * RADIX_TILE + wave * (64 * SCATTER_ITEMS) + lane;
        uint32_t o[SCATTER_ITEMS];
        okm = 0;
#pragma unroll
        for (int r = 0; r < SCATTER_ITEMS; r++) {
            const uint32_t j = wbase + r * 64;
            const bool in = j < ch.n;
            const uint32_t jc = in ? j : ch.n - 1u;
            const bool ok = in && ld.valid(jc);              // a loader that drops elements turns the pass into a stable compaction
            o[r] = ld.pre(jc);
            okm |= ok ? (1u << r) : 0u;
        }
#pragma unroll
        for (int r = 0; r < SCATTER_ITEMS; r++) {
            const uint32_t j = wbase + r * 64;
            raw[r] = ld.fetch(j < ch.n ? j : ch.n - 1u, o[r]);
        }
    };

    {   // this workgroup's first output slot per digit = keys with a smaller digit + keys of this digit in earlier
        // workgroups; SCATTER_PARTS threads per digit split the rows, every load is independent of the others
        const uint32_t d = tid & 255u, half = tid >> 8;      // `half` = which of the SCATTER_PARTS row subsets
        const uint32_t g = ch.id / RADIX_GROUP, groups = (gridDim.x + RADIX_GROUP - 1) / RADIX_GROUP;
        uint32_t before = 0, all = 0;
        if (ch.tile_begin < ch.tile_end) fetch_tile(ch.tile_begin);      // the first tile travels with the table rows
        // batches of 8 independent loads (clamped rows + selects: no exec-masked branch per load); two to four round trips to L2
        for (uint32_t r0 = half; r0 < groups; r0 += 8u * SCATTER_PARTS) {
            uint32_t v[8];
#pragma unroll
            for (uint32_t k = 0; k < 8u; k++) v[k] = ld32(group_hist, min(r0 + k * SCATTER_PARTS, groups - 1u) * RADIX_BINS + d);
#pragma unroll
            for (uint32_t k = 0; k < 8u; k++) {
                const uint32_t r = r0 + k * SCATTER_PARTS;
                all += r < groups ? v[k] : 0u;
                before += r < g ? v[k] : 0u;
            }
        }
        for (uint32_t r0 = g * RADIX_GROUP + half; r0 < ch.id; r0 += 8u * SCATTER_PARTS) {
            uint32_t v[8];
#pragma unroll
            for (uint32_t k = 0; k < 8u; k++) v[k] = ld32(block_hist, min(r0 + k * SCATTER_PARTS, ch.id) * RADIX_BINS + d);
#pragma unroll
            for (uint32_t k = 0; k < 8u; k++) before += (r0 + k * SCATTER_PARTS) < ch.id ? v[k] : 0u;
        }
        uint32_t* s_before = &s_wave[0][0];                  // [PARTS][256], free until the tile loop uses it
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
        uint32_t n_valid;
        const uint32_t smaller = block_excl_scan<SCATTER_WAVES>(tot, s_tmp, &n_valid);    // (two barriers: s_wave is free after it)
        if (tid < RADIX_BINS) s_base[tid] = smaller + mine;  // (read after barrier (1) of the first tile)
        // a compacting pass (a loader that drops elements) publishes how many it keeps: the sum of its digit totals - one plain
        // store, where one atomicAdd per workgroup of the key kernel on a single word cost ~12 ns each (25 us for 2048 of them)
        if (count_out && blockIdx.x == 0 && tid == 0) *count_out = n_valid;
        // A single-pass sort's digit IS the key, so key d ends up in [smaller, smaller + tot): workgroup 0 publishes the
        // ranges of the first `direct_ranges` keys with plain stores.  (The RANGES path's atomicMin / atomicMax pairs all
        // land on ~9 cache lines when there are only 135 keys: 100 us for the 1080p entry sort instead of 30.)
        if (direct_ranges && blockIdx.x == 0 && tid < direct_ranges)
            ranges[tid] = tot ? make_uint2(smaller, smaller + tot) : make_uint2(0xFFFFFFFFu, 0u);
    }
    // every wave zeroes, increments and (after the scan) reads only ITS OWN row of s_wave
    *reinterpret_cast<uint4*>(&s_wave[wave][4 * lane]) = make_uint4(0u, 0u, 0u, 0u);

    RADIX_PROF(1, wall_clock64());
    for (uint32_t tile = ch.tile_begin; tile < ch.tile_end; tile++) {
        const bool prof_tile = tile == ch.tile_begin;
        (void)prof_tile;
        // per element: `a` = what is staged as the value (PACK_OUT: the packed word), `b` = the staged key (PACK_OUT: the digit),
        // `rk` = its rank among the equal digits its wave has seen in this tile
        uint32_t a[SCATTER_ITEMS], b[SCATTER_ITEMS], rk[SCATTER_ITEMS];
#pragma unroll
        for (int r = 0; r < SCATTER_ITEMS; r++) {
            uint32_t k = 0xFFFFFFFFu, v = 0u;
            ld.decode(raw[r], k, v);
            if (PACK_OUT) {
                a[r] = ((k >> (shift + 8)) << val_bits) | v;
                b[r] = (k >> shift) & 255u;
            } else {
                a[r] = v;
                b[r] = k;
            }
        }
        RADIX_PROF_WAIT();
        if (prof_tile) RADIX_PROF(2, wall_clock64());

        if (ATOMIC_RANK) {
#pragma unroll
            for (int r = 0; r < SCATTER_ITEMS; r++) {
                const uint32_t digit = (b[r] >> dshift) & 255u;
                // lanes of this instruction that share `digit` are served in ascending lane order (see header)
                rk[r] = 0u;
                if ((okm >> r) & 1u) rk[r] = atomicAdd(&s_wave[wave][digit], 1u);
            }
        } else {
#pragma unroll
            for (int r = 0; r < SCATTER_ITEMS; r++) {
                const uint32_t digit = (b[r] >> dshift) & 255u;
                const bool ok = (okm >> r) & 1u;
                uint64_t same = __ballot(ok);
#pragma unroll
                for (int bit = 0; bit < 8; bit++) {
                    const uint64_t vote = __ballot(ok && ((digit >> bit) & 1u));
                    same &= ((digit >> bit) & 1u) ? vote : ~vote;
                }
                rk[r] = 0u;
                if (ok) {
                    const uint32_t prior = my_hist[digit];
                    rk[r] = prior + __popcll(same & lt_mask);
                    if ((same >> lane) == 1ull) my_hist[digit] = prior + __popcll(same);   // highest lane of the group
                }
            }
        }
        __syncthreads();                                     // (1) every wave's digit counts are final
        if (prof_tile) RADIX_PROF(3, wall_clock64());

        // ONE wave turns the counts into offsets, four digits per lane (16-byte LDS accesses, the scan on the DPP network): no
        // second barrier, no cross-wave exchange.  s_wave[w][d] becomes the first staging slot of (wave w, digit d); s_gbase[d] =
        // global slot of digit d's first staged key minus its staging slot; s_base[d] = the workgroup's next global slot of d.
        if (wave == 0) {
            uint4 tot = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
            for (int w = 0; w < SCATTER_WAVES; w++) {
                const uint4 c = *reinterpret_cast<const uint4*>(&s_wave[w][4 * lane]);
                tot.x += c.x; tot.y += c.y; tot.z += c.z; tot.w += c.w;
            }
            const uint32_t t4 = tot.x + tot.y + tot.z + tot.w;
            const uint32_t incl = wave_incl_scan_dpp(t4);
            uint4 first;                                     // first staging slot of each of the four digits
            first.x = incl - t4;
            first.y = first.x + tot.x;
            first.z = first.y + tot.y;
            first.w = first.z + tot.z;
            uint4 base = *reinterpret_cast<const uint4*>(&s_base[4 * lane]);
            *reinterpret_cast<uint4*>(&s_gbase[4 * lane]) = make_uint4(base.x - first.x, base.y - first.y, base.z - first.z, base.w - first.w);
            base.x += tot.x; base.y += tot.y; base.z += tot.z; base.w += tot.w;
            *reinterpret_cast<uint4*>(&s_base[4 * lane]) = base;
#pragma unroll
            for (int w = 0; w < SCATTER_WAVES; w++) {
                uint4* row = reinterpret_cast<uint4*>(&s_wave[w][4 * lane]);
                const uint4 c = *row;
                *row = first;
                first.x += c.x; first.y += c.y; first.z += c.z; first.w += c.w;
            }
            if (lane == 63) s_tmp[0] = incl;                 // keys staged by this tile (= its length unless the loader drops some)
        }
        __syncthreads();                                     // (2)
        const uint32_t tile_count = s_tmp[0];
        if (prof_tile) RADIX_PROF(4, wall_clock64());
#pragma unroll
        for (int r = 0; r < SCATTER_ITEMS; r++) {
            if ((okm >> r) & 1u) {
                const uint32_t digit = (b[r] >> dshift) & 255u;
                const uint32_t pos = s_wave[wave][digit] + rk[r];
                s_keys[pos] = (KeyOutT)b[r];
                s_vals[pos] = a[r];
            }
        }
        // the registers are free: the next tile's reads travel while this one is stored
        okm = 0;
        if (tile + 1 < ch.tile_end) fetch_tile(tile + 1);
        __syncthreads();                                     // (3)
        if (prof_tile) RADIX_PROF(5, wall_clock64());
        *reinterpret_cast<uint4*>(&s_wave[wave][4 * lane]) = make_uint4(0u, 0u, 0u, 0u);   // own row, read for the last time above
        if (!RANGES && tile_count == (uint32_t)RADIX_TILE) {
            // a full tile: every LDS read first, then the stores, no predicate
            uint32_t kk[SCATTER_ITEMS], vv[SCATTER_ITEMS], gg[SCATTER_ITEMS];
#pragma unroll
            for (int k = 0; k < SCATTER_ITEMS; k++) {
                kk[k] = s_keys[k * SCATTER_THREADS + tid];
                vv[k] = s_vals[k * SCATTER_THREADS + tid];
            }
#pragma unroll
            for (int k = 0; k < SCATTER_ITEMS; k++) gg[k] = s_gbase[(kk[k] >> dshift) & 255u] + (k * SCATTER_THREADS + tid);
#pragma unroll
            for (int k = 0; k < SCATTER_ITEMS; k++) {
                if (WRITE_KEYS) st32(keys_out, gg[k], (KeyOutT)kk[k]);
                st32(vals_out, gg[k], vv[k]);
            }
        } else {
#pragma unroll
            for (int k = 0; k < SCATTER_ITEMS; k++) {
                const uint32_t e = k * SCATTER_THREADS + tid;
                if (e < tile_count) {
                    const uint32_t kk = s_keys[e];
                    const uint32_t digit = (kk >> dshift) & 255u;
                    const uint32_t g = s_gbase[digit] + e;
                    if (WRITE_KEYS) st32(keys_out, g, (KeyOutT)kk);
                    st32(vals_out, g, s_vals[e]);
                    if (RANGES) {
                        if (e == 0 || (uint32_t)s_keys[e - 1] != kk) atomicMin(&ranges[kk].x, g);
                        if (e + 1 == tile_count || (uint32_t)s_keys[e + 1] != kk) atomicMax(&ranges[kk].y, g + 1u);
                    }
                }
            }
        }
        RADIX_PROF_WAIT();
        if (prof_tile) RADIX_PROF(6, wall_clock64());
        // no barrier here: the staging area is next written after barrier (2) of the next tile, s_gbase and s_tmp after its
        // barrier (1), and a wave passes those only when every wave has left this loop
    }
    RADIX_PROF(7, wall_clock64());
    RADIX_PROF(8, (unsigned long long)(ch.tile_end - ch.tile_begin));
}

// ---------------------------------------------------------------------------------------------------
// chunk-staged scatter (round 4): the depth sort's passes
// ---------------------------------------------------------------------------------------------------
// The tile kernel above stores every 4096-key tile on its own: 256 digit runs of ~16 keys = 64 bytes each, at arbitrary
// alignment, and gfx950's L2 forwards a store to the fabric as it arrives (MI355X_MICROARCH.md: "all bytes leave L2 every pass"),
// so a 64-byte run costs two 64-byte write requests - the counters had pass 0 at 1.95x its algorithmic write bytes
// (profiles/r04c).  Here a workgroup keeps its WHOLE chunk (<= CHUNK_TILES tiles) in LDS: the histogram kernel's row of this
// chunk already holds its digit counts, so every key's final staging slot is known while the tiles are still being ranked, and
// ONE sweep at the end stores digit runs that are CHUNK_TILES times longer (3 tiles: 48 keys = 192 bytes, 1.33 requests per
// useful one).  A tile is rank -> barrier -> per-wave offsets -> barrier -> reorder (two barriers, no store phase, no digit scan);
// the scan over the 256 digits happens once per chunk.  Elements travel as ONE word:
//   PACK_OUT   out word = (key >> (shift + 8)) << val_bits | value (what the next pass reads), digit staged as a byte beside it;
//   otherwise  the last pass after a packing one: staged digit << 24 | value (value < 2^24 whenever a pass packs 8 key bits), out = value.
// Geometry: 512 threads, 8 keys per lane and tile; LDS = 5 (4) bytes per key of the chunk + 8 KB: two workgroups per CU.
#ifndef CHUNK_TILES_CFG
#define CHUNK_TILES_CFG 3
#endif
#ifndef CHUNK_THREADS_CFG
#define CHUNK_THREADS_CFG 512
#endif
#ifndef CHUNK_OCC_CFG
#define CHUNK_OCC_CFG 4
#endif
constexpr int CHUNK_TILES = CHUNK_TILES_CFG;
constexpr int CHUNK_THREADS = CHUNK_THREADS_CFG;
constexpr int CHUNK_WAVES = CHUNK_THREADS / 64;
constexpr int CHUNK_ITEMS = RADIX_TILE / CHUNK_THREADS;
constexpr int CHUNK_PARTS = CHUNK_THREADS / RADIX_BINS;
constexpr int CHUNK_KEYS = CHUNK_TILES * RADIX_TILE;
#ifndef CHUNK_TARGET_BLOCKS_CFG
#define CHUNK_TARGET_BLOCKS_CFG 512
#endif
constexpr int CHUNK_TARGET_BLOCKS = CHUNK_TARGET_BLOCKS_CFG;                   // two workgroups per CU: the chunk length aims at this many of them
static_assert(CHUNK_THREADS % RADIX_BINS == 0 && 2 * CHUNK_PARTS <= CHUNK_WAVES, "offset prologue layout");

inline uint32_t radix_chunk_grid_for(uint32_t n_upper) {    // 0: too long for chunk-sized workgroups (the tile kernel takes it)
    uint32_t tiles = (n_upper + RADIX_TILE - 1) / RADIX_TILE;
    if (tiles < 1) tiles = 1;
    uint32_t per = (tiles + CHUNK_TARGET_BLOCKS - 1) / CHUNK_TARGET_BLOCKS;
    if (per > (uint32_t)CHUNK_TILES) per = CHUNK_TILES;
    const uint32_t grid = (tiles + per - 1) / per;
    return grid <= (uint32_t)RADIX_MAX_BLOCKS ? grid : 0u;
}

template <class Loader, bool PACK_OUT, bool ATOMIC_RANK>
__global__ __launch_bounds__(CHUNK_THREADS, CHUNK_OCC_CFG) void k_radix_scatter_chunk(Loader ld, int shift,
                                                                                     const uint32_t* __restrict__ block_hist,
                                                                                     const uint32_t* __restrict__ group_hist,
                                                                                     uint32_t* __restrict__ out, uint32_t val_bits,
                                                                                     uint32_t* __restrict__ count_out) {
    __shared__ uint32_t s_stage[CHUNK_KEYS];
    __shared__ uint8_t s_digit[PACK_OUT ? CHUNK_KEYS : 4];
    __shared__ __attribute__((aligned(16))) uint32_t s_wave[CHUNK_WAVES][RADIX_BINS];   // per-wave digit counts, then first slot of (wave, digit)
    __shared__ uint32_t s_run[RADIX_BINS];                  // next staging slot of each digit
    __shared__ uint32_t s_gbase[RADIX_BINS];                // global slot of a digit's run minus its staging slot
    __shared__ uint32_t s_tmp[CHUNK_WAVES];

    ld.prepare();
    const RadixChunk ch = radix_chunk(ld.count());
    if (ch.tile_begin >= ch.tile_end && blockIdx.x != 0) return;        // (no tile: see k_radix_scatter)
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    volatile uint32_t* my_hist = s_wave[wave];

    uint32_t okm = 0;
    typename Loader::Raw raw[CHUNK_ITEMS];
    auto fetch_tile = [&](uint32_t tile) {
        const uint32_t wbase = tile * RADIX_TILE + wave * (64 * CHUNK_ITEMS) + lane;
        uint32_t o[CHUNK_ITEMS];
        okm = 0;
#pragma unroll
        for (int r = 0; r < CHUNK_ITEMS; r++) {
            const uint32_t j = wbase + r * 64;
            const bool in = j < ch.n;
            const uint32_t jc = in ? j : ch.n - 1u;
            const bool ok = in && ld.valid(jc);
            o[r] = ld.pre(jc);
            okm |= ok ? (1u << r) : 0u;
        }
#pragma unroll
        for (int r = 0; r < CHUNK_ITEMS; r++) {
            const uint32_t j = wbase + r * 64;
            raw[r] = ld.fetch(j < ch.n ? j : ch.n - 1u, o[r]);
        }
    };
    if (ch.tile_begin < ch.tile_end) fetch_tile(ch.tile_begin);         // travels with the table rows below

    uint32_t chunk_count = 0;
    {   // global slot of this chunk's first key of every digit = keys with a smaller digit + keys of the digit in earlier chunks
        const uint32_t d = tid & 255u, part = tid >> 8;
        const uint32_t g = ch.id / RADIX_GROUP, groups = (gridDim.x + RADIX_GROUP - 1) / RADIX_GROUP;
        uint32_t before = 0, all = 0;
        // batches of 8 independent loads (clamped rows + selects: no exec-masked branch per load); two to four round trips to L2
        for (uint32_t r0 = part; r0 < groups; r0 += 8u * CHUNK_PARTS) {
            uint32_t v[8];
#pragma unroll
            for (uint32_t k = 0; k < 8u; k++) v[k] = ld32(group_hist, min(r0 + k * CHUNK_PARTS, groups - 1u) * RADIX_BINS + d);
#pragma unroll
            for (uint32_t k = 0; k < 8u; k++) {
                const uint32_t r = r0 + k * CHUNK_PARTS;
                all += r < groups ? v[k] : 0u;
                before += r < g ? v[k] : 0u;
            }
        }
        for (uint32_t r0 = g * RADIX_GROUP + part; r0 < ch.id; r0 += 8u * CHUNK_PARTS) {
            uint32_t v[8];
#pragma unroll
            for (uint32_t k = 0; k < 8u; k++) v[k] = ld32(block_hist, min(r0 + k * CHUNK_PARTS, ch.id) * RADIX_BINS + d);
#pragma unroll
            for (uint32_t k = 0; k < 8u; k++) before += (r0 + k * CHUNK_PARTS) < ch.id ? v[k] : 0u;
        }
        // this chunk's keys of digit d.  A workgroup WITHOUT tiles (only workgroup 0 of an empty list gets here) has no row: the
        // histogram kernel wrote none for it, and row 0 still holds the previous sort's counts - reading it made the final sweep
        // store that many never-written staging words through s_gbase of a zero table
        const uint32_t own = (tid < RADIX_BINS && ch.tile_begin < ch.tile_end) ? ld32(block_hist, ch.id * RADIX_BINS + d) : 0u;
        uint32_t* s_before = &s_wave[0][0];                  // [PARTS][256], free until the tile loop uses it
        uint32_t* s_all = &s_wave[CHUNK_PARTS][0];
        s_before[tid] = before;
        s_all[tid] = all;
        __syncthreads();
        uint32_t tot = 0, mine = 0;
        if (tid < RADIX_BINS) {
#pragma unroll
            for (int q = 0; q < CHUNK_PARTS; q++) {
                tot += s_all[q * RADIX_BINS + tid];
                mine += s_before[q * RADIX_BINS + tid];
            }
        }
        uint32_t n_valid;
        const uint32_t smaller = block_excl_scan<CHUNK_WAVES>(tot, s_tmp, &n_valid);
        if (count_out && blockIdx.x == 0 && tid == 0) *count_out = n_valid;              // (see k_radix_scatter)
        const uint32_t first = block_excl_scan<CHUNK_WAVES>(own, s_tmp, &chunk_count);   // first staging slot of digit d
        if (tid < RADIX_BINS) {
            s_run[tid] = first;
            s_gbase[tid] = smaller + mine - first;
        }
    }
    *reinterpret_cast<uint4*>(&s_wave[wave][4 * lane]) = make_uint4(0u, 0u, 0u, 0u);    // every wave zeroes and increments only its own row

    for (uint32_t tile = ch.tile_begin; tile < ch.tile_end; tile++) {
        uint32_t a[CHUNK_ITEMS], dg[CHUNK_ITEMS], rk[CHUNK_ITEMS];
#pragma unroll
        for (int r = 0; r < CHUNK_ITEMS; r++) {
            uint32_t k = 0xFFFFFFFFu, v = 0u;
            ld.decode(raw[r], k, v);
            dg[r] = (k >> shift) & 255u;
            a[r] = PACK_OUT ? (((k >> (shift + 8)) << val_bits) | v) : ((dg[r] << 24) | v);
        }
        if (ATOMIC_RANK) {
#pragma unroll
            for (int r = 0; r < CHUNK_ITEMS; r++) {
                rk[r] = 0u;
                if ((okm >> r) & 1u) rk[r] = atomicAdd(&s_wave[wave][dg[r]], 1u);   // same-digit lanes are served in lane order (header)
            }
        } else {
#pragma unroll
            for (int r = 0; r < CHUNK_ITEMS; r++) {
                const uint32_t digit = dg[r];
                const bool ok = (okm >> r) & 1u;
                uint64_t same = __ballot(ok);
#pragma unroll
                for (int bit = 0; bit < 8; bit++) {
                    const uint64_t vote = __ballot(ok && ((digit >> bit) & 1u));
                    same &= ((digit >> bit) & 1u) ? vote : ~vote;
                }
                rk[r] = 0u;
                if (ok) {
                    const uint32_t prior = my_hist[digit];
                    rk[r] = prior + __popcll(same & lt_mask);
                    if ((same >> lane) == 1ull) my_hist[digit] = prior + __popcll(same);
                }
            }
        }
        __syncthreads();                                     // (1) every wave's counts of this tile are final
        if (tid < RADIX_BINS) {                              // thread d: counts -> first slot of (wave, d), the digit's next slot moves on
            uint32_t run = s_run[tid];
#pragma unroll
            for (int w = 0; w < CHUNK_WAVES; w++) {
                const uint32_t c = s_wave[w][tid];
                s_wave[w][tid] = run;
                run += c;
            }
            s_run[tid] = run;
        }
        __syncthreads();                                     // (2)
#pragma unroll
        for (int r = 0; r < CHUNK_ITEMS; r++) {
            if ((okm >> r) & 1u) {
                const uint32_t pos = s_wave[wave][dg[r]] + rk[r];
                s_stage[pos] = a[r];
                if (PACK_OUT) s_digit[pos] = (uint8_t)dg[r];
            }
        }
        okm = 0;
        if (tile + 1 < ch.tile_end) fetch_tile(tile + 1);   // the registers are free: the next tile's reads travel meanwhile
        *reinterpret_cast<uint4*>(&s_wave[wave][4 * lane]) = make_uint4(0u, 0u, 0u, 0u);   // own row, read for the last time above
    }
    __syncthreads();                                         // the chunk is staged
    const uint32_t vmask = 0x00FFFFFFu;
    for (uint32_t e0 = 0; e0 < chunk_count; e0 += 8u * CHUNK_THREADS) {
        if (e0 + 8u * CHUNK_THREADS <= chunk_count) {        // all LDS reads first, then the stores, no predicate
            uint32_t w[8], g[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const uint32_t e = e0 + k * CHUNK_THREADS + tid;
                w[k] = s_stage[e];
                g[k] = PACK_OUT ? (uint32_t)s_digit[e] : (w[k] >> 24);
            }
#pragma unroll
            for (int k = 0; k < 8; k++) g[k] = s_gbase[g[k]] + (e0 + k * CHUNK_THREADS + tid);
#pragma unroll
            for (int k = 0; k < 8; k++) st32(out, g[k], PACK_OUT ? w[k] : (w[k] & vmask));
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const uint32_t e = e0 + k * CHUNK_THREADS + tid;
                if (e < chunk_count) {
                    const uint32_t w = s_stage[e];
                    const uint32_t d = PACK_OUT ? (uint32_t)s_digit[e] : (w >> 24);
                    st32(out, s_gbase[d] + e, PACK_OUT ? w : (w & vmask));
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// host launcher for one pass
// ---------------------------------------------------------------------------------------------------
inline uint32_t radix_grid_for(uint32_t n_upper) {
    uint32_t tiles = (n_upper + RADIX_TILE - 1) / RADIX_TILE;
    if (tiles < 1) tiles = 1;
    if (tiles <= (uint32_t)RADIX_TILE_GRID) return tiles;
    const uint32_t per = (tiles + RADIX_TILE_GRID - 1) / RADIX_TILE_GRID;
    return (tiles + per - 1) / per;
}

// n_upper: host-side upper bound of the element count (sizes the grid); pass_slot picks the zeroed
// digit_total row (the caller zeroes RadixScratch::digit_total once per frame).
template <class HistLoader, class Loader, class KeyOutT, bool WRITE_KEYS, bool RANGES, bool PACK_OUT>
int radix_pass_ex(const RadixExec& ex, const HistLoader& ld_hist, int hist_shift, const Loader& ld, uint32_t n_upper, int shift,
                  int pass_slot, KeyOutT* keys_out, uint32_t* vals_out, uint2* ranges, uint32_t direct_ranges, uint32_t val_bits,
                  uint32_t* count_out = nullptr) {
    const uint32_t grid = radix_grid_for(n_upper);
    uint32_t* bh = ex.scratch->block_hist.as<uint32_t>();
    uint32_t* dt = ex.scratch->digit_total.as<uint32_t>() + pass_slot * RADIX_MAX_GROUPS * RADIX_BINS;
    hipLaunchKernelGGL((k_radix_hist<HistLoader>), dim3(grid), dim3(HIST_THREADS), 0, ex.stream, ld_hist, hist_shift, bh, dt);
    if (ex.atomic_rank)
        hipLaunchKernelGGL((k_radix_scatter<Loader, KeyOutT, WRITE_KEYS, RANGES, true, PACK_OUT>), dim3(grid), dim3(SCATTER_THREADS), 0,
                           ex.stream, ld, shift, bh, dt, keys_out, vals_out, ranges, direct_ranges, val_bits, count_out);
    else
        hipLaunchKernelGGL((k_radix_scatter<Loader, KeyOutT, WRITE_KEYS, RANGES, false, PACK_OUT>), dim3(grid), dim3(SCATTER_THREADS), 0,
                           ex.stream, ld, shift, bh, dt, keys_out, vals_out, ranges, direct_ranges, val_bits, count_out);
    GS_HIP(hipGetLastError());
    return GS_OK;
}

// A pass of the depth sort between / after packing passes (see k_radix_scatter_chunk); the caller checked radix_chunk_grid_for.
template <class HistLoader, class Loader, bool PACK_OUT>
int radix_pass_chunk(const RadixExec& ex, const HistLoader& ld_hist, int hist_shift, const Loader& ld, uint32_t n_upper, int shift,
                     int pass_slot, uint32_t* out, uint32_t val_bits, uint32_t* count_out = nullptr, bool skip_hist = false) {
    const uint32_t grid = radix_chunk_grid_for(n_upper);
    uint32_t* bh = ex.scratch->block_hist.as<uint32_t>();
    uint32_t* dt = ex.scratch->digit_total.as<uint32_t>() + pass_slot * RADIX_MAX_GROUPS * RADIX_BINS;
    // (skip_hist: the rows were left by the kernel in front of this pass - sorter.hip, k_depth_key_hist)
    if (!skip_hist) hipLaunchKernelGGL((k_radix_hist<HistLoader>), dim3(grid), dim3(HIST_THREADS), 0, ex.stream, ld_hist, hist_shift, bh, dt);
    if (ex.atomic_rank)
        hipLaunchKernelGGL((k_radix_scatter_chunk<Loader, PACK_OUT, true>), dim3(grid), dim3(CHUNK_THREADS), 0, ex.stream, ld, shift, bh, dt, out, val_bits, count_out);
    else
        hipLaunchKernelGGL((k_radix_scatter_chunk<Loader, PACK_OUT, false>), dim3(grid), dim3(CHUNK_THREADS), 0, ex.stream, ld, shift, bh, dt, out, val_bits, count_out);
    GS_HIP(hipGetLastError());
    return GS_OK;
}

template <class Loader, class KeyOutT, bool WRITE_KEYS, bool RANGES = false>
int radix_pass(const RadixExec& ex, const Loader& ld_hist, const Loader& ld, uint32_t n_upper, int shift, int pass_slot,
               KeyOutT* keys_out, uint32_t* vals_out, uint2* ranges = nullptr, uint32_t direct_ranges = 0) {
    return radix_pass_ex<Loader, Loader, KeyOutT, WRITE_KEYS, RANGES, false>(ex, ld_hist, shift, ld, n_upper, shift, pass_slot, keys_out,
                                                                             vals_out, ranges, direct_ranges, 0u);
}
