// Look-back radix pass probe (MI355X, round 5; VERDICT r04 item 2 / DESIGN 12.9 item 1): a stable 8-bit LSD pass WITHOUT a histogram
// kernel in front of it.  A workgroup takes a TICKET (so that it only ever waits for workgroups that are already running - no
// assumption about dispatch order or residency, the GPU may be shared), keeps its whole chunk of 12288 keys in registers, ranks it
// on per-wave LDS histograms (the product's ds_add_rtn ranking), publishes its 256 digit counts as 8-byte {value, epoch} granules
// (one sc1 store each, no fence: data and flag travel together) and gets its offsets from PREDECESSORS ONLY, two levels deep so
// that 500 chunks that are all resident at once do not chain: the last chunk of every group of 32 sums its group and publishes
// the group's row; a chunk adds the rows of the groups before its own and the rows of the chunks before it inside its group.
// The global digit totals (keys with a smaller digit, in ALL chunks) cannot come from predecessors: pass 0 takes them from a
// totals-only kernel (what is left of k_radix_hist), pass 1 from pass 0, which counts the next digit on the way (256 atomics per
// chunk into 16 shards).  Every spin is bounded: a protocol bug becomes an error flag, not a hung box.
//
// The probe sorts (16-bit key, 23-bit payload) pairs the way the depth sort does (pass 0: key + payload -> packed word, pass 1:
// packed word -> payload), checks the result against a stable CPU sort, and times   totals + pass 0 + pass 1   with HIP events.
// build: hipcc -O3 --offload-arch=gfx950 tools/probes/lookback_probe.hip -o tools/probes/lookback_probe.bin
// run:   timeout 60 tools/probes/lookback_probe.bin [n = 5800000] [reps = 20]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <numeric>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int THREADS = 512, WAVES = 8, ITEMS = 24, CHUNK = THREADS * ITEMS, BINS = 256, GROUP = 32, SHARDS = 16;
constexpr uint32_t SPIN_LIMIT = 1u << 20, VAL_BITS = 23;

__device__ __forceinline__ void st_granule(unsigned long long* p, uint32_t value, uint32_t tag) {
    __hip_atomic_store(p, ((unsigned long long)tag << 32) | value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long ld_granule(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// sum over rows [first, first + count) of column d; every row is waited for (tag == epoch); eight polls in flight
__device__ __forceinline__ uint32_t sum_rows(const unsigned long long* rows, uint32_t first, uint32_t count, uint32_t d, uint32_t epoch, uint32_t* fail) {
    uint32_t sum = 0;
    for (uint32_t r0 = first; r0 < first + count; r0 += 8u) {
        const uint32_t nb = min(8u, first + count - r0);
        uint32_t pending = (1u << nb) - 1u, spins = 0;
        while (pending) {
            unsigned long long g[8];
#pragma unroll
            for (uint32_t k = 0; k < 8u; k++) g[k] = ld_granule(rows + (size_t)(r0 + min(k, nb - 1u)) * BINS + d);
#pragma unroll
            for (uint32_t k = 0; k < 8u; k++)
                if (((pending >> k) & 1u) && (uint32_t)(g[k] >> 32) == epoch) { sum += (uint32_t)g[k]; pending &= ~(1u << k); }
            if (pending) {
                if (++spins > SPIN_LIMIT) { *fail = 1u; return sum; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
    }
    return sum;
}
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, uint32_t lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(v, o, 64); if ((int)lane >= o) v += t; }
    return v;
}

// digit totals of pass 0, sharded (what a totals-only k_radix_hist leaves): totals[shard][256]; also resets the tickets
__global__ __launch_bounds__(1024) void k_totals(const uint32_t* __restrict__ in, uint32_t n, uint32_t* __restrict__ totals, uint32_t* __restrict__ tickets) {
    __shared__ uint32_t s_h[4][BINS];
    if (blockIdx.x == 0 && threadIdx.x < 2) tickets[32 * threadIdx.x] = 0u;
    for (uint32_t k = threadIdx.x; k < 4u * BINS; k += 1024u) (&s_h[0][0])[k] = 0u;
    __syncthreads();
    const uint32_t w = (threadIdx.x >> 6) & 3u;
    const uint4* in4 = reinterpret_cast<const uint4*>(in);
    for (uint32_t v = blockIdx.x * 1024u + threadIdx.x; v < n / 4u; v += gridDim.x * 1024u) {
        const uint4 x = in4[v];
        atomicAdd(&s_h[w][x.x & 255u], 1u); atomicAdd(&s_h[w][x.y & 255u], 1u);
        atomicAdd(&s_h[w][x.z & 255u], 1u); atomicAdd(&s_h[w][x.w & 255u], 1u);
    }
    if (blockIdx.x == 0) for (uint32_t i = (n / 4u) * 4u + threadIdx.x; i < n; i += 1024u) atomicAdd(&s_h[w][in[i] & 255u], 1u);
    __syncthreads();
    if (threadIdx.x < BINS) {
        const uint32_t t = s_h[0][threadIdx.x] + s_h[1][threadIdx.x] + s_h[2][threadIdx.x] + s_h[3][threadIdx.x];
        if (t) atomicAdd(&totals[(blockIdx.x % SHARDS) * BINS + threadIdx.x], t);
    }
}

// PASS0: in = 16-bit keys (as u32), payload = perm[j]; digit = key & 255; out word = (key >> 8) << 23 | payload; also counts the
//        NEXT digit (key >> 8) into next_totals.   !PASS0: in = packed words; digit = word >> 23; out = word & (2^23 - 1).
template <bool PASS0>
__global__ __launch_bounds__(THREADS, 4) void k_pass_lb(const uint32_t* __restrict__ in, const uint32_t* __restrict__ perm, uint32_t* __restrict__ out,
                                                        uint32_t n, const uint32_t* __restrict__ totals, uint32_t* __restrict__ next_totals,
                                                        unsigned long long* __restrict__ chunk_rows, unsigned long long* __restrict__ group_rows,
                                                        uint32_t* __restrict__ ticket, uint32_t epoch, uint32_t* __restrict__ fail) {
    __shared__ uint32_t s_stage[CHUNK];                                    // PASS0: the packed word; else digit << 24 | payload
    __shared__ uint8_t s_digit[PASS0 ? CHUNK : 4];
    __shared__ __attribute__((aligned(16))) uint32_t s_wave[WAVES][BINS];
    __shared__ uint32_t s_next[PASS0 ? BINS : 4];
    __shared__ uint32_t s_first[BINS], s_gbase[BINS], s_bg[BINS], s_tmp[WAVES], s_ticket;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tid == 0u) s_ticket = atomicAdd(ticket, 1u);
    *reinterpret_cast<uint4*>(&s_wave[wave][4 * lane]) = make_uint4(0u, 0u, 0u, 0u);
    if (PASS0 && tid < BINS) s_next[tid] = 0u;
    // keys with a smaller digit, anywhere in the list: exclusive scan of the global totals (thread d < 256 keeps its value)
    uint32_t smaller = 0;
    {
        uint32_t tot = 0;
        if (tid < BINS)
#pragma unroll
            for (int s = 0; s < SHARDS; s++) tot += totals[s * BINS + tid];
        const uint32_t incl = wave_incl_scan(tot, lane);
        if (lane == 63u) s_tmp[wave] = incl;
        __syncthreads();                                                   // (also: s_ticket, the zeroed rows)
        uint32_t wb = 0;
        for (uint32_t w = 0; w < wave; w++) wb += s_tmp[w];
        smaller = wb + incl - tot;
    }
    const uint32_t chunk = s_ticket, chunks = (n + CHUNK - 1) / CHUNK;
    if (chunk >= chunks) return;                                           // (uniform)
    const uint32_t base = chunk * CHUNK + wave * (64 * ITEMS) + lane;      // stable order inside the chunk = (wave, r, lane) = ascending j
    uint32_t word[ITEMS], dr[ITEMS];                                       // the staged word; digit | rank << 8 (all ones: beyond the list)
    {
        uint32_t key[ITEMS], pay[ITEMS];
#pragma unroll
        for (int r = 0; r < ITEMS; r++) {
            const uint32_t j = base + r * 64u, jc = j < n ? j : n - 1u;
            key[r] = in[jc];
            pay[r] = PASS0 ? perm[jc] : 0u;
        }
#pragma unroll
        for (int r = 0; r < ITEMS; r++) {
            const uint32_t j = base + r * 64u;
            uint32_t d;
            if (PASS0) { d = key[r] & 255u; word[r] = ((key[r] >> 8) << VAL_BITS) | pay[r]; }
            else { d = key[r] >> VAL_BITS; word[r] = (d << 24) | (key[r] & ((1u << VAL_BITS) - 1u)); }
            uint32_t rk = 0;
            if (j < n) {
                rk = atomicAdd(&s_wave[wave][d], 1u);                      // same-digit lanes are served in lane order (product self-test)
                if (PASS0) atomicAdd(&s_next[(key[r] >> 8) & 255u], 1u);
            }
            dr[r] = j < n ? (d | (rk << 8)) : 0xFFFFFFFFu;
        }
    }
    __syncthreads();
    // thread d: this chunk's count of digit d; s_wave[w][d] becomes the first slot of (wave w, digit d) inside the digit's run
    uint32_t cnt = 0;
    if (tid < BINS) {
#pragma unroll
        for (int w = 0; w < WAVES; w++) { const uint32_t c = s_wave[w][tid]; s_wave[w][tid] = cnt; cnt += c; }
        st_granule(chunk_rows + (size_t)chunk * BINS + tid, cnt, epoch);   // PUBLISH: the chunk's aggregate
        if (PASS0 && s_next[tid]) atomicAdd(&next_totals[(chunk % SHARDS) * BINS + tid], s_next[tid]);
    }
    // first staging slot of every digit: exclusive scan of cnt over the 256 digits
    uint32_t first = 0;
    {
        const uint32_t incl = wave_incl_scan(cnt, lane);
        if (lane == 63u) s_tmp[wave] = incl;                               // (waves 4..7 hold zeros)
        __syncthreads();
        uint32_t wb = 0;
        for (uint32_t w = 0; w < wave; w++) wb += s_tmp[w];
        first = wb + incl - cnt;
        if (tid < BINS) s_first[tid] = first;
    }
    __syncthreads();
    const uint32_t chunk_count = s_tmp[0] + s_tmp[1] + s_tmp[2] + s_tmp[3];
    // stage the chunk in its final (digit-major) order while the predecessors' rows travel
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        if (dr[r] != 0xFFFFFFFFu) {
            const uint32_t d = dr[r] & 255u, pos = s_first[d] + s_wave[wave][d] + (dr[r] >> 8);
            s_stage[pos] = word[r];
            if (PASS0) s_digit[pos] = (uint8_t)d;
        }
    }
    // EXCHANGE, predecessors only.  threads 0..255: the chunks before this one inside its group (and, for the group's last chunk,
    // the group's row);  threads 256..511: the rows of the groups before this one
    const uint32_t g = chunk / GROUP, jg = chunk % GROUP;
    uint32_t before = 0;
    if (tid < BINS) {
        before = sum_rows(chunk_rows, g * GROUP, jg, tid, epoch, fail);
        if (jg == GROUP - 1u) st_granule(group_rows + (size_t)g * BINS + tid, before + cnt, epoch);
    } else {
        s_bg[tid - BINS] = sum_rows(group_rows, 0u, g, tid - BINS, epoch, fail);
    }
    __syncthreads();                                                       // the chunk is staged, s_bg is written
    if (tid < BINS) s_gbase[tid] = smaller + s_bg[tid] + before - first;   // global slot of the digit's run minus its staging slot
    __syncthreads();
    for (uint32_t e = tid; e < chunk_count; e += THREADS) {
        const uint32_t w = s_stage[e];
        if (PASS0) out[s_gbase[s_digit[e]] + e] = w;
        else out[s_gbase[w >> 24] + e] = w & 0x00FFFFFFu;
    }
}

int main(int argc, char** argv) {
    const uint32_t n = argc > 1 ? (uint32_t)atoi(argv[1]) : 5800000u;
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    if (n < 1 || n > (1u << VAL_BITS)) { printf("n outside 1 .. 2^23\n"); return 1; }
    std::mt19937 rng(12345);
    std::normal_distribution<float> nd(32768.0f, 9000.0f);
    std::vector<uint32_t> keys(n), perm(n);
    for (uint32_t i = 0; i < n; i++) { float v = nd(rng); keys[i] = (uint32_t)std::min(65535.0f, std::max(0.0f, v)); }
    std::iota(perm.begin(), perm.end(), 0u);
    std::shuffle(perm.begin(), perm.end(), rng);
    std::vector<uint32_t> idx(n), expect(n);
    std::iota(idx.begin(), idx.end(), 0u);
    std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return keys[a] < keys[b]; });
    for (uint32_t i = 0; i < n; i++) expect[i] = perm[idx[i]];

    const uint32_t chunks = (n + CHUNK - 1) / CHUNK, groups = (chunks + GROUP - 1) / GROUP;
    uint32_t *d_keys, *d_perm, *d_mid, *d_out, *d_tot, *d_tickets, *d_fail;
    unsigned long long *d_crows, *d_grows;
    CK(hipMalloc(&d_keys, (size_t)n * 4)); CK(hipMalloc(&d_perm, (size_t)n * 4)); CK(hipMalloc(&d_mid, (size_t)n * 4)); CK(hipMalloc(&d_out, (size_t)n * 4));
    CK(hipMalloc(&d_tot, 2 * SHARDS * BINS * 4)); CK(hipMalloc(&d_tickets, 256)); CK(hipMalloc(&d_fail, 4));
    CK(hipMalloc(&d_crows, (size_t)chunks * BINS * 8)); CK(hipMalloc(&d_grows, (size_t)(groups + 1) * BINS * 8));
    CK(hipMemcpy(d_keys, keys.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_perm, perm.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    CK(hipMemset(d_crows, 0, (size_t)chunks * BINS * 8)); CK(hipMemset(d_grows, 0, (size_t)(groups + 1) * BINS * 8));
    CK(hipMemset(d_fail, 0, 4)); CK(hipMemset(d_tickets, 0, 256));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("n %u: %u chunks of %d keys in %u groups; %d CUs (2 workgroups of %d threads per CU resident = %d)\n", n, chunks, CHUNK, groups,
           prop.multiProcessorCount, THREADS, 2 * prop.multiProcessorCount);
    hipEvent_t e0, e1, e2, e3; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2)); CK(hipEventCreate(&e3));
    uint32_t epoch = 0;
    std::vector<float> whole, t_tot, t_p0, t_p1;
    const uint32_t tgrid = std::min<uint32_t>(512u, (n / 4u + 1023u) / 1024u + 1u);
    for (int it = 0; it < reps + 3; it++) {
        const bool split = it >= 3 && (it & 1);                            // odd reps: an event between the kernels (per-kernel times)
        CK(hipMemsetAsync(d_tot, 0, 2 * SHARDS * BINS * 4, 0));
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_totals, dim3(tgrid), dim3(1024), 0, 0, d_keys, n, d_tot, d_tickets);
        if (split) CK(hipEventRecord(e1, 0));
        hipLaunchKernelGGL((k_pass_lb<true>), dim3(chunks), dim3(THREADS), 0, 0, d_keys, d_perm, d_mid, n, d_tot, d_tot + SHARDS * BINS, d_crows, d_grows,
                           d_tickets, ++epoch, d_fail);
        if (split) CK(hipEventRecord(e2, 0));
        hipLaunchKernelGGL((k_pass_lb<false>), dim3(chunks), dim3(THREADS), 0, 0, d_mid, (const uint32_t*)nullptr, d_out, n, d_tot + SHARDS * BINS, d_tot,
                           d_crows, d_grows, d_tickets + 32, ++epoch, d_fail);
        CK(hipEventRecord(e3, 0));
        CK(hipEventSynchronize(e3));
        CK(hipGetLastError());
        float ms; CK(hipEventElapsedTime(&ms, e0, e3));
        if (it < 3) continue;
        if (split) {
            float a, b, c; CK(hipEventElapsedTime(&a, e0, e1)); CK(hipEventElapsedTime(&b, e1, e2)); CK(hipEventElapsedTime(&c, e2, e3));
            t_tot.push_back(a); t_p0.push_back(b); t_p1.push_back(c);
        } else whole.push_back(ms);
    }
    uint32_t fail = 0; CK(hipMemcpy(&fail, d_fail, 4, hipMemcpyDeviceToHost));
    std::vector<uint32_t> got(n); CK(hipMemcpy(got.data(), d_out, (size_t)n * 4, hipMemcpyDeviceToHost));
    size_t bad = 0; for (uint32_t i = 0; i < n; i++) bad += got[i] != expect[i];
    auto med = [](std::vector<float> v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.f : v[v.size() / 2]; };
    printf("spin timeouts: %u   mismatches vs the stable CPU sort: %zu of %u\n", fail, bad, n);
    printf("totals + pass 0 + pass 1 (three launches, no events between): median %.1f us  min %.1f us\n", med(whole) * 1e3, *std::min_element(whole.begin(), whole.end()) * 1e3);
    printf("with an event between the kernels: totals %.1f us, pass 0 %.1f us, pass 1 %.1f us\n", med(t_tot) * 1e3, med(t_p0) * 1e3, med(t_p1) * 1e3);
    return (fail || bad) ? 2 : 0;
}
