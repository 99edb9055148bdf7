// selftest.hip — hardware property the fast radix ranking relies on, verified on the actual device at context creation.
//
// Property: when several lanes of ONE wave instruction `ds_add_rtn_u32` hit the same LDS address, the returned values
// follow ascending lane order (lane i gets base + number of lower active lanes with the same address).  It is not an
// architectural promise, so it is checked here (~50 us) and the radix scatter falls back to ballot ranking otherwise.
#include "gs_internal.hpp"

__global__ __launch_bounds__(256) void k_selftest_lds_order(uint32_t trials, uint32_t* mismatches) {
    __shared__ uint32_t s_hist[4][256];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t lt = (1ull << lane) - 1ull;
    uint32_t bad = 0;
    for (uint32_t t = 0; t < trials; t++) {
        for (int nb = 1; nb <= 256; nb <<= 1) {
#pragma unroll
            for (int k = 0; k < 4; k++) s_hist[wave][64 * k + lane] = 0;
            __syncthreads();
            uint32_t h = (lane * 2654435761u) ^ (t * 40503u + blockIdx.x * 9176u + (uint32_t)nb * 77u + wave * 13u);
            h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
            const uint32_t digit = h % (uint32_t)nb;
            const bool active = ((h >> 20) & 7u) != 0u;               // partial exec masks too
            uint64_t same = __ballot(active);
#pragma unroll
            for (int b = 0; b < 8; b++) {
                const uint64_t vote = __ballot(active && ((digit >> b) & 1u));
                same &= ((digit >> b) & 1u) ? vote : ~vote;
            }
            if (active) {
                const uint32_t expect = (uint32_t)__popcll(same & lt);
                const uint32_t got = atomicAdd(&s_hist[wave][digit], 1u);
                const uint32_t got2 = atomicAdd(&s_hist[wave][digit], 1u);
                bad += (got != expect) + (got2 != expect + (uint32_t)__popcll(same));
            }
            __syncthreads();
        }
    }
    if (bad) atomicAdd(mismatches, bad);
}

int gs_selftest_lds_atomic_order(gs_context* ctx, bool* ok) {
    *ok = false;
    uint32_t* d = ctx->radix.digit_total.as<uint32_t>();      // any zeroable device word
    GS_HIP(hipMemsetAsync(d, 0, 4, ctx->stream));
    hipLaunchKernelGGL(k_selftest_lds_order, dim3(512), dim3(256), 0, ctx->stream, 8u, d);
    GS_HIP(hipGetLastError());
    uint32_t bad = 1;
    GS_HIP(hipMemcpyAsync(&bad, d, 4, hipMemcpyDeviceToHost, ctx->stream));
    GS_HIP(hipStreamSynchronize(ctx->stream));
    *ok = (bad == 0);
    return GS_OK;
}
