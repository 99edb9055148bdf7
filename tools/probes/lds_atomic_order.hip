// Probe: does ds_add_rtn_u32 resolve same-address lanes of ONE wave instruction in ascending lane order?
// (If yes, the returned value is the stable rank of the lane among equal digits: one LDS op replaces 8 ballots.)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ void probe(uint32_t trials, uint32_t* mismatches, uint32_t* checked) {
    __shared__ uint32_t s_hist[4][256];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t lt = (1ull << lane) - 1ull;
    uint32_t bad = 0, n = 0;
    for (uint32_t t = 0; t < trials; t++) {
        for (int nb = 1; nb <= 256; nb <<= 1) {
            s_hist[wave][threadIdx.x & 63] = 0; s_hist[wave][64 + lane] = 0; s_hist[wave][128 + lane] = 0; s_hist[wave][192 + lane] = 0;
            __syncthreads();
            uint32_t h = (lane * 2654435761u) ^ (t * 40503u + blockIdx.x * 9176u + nb * 77u + wave * 13u);
            h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
            const uint32_t digit = h % (uint32_t)nb;
            // some lanes inactive to exercise exec masks
            const bool active = ((h >> 20) & 7u) != 0u;
            uint64_t same = __ballot(active);
            for (int b = 0; b < 8; b++) {
                const uint64_t vote = __ballot(active && ((digit >> b) & 1u));
                same &= ((digit >> b) & 1u) ? vote : ~vote;
            }
            if (active) {
                const uint32_t expect = __popcll(same & lt);
                const uint32_t got = atomicAdd(&s_hist[wave][digit], 1u);
                // second round on the same counters: must continue from the group size
                const uint32_t got2 = atomicAdd(&s_hist[wave][digit], 1u);
                bad += (got != expect) + (got2 != expect + (uint32_t)__popcll(same));
                n += 2;
            }
            __syncthreads();
        }
    }
    atomicAdd(mismatches, bad);
    atomicAdd(checked, n);
}

int main() {
    uint32_t *d, h[2] = {0, 0};
    hipMalloc(&d, 8); hipMemset(d, 0, 8);
    hipLaunchKernelGGL(probe, dim3(2048), dim3(256), 0, 0, 200u, d, d + 1);
    hipDeviceSynchronize();
    hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("lds atomic lane order: mismatches=%u of %u checks\n", h[0], h[1]);
    return h[0] != 0;
}
