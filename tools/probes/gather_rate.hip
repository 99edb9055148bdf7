// Random-gather rate probe (MI355X): N gathers of ELEM bytes from a table of T entries, indexes = a random permutation
// (every entry exactly once, like the binner's per-sorted-splat look-up of a table in storage order).
// build: hipcc -O3 --offload-arch=gfx950 tools/probes/gather_rate.hip -o /tmp/gather_rate ; run: /tmp/gather_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include <random>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <class T, int PER>
__global__ __launch_bounds__(256) void k_gather(const uint32_t* __restrict__ idx, const T* __restrict__ table, uint32_t n, uint32_t* __restrict__ out) {
    uint32_t acc = 0;
    const uint32_t stride = gridDim.x * 256u * PER;
    for (uint32_t base = blockIdx.x * 256u * PER + threadIdx.x; base < n; base += stride) {
        uint32_t ix[PER];
        T v[PER];
#pragma unroll
        for (int k = 0; k < PER; k++) ix[k] = idx[min(base + k * 256u, n - 1u)];
#pragma unroll
        for (int k = 0; k < PER; k++) v[k] = table[ix[k]];
#pragma unroll
        for (int k = 0; k < PER; k++) acc += (uint32_t)(sizeof(T) == 8 ? ((const uint32_t*)&v[k])[0] + ((const uint32_t*)&v[k])[1] : ((const uint32_t*)&v[k])[0]);
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void k_stream(const uint4* __restrict__ p, size_t n16, uint32_t* out) {
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { uint4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}

template <class T, int PER>
static void run(const char* what, const uint32_t* d_idx, const void* d_table, uint32_t n, size_t table_bytes, bool warm, void* d_trash, size_t trash_bytes, uint32_t* d_out, int grid) {
    printf("start %s\n", what);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9f, sum = 0;
    for (int it = 0; it < 6; it++) {
        // evict: stream 1 GB of other data through the caches
        hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, 0, (const uint4*)d_trash, trash_bytes / 16, d_out);
        if (warm) hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, 0, (const uint4*)d_table, table_bytes / 16, d_out);
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k_gather<T, PER>), dim3(grid), dim3(256), 0, 0, d_idx, (const T*)d_table, n, d_out);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (it > 0) { best = ms < best ? ms : best; sum += ms; }
    }
    fflush(stdout); printf("%-44s n=%u table=%4zu MB %s grid=%4d per=%d: best %.3f ms  mean %.3f ms  = %.1f G gathers/s\n", what, n, table_bytes >> 20, warm ? "warm" : "cold", grid, PER, best, sum / 5, n / best / 1e6);
}

int main() {
    setvbuf(stdout, NULL, _IONBF, 0);
    const uint32_t n = 16000000;
    std::vector<uint32_t> idx(n);
    for (uint32_t i = 0; i < n; i++) idx[i] = i;
    std::mt19937 rng(1234);
    std::shuffle(idx.begin(), idx.end(), rng);
    uint32_t *d_idx, *d_out; void *d_table, *d_trash;
    const size_t trash = 1ull << 30;
    CK(hipMalloc(&d_idx, (size_t)n * 4)); CK(hipMalloc(&d_table, (size_t)n * 8)); CK(hipMalloc(&d_trash, trash)); CK(hipMalloc(&d_out, 64));
    CK(hipMemcpy(d_idx, idx.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    CK(hipMemset(d_table, 1, (size_t)n * 8)); CK(hipMemset(d_trash, 2, trash));
    for (int warm = 0; warm < 2; warm++) {
        run<uint2, 4>("8-byte gathers", d_idx, d_table, n, (size_t)n * 8, warm, d_trash, trash, d_out, 2048);
        run<uint32_t, 4>("4-byte gathers", d_idx, d_table, n, (size_t)n * 4, warm, d_trash, trash, d_out, 2048);
        run<uint2, 8>("8-byte gathers", d_idx, d_table, n, (size_t)n * 8, warm, d_trash, trash, d_out, 2048);
        run<uint32_t, 8>("4-byte gathers", d_idx, d_table, n, (size_t)n * 4, warm, d_trash, trash, d_out, 2048);
        run<uint16_t, 8>("2-byte gathers", d_idx, d_table, n, (size_t)n * 2, warm, d_trash, trash, d_out, 2048);
    }
    // a quarter of the gathers (C3: 25 % visible) from the same tables
    run<uint2, 4>("8-byte gathers, first 4 M indexes", d_idx, d_table, n / 4, (size_t)n * 8, false, d_trash, trash, d_out, 2048);
    // sequential indexes for reference
    for (uint32_t i = 0; i < n; i++) idx[i] = i;
    CK(hipMemcpy(d_idx, idx.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    run<uint2, 4>("8-byte loads, identity indexes", d_idx, d_table, n, (size_t)n * 8, false, d_trash, trash, d_out, 2048);
    return 0;
}
