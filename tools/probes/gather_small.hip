// Random look-ups into SMALL tables (MI355X): does a table that fits an XCD's 4 MB L2, or the 256 MB MALL, answer faster than the
// ~55 G/s of gather_rate.hip's 30-122 MB tables?  n random positions in [0, M) (with repeats, like a sorted list walking a table in
// storage order), element 1 / 2 / 8 bytes, and a nibble table (two splats per byte).  The binner's question: would a per-splat
// 4-bit bin count (2.9 MB at C3) or a 16-bit coarse rect (11.6 MB) serve k_bin_count / k_bin_emit faster than the 8-byte rect (46 MB)?
// build: hipcc -O3 --offload-arch=gfx950 tools/probes/gather_small.hip -o tools/probes/gather_small.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <class T, int PER, bool NIBBLE>
__global__ __launch_bounds__(256) void k_gather(const uint32_t* __restrict__ idx, const T* __restrict__ table, uint32_t n, uint32_t* __restrict__ out) {
    uint32_t acc = 0;
    const uint32_t stride = gridDim.x * 256u * PER;
    for (uint32_t base = blockIdx.x * 256u * PER + threadIdx.x; base < n; base += stride) {
        uint32_t ix[PER];
        T v[PER];
#pragma unroll
        for (int k = 0; k < PER; k++) ix[k] = idx[min(base + k * 256u, n - 1u)];
#pragma unroll
        for (int k = 0; k < PER; k++) v[k] = table[NIBBLE ? ix[k] >> 1 : ix[k]];
#pragma unroll
        for (int k = 0; k < PER; k++) {
            uint32_t w = sizeof(T) == 8 ? ((const uint32_t*)&v[k])[0] + ((const uint32_t*)&v[k])[1] : (uint32_t)*(const uint8_t*)&v[k];
            if (sizeof(T) == 2) w = (uint32_t)*(const uint16_t*)&v[k];
            if (NIBBLE) w = (w >> ((ix[k] & 1u) * 4u)) & 15u;
            acc += w;
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void k_stream(const uint4* __restrict__ p, size_t n16, uint32_t* out) {
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { uint4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}

template <class T, int PER, bool NIBBLE>
static void run(const char* what, const uint32_t* d_idx, const void* d_table, uint32_t n, size_t table_bytes, void* d_trash, size_t trash_bytes, uint32_t* d_out) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9f, sum = 0;
    for (int it = 0; it < 6; it++) {
        hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, 0, (const uint4*)d_trash, trash_bytes / 16, d_out);   // evict
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k_gather<T, PER, NIBBLE>), dim3(2048), dim3(256), 0, 0, d_idx, (const T*)d_table, n, d_out);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (it > 0) { best = ms < best ? ms : best; sum += ms; }
    }
    printf("%-34s n=%8u table=%7.1f MB: best %6.1f us  mean %6.1f us  = %6.1f G look-ups/s\n", what, n, table_bytes / 1048576.0, best * 1e3, sum / 5 * 1e3, n / best / 1e6);
}

int main(int argc, char** argv) {
    setvbuf(stdout, NULL, _IONBF, 0);
    const uint32_t M = argc > 1 ? (uint32_t)atol(argv[1]) : 5800000u;      // table entries (splats)
    uint32_t *d_idx, *d_out; void *d_table, *d_trash;
    const size_t trash = 1ull << 30;
    CK(hipMalloc(&d_idx, (size_t)M * 4)); CK(hipMalloc(&d_table, (size_t)M * 8 + 64)); CK(hipMalloc(&d_trash, trash)); CK(hipMalloc(&d_out, 64));
    CK(hipMemset(d_table, 1, (size_t)M * 8)); CK(hipMemset(d_trash, 2, trash));
    std::vector<uint32_t> idx(M);
    std::mt19937 rng(99);
    for (uint32_t i = 0; i < M; i++) idx[i] = (uint32_t)(rng() % M);
    CK(hipMemcpy(d_idx, idx.data(), (size_t)M * 4, hipMemcpyHostToDevice));
    printf("M = %u table entries; list of M random positions (a whole sorted list) and of M / 4 (the visible quarter)\n", M);
    for (uint32_t n : {M, M / 4u}) {
        run<uint2, 4, false>("8-byte rect", d_idx, d_table, n, (size_t)M * 8, d_trash, trash, d_out);
        run<uint16_t, 4, false>("2-byte coarse rect", d_idx, d_table, n, (size_t)M * 2, d_trash, trash, d_out);
        run<uint8_t, 4, false>("1-byte count", d_idx, d_table, n, (size_t)M, d_trash, trash, d_out);
        run<uint8_t, 4, true>("4-bit count (nibble table)", d_idx, d_table, n, (size_t)M / 2, d_trash, trash, d_out);
        run<uint8_t, 8, true>("4-bit count, 8 in flight", d_idx, d_table, n, (size_t)M / 2, d_trash, trash, d_out);
    }
    // the list alone (identity positions would stream): how much of the above is the 4-byte list itself
    for (uint32_t i = 0; i < M; i++) idx[i] &= 65535u;
    CK(hipMemcpy(d_idx, idx.data(), (size_t)M * 4, hipMemcpyHostToDevice));
    run<uint8_t, 4, false>("1-byte, entries 0..65535 only", d_idx, d_table, M, 65536, d_trash, trash, d_out);
    return 0;
}
