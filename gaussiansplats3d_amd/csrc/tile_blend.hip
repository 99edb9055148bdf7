// tile_blend.hip — fragment stage + blend of the RENDER SEAM: one wave64 per 16x16-pixel tile.
//
// Restates the reference's fragment shader and blend state
//   /root/reference/src/splatmesh/SplatMaterial3D.js:235-251   A = dot(vPosition,vPosition); A > 8 -> discard;
//                                                              alpha = exp(-0.5*A) * vColor.a
//   /root/reference/src/splatmesh/SplatMaterial3D.js:65-75     NormalBlending, back-to-front into RGBA8 cleared
//                                                              to (0,0,0,0) (src/Viewer.js:358-359)
// as a front-to-back composite over the tile's near->far list:
//   C += T*alpha*rgb ; T *= 1-alpha      =>  rgb_out = C , alpha_out = 1 - T      (identical in exact arithmetic)
// with early termination once T < 1e-4 (bounded error 1e-4, far below 1/255).
//
// CDNA4 mapping: a tile is ONE wave (workgroup = 64 lanes, up to 32 tiles resident per CU, no block barriers
// between waves), each lane owns 4 pixels (x = lane&15, y = (lane>>4) + 4g, g = 0..3).  The 64-entry batches
// of the tile's list are gathered with one 32-byte record load per lane, expanded to fp32 and staged in LDS;
// the inner loop then reads each splat as three wave-uniform ds_read_b128 broadcasts.  Per-entry uniform work
// (bounds tests against the four 16x4 pixel strips) happens once per wave, the per-pixel work is ~14 VALU ops.
#include "gs_internal.hpp"

constexpr float GS_POWER_CUT = 5.7707801636f;    // 4*log2(e)  <=>  A > 8
constexpr float GS_T_EPS = 1e-4f;
constexpr float GS_K_POWER_B = 2.4022448f;

struct __attribute__((aligned(16))) LdsSplat {
    float cx, cy, ax, ay;
    float bx, by, ymin, ymax;      // ymin/ymax: pixel-centre range the ellipse can touch
    float r, g, b, a;
};

__device__ __forceinline__ void stage_entry(LdsSplat* dst, const uint4 lo, const uint4 hi) {
    LdsSplat s;
    s.cx = __uint_as_float(lo.x); s.cy = __uint_as_float(lo.y);
    s.ax = __uint_as_float(lo.z); s.ay = __uint_as_float(lo.w);
    s.bx = __uint_as_float(hi.x); s.by = __uint_as_float(hi.y);
    const float na = s.ax * s.ax + s.ay * s.ay, nb = s.bx * s.bx + s.by * s.by;
    const float b1y = GS_K_POWER_B * s.ay / na, b2y = GS_K_POWER_B * s.by / nb;
    const float ext_y = sqrtf(b1y * b1y + b2y * b2y) * 1.00001f + 1e-3f;
    s.ymin = s.cy - ext_y;
    s.ymax = s.cy + ext_y;
    s.r = (float)(hi.z & 0xFFFFu) * (1.0f / 65535.0f);
    s.g = (float)(hi.z >> 16) * (1.0f / 65535.0f);
    s.b = (float)(hi.w & 0xFFFFu) * (1.0f / 65535.0f);
    s.a = (float)(hi.w >> 16) * (1.0f / 65535.0f);
    *dst = s;
}

__global__ __launch_bounds__(64) void k_tile_blend(const uint2* __restrict__ ranges, const uint32_t* __restrict__ vals,
                                                   const uint4* __restrict__ recs, uint32_t* __restrict__ out,
                                                   uint32_t width, uint32_t height, uint32_t tiles_x, uint32_t row_begin) {
    __shared__ LdsSplat s_batch[64];
    const uint32_t tile = blockIdx.x;
    const uint32_t lane = threadIdx.x;
    const uint32_t tx = tile % tiles_x, ty = tile / tiles_x + row_begin;
    const uint32_t px = tx * GS_TILE + (lane & 15u);
    const uint32_t py0 = ty * GS_TILE + (lane >> 4);
    const float fx = (float)px + 0.5f;
    const float fy0 = (float)py0 + 0.5f;
    const float strip_lo = (float)(ty * GS_TILE) + 0.5f;       // first pixel-centre row of strip 0

    const uint2 range = ranges[tile];
    const uint32_t begin = range.x, n = range.y > range.x ? range.y - range.x : 0u;   // untouched tiles keep (~0, 0)

    float T[4] = {1.0f, 1.0f, 1.0f, 1.0f};
    float Cr[4] = {0, 0, 0, 0}, Cg[4] = {0, 0, 0, 0}, Cb[4] = {0, 0, 0, 0};

    uint4 lo = make_uint4(0, 0, 0, 0), hi = lo;
    if (lane < n) {
        const uint32_t idx = vals[begin + lane];
        lo = recs[2 * (size_t)idx];
        hi = recs[2 * (size_t)idx + 1];
    }
    bool all_done = false;
    for (uint32_t base = 0; base < n && !all_done; base += 64) {
        const uint32_t cnt = min(64u, n - base);
        __syncthreads();                               // previous batch fully consumed (single wave: cheap)
        if (lane < cnt) stage_entry(&s_batch[lane], lo, hi);
        const uint32_t nxt = base + 64 + lane;         // prefetch the next batch while this one is blended
        if (nxt < n) {
            const uint32_t idx = vals[begin + nxt];
            lo = recs[2 * (size_t)idx];
            hi = recs[2 * (size_t)idx + 1];
        }
        __syncthreads();
        for (uint32_t j0 = 0; j0 < cnt; j0 += 16) {
            const uint32_t j1 = min(cnt, j0 + 16);
            for (uint32_t j = j0; j < j1; j++) {
                const float4 q0 = *reinterpret_cast<const float4*>(&s_batch[j].cx);
                const float4 q1 = *reinterpret_cast<const float4*>(&s_batch[j].bx);
                const float4 q2 = *reinterpret_cast<const float4*>(&s_batch[j].r);
                const float dx = fx - q0.x;
                const float adx = q0.z * dx, bdx = q1.x * dx;
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    // wave-uniform: does the ellipse reach this 16x4 strip at all?
                    const float s_lo = strip_lo + (float)(4 * g), s_hi = s_lo + 3.0f;
                    if (q1.w < s_lo || q1.z > s_hi) continue;
                    const float dy = (fy0 + (float)(4 * g)) - q0.y;
                    const float u = fmaf(q0.w, dy, adx);
                    const float w = fmaf(q1.y, dy, bdx);
                    const float pw = fmaf(w, w, u * u);
                    float alpha = __builtin_amdgcn_exp2f(-pw) * q2.w;
                    alpha = pw <= GS_POWER_CUT ? alpha : 0.0f;          // `if (A > 8.0) discard`
                    const float wgt = T[g] * alpha;
                    Cr[g] = fmaf(wgt, q2.x, Cr[g]);
                    Cg[g] = fmaf(wgt, q2.y, Cg[g]);
                    Cb[g] = fmaf(wgt, q2.z, Cb[g]);
                    T[g] -= wgt;
                }
            }
            // freeze saturated pixels; leave when the whole tile is saturated
            bool live = false;
#pragma unroll
            for (int g = 0; g < 4; g++) {
                if (T[g] < GS_T_EPS) T[g] = 0.0f;
                live = live || (T[g] > 0.0f);
            }
            if (__ballot(live) == 0ull) {
                all_done = true;
                break;
            }
        }
    }
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const uint32_t py = py0 + 4u * g;
        if (px < width && py < height) {
            const float a = 1.0f - T[g];
            const uint32_t r8 = (uint32_t)(fminf(fmaxf(Cr[g], 0.0f), 1.0f) * 255.0f + 0.5f);
            const uint32_t g8 = (uint32_t)(fminf(fmaxf(Cg[g], 0.0f), 1.0f) * 255.0f + 0.5f);
            const uint32_t b8 = (uint32_t)(fminf(fmaxf(Cb[g], 0.0f), 1.0f) * 255.0f + 0.5f);
            const uint32_t a8 = (uint32_t)(fminf(fmaxf(a, 0.0f), 1.0f) * 255.0f + 0.5f);
            out[(size_t)(py - row_begin * GS_TILE) * width + px] = r8 | (g8 << 8) | (b8 << 16) | (a8 << 24);
        }
    }
}

int gs_launch_blend(gs_mesh* m, const ProjectParams& pp, uint8_t* out_dev) {
    const uint32_t rows = pp.row_end - pp.row_begin;
    const uint32_t tiles = pp.tiles_x * rows;
    if (tiles == 0) return GS_OK;
    const uint32_t* vals = (m->sorted_buf ? m->evalB : m->evalA).as<uint32_t>();
    hipLaunchKernelGGL(k_tile_blend, dim3(tiles), dim3(64), 0, m->ctx->stream, m->tile_ranges.as<uint2>(), vals,
                       m->recs.as<uint4>(), reinterpret_cast<uint32_t*>(out_dev), (uint32_t)pp.width, (uint32_t)pp.height,
                       pp.tiles_x, pp.row_begin);
    GS_HIP(hipGetLastError());
    return GS_OK;
}
