// tile_blend.hip — fragment stage + blend of the RENDER SEAM: one 256-thread workgroup per 32x32-pixel bin,
// one wave64 per 16x16-pixel quadrant.
//
// Restates the reference's fragment shader and blend state
//   /root/reference/src/splatmesh/SplatMaterial3D.js:235-251   A = dot(vPosition,vPosition); A > 8 -> discard;
//                                                              alpha = exp(-0.5*A) * vColor.a
//   /root/reference/src/splatmesh/SplatMaterial3D.js:65-75     NormalBlending, back-to-front into RGBA8 cleared
//                                                              to (0,0,0,0) (src/Viewer.js:358-359)
// as a front-to-back composite over the bin's near->far list:
//   C += T*alpha*rgb ; T *= 1-alpha      =>  rgb_out = C , alpha_out = 1 - T      (identical in exact arithmetic)
// with early termination once T < 1e-4 (bounded error 1e-4, far below 1/255).
//
// CDNA4 mapping.  The entry lists are per list bin (128 px, or 32 px for scenes of tiny splats), the pixel work is per
// 16x16 quadrant: the workgroup scans its parent list in batches of 256 entries (one entry per thread: slot -> the
// vertex stage's tile rect and the 32-byte record), keeps the entries whose rect touches its own 32-px bin, expands them
// to fp32 in LDS next to the 4-bit "which quadrants" mask, and every wave walks only its own survivors of the batch
// (4 ballots + s_ff1 per 256 entries).  Each lane owns 4 pixels (x = lane&15, y = (lane>>4) + 4g); the inner loop reads
// a splat as three wave-uniform ds_read_b128 broadcasts and runs the four 16x4 strips of a lane as two packed,
// branch-free chains (v_pk_*_f32).  Every bin of a 1080p frame is resident at once (8 workgroups per CU) and a wave walks
// only ~90 splats before its quadrant saturates, so the kernel is bound by VALU issue slots per walked splat, not by the
// length of the lists (tools/blend_profile.py).
#include "gs_internal.hpp"

constexpr float GS_POWER_CUT = 5.7707801636f;    // 4*log2(e)  <=>  A > 8
constexpr float GS_T_EPS = 1e-4f;
constexpr float GS_HUGE = 1.2676506e30f;         // 2^100
constexpr int BLEND_THREADS = 256;

typedef float v2f __attribute__((ext_vector_type(2)));
// clamp(a * b + c, 0, 1) on both lanes in one VALU slot
__device__ __forceinline__ v2f pk_fma_sat(v2f a, v2f b, v2f c) {
    v2f d;
    asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(d) : "v"(a), "s"(b), "v"(c));   // b: a wave-uniform constant pair
    return d;
}

// A staged splat, relative to the origin of the 32-px bin that staged it: the two rows of the pixel -> ellipse-space map are
//     u = ax * x + ay * y + cu ,  w = bx * x + by * y + cw      (x, y = pixel centre - bin origin, |x|, |y| < 32)
// with cu = -(ax * cx' + ay * cy'), (cx', cy') = splat centre - bin origin, so the inner loop never forms pixel - centre.
struct __attribute__((aligned(16))) LdsSplat {
    float ax, ay, cu, slab;        // slab: the entry's depth slab (MODE_SEQ), as raw bits
    float bx, by, cw, a;           // 48-byte stride keeps the three reads of a splat 16-byte aligned
    float r, g, b, pad;
};

// which of the 2x2 tiles of the 32-px bin (bx, by) the splat's 16-px tile rect touches: bit (qx + 2*qy).  The rect is the
// vertex stage's own (k_project), so the blend takes exactly its per-tile decisions (a strip of a multi-GPU draw then
// reproduces the full frame bit for bit).
__device__ __forceinline__ uint32_t quadrant_mask(uint2 r16, uint32_t bx, uint32_t by) {
    const uint32_t x0 = r16.x & 0xFFFFu, y0 = r16.x >> 16, x1 = r16.y & 0xFFFFu, y1 = r16.y >> 16;
    const uint32_t cx = 2u * bx, cy = 2u * by;
    const uint32_t mx = ((cx >= x0 && cx <= x1) ? 1u : 0u) | ((cx + 1u >= x0 && cx + 1u <= x1) ? 2u : 0u);
    const uint32_t my = ((cy >= y0 && cy <= y1) ? 1u : 0u) | ((cy + 1u >= y0 && cy + 1u <= y1) ? 2u : 0u);
    return (mx & (my & 1u ? 3u : 0u)) | ((mx & (my & 2u ? 3u : 0u)) << 2);
}

// Exact refinement of the rect's quadrant mask, per HALF quadrant.  A splat reaches a pixel only where
// power = (a.d)^2 + (b.d)^2 <= 4*log2(e) (d = pixel centre - splat centre), a convex quadratic.  Its minimum over the box of
// a half quadrant's pixel centres (16 x 8) is 0 when the centre lies inside, and otherwise lies on an edge facing the centre
// (moving from the minimiser towards the centre lowers the form, so that direction must leave the box): at most two 1-D
// minimisations.  The bounding rect of a long diagonal ellipse covers many tiles the ellipse never enters, and most splats
// of a capture are a few pixels across and touch one half of a quadrant only; a wave that skips a half saves ~20 of its
// ~45 VALU instructions per splat (the two halves are the two packed strip pairs of the inner loop), a skipped quadrant all
// of them - and a skipped half would have been discarded at every pixel (keep = 0: C, T unchanged), so the frame does not
// change and strips of a multi-GPU draw stay bit-exact.
// The edge minimum is evaluated in the per-pixel test's own factored form (a.d)^2 + (b.d)^2 at the clamped minimiser, not as
// m00 x^2 + 2 m01 x y + m11 y^2: for an elongated diagonal splat that expansion cancels catastrophically (relative error
// ~ eps * m00 * m11 / det, i.e. above the margin beyond a 50:1 aspect ratio).  An inexact minimiser only moves the sample
// point along the edge by ~eps * sqrt(m00 / m11) * |x|, a second-order change of the form; the 1e-4 relative margin covers
// it and the fp32 rounding of the per-pixel evaluation.
// Result: bit (2*q + h) = half h (rows 8h .. 8h+7) of quadrant q.
#ifndef GS_BLEND_EXACT
#define GS_BLEND_EXACT 1
#endif
// Skipping the half of a quadrant a splat cannot reach (task 3) was built and measured, and is OFF: the lane
// counters (tools/blend_lanes.py, profiles/r03b_blend_lanes.txt) say 93 % (C3), 97 % (C5), 70 % (C3T), 65 % (C2) of the
// evaluated pixel lanes pass `A <= 8` - the splats a quadrant walks before it saturates are the large near ones, only 3 % /
// 12 % / 13 % of the halves are skippable there - and the branches cost more than they save (same-box r03b, blend ms with
// whole-quadrant masks / half masks + skip: C3 0.070 / 0.076-0.083, C3T 0.477 / 0.507-0.585, C2 0.190 / 0.203-0.226).  Even
// on the capture-like C3S scene (19 % of the lanes kept, 36 % of the halves skippable) it gains nothing (4.38 / 4.30 ms):
// that frame is bound by a few bins with very long lists, not by VALU slots.
#ifndef GS_BLEND_HALF_SKIP
#define GS_BLEND_HALF_SKIP 0          // 1: skip the half (16 x 8 px) of a reached quadrant the ellipse does not enter
#endif
__device__ __forceinline__ uint32_t spread_quadrants(uint32_t qm) {          // 4 quadrant bits -> both half bits of each
    const uint32_t s = (qm & 1u) | ((qm & 2u) << 1) | ((qm & 4u) << 2) | ((qm & 8u) << 3);
    return s | (s << 1);
}
__device__ __forceinline__ uint32_t exact_halves(uint32_t qm, const uint4 lo, const uint4 hi, uint32_t bx, uint32_t by) {
    const float cx = __uint_as_float(lo.x), cy = __uint_as_float(lo.y);
    const float ax = __uint_as_float(lo.z), ay = __uint_as_float(lo.w), ex = __uint_as_float(hi.x), ey = __uint_as_float(hi.y);
    const float m00 = ax * ax + ex * ex, m01 = ax * ay + ex * ey, m11 = ay * ay + ey * ey;
    const float r00 = __builtin_amdgcn_rcpf(m00), r11 = __builtin_amdgcn_rcpf(m11);
    const float limit = GS_POWER_CUT * 1.0001f + 1e-6f;
    uint32_t out = 0;
#ifndef BLEND_EXACT_UNROLL
#define BLEND_EXACT_UNROLL 4
#endif
#ifndef GS_BLEND_EXACT_HALVES
#define GS_BLEND_EXACT_HALVES GS_BLEND_HALF_SKIP   // 0: test whole quadrants (16 x 16); both half bits of a reached one are set
#endif
    const float Xq[2] = {(float)(bx * GS_BIN) + 0.5f - cx, (float)(bx * GS_BIN + GS_TILE) + 0.5f - cx};
#pragma unroll BLEND_EXACT_UNROLL
    for (uint32_t r = 0; r < 4u; r += GS_BLEND_EXACT_HALVES ? 1u : 2u) {   // band r = rows 8r .. 8r+7 of the 32-px bin
        const float Y0 = (float)(by * GS_BIN + r * 8u) + 0.5f - cy, Y1 = Y0 + (GS_BLEND_EXACT_HALVES ? 7.0f : 15.0f);
        const float yb = Y0 > 0.0f ? Y0 : (Y1 < 0.0f ? Y1 : 0.0f);     // bound between the box and the centre, 0 = none
#pragma unroll
        for (uint32_t c = 0; c < 2u; c++) {                               // column c = x 16c .. 16c+15
            const float X0 = Xq[c], X1 = X0 + (float)(GS_TILE - 1u);
            const float xb = X0 > 0.0f ? X0 : (X1 < 0.0f ? X1 : 0.0f);
            float qmin = 0.0f;
            if (xb != 0.0f || yb != 0.0f) {
                const float dy = fminf(fmaxf(-(m01 * xb) * r11, Y0), Y1);     // along the edge x = xb
                const float u1 = ax * xb + ay * dy, w1 = ex * xb + ey * dy;
                const float q1 = xb != 0.0f ? u1 * u1 + w1 * w1 : GS_HUGE;
                const float dx = fminf(fmaxf(-(m01 * yb) * r00, X0), X1);     // along the edge y = yb
                const float u2 = ax * dx + ay * yb, w2 = ex * dx + ey * yb;
                const float q2 = yb != 0.0f ? u2 * u2 + w2 * w2 : GS_HUGE;
                qmin = fminf(q1, q2);
            }
            // quadrant q = c + 2 * (r >> 1), half h = r & 1
            if (!(qmin > limit)) out |= (GS_BLEND_EXACT_HALVES ? 1u : 3u) << (2u * (c + 2u * (r >> 1)) + (r & 1u));   // NaN keeps the half
        }
    }
    return qm & out;
}

// expands one record
__device__ __forceinline__ void stage_entry(LdsSplat* dst, const uint4 lo, const uint4 hi, float bin_x0, float bin_y0, uint32_t slab) {
    LdsSplat s;
    const float cx = __uint_as_float(lo.x) - bin_x0, cy = __uint_as_float(lo.y) - bin_y0;
    s.ax = __uint_as_float(lo.z); s.ay = __uint_as_float(lo.w);
    s.bx = __uint_as_float(hi.x); s.by = __uint_as_float(hi.y);
    s.cu = -__builtin_fmaf(s.ax, cx, s.ay * cy);
    s.cw = -__builtin_fmaf(s.bx, cx, s.by * cy);
    s.slab = __uint_as_float(slab); s.pad = 0.0f;
    s.r = (float)(hi.z & 0xFFFFu) * (1.0f / 65535.0f);
    s.g = (float)(hi.z >> 16) * (1.0f / 65535.0f);
    s.b = (float)(hi.w & 0xFFFFu) * (1.0f / 65535.0f);
    s.a = (float)(hi.w >> 16) * (1.0f / 65535.0f);
    *dst = s;
}

#ifdef GS_BLEND_PROFILE
// tools/blend_profile.py: per bin {start, end} of s_memrealtime (100 MHz), list length, survivors walked by wave 0..3
// ... and [8] lane evaluations (128 per evaluated half), [9] lanes that passed `keep` (A <= 8), [10] lanes that passed it on
// a pixel still accumulating (T > 0), [11] halves evaluated: what fraction of the blend's pixel work can hit anything
constexpr unsigned BLEND_PROF_BINS = 40960, BLEND_PROF_WORDS = 12;
__device__ unsigned long long g_blend_prof[BLEND_PROF_WORDS * BLEND_PROF_BINS];
extern "C" int gs_debug_blend_prof(void* dst, unsigned bins) {
    if (bins > BLEND_PROF_BINS) bins = BLEND_PROF_BINS;
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_blend_prof), (size_t)bins * BLEND_PROF_WORDS * 8, 0, hipMemcpyDeviceToHost);
}
#endif

// 6 workgroups per CU: the exact quadrant test needs 80 VGPRs (at 8 per CU = 64 VGPRs it spilled 50 of them to scratch:
// 132 MB of HBM traffic per launch instead of 34), and with 2040 bins at 1080p a quarter of the workgroups then start late,
// into whatever CU frees up first
#ifndef BLEND_OCC
#define BLEND_OCC 6
#endif
#ifndef GS_BLEND_CHECK
#define GS_BLEND_CHECK 4u             // default mode: a wave tests its quadrant for saturation after every 4th walked splat
#endif
#ifndef GS_BLEND_BRANCH_STYLE
#define GS_BLEND_BRANCH_STYLE 1       // 0: two independent ifs, 1: both / first / second as three blocks (A/B)
#endif
// DEPTH SLABS (GS_CAM_DEPTH_SLABS).  A pixel's composite is sequential in its list, so a bin whose list is tens of thousands
// of entries deep and does not saturate (a surface seen at a grazing angle, a pile of translucent splats) runs on four waves
// for milliseconds while the rest of the GPU idles (the capture-like C3S scene: 8 bins of 2040 take > 4 ms,
// profiles/r03c_blend_profile_C3S.txt).  In slab mode the composite is DEFINED as a two-level fold: the depth sort's buckets are
// cut into GS_SLABS slabs (a splat's slab is a property of the splat and the camera: the top bits of its sort bucket); inside a
// slab a pixel composites its splats from T = 1, and the slabs are merged near -> far with
//     C = fma(T, C_s, C) ;  T = T * T_s  (frozen to 0 at T <= 1e-4, like the per-splat rule).
// A slab that holds nothing for a pixel is exactly neutral and a slab behind T = 0 contributes exactly nothing, so the value of
// a pixel depends only on its own ordered splats and their slabs - not on lists, strips or on WHO executes the fold:
//   MODE_SEQ   one workgroup per bin walks its list as before and closes a slab (merges, resets) whenever the next splat of a
//              wave belongs to another one; early termination works across slabs (running T == 0);
//   MODE_PART  the bins the previous draw found very deep (k_bin_emit's ordering workgroup: cost > 8x the mean) are drawn by
//              one workgroup per (bin, slab) instead, each from T = 1 into a partial {C, T} per pixel, merged by k_slab_fold.
// Both give the same bits (tests/test_gpu_slabs.py), so the choice is pure scheduling.  The frame differs from the default
// single fold by fp32 rounding only (same tolerance against the oracle); draws from host-supplied index lists have no buckets
// and fold as one slab.  (First cut, r03l: EVERY (bin, slab) on a workgroup of its own - early termination across slabs is
// lost, deeper slabs walk what nearer ones already hide: C3 0.30 -> 2.3 ms, C3S 4.6 -> 6.9 ms.  Hence the two modes.)
constexpr int MODE_DEFAULT = 0, MODE_SEQ = 1, MODE_PART = 2;
constexpr uint32_t GS_DEEP_MAX = GS_DEEP_MAX_BINS;        // bins drawn slab-parallel per draw, at most
constexpr uint32_t GS_DEEP_NONE = 0xFFFFFFFFu;
struct SlabArgs {
    float4* partial;            // [GS_DEEP_MAX * GS_SLABS][1024]: {C.r, C.g, C.b, T} per pixel of a deep bin and slab
    uint32_t* opaque_upto;      // [GS_DEEP_MAX]: smallest slab whose own composite saturated every pixel of the bin
    uint32_t* valid;            // [GS_DEEP_MAX * GS_SLABS]: the partial was written
    const uint32_t* deep_list;  // [GS_DEEP_MAX]: the deep bins of this draw
    const uint32_t* deep_count;
    const uint32_t* deep_of;    // [bins]: index in deep_list, or GS_DEEP_NONE
};

template <int MODE>
__device__ __forceinline__ void blend_body(const uint2* __restrict__ ranges, const uint32_t* __restrict__ vals,
                                           const uint4* __restrict__ recs, const uint2* __restrict__ rects,
                                           uint32_t* __restrict__ out, uint32_t width, uint32_t y0, uint32_t y1,
                                           uint32_t bins_x, uint32_t bin_row_begin, uint32_t lists_x,
                                           uint32_t list_row_begin, uint32_t list_shift,
                                           uint2* __restrict__ bin_stats, uint32_t* __restrict__ bin_pairs,
                                           const uint32_t* __restrict__ bin_order, const SlabArgs& sa, const uint32_t wg) {
    __shared__ LdsSplat s_batch[BLEND_THREADS];
    __shared__ uint32_t s_qmask[BLEND_THREADS];
    __shared__ uint32_t s_live;
    __shared__ uint32_t s_abort;
    __shared__ uint32_t s_walked[4], s_halves[4];
    constexpr bool SLAB = MODE == MODE_PART;                // one workgroup per (deep bin, slab), from T = 1 into a partial
    constexpr bool FREEZE = MODE != MODE_DEFAULT;           // per-pixel freeze at saturation (see the inner loop)
    const uint32_t slab = SLAB ? wg % GS_SLABS : 0u;
    uint32_t bin;
    if (SLAB) {
        if (wg / GS_SLABS >= *sa.deep_count) return;
        bin = sa.deep_list[wg / GS_SLABS];
    } else {
        bin = bin_order ? bin_order[wg] : wg;             // heaviest bins of the previous draw first (k_bin_emit)
        if (MODE == MODE_SEQ && sa.deep_of[bin] != GS_DEEP_NONE) return;   // drawn slab-parallel (MODE_PART + k_slab_fold)
    }
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));     // wave-uniform, and known to be
    const uint32_t bx = bin % bins_x, by = bin / bins_x + bin_row_begin;
    const uint32_t qx0 = bx * GS_BIN + (wave & 1u) * GS_TILE, qy0 = by * GS_BIN + (wave >> 1) * GS_TILE;   // quadrant origin
    const uint32_t px = qx0 + (lane & 15u);
    const uint32_t py0 = qy0 + (lane >> 4);
    // pixel centres relative to the bin's origin (what the staged splats are expressed in)
    const float bin_x0 = (float)(bx * GS_BIN), bin_y0 = (float)(by * GS_BIN);
    const float fx = (float)((wave & 1u) * GS_TILE + (lane & 15u)) + 0.5f;
    const float fy0 = (float)((wave >> 1) * GS_TILE + (lane >> 4)) + 0.5f;

#ifdef GS_BLEND_PROFILE
    const unsigned long long t_start = wall_clock64();
    uint32_t batches = 0;
#endif
    uint32_t walked = 0, halves = 0, scanned = 0;      // statistics: wave-uniform, kept in scalar registers
    uint32_t since_check = 0;                          // splats walked since the last saturation test (default mode: never reset
                                                       // by a batch or group boundary - see the test)
#ifdef GS_BLEND_PROFILE
    uint32_t p_kept = 0, p_useful = 0;
#endif
    // the entry list of the list bin this 32-px bin lies in
    const uint32_t per_list = list_shift - GS_BIN_SHIFT;
    const uint32_t list_id = ((by >> per_list) - list_row_begin) * lists_x + (bx >> per_list);
    uint2 range;
    if (MODE == MODE_SEQ) {
        // the list's GS_SLABS per-slab ranges lie back to back in the sorted entries (keys list * GS_SLABS + slab): their union
        __shared__ uint32_t s_rng[2];
        if (tid == 0u) { s_rng[0] = 0xFFFFFFFFu; s_rng[1] = 0u; }
        __syncthreads();
        if (tid < GS_SLABS) {
            const uint2 r = ranges[list_id * GS_SLABS + tid];
            if (r.y > r.x) { atomicMin(&s_rng[0], r.x); atomicMax(&s_rng[1], r.y); }
        }
        __syncthreads();
        range = make_uint2(s_rng[0], s_rng[1]);
    } else {
        range = ranges[SLAB ? list_id * GS_SLABS + slab : list_id];
    }
    if (SLAB && !(range.y > range.x)) return;              // nothing of this slab reaches the list: no partial (the fold skips it)
    const uint32_t begin = range.x, n = range.y > range.x ? range.y - range.x : 0u;   // untouched bins keep (~0, 0)

    // a quadrant outside the viewport / this rank's strip of pixel rows has nothing to draw
    bool live_wave = qx0 < width && qy0 < y1 && qy0 + GS_TILE > y0;

    // strips g = 0..3 of a lane (rows py0 + 4g) as two packed pairs: [h].x = strip 2h, [h].y = strip 2h + 1
    v2f T[2] = {{1.0f, 1.0f}, {1.0f, 1.0f}};
    v2f Cr[2] = {{0, 0}, {0, 0}}, Cg[2] = {{0, 0}, {0, 0}}, Cb[2] = {{0, 0}, {0, 0}};
    const v2f fy[2] = {{fy0, fy0 + 4.0f}, {fy0 + 8.0f, fy0 + 12.0f}};
    // MODE_SEQ: T / C above are the CURRENT slab's composite (from T = 1); these are the fold of the slabs closed so far
    v2f Tr[2] = {{1.0f, 1.0f}, {1.0f, 1.0f}};
    v2f Rr[2] = {{0, 0}, {0, 0}}, Rg[2] = {{0, 0}, {0, 0}}, Rb[2] = {{0, 0}, {0, 0}};
    uint32_t cur_slab = 0;                                  // wave-uniform
    auto close_slab = [&]() {                               // the arithmetic of k_slab_fold, component by component
#pragma unroll
        for (int h = 0; h < 2; h++) {
            Rr[h].x = __builtin_fmaf(Tr[h].x, Cr[h].x, Rr[h].x); Rr[h].y = __builtin_fmaf(Tr[h].y, Cr[h].y, Rr[h].y);
            Rg[h].x = __builtin_fmaf(Tr[h].x, Cg[h].x, Rg[h].x); Rg[h].y = __builtin_fmaf(Tr[h].y, Cg[h].y, Rg[h].y);
            Rb[h].x = __builtin_fmaf(Tr[h].x, Cb[h].x, Rb[h].x); Rb[h].y = __builtin_fmaf(Tr[h].y, Cb[h].y, Rb[h].y);
            const float tx = Tr[h].x * T[h].x, ty = Tr[h].y * T[h].y;
            Tr[h].x = tx > GS_T_EPS ? tx : 0.0f;
            Tr[h].y = ty > GS_T_EPS ? ty : 0.0f;
            T[h] = v2f{1.0f, 1.0f};
            Cr[h] = v2f{0, 0}; Cg[h] = v2f{0, 0}; Cb[h] = v2f{0, 0};
        }
    };

    // entry payload = record slot (k_bin_emit); the slot also names the splat's tile rect, which says whether and where the
    // splat touches THIS bin - most entries of a 128-px list do not, and only the others are expanded into LDS.
    // Software pipeline over batches of 256 entries: the entry word is fetched two batches ahead and the record it names
    // one batch ahead, so a batch waits for ONE gather latency, not for two dependent ones.  Long lists whose pixels do not
    // saturate are bound by exactly that latency (tools/blend_profile.py: ~8 us per batch before, a wave only walks
    // ~10 survivors of a batch).
#ifndef BLEND_PREFETCH
#define BLEND_PREFETCH 1
#endif
    uint4 lo = make_uint4(0, 0, 0, 0), hi = lo;
    uint2 rect = make_uint2(0xFFFFu, 0u);              // empty
    uint32_t v_next = 0;
    uint32_t v_slab = 0, v_slab_next = 0;              // slab of the entry in (lo, hi) / of the entry word in v_next (MODE_SEQ)
    if (BLEND_PREFETCH) {
        if (tid < n) {
            const uint32_t raw = vals[begin + tid];                    // (top bits: the entry's depth slab)
            const uint32_t slot = MODE == MODE_DEFAULT ? raw : raw & GS_SLOT_MASK;
            v_slab = raw >> GS_SLAB_SHIFT;
            rect = rects[slot];
            lo = recs[2 * (size_t)slot];
            hi = recs[2 * (size_t)slot + 1];
        }
        if (BLEND_THREADS + tid < n) { const uint32_t raw = vals[begin + BLEND_THREADS + tid]; v_next = MODE == MODE_DEFAULT ? raw : raw & GS_SLOT_MASK; v_slab_next = raw >> GS_SLAB_SHIFT; }
    } else if (tid < n) {
        const uint32_t raw = vals[begin + tid]; v_next = MODE == MODE_DEFAULT ? raw : raw & GS_SLOT_MASK; v_slab_next = raw >> GS_SLAB_SHIFT;
    }
    for (uint32_t base = 0; base < n; base += BLEND_THREADS) {
        const uint32_t cnt = min((uint32_t)BLEND_THREADS, n - base);
#ifdef GS_BLEND_PROFILE
        batches++;
#endif
        scanned += cnt;
        __syncthreads();                               // previous batch fully consumed, s_live read by everyone
        if (!BLEND_PREFETCH) {                         // only the entry word travels across the inner loop
            rect = make_uint2(0xFFFFu, 0u);
            if (tid < cnt) {
                rect = rects[v_next];
                lo = recs[2 * (size_t)v_next];
                hi = recs[2 * (size_t)v_next + 1];
                v_slab = v_slab_next;
            }
            const uint32_t nx = base + BLEND_THREADS + tid;
            if (nx < n) { const uint32_t raw = vals[begin + nx]; v_next = MODE == MODE_DEFAULT ? raw : raw & GS_SLOT_MASK; v_slab_next = raw >> GS_SLAB_SHIFT; }
        }
        uint32_t qm = tid < cnt ? spread_quadrants(quadrant_mask(rect, bx, by)) : 0u;   // bit 2q + h: half h of quadrant q
        if (GS_BLEND_EXACT && qm) qm = exact_halves(qm, lo, hi, bx, by);
        s_qmask[tid] = qm;
        if (qm) stage_entry(&s_batch[tid], lo, hi, bin_x0, bin_y0, v_slab);           // (v_slab: the payload's top bits)
        if (tid == 0) {
            s_live = 0u;
            // a nearer slab of this bin has saturated every pixel by itself: whatever this one composites is multiplied by 0
            if (SLAB) s_abort = __hip_atomic_load(&sa.opaque_upto[wg / GS_SLABS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < slab ? 1u : 0u;
        }
        const uint32_t nxt = base + BLEND_THREADS + tid;   // prefetch while this batch is blended
        if (BLEND_PREFETCH) {
            if (nxt < n) {
                rect = rects[v_next];
                lo = recs[2 * (size_t)v_next];
                hi = recs[2 * (size_t)v_next + 1];
                v_slab = v_slab_next;
            }
            if (nxt + BLEND_THREADS < n) { const uint32_t raw = vals[begin + nxt + BLEND_THREADS]; v_next = MODE == MODE_DEFAULT ? raw : raw & GS_SLOT_MASK; v_slab_next = raw >> GS_SLAB_SHIFT; }
        }
        __syncthreads();
        if (SLAB && __builtin_amdgcn_readfirstlane((int)s_abort)) return;                       // (uniform: no partial is written, the fold never gets this far)
        if (live_wave) {
            if (FREEZE) since_check = 0;
            for (uint32_t g0 = 0; g0 < cnt && live_wave; g0 += 64) {
                // this wave's survivors among staged entries [g0, g0+64), per half of its quadrant (wave-uniform masks)
                const uint32_t mine = s_qmask[g0 + lane] >> (2u * wave);
                const unsigned long long mh0 = __ballot(mine & 1u), mh1 = __ballot(mine & 2u);
                unsigned long long m = mh0 | mh1;
                while (m) {
                    const uint32_t bit = (uint32_t)__builtin_ctzll(m);
                    const uint32_t j = g0 + bit;
                    m &= m - 1ull;
                    walked++;
                    const float4 q0 = *reinterpret_cast<const float4*>(&s_batch[j].ax);
                    const float4 q1 = *reinterpret_cast<const float4*>(&s_batch[j].bx);
                    const float4 q2 = *reinterpret_cast<const float4*>(&s_batch[j].r);
                    if (MODE == MODE_SEQ) {
                        const uint32_t sj = (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(q0.w));
                        if (sj != cur_slab) {              // this wave's next splat opens another slab: merge the finished one
                            close_slab();
                            cur_slab = sj;
                            bool open = false;
#pragma unroll
                            for (int h = 0; h < 2; h++) open = open || (Tr[h].x > 0.0f) || (Tr[h].y > 0.0f);
                            if (__ballot(open) == 0ull) {  // every pixel of the quadrant is saturated by the slabs merged so far
                                live_wave = false;
                                break;
                            }
                        }
                    }
                    const float ux = __builtin_fmaf(q0.x, fx, q0.z), wx = __builtin_fmaf(q1.x, fx, q1.z);
                    // The four 16x4 strips of a lane are two packed pairs (v_pk_*_f32 does two fp32 lanes per VALU slot): pair
                    // h = rows 8h .. 8h+7 of the quadrant, and a pair the splat cannot reach is skipped with a scalar branch.
                    // `if (A > 8.0) discard` and the freeze are saturated multiply-adds instead of compare + select pairs
                    // (which do not pack and stall on VCC): keep = sat((CUT - pw) * 2^100) is exactly 1 for pw < CUT and 0
                    // for pw >= CUT - fp32 cannot represent a positive difference below 2^-100 here.
                    auto half = [&](const int h) {
                        const v2f u = q0.y * fy[h] + ux;                     // contracted to v_pk_fma_f32
                        const v2f w = q1.y * fy[h] + wx;
                        const v2f pw = w * w + u * u;
                        v2f e;
                        e.x = __builtin_amdgcn_exp2f(-pw.x);
                        e.y = __builtin_amdgcn_exp2f(-pw.y);
                        const v2f keep = pk_fma_sat(pw, v2f{-GS_HUGE, -GS_HUGE}, v2f{GS_POWER_CUT * GS_HUGE, GS_POWER_CUT * GS_HUGE});
#ifdef GS_BLEND_PROFILE
                        p_kept += (uint32_t)__popcll(__ballot(keep.x > 0.0f)) + (uint32_t)__popcll(__ballot(keep.y > 0.0f));
                        p_useful += (uint32_t)__popcll(__ballot(keep.x > 0.0f && T[h].x > 0.0f)) +
                                    (uint32_t)__popcll(__ballot(keep.y > 0.0f && T[h].y > 0.0f));
#endif
                        const v2f alpha = e * (q1.w * keep);
                        const v2f wgt = T[h] * alpha;
                        Cr[h] += wgt * q2.x;
                        Cg[h] += wgt * q2.y;
                        Cb[h] += wgt * q2.z;
                        const v2f t_new = T[h] - wgt;
                        // slab modes: a pixel freezes the moment it saturates (T <= 1e-4 -> 0), so that its value depends only
                        // on its own ordered splats and their slabs, whoever executes the fold
                        if (FREEZE) T[h] = t_new * pk_fma_sat(t_new, v2f{GS_HUGE, GS_HUGE}, v2f{-GS_T_EPS * GS_HUGE, -GS_T_EPS * GS_HUGE});
                        else T[h] = t_new;
                    };
                    // (both halves as ONE straight-line block: the scheduler interleaves the two independent chains exactly as
                    // before there was anything to skip - the common case for large splats; r03a: with two separately
                    // branched blocks the C3 blend went 0.066 -> 0.080 ms while only 3 % of its halves could be skipped)
                    const bool do0 = !GS_BLEND_HALF_SKIP || ((mh0 >> bit) & 1ull), do1 = !GS_BLEND_HALF_SKIP || ((mh1 >> bit) & 1ull);
#if GS_BLEND_BRANCH_STYLE == 0
                    if (do0) { half(0); halves++; }
                    if (do1) { half(1); halves++; }
#else
                    if (do0 && do1) {
                        half(0);
                        half(1);
                        halves += 2u;
                    } else if (do0) {
                        half(0);
                        halves++;
                    } else {
                        half(1);
                        halves++;
                    }
#endif
                    // Retire the wave when its whole quadrant is saturated (every T <= 1e-4).  Default mode: tested after every
                    // GS_BLEND_CHECK-th splat the wave walks and nowhere else, so a quadrant composites exactly the first K of its
                    // own ordered survivors (K = the first multiple of GS_BLEND_CHECK at which all 256 pixels are saturated) - a
                    // function of that sequence alone, not of how the list is batched: strips of a multi-GPU draw reproduce the
                    // full frame bit for bit WITHOUT a per-pixel freeze in the chain (r03: two packed VALU slots per half).
                    if (++since_check == (FREEZE ? 16u : GS_BLEND_CHECK) || (FREEZE && m == 0ull)) {
                        since_check = 0;
                        const float thr = FREEZE ? 0.0f : GS_T_EPS;
                        const float tmax = fmaxf(fmaxf(T[0].x, T[0].y), fmaxf(T[1].x, T[1].y));
                        if (__ballot(tmax > thr) == 0ull) {
                            live_wave = false;
                            break;
                        }
                    }
                }
            }
            // (every lane stores the same word: a `lane == 0` guard here makes live_wave - and with it every counter and branch of
            // the loops above - divergent in the compiler's eyes: exec-mask bookkeeping and VALU counters in the inner loop)
            if (live_wave) s_live = 1u;
        }
        __syncthreads();
        // (LDS words read through readfirstlane: the compiler cannot know they are wave-uniform, and one divergent-looking exit
        // turns every counter of these loops into a VGPR and every branch into exec-mask bookkeeping)
        if (__builtin_amdgcn_readfirstlane((int)s_live) == 0) break;   // every quadrant saturated (or clipped): skip the rest of the list
    }
#ifdef GS_BLEND_PROFILE
    __shared__ unsigned int s_prof[3];
    if (tid < 3u) s_prof[tid] = 0u;
    __syncthreads();
    if (lane == 0u) {
        atomicAdd(&s_prof[0], p_kept);
        atomicAdd(&s_prof[1], p_useful);
        atomicAdd(&s_prof[2], halves);
    }
    if (bin < BLEND_PROF_BINS && lane == 0u) {
        if (wave == 0u) {
            g_blend_prof[BLEND_PROF_WORDS * bin + 0] = t_start;
            g_blend_prof[BLEND_PROF_WORDS * bin + 2] = n;
            g_blend_prof[BLEND_PROF_WORDS * bin + 3] = batches;
        }
        g_blend_prof[BLEND_PROF_WORDS * bin + 4 + wave] = walked;
    }
    __syncthreads();
    if (bin < BLEND_PROF_BINS && tid == 0u) {
        g_blend_prof[BLEND_PROF_WORDS * bin + 1] = wall_clock64();
        g_blend_prof[BLEND_PROF_WORDS * bin + 8] = 128ull * s_prof[2];
        g_blend_prof[BLEND_PROF_WORDS * bin + 9] = s_prof[0];
        g_blend_prof[BLEND_PROF_WORDS * bin + 10] = s_prof[1];
        g_blend_prof[BLEND_PROF_WORDS * bin + 11] = s_prof[2];
    }
#endif
    // statistics: one plain 8-byte store per workgroup, summed by the host when somebody asks (8160 same-address atomics
    // at the end of the kernel cost 60 us: a device-scope counter retires ~88 atomics per microsecond)
    // {entries staged, half quadrants evaluated} per bin (the blend's cost: what orders the next draw's workgroups and balances
    // multi-GPU strips), and the (splat, quadrant) pairs in a plane of their own behind them
    if (lane == 0u) { s_walked[wave] = walked; s_halves[wave] = halves; }
    __syncthreads();
    if (SLAB) {
        // this slab's composite of the bin: {C, T} per pixel (pixels it never touched keep the neutral (0, 0, 0, 1)); the fold
        // (k_slab_fold) merges the slabs in order and writes the frame.  s_live == 0 after the last batch: every quadrant of the
        // bin is saturated (or clipped) by this slab alone - farther slabs need not finish.
        float4* part = sa.partial + (size_t)wg * 1024u + wave * 256u + lane;
#pragma unroll
        for (int g = 0; g < 4; g++) part[64 * g] = make_float4(Cr[g >> 1][g & 1], Cg[g >> 1][g & 1], Cb[g >> 1][g & 1], T[g >> 1][g & 1]);
        if (tid == 0u) {
            atomicAdd(&bin_stats[bin].x, scanned);
            atomicAdd(&bin_stats[bin].y, s_halves[0] + s_halves[1] + s_halves[2] + s_halves[3]);
            atomicAdd(&bin_pairs[bin], s_walked[0] + s_walked[1] + s_walked[2] + s_walked[3]);
            if (s_live == 0u) atomicMin(&sa.opaque_upto[wg / GS_SLABS], slab);
        }
        __threadfence();                                    // the partial is visible before its flag
        __syncthreads();
        if (tid == 0u) __hip_atomic_store(&sa.valid[wg], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    if (MODE == MODE_SEQ) {                                 // the last open slab, then the frame comes from the fold
        close_slab();
#pragma unroll
        for (int h = 0; h < 2; h++) { T[h] = Tr[h]; Cr[h] = Rr[h]; Cg[h] = Rg[h]; Cb[h] = Rb[h]; }
    }
    if (tid == 0u) {
        bin_stats[bin] = make_uint2(scanned, s_halves[0] + s_halves[1] + s_halves[2] + s_halves[3]);
        bin_pairs[bin] = s_walked[0] + s_walked[1] + s_walked[2] + s_walked[3];
    }
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const uint32_t py = py0 + 4u * g;
        if (px < width && py >= y0 && py < y1) {
            const float a = 1.0f - T[g >> 1][g & 1];
            const uint32_t r8 = (uint32_t)(fminf(fmaxf(Cr[g >> 1][g & 1], 0.0f), 1.0f) * 255.0f + 0.5f);
            const uint32_t g8 = (uint32_t)(fminf(fmaxf(Cg[g >> 1][g & 1], 0.0f), 1.0f) * 255.0f + 0.5f);
            const uint32_t b8 = (uint32_t)(fminf(fmaxf(Cb[g >> 1][g & 1], 0.0f), 1.0f) * 255.0f + 0.5f);
            const uint32_t a8 = (uint32_t)(fminf(fmaxf(a, 0.0f), 1.0f) * 255.0f + 0.5f);
            out[(size_t)(py - y0) * width + px] = r8 | (g8 << 8) | (b8 << 16) | (a8 << 24);
        }
    }
}

// the default frame: one workgroup per bin, the single fold
__global__ __launch_bounds__(BLEND_THREADS, BLEND_OCC) void k_tile_blend(const uint2* __restrict__ ranges, const uint32_t* __restrict__ vals,
                                                              const uint4* __restrict__ recs, const uint2* __restrict__ rects,
                                                              uint32_t* __restrict__ out, uint32_t width, uint32_t y0, uint32_t y1,
                                                              uint32_t bins_x, uint32_t bin_row_begin, uint32_t lists_x,
                                                              uint32_t list_row_begin, uint32_t list_shift,
                                                              uint2* __restrict__ bin_stats, uint32_t* __restrict__ bin_pairs,
                                                              const uint32_t* __restrict__ bin_order) {
    const SlabArgs none = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    blend_body<MODE_DEFAULT>(ranges, vals, recs, rects, out, width, y0, y1, bins_x, bin_row_begin, lists_x, list_row_begin, list_shift,
                             bin_stats, bin_pairs, bin_order, none, blockIdx.x);
}

// slab mode, one launch: the first GS_DEEP_MAX * GS_SLABS workgroups are the (deep bin, slab) slots - the long ones start
// first -, the rest draw one bin each (and leave the deep bins alone).  5 workgroups per CU: the sequential mode carries the
// merged {C, T} next to the open slab's (92 VGPRs).
__global__ __launch_bounds__(BLEND_THREADS, 5) void k_tile_blend_slabs(const uint2* __restrict__ ranges, const uint32_t* __restrict__ vals,
                                                              const uint4* __restrict__ recs, const uint2* __restrict__ rects,
                                                              uint32_t* __restrict__ out, uint32_t width, uint32_t y0, uint32_t y1,
                                                              uint32_t bins_x, uint32_t bin_row_begin, uint32_t lists_x,
                                                              uint32_t list_row_begin, uint32_t list_shift,
                                                              uint2* __restrict__ bin_stats, uint32_t* __restrict__ bin_pairs,
                                                              const uint32_t* __restrict__ bin_order, SlabArgs sa) {
    if (blockIdx.x < GS_DEEP_MAX * GS_SLABS)
        blend_body<MODE_PART>(ranges, vals, recs, rects, out, width, y0, y1, bins_x, bin_row_begin, lists_x, list_row_begin, list_shift,
                              bin_stats, bin_pairs, nullptr, sa, blockIdx.x);
    else
        blend_body<MODE_SEQ>(ranges, vals, recs, rects, out, width, y0, y1, bins_x, bin_row_begin, lists_x, list_row_begin, list_shift,
                             bin_stats, bin_pairs, bin_order, sa, blockIdx.x - GS_DEEP_MAX * GS_SLABS);
}

// The second level of the slab-mode composite for the deep bins: per pixel, the slabs' partials merged near -> far,
//     C = fma(T, C_s, C) ;  T = T * T_s  (frozen to 0 at T <= 1e-4, like the per-splat rule)
// exactly what MODE_SEQ does when it closes a slab.  A (bin, slab) whose list range is empty has no partial and is skipped
// (exactly neutral); one whose partial is missing can only lie behind a slab that saturated the whole bin (its workgroup gave
// up because of that), where T is 0 for every pixel: the fold stops there.
__global__ __launch_bounds__(BLEND_THREADS) void k_slab_fold(const uint2* __restrict__ ranges, SlabArgs sa, uint32_t* __restrict__ out,
                                                              uint32_t width, uint32_t y0, uint32_t y1, uint32_t bins_x,
                                                              uint32_t bin_row_begin, uint32_t lists_x, uint32_t list_row_begin,
                                                              uint32_t list_shift) {
    if (blockIdx.x >= *sa.deep_count) return;
    const uint32_t bin = sa.deep_list[blockIdx.x], tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t bx = bin % bins_x, by = bin / bins_x + bin_row_begin;
    const uint32_t px = bx * GS_BIN + (wave & 1u) * GS_TILE + (lane & 15u), py0 = by * GS_BIN + (wave >> 1) * GS_TILE + (lane >> 4);
    const uint32_t per_list = list_shift - GS_BIN_SHIFT;
    const uint32_t list_id = ((by >> per_list) - list_row_begin) * lists_x + (bx >> per_list);
    float T[4] = {1.0f, 1.0f, 1.0f, 1.0f}, C[4][3] = {};
    for (uint32_t s = 0; s < GS_SLABS; s++) {
        const uint2 range = ranges[list_id * GS_SLABS + s];
        if (!(range.y > range.x)) continue;
        if (__hip_atomic_load(&sa.valid[blockIdx.x * GS_SLABS + s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) break;
        const float4* part = sa.partial + ((size_t)blockIdx.x * GS_SLABS + s) * 1024u + wave * 256u + lane;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const float4 p = part[64 * g];
            C[g][0] = __builtin_fmaf(T[g], p.x, C[g][0]);
            C[g][1] = __builtin_fmaf(T[g], p.y, C[g][1]);
            C[g][2] = __builtin_fmaf(T[g], p.z, C[g][2]);
            const float t = T[g] * p.w;
            T[g] = t > GS_T_EPS ? t : 0.0f;
        }
    }
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const uint32_t py = py0 + 4u * g;
        if (px < width && py >= y0 && py < y1) {
            const float a = 1.0f - T[g];
            const uint32_t r8 = (uint32_t)(fminf(fmaxf(C[g][0], 0.0f), 1.0f) * 255.0f + 0.5f);
            const uint32_t g8 = (uint32_t)(fminf(fmaxf(C[g][1], 0.0f), 1.0f) * 255.0f + 0.5f);
            const uint32_t b8 = (uint32_t)(fminf(fmaxf(C[g][2], 0.0f), 1.0f) * 255.0f + 0.5f);
            const uint32_t a8 = (uint32_t)(fminf(fmaxf(a, 0.0f), 1.0f) * 255.0f + 0.5f);
            out[(size_t)(py - y0) * width + px] = r8 | (g8 << 8) | (b8 << 16) | (a8 << 24);
        }
    }
}

int gs_launch_blend(gs_mesh* m, const ProjectParams& pp, uint8_t* out_dev) {
    const uint32_t bins = pp.bins_x * (pp.bin_row_end - pp.bin_row_begin);
    if (bins == 0) return GS_OK;
    const uint32_t* vals = (m->sorted_buf ? m->evalB : m->evalA).as<uint32_t>();
    GS_TRY(m->blend_stats.ensure((size_t)bins * 12));     // uint2 [bins] {staged, halves} | uint32 [bins] pairs
    m->blend_bins = bins;
    const uint32_t* order = m->blend_order_valid ? m->blend_order.as<uint32_t>() : nullptr;
    if (pp.slabs) {
        // (the buffers were sized and reset by the binner's launches: gs_launch_binning)
        uint32_t* flags = m->slab_flags.as<uint32_t>();      // layout: GS_FLAG_* in gs_internal.hpp
        SlabArgs sa;
        sa.partial = m->slab_partial.as<float4>();
        sa.opaque_upto = flags;
        sa.valid = flags + GS_FLAG_VALID;
        sa.deep_list = flags + GS_FLAG_LIST;
        sa.deep_count = flags + GS_FLAG_COUNT;
        sa.deep_of = flags + GS_FLAG_OF;
        hipLaunchKernelGGL(k_tile_blend_slabs, dim3(GS_DEEP_MAX * GS_SLABS + bins), dim3(BLEND_THREADS), 0, m->ctx->stream,
                           m->tile_ranges.as<uint2>(), vals, m->recs.as<uint4>(), m->rects.as<uint2>(), reinterpret_cast<uint32_t*>(out_dev),
                           (uint32_t)pp.width, pp.y0, pp.y1, pp.bins_x, pp.bin_row_begin, pp.lists_x, pp.list_row_begin, pp.list_shift,
                           m->blend_stats.as<uint2>(), m->blend_stats.as<uint32_t>() + 2 * (size_t)bins, order, sa);
        hipLaunchKernelGGL(k_slab_fold, dim3(GS_DEEP_MAX), dim3(BLEND_THREADS), 0, m->ctx->stream, m->tile_ranges.as<uint2>(), sa,
                           reinterpret_cast<uint32_t*>(out_dev), (uint32_t)pp.width, pp.y0, pp.y1, pp.bins_x, pp.bin_row_begin,
                           pp.lists_x, pp.list_row_begin, pp.list_shift);
    } else {
        hipLaunchKernelGGL(k_tile_blend, dim3(bins), dim3(BLEND_THREADS), 0, m->ctx->stream, m->tile_ranges.as<uint2>(), vals,
                           m->recs.as<uint4>(), m->rects.as<uint2>(), reinterpret_cast<uint32_t*>(out_dev), (uint32_t)pp.width, pp.y0,
                           pp.y1, pp.bins_x, pp.bin_row_begin, pp.lists_x, pp.list_row_begin, pp.list_shift,
                           m->blend_stats.as<uint2>(), m->blend_stats.as<uint32_t>() + 2 * (size_t)bins, order);
    }
    m->blend_row_begin = pp.bin_row_begin;
    m->blend_width = (uint32_t)pp.width;
    GS_HIP(hipGetLastError());
    return GS_OK;
}
