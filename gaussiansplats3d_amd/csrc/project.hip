// project.hip — vertex stage of the RENDER SEAM: one thread per splat in STORAGE order.
//
// Restates (never copies) the reference's GLSL vertex shader, executed there 4x per splat by an instanced
// quad draw:  /root/reference/src/splatmesh/SplatMaterial.js:112-341 (fetch, view/clip transform, 1.2x
// frustum reject, SH degree 1/2 colour) and /root/reference/src/splatmesh/SplatMaterial3D.js:81-217
// (3D->2D covariance, +kernel2DSize, eigen basis, 1024-px clamp).
//
// MI355X layout: every input is an SoA *plane* whose element is the widest vector the attribute allows
// (float for x/y/z, float4+float2 for the covariance, uint4 x3 for the 24 fp16 SH coefficients), so a
// wave64 reads 256 B..1 KiB contiguous per instruction and each byte is fetched from HBM exactly once.
// Storage order (not depth order) keeps those reads streaming; the depth sort runs concurrently on the
// keys alone and meets this stage's output only through 8-byte tile rects and 32-byte records.
//
// Outputs per splat: SplatRec (32 B, what the blend gathers) and a tile rect (8 B, what the binner gathers).
// Built with -ffp-contract=off and written in the oracle's operation order so accept/reject decisions agree.
#include "gs_internal.hpp"

struct MeshPlanes {
    const float* __restrict__ px;
    const float* __restrict__ py;
    const float* __restrict__ pz;
    const void* __restrict__ covA;      // fp32: float4 (c0..c3)   | fp16: uint2 (c0..c3)
    const void* __restrict__ covB;      // fp32: float2 (c4,c5)    | fp16: uint  (c4,c5)
    const float* __restrict__ cov_bound;   // spectral-radius bound of the covariance (mesh.hip, cov_spectral_bound)
    const float* __restrict__ block_box;   // per 256-splat storage block: {min xyz, max xyz, max cov_bound, -} (mesh.hip, k_block_boxes)
    const uint32_t* __restrict__ rgba;
    const uint4* __restrict__ sh0;      // halfs 0..7
    const void* __restrict__ sh1;       // SH2: uint4 halfs 8..15 | SH1: uint (half 8)
    const uint4* __restrict__ sh2;      // SH2: halfs 16..23
    const uint32_t* __restrict__ scene_idx;          // per-splat scene (EXT, scene_count > 1)
    const gs_scene_params* __restrict__ scenes;      // per-scene uniforms (EXT)
};

// One storage block's verdict: can NO splat of the block reach the frame (or this rank's strip)?  See k_project's header comment
// for the argument; `corner` evaluates one of the box's eight corners, the reductions over the corners are the caller's.
struct BlockCorner {
    float q[4], v[3];
    bool rej[6];            // this corner satisfies reject k (x > 1.2 w, x < -1.2 w, y > .., y < .., z < -1.2 w, w < 0) beyond the margin
    bool front;             // properly in front of the camera
    float ypx;              // its window y
};
__device__ __forceinline__ BlockCorner block_corner(const ProjectParams& pp, const float* bb, uint32_t c) {
    BlockCorner o;
    const float x = (c & 1u) ? bb[3] : bb[0], y = (c & 2u) ? bb[4] : bb[1], z = (c & 4u) ? bb[5] : bb[2];
    const float* MV = pp.view;
    const float* P = pp.proj;
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; r++) v[r] = MV[r] * x + MV[4 + r] * y + MV[8 + r] * z + MV[12 + r];
#pragma unroll
    for (int r = 0; r < 4; r++) o.q[r] = P[r] * v[0] + P[4 + r] * v[1] + P[8 + r] * v[2] + P[12 + r] * v[3];
    o.v[0] = v[0]; o.v[1] = v[1]; o.v[2] = v[2];
    const float clip = 1.2f * o.q[3], tol = 1e-4f * (fabsf(o.q[0]) + fabsf(o.q[1]) + fabsf(o.q[2]) + fabsf(clip) + 1.0f);
    o.rej[0] = o.q[0] - clip > tol; o.rej[1] = -o.q[0] - clip > tol;
    o.rej[2] = o.q[1] - clip > tol; o.rej[3] = -o.q[1] - clip > tol;
    o.rej[4] = -o.q[2] - clip > tol; o.rej[5] = -o.q[3] > tol;
    o.front = o.q[3] > 1e-6f && v[2] < -1e-6f;
    o.ypx = (o.q[1] / o.q[3] * 0.5f + 0.5f) * pp.height;
    return o;
}
// the strip half of the verdict, from the reductions over the eight corners (all of them in front of the camera)
__device__ __forceinline__ bool block_misses_strip(const ProjectParams& pp, float cov_bound_max, float ymin, float ymax, float zmin,
                                                   float axmax, float aymax) {
    const float ks = fabsf(pp.splat_scale * pp.inv_focal_adj);
    float reach = pp.max_splat_px * ks * 1.001f + 2.0f;
    const float iz = 1.0f / zmin, s2 = iz * iz;
    const float t0 = fabsf(pp.focal_x) * iz * pp.mv_row_norm[0] + fabsf(pp.focal_x) * axmax * s2 * pp.mv_row_norm[2];
    const float t1 = fabsf(pp.focal_y) * iz * pp.mv_row_norm[1] + fabsf(pp.focal_y) * aymax * s2 * pp.mv_row_norm[2];
    const float t = fmaxf(t0, t1) * 1.001f;
    float l = cov_bound_max * t * t + pp.kernel2d + 0.3163f;
    if (pp.flags & GS_CAM_POINT_CLOUD) l = fmaxf(l, 0.2f);
    const float tight = ks * sqrtf(8.0f * l) * 1.002f + 2.0f;
    if (tight < reach) reach = tight;                     // false for NaN: the cap stays
    const float slack = 0.05f + 1e-5f * pp.height;       // the corners' own fp32 projection
    return ymax + reach + slack < (float)(pp.row_begin * GS_TILE) || ymin - reach - slack > (float)(pp.row_end * GS_TILE);
}

// The block test as a kernel of its own (round 5): eight lanes per storage block run the eight-corner test; a dead block gets its
// empty masks here and block_any = 0, a live one block_any = 1 (k_project overwrites it with what it finds), and k_project's
// workgroup b leaves at once when block_any[b] == 0 - one byte, no box read, no corner arithmetic, no barrier.  Rounds 2-4 had every
// one of the 22.6 k workgroups of a C3 draw evaluate its own box: the vertex stage 59.6 -> 52 us at C3, a rank's strip of eight
// 41.6 -> 32 us (r05a-c, same box).  No list and no counter: the first version appended the live blocks to a list through one
// wave-aggregated atomic per wave and let k_project walk it - 354 same-address atomics were fine, but with eight lanes per block
// (2829 waves) they alone took 17 us (one address retires ~88 atomics per microsecond; r05g rank_C3_n1 table).
__global__ __launch_bounds__(256) void k_block_test(ProjectParams pp, const float* __restrict__ block_box, uint32_t blocks,
                                                    unsigned long long* __restrict__ vis_mask, uint2* __restrict__ vis32,
                                                    uint8_t* __restrict__ block_any) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x, b = t >> 3, c = threadIdx.x & 7u;
    const bool in = b < blocks;
    const float* bb = block_box + 8u * (size_t)(in ? b : 0u);
    const BlockCorner k = block_corner(pp, bb, c);
    // AND / min / max over the block's eight lanes
    // (bit masks and `&`, never `&&`: a short-circuited `x && __shfl_xor(x)` takes the lanes whose x is false out of the shuffle,
    // and what the others then read from them is undefined - the first version of this kernel declared 4 % of C3's visible splats
    // dead that way, r05e)
    uint32_t flags = (k.rej[0] ? 1u : 0u) | (k.rej[1] ? 2u : 0u) | (k.rej[2] ? 4u : 0u) | (k.rej[3] ? 8u : 0u) | (k.rej[4] ? 16u : 0u) |
                     (k.rej[5] ? 32u : 0u) | (k.front ? 64u : 0u);
    float ymin = k.ypx, ymax = k.ypx, zmin = -k.v[2], axmax = fabsf(k.v[0]), aymax = fabsf(k.v[1]);
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
        flags &= (uint32_t)__shfl_xor((int)flags, o, 64);
        ymin = fminf(ymin, __shfl_xor(ymin, o, 64)); ymax = fmaxf(ymax, __shfl_xor(ymax, o, 64));
        zmin = fminf(zmin, __shfl_xor(zmin, o, 64));
        axmax = fmaxf(axmax, __shfl_xor(axmax, o, 64)); aymax = fmaxf(aymax, __shfl_xor(aymax, o, 64));
    }
    bool dead = (flags & 63u) != 0u;                         // every corner satisfies one of the six rejects
    const bool strip = pp.row_begin > 0u || pp.row_end < pp.tiles_y, all_front = (flags & 64u) != 0u;
    if (!dead && strip && !(pp.flags & GS_CAM_ORTHOGRAPHIC) && all_front) dead = block_misses_strip(pp, bb[6], ymin, ymax, zmin, axmax, aymax);
    if (!in) return;
    if (dead) {                                              // nothing of this block draws: empty masks, no records (lane c: its share)
        if (c < 4u) vis_mask[4u * b + c] = 0ull;
        vis32[8u * b + c] = make_uint2(0u, b * 256u);
    }
    if (c == 0u) block_any[b] = dead ? 0 : 1;
}

__device__ __forceinline__ float h2f(uint32_t bits16) {
    return (float)__builtin_bit_cast(_Float16, (unsigned short)(bits16 & 0xFFFFu));
}
__device__ __forceinline__ float clamp01(float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); }
__device__ __forceinline__ uint32_t unorm16(float v) { return (uint32_t)(clamp01(v) * 65535.0f + 0.5f); }

constexpr float GS_K_POWER = 2.4022448f;          // sqrt(4*log2(e)): alpha = exp2(-|K*q|^2) == exp(-0.5*A)
constexpr uint32_t RECT_EMPTY_LO = 0x0000FFFFu;   // x0 = 0xFFFF > x1 = 0 -> zero tiles

// View-dependent colour (SplatMaterial.js:179-337): evaluated only for splats that reach the frame - the SH planes are
// the widest read of the vertex stage (48 B/splat at SH-2), and a splat dropped by the eigenvalue floor, an empty pixel
// rect or another rank's strip of tile rows never needs them.
#ifndef GS_SH_EARLY
#define GS_SH_EARLY 0
#endif
#ifndef GS_COV_EARLY
#define GS_COV_EARLY 1
#endif
#ifndef GS_BLOCK_TEST_PER_WAVE
#define GS_BLOCK_TEST_PER_WAVE 0      // 1: every wave of a block evaluates the block's box itself (A/B)
#endif
struct ShPre {                  // fp16 SH planes fetched together with the covariance (GS_SH_EARLY): one dependent round trip less
    uint4 a, b, c;
    bool have;
};
template <bool EXT>
__device__ __forceinline__ void sh_colour(const ProjectParams& pp, const MeshPlanes& mp, uint32_t i, uint32_t scene, float c0,
                                          float c1, float c2, float* col, const ShPre& pre) {
    if (pp.sh_stored >= 1 && pp.sh_degree >= 1) {
        float cp0 = pp.cam_pos[0], cp1 = pp.cam_pos[1], cp2 = pp.cam_pos[2];
        if (EXT && (pp.flags & GS_CAM_DYNAMIC)) {                // :179-183 camera in the scene's own frame
            cp0 = mp.scenes->inv_cam_pos[scene][0]; cp1 = mp.scenes->inv_cam_pos[scene][1];
            cp2 = mp.scenes->inv_cam_pos[scene][2];
        }
        float d0 = c0 - cp0, d1 = c1 - cp1, d2 = c2 - cp2;                                 // :185
        const float inv = 1.0f / sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
        const float x = d0 * inv, y = d1 * inv, z = d2 * inv;
        float sh[24];
        const uint4 a = pre.have ? pre.a : mp.sh0[i];
        if (EXT && pp.sh_u8) {
            // 8-bit SH (:150-154,265-269): texel = v/255 (unorm8), sh = texel*range + min; sh0 = bytes 0..15
            const float mn = mp.scenes->sh8_min[scene], range = mp.scenes->sh8_max[scene] - mn;
            const uint32_t w0[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int k = 0; k < 16; k++) sh[k] = ((float)((w0[k >> 2] >> (8 * (k & 3))) & 255u) / 255.0f) * range + mn;
            if (pp.sh_stored >= 2) {
                const uint2 b = reinterpret_cast<const uint2*>(mp.sh1)[i];
                const uint32_t w1[2] = {b.x, b.y};
#pragma unroll
                for (int k = 0; k < 8; k++) sh[16 + k] = ((float)((w1[k >> 2] >> (8 * (k & 3))) & 255u) / 255.0f) * range + mn;
            }
        } else {
        sh[0] = h2f(a.x); sh[1] = h2f(a.x >> 16); sh[2] = h2f(a.y); sh[3] = h2f(a.y >> 16);
        sh[4] = h2f(a.z); sh[5] = h2f(a.z >> 16); sh[6] = h2f(a.w); sh[7] = h2f(a.w >> 16);
        if (pp.sh_stored >= 2) {
            const uint4 b = pre.have ? pre.b : reinterpret_cast<const uint4*>(mp.sh1)[i];
            sh[8] = h2f(b.x); sh[9] = h2f(b.x >> 16); sh[10] = h2f(b.y); sh[11] = h2f(b.y >> 16);
            sh[12] = h2f(b.z); sh[13] = h2f(b.z >> 16); sh[14] = h2f(b.w); sh[15] = h2f(b.w >> 16);
        } else {
            sh[8] = h2f(reinterpret_cast<const uint32_t*>(mp.sh1)[i]);
        }
        }
        const float SH_C1 = 0.4886025119029199f;                                            // :273
#pragma unroll
        for (int ch = 0; ch < 3; ch++) col[ch] += SH_C1 * (-sh[0 + ch] * y + sh[3 + ch] * z - sh[6 + ch] * x);
        if (pp.sh_stored >= 2 && pp.sh_degree >= 2) {                                       // :308-330
            if (!(EXT && pp.sh_u8)) {
                const uint4 cc = pre.have ? pre.c : mp.sh2[i];
                sh[16] = h2f(cc.x); sh[17] = h2f(cc.x >> 16); sh[18] = h2f(cc.y); sh[19] = h2f(cc.y >> 16);
                sh[20] = h2f(cc.z); sh[21] = h2f(cc.z >> 16); sh[22] = h2f(cc.w); sh[23] = h2f(cc.w >> 16);
            }
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            const float C0 = 1.0925484f, C1 = -1.0925484f, C2 = 0.3153916f, C3 = -1.0925484f, C4 = 0.5462742f;
#pragma unroll
            for (int ch = 0; ch < 3; ch++)
                col[ch] += (C0 * xy) * sh[9 + ch] + (C1 * yz) * sh[12 + ch] + (C2 * (2.0f * zz - xx - yy)) * sh[15 + ch] +
                           (C3 * xz) * sh[18 + ch] + (C4 * (xx - yy)) * sh[21 + ch];
        }
#pragma unroll
        for (int ch = 0; ch < 3; ch++) col[ch] = clamp01(col[ch]);                          // :337
    }
}

// EXT = false: the static perspective scene with fp16 SH (the benchmark path).  EXT = true adds the reference's shader
// permutations: orthographic J, per-scene transforms (dynamicMode), per-scene opacity / visibility
// (enableOptionalEffects), 8-bit SH, distance fade-in.
// One storage block `blk` of 256 splats by one workgroup (k_project below says which).
// DEPTH: a destination depth is set (gs_mesh_set_destination) and the blend wants every survivor's window depth - a compile-time
// switch, because one more pointer and two more uniforms pushed the plain kernel over its scalar-register budget (106 SGPRs, 24
// spilled: the all-visible C4 launch went 250 -> 278 us, r05m).
template <bool EXT, bool TEST, bool DEPTH>
__device__ __forceinline__ void project_block(const ProjectParams& pp, const MeshPlanes& mp, const uint32_t blk, SplatRec* __restrict__ recs,
                                              uint2* __restrict__ rects, unsigned long long* __restrict__ vis_mask,
                                              uint2* __restrict__ vis32, uint32_t* __restrict__ vis_orig,
                                              const uint32_t* __restrict__ inv_perm, uint8_t* __restrict__ block_any,
                                              uint2* __restrict__ prect, float* __restrict__ zrec) {
    const uint32_t i = blk * 256u + threadIdx.x;
    // Block-level cull.  Storage order is Morton order, so a block of 256 splats is a small box in space; its eight corners
    // (one lane each, wave 0) decide whether EVERY splat inside must fail the vertex stage - then nothing of the block is
    // read: not even its centres (12 bytes x 256), which is all a rank of a multi-GPU draw still paid for the ~85 % of the
    // scene that cannot reach its strip, and what the 75 % of a scene outside the frustum cost a single GPU.
    //  * frustum: each reject of SplatMaterial.js:160-164 (x > 1.2 w, x < -1.2 w, the same for y, z < -1.2 w; w < 0 fails them
    //    all) is a linear inequality in the position, so if all eight corners satisfy ONE of them - with a margin far above
    //    the fp32 rounding of either evaluation - so does every point of the box;
    //  * strip (tile_row_begin / end): with every corner in front of the camera the box projects into the hull of its corners;
    //    the reach of any splat inside is bounded as in the per-splat pre-test below, from the block's largest covariance
    //    bound, its nearest depth and its largest |x|, |y| in view space.
    // A dead block publishes empty masks and returns; the frame cannot change (every splat it skips would have been rejected).
    // What it buys (r03, tools/project_floor.py, C3): full frame 59.8 -> 56.5 us, one strip of eight 39.7 -> 38.8 us, a camera
    // that sees nothing 26.0 -> 20.2 us: the floor is now the dispatch of 22.6 k workgroups, not their memory traffic.  Both
    // ways around that floor were built and measured slower: a fixed grid of workgroups looping over the blocks (the loop needs
    // 76 VGPRs instead of 46: 72.5 / 51.9 / 25.4 us) and four blocks per 1024-thread workgroup (92.7 / 71.9 / 36.1 us: sixteen
    // waves have to find room at once and wait for each other at the barrier).  The same test again per WAVE (a box per 64 splats,
    // dead waves of live blocks skip their centre loads and the shader) is slower as well: the test costs every live wave ~60
    // instructions and a 32-byte read, and Morton blocks are already nearly all-or-nothing (C3 59.5 -> 62.8 us, C2 34.0 -> 37.5,
    // C4 262 -> 300).
    if (TEST && pp.block_cull) {                   // (MODE 0 only: in MODE 1 k_block_test has decided already)
        __shared__ uint32_t s_dead;
        bool wave_dead = false;
        if (GS_BLOCK_TEST_PER_WAVE || threadIdx.x < 64u) {
            const float* bb = mp.block_box + 8u * (size_t)blk;
            const BlockCorner k = block_corner(pp, bb, threadIdx.x & 7u);
            const unsigned long long lanes8 = 0xFFull;
            const bool dead_frustum = (__ballot(k.rej[0]) & lanes8) == lanes8 || (__ballot(k.rej[1]) & lanes8) == lanes8 ||
                                      (__ballot(k.rej[2]) & lanes8) == lanes8 || (__ballot(k.rej[3]) & lanes8) == lanes8 ||
                                      (__ballot(k.rej[4]) & lanes8) == lanes8 || (__ballot(k.rej[5]) & lanes8) == lanes8;
            bool dead = dead_frustum;
            const bool strip = pp.row_begin > 0u || pp.row_end < pp.tiles_y;
            if (!dead && strip && !(pp.flags & GS_CAM_ORTHOGRAPHIC)) {
                // every corner properly in front of the camera (w > 0 and view-space z < 0)?
                float ymin = k.ypx, ymax = k.ypx, zmin = -k.v[2], axmax = fabsf(k.v[0]), aymax = fabsf(k.v[1]);
#pragma unroll
                for (int o = 1; o < 8; o <<= 1) {
                    ymin = fminf(ymin, __shfl_xor(ymin, o, 64)); ymax = fmaxf(ymax, __shfl_xor(ymax, o, 64));
                    zmin = fminf(zmin, __shfl_xor(zmin, o, 64));
                    axmax = fmaxf(axmax, __shfl_xor(axmax, o, 64)); aymax = fmaxf(aymax, __shfl_xor(aymax, o, 64));
                }
                if ((__ballot(k.front) & lanes8) == lanes8) dead = block_misses_strip(pp, bb[6], ymin, ymax, zmin, axmax, aymax);
            }
            if (GS_BLOCK_TEST_PER_WAVE) wave_dead = __builtin_amdgcn_readfirstlane((int)dead) != 0;     // (every wave evaluated the same box: no LDS word, no barrier)
            else if (threadIdx.x == 0u) s_dead = dead ? 1u : 0u;
        }
        if (!GS_BLOCK_TEST_PER_WAVE) {
            __syncthreads();
            wave_dead = s_dead != 0u;
        }
        if (wave_dead) {                                      // nothing of this block draws: empty masks, no records
            const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
            if (lane == 0u) vis_mask[blk * 4u + wave] = 0ull;
            if ((lane & 31u) == 0u) vis32[i >> 5] = make_uint2(0u, blk * 256u);
            if (threadIdx.x == 0u) block_any[blk] = 0;
            return;
        }
    }
    bool visible = false;
    SplatRec rec;
    rec.cx = rec.cy = rec.ax = rec.ay = rec.bx = rec.by = 0.0f;
    rec.c0 = rec.c1 = 0u;
    uint2 rect = make_uint2(RECT_EMPTY_LO, 0u);
    float zwin = 0.0f;          // window-space depth of the centre = of the whole quad (SplatMaterial3D.js:206-210)

    if (i < pp.count) {
        const float* MV = pp.view;
        const float* P = pp.proj;
        const float c0 = mp.px[i], c1 = mp.py[i], c2 = mp.pz[i];
        // A full-frame draw fetches the covariance and the colour WITH the centre: Morton blocks are all-or-nothing (C3: 1.44 M
        // visible splats in 5.6 k live blocks of 256), so in a block that survived the block test nearly every lane needs them,
        // and the visible path loses one of its three dependent memory round trips.  A rank's strip of a multi-GPU draw keeps the
        // lazy fetch: there the per-splat pre-test exists precisely to leave the covariance of the other strips' splats unread.
        const bool early = GS_COV_EARLY && !(pp.row_begin > 0u || pp.row_end < pp.tiles_y);
        uint32_t e_rgba = 0;
        uint4 e_a = make_uint4(0u, 0u, 0u, 0u);
        uint2 e_b = make_uint2(0u, 0u);
        ShPre pre;
        pre.have = false;
        if (early) {
            if (GS_SH_EARLY == 2 && !EXT && pp.sh_stored >= 2 && pp.sh_degree >= 1) {   // (A/B: the SH planes with the centre as well)
                pre.a = mp.sh0[i];
                pre.b = reinterpret_cast<const uint4*>(mp.sh1)[i];
                pre.c = mp.sh2[i];
                pre.have = true;
            }
            e_rgba = mp.rgba[i];
            if (pp.cov_half) {
                const uint2 a = reinterpret_cast<const uint2*>(mp.covA)[i];
                e_a.x = a.x; e_a.y = a.y;
                e_b.x = reinterpret_cast<const uint32_t*>(mp.covB)[i];
            } else {
                e_a = reinterpret_cast<const uint4*>(mp.covA)[i];
                e_b = reinterpret_cast<const uint2*>(mp.covB)[i];
            }
        }
        uint32_t scene = 0;
        float opacity_from_scene = 1.0f;
        bool scene_ok = true;
        float MVd[16];
        if (EXT) {
            if (pp.scene_count > 1) scene = mp.scene_idx[i];                          // SplatMaterial.js:123-126
            if (pp.flags & GS_CAM_SCENE_EFFECTS) {                                    // :129-137
                opacity_from_scene = mp.scenes->opacity[scene];
                scene_ok = !(opacity_from_scene <= 0.01f || mp.scenes->visible[scene] == 0u);
            }
            if (pp.flags & GS_CAM_DYNAMIC) {                                          // :140-144 viewMatrix * transform
                const float* A = pp.view_matrix;
                const float* B = mp.scenes->transforms[scene];
#pragma unroll
                for (int col = 0; col < 4; col++)
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        MVd[4 * col + r] = A[r] * B[4 * col] + A[4 + r] * B[4 * col + 1] + A[8 + r] * B[4 * col + 2] +
                                           A[12 + r] * B[4 * col + 3];
                MV = MVd;
            }
        }
        // SplatMaterial.js:156,158
        float v[4], q[4];
#pragma unroll
        for (int r = 0; r < 4; r++) v[r] = MV[r] * c0 + MV[4 + r] * c1 + MV[8 + r] * c2 + MV[12 + r];
#pragma unroll
        for (int r = 0; r < 4; r++) q[r] = P[r] * v[0] + P[4 + r] * v[1] + P[8 + r] * v[2] + P[12 + r] * v[3];
        const float clip = 1.2f * q[3];                                               // :160-164
        bool ok = scene_ok && !(q[2] < -clip || q[0] < -clip || q[0] > clip || q[1] < -clip || q[1] > clip);
        const float ndcx = q[0] / q[3], ndcy = q[1] / q[3], ndcz = q[2] / q[3];       // :166
        ok = ok && (ndcz >= -1.0f && ndcz <= 1.0f);       // quad z == centre z (SplatMaterial3D.js:209): GL clip
        // what the depth test of a draw with a destination compares (gs_mesh_set_destination): glDepthRange(0, 1), and for a
        // fixed-point depth buffer the value it would be converted to
        if (DEPTH) {
            zwin = ndcz * 0.5f + 0.5f;
            if (pp.depth_mode == 2u) zwin = (float)floor((double)zwin * 16777215.0 + 0.5);   // (fp64: exact round to nearest; < 2^24, exact as a float)
        }
        if (pp.row_begin > 0u || pp.row_end < pp.tiles_y) {
            // a rank's strip of a multi-GPU draw: no splat reaches farther than maxScreenSpaceSplatSize from its centre, so one
            // whose centre is farther than that from the strip is dropped before its covariance is fetched (the exact rect
            // clip below decides the rest; this only saves the reads)
            const float ks = fabsf(pp.splat_scale * pp.inv_focal_adj);
            float reach = pp.max_splat_px * ks * 1.001f + 2.0f;
            // ... and usually far less.  The quad's vertical half-extent squared is 8 k^2 (l1 e1y^2 + l2 e2y^2): with the exact
            // eigen pairs of cov2D that is 8 k^2 d, and when the shader floors the discriminant at 0.1 it is at most
            // 8 k^2 l1 = 8 k^2 ((a + d) / 2 + sqrt(0.1)); both are <= 8 k^2 (max(a, d) + 0.3163).  a = T0'VT0 + kernel,
            // d = T1'VT1 + kernel, x'Vx <= rho(V) |x|^2, T0 = j00 row0 + j20 row2 and T1 = j11 row1 + j21 row2 of
            // mat3(modelView).  rho(V) is bounded by a 4-byte plane written at upload, so the covariance of a splat that
            // cannot reach the strip is never read (the exact rect clip below still decides everything that passes).
            if (ok && !(EXT && (pp.flags & GS_CAM_DYNAMIC))) {
                float j00, j20, j11, j21;
                if (EXT && (pp.flags & GS_CAM_ORTHOGRAPHIC)) {
                    j00 = pp.ortho_zoom; j11 = pp.ortho_zoom; j20 = 0.0f; j21 = 0.0f;
                } else {
                    const float s = 1.0f / (v[2] * v[2]);
                    j00 = pp.focal_x / v[2]; j20 = (pp.focal_x * v[0]) * s;
                    j11 = pp.focal_y / v[2]; j21 = (pp.focal_y * v[1]) * s;
                }
                const float t0 = fabsf(j00) * pp.mv_row_norm[0] + fabsf(j20) * pp.mv_row_norm[2];
                const float t1 = fabsf(j11) * pp.mv_row_norm[1] + fabsf(j21) * pp.mv_row_norm[2];
                const float t = fmaxf(t0, t1) * 1.0001f;
                float l = mp.cov_bound[i] * t * t + pp.kernel2d + 0.3163f;
                if (pp.flags & GS_CAM_POINT_CLOUD) l = fmaxf(l, 0.2f);
                const float tight = ks * sqrtf(8.0f * l) * 1.001f + 2.0f;
                if (tight < reach) reach = tight;                     // false for NaN: the cap stays
            }
            const float cyc = (ndcy * 0.5f + 0.5f) * pp.height;
            ok = ok && !(cyc + reach < (float)(pp.row_begin * GS_TILE) || cyc - reach > (float)(pp.row_end * GS_TILE));
        }

        if (ok) {
            const uint32_t packed = early ? e_rgba : mp.rgba[i];
            if (GS_SH_EARLY && !pre.have && !EXT && pp.sh_stored >= 2 && pp.sh_degree >= 1) {     // static fp16 SH-2 scene (the benchmark path)
                pre.a = mp.sh0[i];
                pre.b = reinterpret_cast<const uint4*>(mp.sh1)[i];
                pre.c = mp.sh2[i];
                pre.have = true;
            }
            float col[3] = {(float)(packed & 255u) * (1.0f / 255.0f), (float)((packed >> 8) & 255u) * (1.0f / 255.0f),
                            (float)((packed >> 16) & 255u) * (1.0f / 255.0f)};                 // :169
            float alpha = (float)(packed >> 24) * (1.0f / 255.0f);


            // SplatMaterial3D.js:87-109 covariance fetch
            float V00, V01, V02, V11, V12, V22;
            if (pp.cov_half) {
                uint2 a = make_uint2(e_a.x, e_a.y);
                uint32_t b = e_b.x;
                if (!early) {
                    a = reinterpret_cast<const uint2*>(mp.covA)[i];
                    b = reinterpret_cast<const uint32_t*>(mp.covB)[i];
                }
                V00 = h2f(a.x); V01 = h2f(a.x >> 16); V02 = h2f(a.y); V11 = h2f(a.y >> 16); V12 = h2f(b); V22 = h2f(b >> 16);
            } else {
                uint4 a = e_a;
                uint2 b = e_b;
                if (!early) {
                    a = reinterpret_cast<const uint4*>(mp.covA)[i];
                    b = reinterpret_cast<const uint2*>(mp.covB)[i];
                }
                V00 = __uint_as_float(a.x); V01 = __uint_as_float(a.y); V02 = __uint_as_float(a.z); V11 = __uint_as_float(a.w);
                V12 = __uint_as_float(b.x); V22 = __uint_as_float(b.y);
            }
            // :120-134  J, W = transpose(mat3(MV)), T = W*J, cov2D = T^T * Vrk * T
            float j00, j20, j11, j21;
            if (EXT && (pp.flags & GS_CAM_ORTHOGRAPHIC)) {       // SplatMaterial3D.js:112-117: J = diag(zoom, zoom, 0)
                j00 = pp.ortho_zoom; j11 = pp.ortho_zoom; j20 = 0.0f; j21 = 0.0f;
            } else {
                const float s = 1.0f / (v[2] * v[2]);
                j00 = pp.focal_x / v[2]; j20 = -(pp.focal_x * v[0]) * s;
                j11 = pp.focal_y / v[2]; j21 = -(pp.focal_y * v[1]) * s;
            }
            float T0[3], T1[3];
#pragma unroll
            for (int r = 0; r < 3; r++) {
                const float w0 = MV[4 * r + 0], w1 = MV[4 * r + 1], w2 = MV[4 * r + 2];
                T0[r] = w0 * j00 + w2 * j20;
                T1[r] = w1 * j11 + w2 * j21;
            }
            const float VT0[3] = {V00 * T0[0] + V01 * T0[1] + V02 * T0[2], V01 * T0[0] + V11 * T0[1] + V12 * T0[2],
                                  V02 * T0[0] + V12 * T0[1] + V22 * T0[2]};
            const float VT1[3] = {V00 * T1[0] + V01 * T1[1] + V02 * T1[2], V01 * T1[0] + V11 * T1[1] + V12 * T1[2],
                                  V02 * T1[0] + V12 * T1[1] + V22 * T1[2]};
            float a = T0[0] * VT0[0] + T0[1] * VT0[1] + T0[2] * VT0[2];
            const float bb = T0[0] * VT1[0] + T0[1] * VT1[1] + T0[2] * VT1[2];
            float d = T1[0] * VT1[0] + T1[1] * VT1[1] + T1[2] * VT1[2];
            if (pp.flags & GS_CAM_ANTIALIASED) {                                                    // :137-144
                const float det0 = a * d - bb * bb;
                a += pp.kernel2d; d += pp.kernel2d;
                const float det1 = a * d - bb * bb;
                const float ratio = det0 / det1;
                alpha *= sqrtf(ratio > 0.0f ? ratio : 0.0f);
                if (alpha < (1.0f / 255.0f)) ok = false;
            } else {                                                                                // :147-150
                a += pp.kernel2d; d += pp.kernel2d;
            }
            // :174-196 eigen decomposition and basis
            const float D = a * d - bb * bb;
            const float half_tr = 0.5f * (a + d);
            const float disc = half_tr * half_tr - D;
            const float term2 = sqrtf(disc > 0.1f ? disc : 0.1f);
            float l1 = half_tr + term2, l2 = half_tr - term2;
            if (pp.flags & GS_CAM_POINT_CLOUD) l1 = l2 = 0.2f;
            if (l2 <= 0.0f) ok = false;
            const float ex = bb, ey = l1 - a;
            const float elen = sqrtf(ex * ex + ey * ey);
            const float e1x = ex / elen, e1y = ey / elen;
            if (!(e1x == e1x) || !(e1y == e1y)) ok = false;          // normalize(vec2(0)) -> NaN -> nothing drawn
            const float e2x = e1y, e2y = -e1x;
            const float sqrt8 = sqrtf(8.0f);
            float h1 = sqrt8 * sqrtf(l1); if (h1 > pp.max_splat_px) h1 = pp.max_splat_px;
            float h2 = sqrt8 * sqrtf(l2); if (h2 > pp.max_splat_px) h2 = pp.max_splat_px;
            if (EXT) {
                if (pp.flags & GS_CAM_SCENE_EFFECTS) alpha *= opacity_from_scene;          // SplatMaterial3D.js:199-203
                if (pp.flags & GS_CAM_FADE_IN) {                                           // SplatMaterial.js:347-363
                    const float fx_ = c0 - pp.scene_center[0], fy_ = c1 - pp.scene_center[1], fz_ = c2 - pp.scene_center[2];
                    const float center_dist = sqrtf(fx_ * fx_ + fy_ * fy_ + fz_ * fz_);
                    float f = center_dist < pp.fade_start ? 0.0f : 1.0f;                   // step(edge, x)
                    float t = (center_dist - pp.fade_start) / 0.75f;
                    t = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);
                    f = (1.0f - f) + (1.0f - t) * f;
                    alpha *= 1.0f * f;
                }
            }
            const float k = pp.splat_scale * pp.inv_focal_adj;       // pixel offset = (q.x*b1 + q.y*b2)*invFocalAdj
            const float b1x = e1x * k * h1, b1y = e1y * k * h1;
            const float b2x = e2x * k * h2, b2y = e2y * k * h2;
            const float cx = (ndcx * 0.5f + 0.5f) * pp.width;
            const float cy = (ndcy * 0.5f + 0.5f) * pp.height;

            if (ok) {
                // conservative pixel bounds of the ellipse {c + qx*b1 + qy*b2 : |q| <= 1}  (A <= 8)
                const float ext_x = sqrtf(b1x * b1x + b2x * b2x) * 1.00001f + 1e-3f;
                const float ext_y = sqrtf(b1y * b1y + b2y * b2y) * 1.00001f + 1e-3f;
                float fx0 = ceilf(cx - ext_x - 0.5f), fx1 = floorf(cx + ext_x - 0.5f);
                float fy0 = ceilf(cy - ext_y - 0.5f), fy1 = floorf(cy + ext_y - 0.5f);
                const float ymin = (float)(pp.row_begin * GS_TILE);
                const float ymax = fminf(pp.height, (float)(pp.row_end * GS_TILE)) - 1.0f;
                fx0 = fmaxf(fx0, 0.0f); fx1 = fminf(fx1, pp.width - 1.0f);
                fy0 = fmaxf(fy0, ymin); fy1 = fminf(fy1, ymax);
                if (fx0 <= fx1 && fy0 <= fy1) {           // false for NaN as well
                    visible = true;
                    const uint32_t tx0 = (uint32_t)fx0 / GS_TILE, tx1 = (uint32_t)fx1 / GS_TILE;
                    const uint32_t ty0 = (uint32_t)fy0 / GS_TILE, ty1 = (uint32_t)fy1 / GS_TILE;
                    rect = make_uint2(tx0 | (ty0 << 16), tx1 | (ty1 << 16));
                    const float n1 = b1x * b1x + b1y * b1y, n2 = b2x * b2x + b2y * b2y;
                    rec.cx = cx; rec.cy = cy;
                    rec.ax = GS_K_POWER * (b1x / n1); rec.ay = GS_K_POWER * (b1y / n1);
                    rec.bx = GS_K_POWER * (b2x / n2); rec.by = GS_K_POWER * (b2y / n2);
                    sh_colour<EXT>(pp, mp, i, scene, c0, c1, c2, col, pre);
                    rec.c0 = unorm16(col[0]) | (unorm16(col[1]) << 16);
                    rec.c1 = unorm16(col[2]) | (unorm16(alpha) << 16);
                }
            }
        }
    }
    // Survivors are COMPACTED inside their 256-splat block: splat i lands in slot (block base) + (number of visible splats of
    // the block before it).  A wave's survivors therefore write consecutive 32-byte records (whole cache lines instead of
    // scattered 32-byte sectors: -20 % kernel time at 33 % visibility); consumers get the slot from vis32 below.
    // Tried and dropped in round 2 (all pixel-identical, all slower):
    //  - compacting ACROSS blocks as well (one dense array through a single-pass chained scan with agent-scope look-back
    //    words): the scan chain made this kernel 54 -> 172 us and the binner's gathers gained nothing;
    //  - a cull phase over 4 splats per lane, then the rest of the shader on LDS-compacted dense lanes: 55 -> 65 us (Morton
    //    order already makes the lanes of a wave pass or fail together; the barriers and the recomputation are pure cost);
    //  - one wave per 256-splat block in four rounds, centres of all rounds fetched up front, no LDS / barriers: 55 -> 64 us
    //    (71 VGPRs -> 7 waves per SIMD, and a wave's four rounds run back to back instead of on four waves at once).
    // 252 MB in 55 us is 4.6 TB/s of mixed read / write traffic against the 6.3 TB/s a pure copy reaches.
    __shared__ uint32_t s_cnt[4];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned long long vis = __ballot(visible);
    if (lane == 0u) {
        s_cnt[wave] = (uint32_t)__popcll(vis);
        vis_mask[i >> 6] = vis;                      // the buffer covers whole blocks
    }
    __syncthreads();
    // one byte per block: does ANY of its 256 splats reach the frame?  Morton order makes most blocks all-or-nothing, and the
    // binner tests this (from LDS) before it spends an L2 gather on a splat's visibility word
    if (threadIdx.x == 0) block_any[blk] = (s_cnt[0] | s_cnt[1] | s_cnt[2] | s_cnt[3]) ? 1 : 0;
    const uint32_t block_base = blk * 256u;
    uint32_t wave_base = block_base;
#pragma unroll
    for (uint32_t w = 0; w < 3; w++) wave_base += (w < wave) ? s_cnt[w] : 0u;
    // the binner's look-up table: per 32 splats one 8-byte record {mask, slot of the first visible one} - a single
    // 8-byte load per sorted index there (1.45 MB for 5.8 M splats, L2 resident)
    if ((lane & 31u) == 0u) {
        const uint32_t lo = (uint32_t)vis, hi = (uint32_t)(vis >> 32);
        vis32[i >> 5] = lane == 0u ? make_uint2(lo, wave_base) : make_uint2(hi, wave_base + (uint32_t)__popc(lo));
    }
    // The binner's look-up, one 8-byte word per splat POSITION of a live block: {x0:12 | y0:12 | slot in the block:8,
    // x1:12 | y1:12 | visible:1} - visibility, record slot and tile rect in ONE gather per sorted index (r02: a visibility word
    // and then the rect, two dependent 64-byte sectors per visible splat: 2.2 GB of the 6.0 GB a C4 frame moved).  Dead blocks
    // write nothing: the binner never looks past their block_any byte.
    {
        const uint32_t slot = wave_base + (uint32_t)__popcll(vis & ((1ull << lane) - 1ull));
        const uint32_t x0 = rect.x & 0xFFFFu, y0 = rect.x >> 16, x1 = rect.y & 0xFFFFu, y1 = rect.y >> 16;
        if (i < pp.count)
            prect[i] = visible ? make_uint2(x0 | (y0 << 12) | ((slot - block_base) << 24), x1 | (y1 << 12) | (1u << 24)) : make_uint2(0u, 0u);
    }
    if (visible) {
        const uint32_t slot = wave_base + (uint32_t)__popcll(vis & ((1ull << lane) - 1ull));
        recs[slot] = rec;
        rects[slot] = rect;
        if (DEPTH) zrec[slot] = zwin;
        // gs_mesh_project: the same bit by ORIGINAL splat index, for a sort that keeps only what this frame draws (the mask
        // was zeroed before the launch; only survivors pay the atomic)
        if (vis_orig) {
            const uint32_t orig = inv_perm ? inv_perm[i] : i;
            atomicOr(&vis_orig[orig >> 5], 1u << (orig & 31u));
        }
    }
}

// The vertex stage's launch (the modes are named at the kernel below): k_block_test has decided and a workgroup whose block is dead
// leaves on one byte - or, for $GSPLAT_NO_BLOCK_LIST, workgroup b tests storage block b itself as in rounds 2-4.
// Measured and dropped (r05b/c, project_floor / ab_libs, same box): (a) ONE kernel whose workgroups loop over a list of the live
// blocks behind a grid sized from an earlier draw's count - the loop costs 80 instead of 48 VGPRs, C3 51.8 -> 60.0 us; (b) the
// same grid bound with a separate 64-workgroup tail launch for the entries beyond it - 52.0 -> 54.3 us, and the floor (a pose
// that sees next to nothing) 16.7 -> 17.5 us: the empty workgroups of a full grid are NOT what the floor is made of (the kernel
// boundaries and the test kernel are), so the full grid stays.
// amdgpu_num_sgpr(80): left alone the compiler takes 106 scalar registers, and MI355X admits 256-thread workgroups per CU by
// floor(800 / (ceil(sgpr / 16) * 16 + 16)) - 6 at 106, 7 at 86 (round 4's kernel), 8 at <= 80 (MI355X_MICROARCH.md, residency).
// MODE 0: every workgroup tests its own block (rounds 2-4; a full-frame draw of a scene that is 55-90 % in view, or
// $GSPLAT_NO_BLOCK_LIST);  1: k_block_test has decided (the default, any strip, $GSPLAT_BLOCK_TEST_ALWAYS);  2: no block test at all
// (per-scene transforms, or a full-frame draw of a scene that was > 90 % in view at its last full-frame draw: the test finds nothing
// to drop there, and its code costs the all-visible launch scalar registers - C4 250 -> 278 us with it compiled in, r05m).  The
// rule and its measurements: gs_launch_project below.
template <bool EXT, int MODE, bool DEPTH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80))) void k_project(ProjectParams pp, MeshPlanes mp, SplatRec* __restrict__ recs,
                                                 uint2* __restrict__ rects, unsigned long long* __restrict__ vis_mask,
                                                 uint2* __restrict__ vis32, uint32_t* __restrict__ vis_orig,
                                                 const uint32_t* __restrict__ inv_perm, uint8_t* __restrict__ block_any,
                                                 uint2* __restrict__ prect, float* __restrict__ zrec) {
    if (MODE == 1) {
        if (block_any[blockIdx.x] == 0) return;
        project_block<EXT, false, DEPTH>(pp, mp, blockIdx.x, recs, rects, vis_mask, vis32, vis_orig, inv_perm, block_any, prect, zrec);
    } else if (MODE == 2) {
        project_block<EXT, false, DEPTH>(pp, mp, blockIdx.x, recs, rects, vis_mask, vis32, vis_orig, inv_perm, block_any, prect, zrec);
    } else {
        project_block<EXT, true, DEPTH>(pp, mp, blockIdx.x, recs, rects, vis_mask, vis32, vis_orig, inv_perm, block_any, prect, zrec);
    }
}

int gs_launch_project(gs_mesh* m, const ProjectParams& pp, bool orig_mask, hipEvent_t ev_before, hipEvent_t ev_after, bool whole_stage) {
    MeshPlanes mp;
    mp.px = m->px.as<float>(); mp.py = m->py.as<float>(); mp.pz = m->pz.as<float>();
    mp.covA = m->covA.p; mp.covB = m->covB.p;
    mp.cov_bound = m->cov_bound.as<float>();
    mp.block_box = m->block_box.as<float>();
    mp.rgba = m->rgba.as<uint32_t>();
    mp.sh0 = m->sh0.as<uint4>(); mp.sh1 = m->sh1.p; mp.sh2 = m->sh2.as<uint4>();
    mp.scene_idx = m->scene_idx.as<uint32_t>();
    mp.scenes = m->scene_dev.as<gs_scene_params>();
    if (pp.count == 0) {
        if (ev_before) GS_HIP(hipEventRecord(ev_before, m->ctx->aux));
        if (ev_after) GS_HIP(hipEventRecord(ev_after, m->ctx->aux));
        return GS_OK;
    }
    const uint32_t blocks = (pp.count + 255u) / 256u;
    // Where the block test runs, by the share of the splats that were in view at the mesh's last MEASURED full-frame draw (same box,
    // tools/ab_libs.py, vertex stage): the separate k_block_test pays a launch (~5 us) to spare every dead block its workgroup's set-up,
    // so it wins where most blocks are dead (C3, V/N 0.25: 58.0 -> 52.5 us; C2, 0.48: 35.0 against 35.9 inside the workgroups; any strip);
    // a scene that is mostly in view keeps the test inside each workgroup (C3S, 0.65: 95.3 there vs 107.7 with no test at all); one
    // that is all in view runs no test (C4, 0.998: 243.2 -> 219.2 us - the test's registers cost k_project a wave per SIMD).
    // (profiles/r05z_ab_r03_vs_this_tree.txt, r05p_ab.txt)
    const bool measured = m->measured_count > 0u;
    const bool all_live = measured && (uint64_t)m->measured_visible * 10u > (uint64_t)m->measured_count * 9u;
    const bool half_live = measured && !all_live && (uint64_t)m->measured_visible * 20u > (uint64_t)m->measured_count * 11u;
    const bool strip = pp.row_begin > 0u || pp.row_end < pp.tiles_y;     // (a strip of a scene that is mostly in view still drops most blocks)
    const bool test = pp.block_cull && (!all_live || strip || m->block_test_always);
    const bool pretest = test && !m->no_block_list && (!half_live || strip || m->block_test_always);
    const int mode = pretest ? 1 : (test ? 0 : 2);
    m->last_project_mode = (uint32_t)mode;
    if (ev_before && whole_stage) GS_HIP(hipEventRecord(ev_before, m->ctx->aux));
    if (pretest)
        hipLaunchKernelGGL(k_block_test, dim3((blocks + 31u) / 32u), dim3(256), 0, m->ctx->aux, pp, mp.block_box, blocks,
                           m->vis_mask.as<unsigned long long>(), m->vis32.as<uint2>(), m->block_any.as<uint8_t>());
    if (pp.depth_mode) GS_TRY(m->zrec.ensure((size_t)m->max_count * 4 + 16));
    uint32_t* vis_orig = nullptr;
    if (orig_mask) {
        const bool fresh = m->vis_orig.p == nullptr;
        GS_TRY(m->vis_orig.ensure(((size_t)m->max_count + 31) / 32 * 4 + 64));
        // the visibility-culled sort that consumes the mask leaves it zeroed (k_mask_compact); only a mask nobody consumed
        // (two gs_mesh_project in a row, a sort over fewer splats) is cleared here
        if (fresh || m->vis_orig_dirty)
            GS_HIP(hipMemsetAsync(m->vis_orig.p, 0, ((size_t)m->max_count + 31) / 32 * 4, m->ctx->aux));
        m->vis_orig_dirty = true;
        m->vis_orig_count = pp.count;
        vis_orig = m->vis_orig.as<uint32_t>();
    }
    const uint32_t* inv_perm = m->reorder ? m->inv_perm.as<uint32_t>() : nullptr;
    const bool ext = pp.sh_u8 || pp.scene_count > 1 ||
                     (pp.flags & (GS_CAM_ORTHOGRAPHIC | GS_CAM_FADE_IN | GS_CAM_SCENE_EFFECTS | GS_CAM_DYNAMIC));
    auto launch = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, m->ctx->aux, pp, mp, m->recs.as<SplatRec>(), m->rects.as<uint2>(),
                           m->vis_mask.as<unsigned long long>(), m->vis32.as<uint2>(), vis_orig, inv_perm, m->block_any.as<uint8_t>(),
                           m->prect.as<uint2>(), m->zrec.as<float>());
    };
    const bool depth = pp.depth_mode != 0u;
    if (ev_before && !whole_stage) GS_HIP(hipEventRecord(ev_before, m->ctx->aux));
#define GS_PROJECT_LAUNCH(E, D)                                                                                  \
    do {                                                                                                         \
        if (mode == 1) launch(k_project<E, 1, D>);                                                               \
        else if (mode == 2) launch(k_project<E, 2, D>);                                                          \
        else launch(k_project<E, 0, D>);                                                                         \
    } while (0)
    if (ext && depth) GS_PROJECT_LAUNCH(true, true);
    else if (ext) GS_PROJECT_LAUNCH(true, false);
    else if (depth) GS_PROJECT_LAUNCH(false, true);
    else GS_PROJECT_LAUNCH(false, false);
#undef GS_PROJECT_LAUNCH
    if (ev_after) GS_HIP(hipEventRecord(ev_after, m->ctx->aux));
    GS_HIP(hipGetLastError());
    return GS_OK;
}
