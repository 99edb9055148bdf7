/*
 * oracle/raster_oracle.c — TEST INFRASTRUCTURE ONLY (never linked into or called by the product path).
 *
 * fp32 CPU restatement of the reference's splat rasterisation = GLSL vertex + fragment shader + GL blend:
 *   vertex, common half   /root/reference/src/splatmesh/SplatMaterial.js:112-341
 *   vertex, 3D half       /root/reference/src/splatmesh/SplatMaterial3D.js:81-217
 *   fragment              /root/reference/src/splatmesh/SplatMaterial3D.js:235-251
 *   blend / target        /root/reference/src/splatmesh/SplatMaterial3D.js:65-75, src/Viewer.js:358-359
 *   uniforms              /root/reference/src/splatmesh/SplatMesh.js:1248-1280, src/Viewer.js:651-677
 *   quad                  /root/reference/src/splatmesh/SplatGeometry.js:14-23
 *
 * Parity status: PINNED since round 2.  The reference has no tests, goldens or CPU rasteriser and WebGL cannot run
 * here, so its own shader text is executed instead: oracle/make_golden_raster.py takes the GLSL that
 * SplatMaterial3D.build() returns (7 permutations), rewrites tokens only, compiles it against oracle/glsl_shim.hpp
 * (-O1 -ffp-contract=off) and records gl_Position / vColor / vPosition for 12 seeded cases and the fragment rule for 1505
 * fragments into tests/golden/raster_ref.npz; tests/test_raster_ref.py compares project_one / blend_one below (and the HIP
 * vertex stage) with them.  This file is a line-by-line arithmetic restatement of the shader source in IEEE fp32
 * (compiled with -ffp-contract=off).  What stays unobservable from source is the order of operations inside a GPU's ROPs
 * (RGBA8 target): the rop8 mode below emulates it and DESIGN.md section 2 reports the gap.
 *
 * Conventions: matrices are column-major float[16] like three.js / GLSL; framebuffer row 0 is the
 * BOTTOM row (GL window coordinates); pixel (x,y) is sampled at its centre (x+0.5, y+0.5).
 * Splats are composited in the order given (index 0 drawn first = farthest), NormalBlending:
 *   rgb = a*src + (1-a)*rgb ; alpha = a + (1-a)*alpha      (three r160 blendFuncSeparate)
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

typedef struct {
    float view[16];        /* modelViewMatrix = view * meshWorld                         */
    float proj[16];        /* projectionMatrix                                          */
    float cam_pos[3];      /* cameraPosition (world)                                    */
    float focal[2];        /* SplatMesh.updateUniforms: proj[0]*0.5*W , proj[5]*0.5*H   */
    float viewport[2];     /* W, H in pixels                                            */
    float splat_scale;     /* 1                                                         */
    float kernel2d;        /* 0.3                                                       */
    float max_splat_px;    /* 1024                                                      */
    float inv_focal_adj;   /* 1                                                         */
    int32_t sh_degree;     /* degree evaluated (0..2), <= degree stored                 */
    int32_t sh_stored;     /* degree stored: 0,1,2 -> 0,9,24 floats per splat           */
    int32_t antialiased;   /* 0                                                         */
    int32_t point_cloud;   /* 0                                                         */
    /* --- shader permutations (all off = the static perspective path above) ---------------------------- */
    int32_t orthographic;  /* orthographicMode, SplatMaterial3D.js:112-117                 */
    float ortho_zoom;      /* orthoZoom                                                   */
    int32_t fade_in;       /* fadeInComplete == 0, SplatMaterial.js:347-363               */
    float scene_center[3];
    float fade_start;      /* visibleRegionFadeStartRadius                                */
    int32_t effects;       /* enableOptionalEffects: sceneOpacity / sceneVisibility       */
    int32_t dynamic;       /* dynamicMode: transformModelViewMatrix = viewMatrix * transforms[scene] */
    int32_t sh8;           /* sphericalHarmonics8BitMode: sh = (u8/255)*range + min       */
    int32_t scene_count;
    float view_matrix[16];              /* viewMatrix (dynamic mode; `view` is unused then)                   */
    float transforms[32][16];
    float inv_cam_pos[32][3];           /* inverse(transform) * cameraPosition, precomputed by the host in fp64 */
    float opacity[32];
    int32_t visible[32];
    float sh8_min[32], sh8_max[32];
} gro_camera;

/* Per-splat result of the vertex stage. */
typedef struct {
    int32_t visible;       /* 0 = rejected (clip / eigen / NaN)                          */
    float cx, cy;          /* centre in pixels, GL window coords                         */
    float b1x, b1y;        /* basisVector1 in pixels (quad half-axis)                    */
    float b2x, b2y;        /* basisVector2 in pixels                                     */
    float r, g, b, a;      /* vColor                                                    */
    float ndcz;
} gro_splat2d;

static float clamp01(float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); }

/* Vertex stage for one splat.  Every line cites the shader line it restates. */
static void project_one(const gro_camera* cam, const float* c, const float* cov, const uint8_t* rgba,
                        const float* sh_in, uint32_t scene, gro_splat2d* o) {
    const float* MV = cam->view;
    const float* P = cam->proj;
    memset(o, 0, sizeof(*o));
    if (cam->scene_count <= 1) scene = 0;                      /* SplatMaterial.js:123-126 */
    float opacity_from_scene = 1.0f;
    if (cam->effects) {                                        /* :129-137 */
        opacity_from_scene = cam->opacity[scene];
        if (opacity_from_scene <= 0.01f || cam->visible[scene] == 0) return;
    }
    float MVd[16];
    if (cam->dynamic) {                                        /* :140-144 viewMatrix * transform */
        const float* A = cam->view_matrix;
        const float* B = cam->transforms[scene];
        for (int col = 0; col < 4; col++)
            for (int r = 0; r < 4; r++)
                MVd[4 * col + r] = A[r] * B[4 * col] + A[4 + r] * B[4 * col + 1] + A[8 + r] * B[4 * col + 2] + A[12 + r] * B[4 * col + 3];
        MV = MVd;
    }
    /* :150-154 8-bit SH: texel (unorm8 -> v/255) * range + min, with range = max - min */
    float shbuf[24];
    const float* sh = sh_in;
    if (cam->sh8 && sh_in) {
        const float range = cam->sh8_max[scene] - cam->sh8_min[scene];
        const int shn = cam->sh_stored == 0 ? 0 : (cam->sh_stored == 1 ? 9 : 24);
        for (int k = 0; k < shn; k++) shbuf[k] = (sh_in[k] / 255.0f) * range + cam->sh8_min[scene];
        sh = shbuf;
    }

    /* SplatMaterial.js:156  viewCenter = MV * vec4(c,1) */
    float v[4], q[4];
    for (int r = 0; r < 4; r++) v[r] = MV[r] * c[0] + MV[4 + r] * c[1] + MV[8 + r] * c[2] + MV[12 + r];
    /* :158 clipCenter = P * viewCenter */
    for (int r = 0; r < 4; r++) q[r] = P[r] * v[0] + P[4 + r] * v[1] + P[8 + r] * v[2] + P[12 + r] * v[3];
    /* :160-164 1.2x frustum reject */
    const float clip = 1.2f * q[3];
    if (q[2] < -clip || q[0] < -clip || q[0] > clip || q[1] < -clip || q[1] > clip) return;
    /* :166 */
    const float ndcx = q[0] / q[3], ndcy = q[1] / q[3], ndcz = q[2] / q[3];
    /* quad z == centre z (SplatMaterial3D.js:209) -> GL clips the whole quad on ndc z */
    if (!(ndcz >= -1.0f && ndcz <= 1.0f)) return;

    /* :169 vColor = rgba/255 */
    float col[3] = {(float)rgba[0] * (1.0f / 255.0f), (float)rgba[1] * (1.0f / 255.0f),
                    (float)rgba[2] * (1.0f / 255.0f)};
    float alpha = (float)rgba[3] * (1.0f / 255.0f);

    if (cam->sh_stored >= 1 && cam->sh_degree >= 1) {
        /* :185 worldViewDir = normalize(splatCenter - cameraPosition); dynamic mode (:179-183): the camera position
         * in the scene's frame, inverse(transform) * cameraPosition */
        const float* cp = cam->dynamic ? cam->inv_cam_pos[scene] : cam->cam_pos;
        float d[3] = {c[0] - cp[0], c[1] - cp[1], c[2] - cp[2]};
        const float inv = 1.0f / sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        const float x = d[0] * inv, y = d[1] * inv, z = d[2] * inv;
        const float SH_C1 = 0.4886025119029199f;
        /* :273  sh1..sh3 are RGB triples, coefficient-major */
        for (int ch = 0; ch < 3; ch++)
            col[ch] += SH_C1 * (-sh[0 + ch] * y + sh[3 + ch] * z - sh[6 + ch] * x);
        if (cam->sh_stored >= 2 && cam->sh_degree >= 2) {
            /* :308-330 */
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            const float C0 = 1.0925484f, C1 = -1.0925484f, C2 = 0.3153916f, C3 = -1.0925484f, C4 = 0.5462742f;
            for (int ch = 0; ch < 3; ch++)
                col[ch] += (C0 * xy) * sh[9 + ch] + (C1 * yz) * sh[12 + ch] +
                           (C2 * (2.0f * zz - xx - yy)) * sh[15 + ch] + (C3 * xz) * sh[18 + ch] +
                           (C4 * (xx - yy)) * sh[21 + ch];
        }
        /* :337 */
        for (int ch = 0; ch < 3; ch++) col[ch] = clamp01(col[ch]);
    }

    /* SplatMaterial3D.js:105-109  Vrk (symmetric) */
    const float V00 = cov[0], V01 = cov[1], V02 = cov[2], V11 = cov[3], V12 = cov[4], V22 = cov[5];
    /* :120-126  J (GLSL column-major constructor) as math matrix Jm[row][col]:
     *   col0 = (fx/z, 0, -(fx*x)*s), col1 = (0, fy/z, -(fy*y)*s), col2 = 0 */
    float j00, j20, j11, j21;
    if (cam->orthographic) {                                   /* :112-117 J = diag(zoom, zoom, 0) */
        j00 = cam->ortho_zoom; j11 = cam->ortho_zoom; j20 = 0.0f; j21 = 0.0f;
    } else {
        const float s = 1.0f / (v[2] * v[2]);
        j00 = cam->focal[0] / v[2]; j20 = -(cam->focal[0] * v[0]) * s;
        j11 = cam->focal[1] / v[2]; j21 = -(cam->focal[1] * v[1]) * s;
    }
    /* :130 W = transpose(mat3(MV)) -> Wm[r][c] = MV3[c][r] = MV[4*r + c]  (MV[4*col+row]) */
    /* :131 T = W * J : T[r][c] = sum_k Wm[r][k] * Jm[k][c]; only columns 0 and 1 are non-zero */
    float T0[3], T1[3];
    for (int r = 0; r < 3; r++) {
        const float w0 = MV[4 * r + 0], w1 = MV[4 * r + 1], w2 = MV[4 * r + 2];
        T0[r] = w0 * j00 + w2 * j20;
        T1[r] = w1 * j11 + w2 * j21;
    }
    /* :134 cov2Dm = transpose(T) * Vrk * T  -> 2x2 upper-left */
    const float VT0[3] = {V00 * T0[0] + V01 * T0[1] + V02 * T0[2], V01 * T0[0] + V11 * T0[1] + V12 * T0[2],
                          V02 * T0[0] + V12 * T0[1] + V22 * T0[2]};
    const float VT1[3] = {V00 * T1[0] + V01 * T1[1] + V02 * T1[2], V01 * T1[0] + V11 * T1[1] + V12 * T1[2],
                          V02 * T1[0] + V12 * T1[1] + V22 * T1[2]};
    float a = T0[0] * VT0[0] + T0[1] * VT0[1] + T0[2] * VT0[2];
    const float bb = T0[0] * VT1[0] + T0[1] * VT1[1] + T0[2] * VT1[2];
    float d = T1[0] * VT1[0] + T1[1] * VT1[1] + T1[2] * VT1[2];

    if (cam->antialiased) {                                    /* :137-144 */
        const float det0 = a * d - bb * bb;
        a += cam->kernel2d; d += cam->kernel2d;
        const float det1 = a * d - bb * bb;
        const float ratio = det0 / det1;
        alpha *= sqrtf(ratio > 0.0f ? ratio : 0.0f);
        if (alpha < (1.0f / 255.0f)) return;
    } else {                                                   /* :147-150 */
        a += cam->kernel2d; d += cam->kernel2d;
    }
    /* :174-182 */
    const float D = a * d - bb * bb;
    const float half_tr = 0.5f * (a + d);
    const float disc = half_tr * half_tr - D;
    const float term2 = sqrtf(disc > 0.1f ? disc : 0.1f);
    float l1 = half_tr + term2, l2 = half_tr - term2;
    if (cam->point_cloud) l1 = l2 = 0.2f;                      /* :184-186 */
    if (l2 <= 0.0f) return;                                    /* :188 */
    /* :190-192  normalize(vec2(b, l1 - a)); (0,0) -> NaN -> nothing is rasterised */
    const float ex = bb, ey = l1 - a;
    const float elen = sqrtf(ex * ex + ey * ey);
    const float e1x = ex / elen, e1y = ey / elen;
    if (!(e1x == e1x) || !(e1y == e1y)) return;
    const float e2x = e1y, e2y = -e1x;
    /* :195-196 */
    const float sqrt8 = sqrtf(8.0f);
    float h1 = sqrt8 * sqrtf(l1); if (h1 > cam->max_splat_px) h1 = cam->max_splat_px;
    float h2 = sqrt8 * sqrtf(l2); if (h2 > cam->max_splat_px) h2 = cam->max_splat_px;
    if (cam->effects) alpha *= opacity_from_scene;             /* SplatMaterial3D.js:199-203 */
    if (cam->fade_in) {                                        /* SplatMaterial.js:347-363 */
        const float dx = c[0] - cam->scene_center[0], dy = c[1] - cam->scene_center[1], dz = c[2] - cam->scene_center[2];
        const float center_dist = sqrtf(dx * dx + dy * dy + dz * dz);
        const float fade_distance = 0.75f;
        float f = center_dist < cam->fade_start ? 0.0f : 1.0f;              /* step(edge, x) */
        float t = (center_dist - cam->fade_start) / fade_distance;
        t = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);
        f = (1.0f - f) + (1.0f - t) * f;
        alpha *= 1.0f * f;
    }
    /* ndcOffset = (q.x*b1 + q.y*b2) * (1/viewport) * 2 * invFocalAdj  (:206-207)  => pixel offset =
     * ndcOffset * viewport/2 = (q.x*b1 + q.y*b2) * invFocalAdj */
    const float k = cam->splat_scale * cam->inv_focal_adj;
    o->b1x = e1x * k * h1; o->b1y = e1y * k * h1;
    o->b2x = e2x * k * h2; o->b2y = e2y * k * h2;
    o->cx = (ndcx * 0.5f + 0.5f) * cam->viewport[0];
    o->cy = (ndcy * 0.5f + 0.5f) * cam->viewport[1];
    o->r = col[0]; o->g = col[1]; o->b = col[2]; o->a = alpha;
    o->ndcz = ndcz;
    o->visible = 1;
}

/* Vertex stage for splats order[0..count) (order==NULL -> identity). */
static const uint32_t* g_scene_idx = 0;     /* per-splat scene indexes for the next call (set by gro_set_scene_indexes) */
void gro_set_scene_indexes(const uint32_t* scene_idx) { g_scene_idx = scene_idx; }

void gro_project(const gro_camera* cam, const float* centers, const float* cov, const uint8_t* rgba,
                 const float* sh, const uint32_t* order, uint32_t count, gro_splat2d* out) {
    const int shn = cam->sh_stored == 0 ? 0 : (cam->sh_stored == 1 ? 9 : 24);
    for (uint32_t i = 0; i < count; i++) {
        const size_t g = order ? order[i] : i;
        project_one(cam, centers + 3 * g, cov + 6 * g, rgba + 4 * g, sh ? sh + (size_t)shn * g : NULL,
                    g_scene_idx ? g_scene_idx[g] : 0u, out + i);
    }
}

/* The DESTINATION's depth (nullable): what other scene geometry left in the depth buffer before the splats are drawn with
 * `depthTest: true, depthWrite: false` (/root/reference/src/splatmesh/SplatMaterial3D.js:72-73; draw order
 * /root/reference/src/Viewer.js:1610-1616; drop-in mode /root/reference/src/DropInViewer.js:34-42).  float [H][W], window
 * depth in [0, 1], row 0 = bottom.  The quad of a splat is flat at its centre's depth (gl_Position.z = ndcCenter.z,
 * SplatMaterial3D.js:206-210), so every fragment of a splat carries z_w = 0.5 * ndc.z + 0.5 (glDepthRange 0..1) and is kept
 * where z_w <= stored depth (three r160's default depthFunc, LessEqualDepth); nothing is written back.  unorm24 != 0: both
 * sides are first converted as a 24-bit fixed-point depth buffer stores them, round(z * (2^24 - 1)).
 * The destination COLOUR needs no state: the caller initialises fb with it instead of zeros. */
static const float* g_dst_depth = 0;
static int g_dst_unorm24 = 0;
void gro_set_destination_depth(const float* depth, int unorm24) { g_dst_depth = depth; g_dst_unorm24 = unorm24; }
/* (the conversion in fp64: round to nearest exactly - in fp32 the product already sits at 1-LSB granularity above z = 0.5; the
 * result is an integer < 2^24, exact as a float) */
static float depth_cmp_value(float z) { return g_dst_unorm24 ? (float)floor((double)z * 16777215.0 + 0.5) : z; }

/* How far an RGBA8 render target can drift from the exact composite (row a14): every blend into the target rounds each channel
 * to 8 bits (an error of at most half a step), and every later splat scales what has accumulated by (1 - alpha):
 *     e <- (1 - alpha) * e + 0.5            per kept fragment, in units of 1/255, e = 0 on the cleared target.
 * With a plane per window set here (nullable; [h][w] floats, zeroed by the caller), blend_one keeps that recursion next to the
 * composite.  The fp32 composite rounded ONCE (the engine) then differs from the per-splat-rounded one (the reference's ROPs) by
 * at most e + 0.5: a bound that grows with the composited depth of a pixel instead of a limit fitted to the scenes at hand. */
#define GRO_MAX_BOUND_PLANES 64
static float* g_err_bound[GRO_MAX_BOUND_PLANES];
static uint32_t g_err_bound_count = 0;
void gro_set_error_bound_planes(uint32_t n, float** planes) {
    g_err_bound_count = n > GRO_MAX_BOUND_PLANES ? GRO_MAX_BOUND_PLANES : n;
    for (uint32_t k = 0; k < g_err_bound_count; k++) g_err_bound[k] = planes[k];
}

/* Composites one projected splat over the window [wx0, wx0+ww) x [wy0, wy0+wh) of the W x H frame; fb / ambig are the
 * window's own [wh][ww] arrays. */
static uint64_t blend_one(const gro_splat2d* s, int W, int H, int wx0, int wy0, int ww, int wh, int rop8, float amb_eps,
                          float* fb, uint8_t* ambig, float* ebound) {
    uint64_t frags = 0;
    /* bounding box of the quad centre +- b1 +- b2 */
    const float ext_x = fabsf(s->b1x) + fabsf(s->b2x), ext_y = fabsf(s->b1y) + fabsf(s->b2y);
    float fx0 = floorf(s->cx - ext_x - 1.0f), fx1 = ceilf(s->cx + ext_x + 1.0f);
    float fy0 = floorf(s->cy - ext_y - 1.0f), fy1 = ceilf(s->cy + ext_y + 1.0f);
    const int xe = (wx0 + ww < W ? wx0 + ww : W) - 1, ye = (wy0 + wh < H ? wy0 + wh : H) - 1;
    if (fx0 < (float)wx0) fx0 = (float)wx0;
    if (fy0 < (float)wy0) fy0 = (float)wy0;
    if (fx1 > (float)xe) fx1 = (float)xe;
    if (fy1 > (float)ye) fy1 = (float)ye;
    if (!(fx0 <= fx1) || !(fy0 <= fy1)) return 0;
    const float n1 = s->b1x * s->b1x + s->b1y * s->b1y, n2 = s->b2x * s->b2x + s->b2y * s->b2y;
    const float zfrag = depth_cmp_value(s->ndcz * 0.5f + 0.5f);
    for (int py = (int)fy0; py <= (int)fy1; py++) {
        for (int px = (int)fx0; px <= (int)fx1; px++) {
            /* depthTest against the destination (the fixed-function test runs whether or not the shader discards) */
            if (g_dst_depth && !(zfrag <= depth_cmp_value(g_dst_depth[(size_t)py * W + px]))) continue;
            const float dx = ((float)px + 0.5f) - s->cx, dy = ((float)py + 0.5f) - s->cy;
            /* quad-local coordinates q in [-1,1]^2 (b1 is orthogonal to b2) */
            const float qx = (dx * s->b1x + dy * s->b1y) / n1;
            const float qy = (dx * s->b2x + dy * s->b2y) / n2;
            /* vPosition = q*sqrt8 (SplatMaterial3D.js:213); A = dot(vPosition,vPosition) (:237) */
            const float A = 8.0f * (qx * qx + qy * qy);
            const size_t at = (size_t)(py - wy0) * ww + (px - wx0);
            if (ambig && fabsf(A - 8.0f) <= amb_eps) ambig[at] = 1;
            if (!(A <= 8.0f)) continue;                    /* :242 `if (A > 8.0) discard` */
            const float al = expf(-0.5f * A) * s->a;       /* :249 */
            float* dst = fb + 4 * at;
            const float om = 1.0f - al;
            dst[0] = al * s->r + om * dst[0];
            dst[1] = al * s->g + om * dst[1];
            dst[2] = al * s->b + om * dst[2];
            dst[3] = al + om * dst[3];
            if (ebound) ebound[at] = om * ebound[at] + 0.5f;
            if (rop8)
                for (int ch = 0; ch < 4; ch++) dst[ch] = floorf(clamp01(dst[ch]) * 255.0f + 0.5f) * (1.0f / 255.0f);
            frags++;
        }
    }
    return frags;
}

/*
 * Full frame.  fb = float RGBA [H][W][4], row 0 = bottom, must be zeroed by the caller (clear colour
 * (0,0,0,0), Viewer.js:358-359).  rop8 != 0 emulates the RGBA8 render target by rounding dst to unorm8
 * after every splat (what the reference's ROP really does); rop8 == 0 keeps fp32 (the parity target).
 * ambig (nullable, [H][W] bytes): set to 1 where some splat's A fell within 8 +- amb_eps, i.e. where
 * the discontinuous `A > 8 -> discard` may legitimately flip under different fp32 evaluation orders.
 * Returns the number of (pixel, splat) fragments that survived the discard.
 */
uint64_t gro_render(const gro_camera* cam, const float* centers, const float* cov, const uint8_t* rgba,
                    const float* sh, const uint32_t* order, uint32_t count, int rop8, float amb_eps,
                    float* fb, uint8_t* ambig) {
    const int W = (int)cam->viewport[0], H = (int)cam->viewport[1];
    const int shn = cam->sh_stored == 0 ? 0 : (cam->sh_stored == 1 ? 9 : 24);
    uint64_t frags = 0;
    for (uint32_t i = 0; i < count; i++) {
        const size_t g = order ? order[i] : i;
        gro_splat2d s;
        project_one(cam, centers + 3 * g, cov + 6 * g, rgba + 4 * g, sh ? sh + (size_t)shn * g : NULL,
                    g_scene_idx ? g_scene_idx[g] : 0u, &s);
        if (!s.visible) continue;
        frags += blend_one(&s, W, H, 0, 0, W, H, rop8, amb_eps, fb, ambig, g_err_bound_count ? g_err_bound[0] : NULL);
    }
    return frags;
}

/*
 * The same composite restricted to `nwin` windows of the frame (wins = {x0, y0, w, h} per window, GL window
 * coordinates; fbs[k] / ambigs[k] = that window's zeroed [h][w][4] floats / [h][w] bytes): every splat of the list is
 * projected once and composited into the windows it reaches, so crops of a full-size frame (5.8 M splats at 1080p ... 8K)
 * cost one projection pass instead of a full rasterisation.  A window's pixels equal the same pixels of gro_render.
 * sh_f16 != 0: `sh` holds IEEE half bits (uint16), widened here - the full-size scenes keep their SH as stored.
 */
static float half_bits_to_float(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 1023u;
    uint32_t bits;
    if (e == 0) {
        if (m == 0) bits = sign;
        else {
            int sh_ = 0;
            uint32_t mm = m;
            while (!(mm & 1024u)) { mm <<= 1; sh_++; }
            bits = sign | ((uint32_t)(113 - sh_) << 23) | ((mm & 1023u) << 13);
        }
    } else if (e == 31) bits = sign | 0x7F800000u | (m << 13);
    else bits = sign | ((e + 112u) << 23) | (m << 13);
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

uint64_t gro_render_windows(const gro_camera* cam, const float* centers, const float* cov, const uint8_t* rgba,
                            const void* sh, int sh_f16, const uint32_t* order, uint32_t count, int rop8, float amb_eps,
                            uint32_t nwin, const int32_t* wins, float** fbs, uint8_t** ambigs) {
    const int W = (int)cam->viewport[0], H = (int)cam->viewport[1];
    const int shn = cam->sh_stored == 0 ? 0 : (cam->sh_stored == 1 ? 9 : 24);
    uint64_t frags = 0;
    for (uint32_t i = 0; i < count; i++) {
        const size_t g = order ? order[i] : i;
        gro_splat2d s;
        float shf[24];
        const float* shp = NULL;
        if (sh && shn) {
            if (sh_f16) {
                const uint16_t* h = (const uint16_t*)sh + (size_t)shn * g;
                for (int k = 0; k < shn; k++) shf[k] = half_bits_to_float(h[k]);
                shp = shf;
            } else shp = (const float*)sh + (size_t)shn * g;
        }
        project_one(cam, centers + 3 * g, cov + 6 * g, rgba + 4 * g, shp, g_scene_idx ? g_scene_idx[g] : 0u, &s);
        if (!s.visible) continue;
        for (uint32_t k = 0; k < nwin; k++)
            frags += blend_one(&s, W, H, wins[4 * k], wins[4 * k + 1], wins[4 * k + 2], wins[4 * k + 3], rop8, amb_eps,
                               fbs[k], ambigs ? ambigs[k] : NULL, k < g_err_bound_count ? g_err_bound[k] : NULL);
    }
    return frags;
}

/* unorm8 conversion of a float framebuffer, round-to-nearest like GL. */
void gro_quantize(const float* fb, uint64_t count, uint8_t* out) {
    for (uint64_t i = 0; i < count; i++) out[i] = (uint8_t)floorf(clamp01(fb[i]) * 255.0f + 0.5f);
}
