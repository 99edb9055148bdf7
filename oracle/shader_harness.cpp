// oracle/shader_harness.cpp — TEST INFRASTRUCTURE.  Runs the reference's OWN vertex / fragment shader text on the CPU.
// SHADER_VERT / SHADER_FRAG are files produced by oracle/make_golden_raster.py from the strings the reference's
// SplatMaterial3D.build() returned (token rewrites only, see there); they compile here against oracle/glsl_shim.hpp.  This
// file supplies what a WebGL2 draw call supplies around them: the data textures laid out as
// /root/reference/src/splatmesh/SplatMesh.js:637-898 lays them out, the uniforms of SplatMesh.updateUniforms (:1248-1280)
// and three.js' built-ins (modelViewMatrix, projectionMatrix, viewMatrix, cameraPosition, the quad attribute `position`,
// SplatGeometry.js:14-23), and collects gl_Position / varyings per (splat, quad corner) and gl_FragColor / discard per
// fragment.  Built by make_golden_raster.py into oracle/_ref/ (never committed: it embeds reference text).
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "glsl_shim.hpp"

#define discard do { gs_discarded = true; return; } while (0)

namespace VS {
using namespace glsl;
// three.js built-ins of a ShaderMaterial vertex shader
mat4 modelViewMatrix, projectionMatrix, viewMatrix;
vec3 cameraPosition;
vec3 position;
vec4 gl_Position;
#include SHADER_VERT
}  // namespace VS

namespace FS {
using namespace glsl;
vec4 gl_FragColor;
bool gs_discarded;
#include SHADER_FRAG
}  // namespace FS

extern "C" {

struct HarnessScene {
    uint32_t count, sh_degree_stored, cov_half, sh_u8;
    const float* centers;        // [3n]
    const uint8_t* rgba;         // [4n]
    const float* cov;            // [6n] fp32 (full-precision covariances)
    const uint16_t* cov16;       // [6n] half bits (cov_half)
    const float* sh;             // [ncoef*n]: what the sampler returns - fp16 values widened (RGBA16F) or byte/255 (RGBA8)
    const uint32_t* scene_idx;   // [n] or NULL
};

struct HarnessUniforms {
    float model_view[16], projection[16], view_matrix[16], camera_position[3];
    float focal[2], viewport[2], ortho_zoom, inverse_focal_adjustment, splat_scale;
    int32_t orthographic, point_cloud, sh_degree, sh_8bit, fade_in_complete, scene_count;
    float scene_center[3], fade_start_radius;
    float transforms[32][16], scene_opacity[32], sh8_min[32], sh8_max[32];
    int32_t scene_visibility[32];
};

static glsl::mat4 to_mat4(const float* m) {
    glsl::mat4 r;
    for (int j = 0; j < 4; j++) for (int i = 0; i < 4; i++) r[j][i] = m[4 * j + i];
    return r;
}
static int pow2_at_least(size_t v) { int p = 1; while ((size_t)p < v) p <<= 1; return p; }

// out per (splat, corner): gl_Position[4], vColor[4], vPosition[2]  (10 floats); corners in SplatGeometry order
// (-1,-1), (-1,1), (1,1), (1,-1)
void harness_run_vertex(const HarnessScene* sc, const HarnessUniforms* u, float* out) {
    using namespace glsl;
    const uint32_t n = sc->count;
    const uint32_t ncoef = sc->sh_degree_stored == 0 ? 0 : (sc->sh_degree_stored == 1 ? 9 : 24);
    // data textures (SplatMesh.js:637-898): power-of-two sizes keep index -> uv -> texel exact
    const int W = 64;
    auto height = [&](size_t texels) { return pow2_at_least((texels + W - 1) / W); };
    std::vector<uint32_t> cc(4 * (size_t)W * height(n), 0u);             // RGBA32UI: {rgba8 packed, bits(x), bits(y), bits(z)}
    for (uint32_t i = 0; i < n; i++) {
        const uint8_t* c = sc->rgba + 4 * i;
        cc[4 * i] = (uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16) | ((uint32_t)c[3] << 24);      // Util.js:53-55
        memcpy(&cc[4 * i + 1], sc->centers + 3 * i, 12);
    }
    VS::centersColorsTexture.data = cc.data(); VS::centersColorsTexture.w = W; VS::centersColorsTexture.h = height(n);
    VS::centersColorsTexture.channels = 4;
    VS::centersColorsTextureSize = vec2((float)W, (float)height(n));
    std::vector<float> covf;                                               // RGBA32F, 6 floats per splat, 1.5 texels (:741)
    std::vector<uint32_t> covh;                                            // RGBA32UI, one texel per splat: 3 packed half pairs (:735-739)
    if (sc->cov_half) {
        covh.assign(4 * (size_t)W * height(n), 0u);
        for (uint32_t i = 0; i < n; i++)
            for (int k = 0; k < 3; k++)
                covh[4 * i + k] = (uint32_t)sc->cov16[6 * i + 2 * k] | ((uint32_t)sc->cov16[6 * i + 2 * k + 1] << 16);
        VS::covariancesTextureHalfFloat.data = covh.data(); VS::covariancesTextureHalfFloat.w = W;
        VS::covariancesTextureHalfFloat.h = height(n); VS::covariancesTextureHalfFloat.channels = 4;
        VS::covariancesTextureSize = vec2((float)W, (float)height(n));
        VS::covariancesAreHalfFloat = 1;
    } else {
        const size_t texels = ((size_t)n * 6 + 3) / 4 + 2;
        covf.assign(4 * (size_t)W * height(texels), 0.0f);
        memcpy(covf.data(), sc->cov, sizeof(float) * 6 * n);
        VS::covariancesTexture.data = covf.data(); VS::covariancesTexture.w = W; VS::covariancesTexture.h = height(texels);
        VS::covariancesTextureSize = vec2((float)W, (float)height(texels));
        VS::covariancesAreHalfFloat = 0;
    }
    std::vector<float> shf;                                                // single texture: 9 -> 10 / 24 -> 24 padded components (:797-867)
    if (ncoef) {
        const uint32_t stride = ncoef == 9 ? 10 : 24;
        const size_t texels = ((size_t)n * stride + 3) / 4 + 2;
        shf.assign(4 * (size_t)W * height(texels), 0.0f);
        for (uint32_t i = 0; i < n; i++) memcpy(&shf[(size_t)i * stride], sc->sh + (size_t)i * ncoef, sizeof(float) * ncoef);
        VS::sphericalHarmonicsTexture.data = shf.data(); VS::sphericalHarmonicsTexture.w = W;
        VS::sphericalHarmonicsTexture.h = height(texels);
        VS::sphericalHarmonicsTextureSize = vec2((float)W, (float)height(texels));
    }
    VS::sphericalHarmonicsMultiTextureMode = 0;
    std::vector<uint32_t> sidx(1, 0u);
    if (sc->scene_idx) {                                                   // R32UI (:881-897)
        sidx.assign((size_t)W * height(n), 0u);
        memcpy(sidx.data(), sc->scene_idx, 4 * (size_t)n);
        VS::sceneIndexesTexture.data = sidx.data(); VS::sceneIndexesTexture.w = W; VS::sceneIndexesTexture.h = height(n);
        VS::sceneIndexesTexture.channels = 1;
        VS::sceneIndexesTextureSize = vec2((float)W, (float)height(n));
    }
    // uniforms (SplatMesh.js:1248-1280, Viewer.js:651-677)
    VS::modelViewMatrix = to_mat4(u->model_view); VS::projectionMatrix = to_mat4(u->projection);
    VS::viewMatrix = to_mat4(u->view_matrix);
    VS::cameraPosition = vec3(u->camera_position[0], u->camera_position[1], u->camera_position[2]);
    VS::focal = vec2(u->focal[0], u->focal[1]);
    VS::viewport = vec2(u->viewport[0], u->viewport[1]);
    VS::basisViewport = vec2(1.0f / u->viewport[0], 1.0f / u->viewport[1]);
    VS::orthoZoom = u->ortho_zoom; VS::orthographicMode = u->orthographic; VS::pointCloudModeEnabled = u->point_cloud;
    VS::inverseFocalAdjustment = u->inverse_focal_adjustment; VS::splatScale = u->splat_scale;
    VS::sphericalHarmonicsDegree = u->sh_degree; VS::sphericalHarmonics8BitMode = u->sh_8bit;
    VS::fadeInComplete = u->fade_in_complete; VS::sceneCount = u->scene_count;
    VS::sceneCenter = vec3(u->scene_center[0], u->scene_center[1], u->scene_center[2]);
    VS::visibleRegionFadeStartRadius = u->fade_start_radius; VS::visibleRegionRadius = 0.0f;
    VS::currentTime = 0.0f; VS::firstRenderTime = 0.0f;
    for (int s = 0; s < 32; s++) {
        VS::sphericalHarmonics8BitCompressionRangeMin[s] = u->sh8_min[s];
        VS::sphericalHarmonics8BitCompressionRangeMax[s] = u->sh8_max[s];
#ifdef SHADER_DYNAMIC
        VS::transforms[s] = to_mat4(u->transforms[s]);
#endif
#ifdef SHADER_EFFECTS
        VS::sceneOpacity[s] = u->scene_opacity[s];
        VS::sceneVisibility[s] = u->scene_visibility[s];
#endif
    }
    static const float corners[4][2] = {{-1, -1}, {-1, 1}, {1, 1}, {1, -1}};
    const float nan = nanf("");
    for (uint32_t i = 0; i < n; i++)
        for (int c = 0; c < 4; c++) {
            VS::splatIndex = i;
            VS::position = vec3(corners[c][0], corners[c][1], 0.0f);
            VS::gl_Position = vec4(nan, nan, nan, nan);                    // an early `return` leaves it undefined in GL
            VS::vColor = vec4(0, 0, 0, 0);
            VS::vPosition = vec2(0, 0);
            VS::main();
            float* o = out + 10 * ((size_t)4 * i + c);
            for (int k = 0; k < 4; k++) { o[k] = VS::gl_Position[k]; o[4 + k] = VS::vColor[k]; }
            o[8] = VS::vPosition.x; o[9] = VS::vPosition.y;
        }
}

// per fragment: in vPosition[2], vColor[4]; out gl_FragColor[4] and a discard flag
void harness_run_fragment(uint32_t count, const float* v_position, const float* v_color, float* frag_color, uint8_t* discarded) {
    for (uint32_t i = 0; i < count; i++) {
        FS::vPosition = glsl::vec2(v_position[2 * i], v_position[2 * i + 1]);
        FS::vColor = glsl::vec4(v_color[4 * i], v_color[4 * i + 1], v_color[4 * i + 2], v_color[4 * i + 3]);
        FS::gl_FragColor = glsl::vec4(0, 0, 0, 0);
        FS::gs_discarded = false;
        FS::main();
        for (int k = 0; k < 4; k++) frag_color[4 * i + k] = FS::gl_FragColor[k];
        discarded[i] = FS::gs_discarded ? 1 : 0;
    }
}

}  // extern "C"
