/*
 * gsplat_hip.h — C ABI of libgsplat_hip.so: the MI355X (gfx950) sort-and-rasterize engine that replaces
 * the two device-facing seams of mkkellogg/GaussianSplats3D.  Plain pointers and sizes only.
 *
 * Every entry point cites the reference interface it replaces (paths relative to /root/reference).
 * The reference has no FFI today; INTEGRATION.md shows the N-API / ctypes bindings a maintainer adds.
 *
 *   SORT SEAM    src/worker/SortWorker.js:202-256 (createSortWorker + message protocol) over the WASM C ABI
 *                src/worker/sorter.cpp:17-22 (sortIndexes)                              -> gs_sorter_*
 *   RENDER SEAM  src/splatmesh/SplatMesh.js (setupDataTextures :637-898, updateRenderIndexes :1228-1235,
 *                updateUniforms :1248-1280) + the GLSL pair SplatMaterial.js:112-341 /
 *                SplatMaterial3D.js:81-255 drawn by renderer.render (src/Viewer.js:1616)  -> gs_mesh_*
 *
 * Threading: calls on one gs_context (and the objects created from it) must be serialised by the caller,
 * like the reference's single sort worker + single GL context.  All device work of a context runs on one
 * HIP stream.  Calls that return data to host memory block until it is there; calls that leave their
 * result on the device only enqueue work.
 *
 * There is NO CPU fallback: every call fails with GS_ERR_HIP if no gfx950 device is usable.
 */
#ifndef GSPLAT_HIP_H
#define GSPLAT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GS_ABI_VERSION 5   /* 5: + gs_mesh_set_draw_mode (round 6); 4: + gs_mesh_set_destination (round 5); earlier entry points unchanged */

/* status codes (negative = error, positive = warning, result still defined) */
#define GS_OK 0
#define GS_WARN_KEY_CLAMPED 1   /* a depth bucket fell outside [0,range): the reference corrupts memory here  */
#define GS_WARN_FRAME_TRUNCATED 2 /* gs_mesh_render: an EARLIER asynchronous draw overflowed its entry buffer (its frame
                                     lacks the farthest entries of some lists); the buffers have been grown, this draw and
                                     the following ones are complete                                                   */
#define GS_ERR_INVALID (-1)     /* bad argument                                                               */
#define GS_ERR_HIP (-2)         /* HIP runtime / device failure (see gs_last_error)                            */
#define GS_ERR_NOMEM (-3)
#define GS_ERR_CAPACITY (-4)    /* tile-entry buffer overflow even after growing to the configured limit       */
#define GS_ERR_UNSUPPORTED (-5)

typedef struct gs_context gs_context;
typedef struct gs_sorter gs_sorter;
typedef struct gs_mesh gs_mesh;
typedef struct gs_tree gs_tree;

/* Message of the last failing call on this thread (the reference throws JS Errors, SplatMesh.js:1518-1533). */
const char* gs_last_error(void);
int gs_abi_version(void);
/* Number of visible HIP devices, or a negative status. */
int gs_device_count(void);

/* One context per GPU.  hip_stream: an existing hipStream_t to enqueue on (e.g. PyTorch's current
 * stream) or NULL to let the library create its own. */
int gs_context_create(int device, void* hip_stream, gs_context** out);
/* The same with explicit flags.  GS_CTX_SINGLE_STREAM: sorts, the vertex stage and the rest of a draw all run on the one
 * stream (no worker / aux streams), i.e. a frame is strictly sort -> draw.  gs_context_create takes this flag from the
 * environment (GSPLAT_SERIAL=1); the default is the reference's shape: the sort runs concurrently with drawing, like its
 * Web Worker (src/worker/SortWorker.js). */
#define GS_CTX_SINGLE_STREAM 1u
#define GS_CTX_STAGE_TIMING 2u /* bracket the stages of EVERY sort / draw with timing events, so that gs_sorter_last_stats /
                                  gs_mesh_last_stats can time calls that were made without a stats pointer.  By default only
                                  calls that pass `stats` (and therefore synchronise anyway) are timed and the *_ms fields
                                  read 0 after any other call: an event record is a barrier packet on the stream, nine of them
                                  per frame cost 35 us of a 0.33 ms frame on MI355X */
#define GS_CTX_FORK_JOIN 4u    /* multi-stream context whose FRAMES stay serial: the sort and the vertex stage of a frame run side by
                                  side on their own streams (they share nothing: the sort reads the centres, the vertex stage
                                  the mesh planes) and join where the binner needs both, but no sort starts before everything the
                                  context's stream held when it was called has finished - frame k + 1 never overlaps frame k.
                                  The default context lets the next sort overlap the tail of the previous draw (throughput);
                                  this one shortens the frame itself: t_frame = max(t_sort, t_vertex) + t_bin + t_blend */
int gs_context_create_ex(int device, void* hip_stream, uint32_t flags, gs_context** out);
void gs_context_destroy(gs_context* ctx);
int gs_context_synchronize(gs_context* ctx);
/* Turn GS_CTX_STAGE_TIMING on (enable != 0) or off for the sorts / draws that follow (e.g. while a profiling overlay is up). */
int gs_context_set_stage_timing(gs_context* ctx, int enable);

/* ------------------------------------------------------------------------------------------------ *
 * SORT SEAM
 * ------------------------------------------------------------------------------------------------ */
#define GS_SORT_INTEGER 1u  /* integerBasedSort (Viewer option, src/Viewer.js:95-98): int32 centres x1000      */
#define GS_SORT_DYNAMIC 2u  /* dynamicMode: per-splat scene index + per-scene transform (sorter.cpp:41-62)     */
#define GS_MAX_SCENES 32u   /* src/Constants.js:7 */

/* createSortWorker(splatCount, useSharedMemory, enableSIMDInSort, integerBasedSort, dynamicMode,
 * splatSortDistanceMapPrecision) — src/worker/SortWorker.js:202-256.  Shared-memory / SIMD flavours of
 * the WASM module have no meaning here.  precision_bits: 10..20 (integer) / 10..24 (float),
 * src/Viewer.js:208-210. */
int gs_sorter_create(gs_context* ctx, uint32_t max_splat_count, uint32_t flags, uint32_t precision_bits,
                     gs_sorter** out);
void gs_sorter_destroy(gs_sorter* s);

/* The {centers, sceneIndexes, range:{from,count}} message — src/worker/SortWorker.js:84-98.
 * centers_aos4: int32[4*count] (GS_SORT_INTEGER; SplatMesh.getIntegerCenters(padFour), SplatMesh.js:1912-1926)
 * or float[4*count].  scene_indexes: uint32[count], required iff GS_SORT_DYNAMIC. */
int gs_sorter_upload_centers(gs_sorter* s, uint32_t from, uint32_t count, const void* centers_aos4,
                             const uint32_t* scene_indexes);

typedef struct gs_sort_stats {
    float device_ms;        /* sortTime of the sortDone message (SortWorker.js:76-78), device clock; 0 if the
                               sort was made without `stats` and without GS_CTX_STAGE_TIMING                 */
    int32_t key_min;        /* minDistance / maxDistance of sorter.cpp:24-25,72-73                           */
    int32_t key_max;
    uint32_t clamped;       /* buckets forced into [0,range)                                                  */
    uint32_t passes;        /* 8-bit LSD radix passes used                                                    */
    uint32_t result_count;  /* length of the sorted result: splatRenderCount, or the kept count under
                               gs_sorter_set_frustum_cull                                                     */
} gs_sort_stats;

/* The {sort:{modelViewProj, splatRenderCount, splatSortCount, usePrecomputedDistances, indexesToSort,
 * transforms, precomputedDistances}} message -> sortIndexes (sorter.cpp:17-22) -> {sortDone, sortedIndexes}.
 *   mvp              float[16] column-major (fp32 as written into WASM memory, SortWorker.js:54)
 *   indexes_to_sort  uint32[render_count] host, or NULL = identity list (Viewer.js:2061-2073)
 *   precomputed      int32/float[splat_count] host or NULL (usePrecomputedDistances)
 *   transforms       float[16*GS_MAX_SCENES] host, required iff GS_SORT_DYNAMIC
 *   sorted_out       uint32[render_count] host or NULL (result stays on the device for gs_mesh_render)
 * Result: [0, render-sort) copied; tail = far->near buckets, ties in reverse input order. */
int gs_sorter_sort(gs_sorter* s, const float* mvp, const uint32_t* indexes_to_sort, uint32_t sort_count,
                   uint32_t render_count, const void* precomputed, const float* transforms,
                   uint32_t* sorted_out, gs_sort_stats* stats);

/* The same sort, fed by the device-resident indexesToSort list (and splatRenderCount) that the last
 * gs_tree_gather(tree, ..., s, ...) left in this sorter: no index list crosses PCIe. */
int gs_sorter_sort_gathered(gs_sorter* s, const float* mvp, uint32_t sort_count, const void* precomputed,
                            const float* transforms, uint32_t* sorted_out, gs_sort_stats* stats);

/* Optional coupling of the two seams: a sorter bound to a mesh leaves its device-resident result as positions in
 * that mesh's internal storage order, so gs_mesh_render(m, ..., sorter = s, ...) needs no per-frame index translation.
 * Host-visible results (sorted_out, gs_sorter_debug_read) are always the caller's splat indexes.  m = NULL unbinds. */
int gs_sorter_bind_mesh(gs_sorter* s, gs_mesh* m);

/* Per-splat frustum cull fused into the sort (no counterpart in the reference, whose cull works on octree nodes,
 * src/Viewer.js:1969-2077; composes with gs_tree_gather).  With enable != 0 a full sort (sort_count == render_count,
 * static scene, no precomputed distances) keeps only the list positions whose centre passes
 *     q = mvp * (x, y, z, 1) in fp32;  |q.x| <= 1.25 q.w + 0.01,  |q.y| <= 1.25 q.w + 0.01,  |q.z| <= 1.01 q.w + 0.01
 * which for the same camera is a superset of what the vertex stage can draw (it drops at 1.2 w, SplatMaterial.js:160-164).
 * Keys, min / max and buckets are still taken over every list position, so the result is exactly the reference's sorted
 * list with the dropped splats removed and a frame drawn from it is bit-identical to one drawn from the full list.
 * The result's length lives on the device (gs_sort_stats.result_count after a sync); gs_mesh_render reads it there:
 * pass the sort's render_count as usual. */
int gs_sorter_set_frustum_cull(gs_sorter* s, int enable);

/* Visibility cull: the exact version of the above, for a sorter bound to a mesh (gs_sorter_bind_mesh).  After
 * gs_mesh_project(m, cam) a full sort keeps exactly the list positions whose splat survived the mesh's vertex stage for
 * `cam` - every reject of the shaders (SplatMaterial.js:160-164, SplatMaterial3D.js:188), an empty pixel footprint, and the
 * camera's strip of tile rows (gs_camera.tile_row_begin / end).  Keys, min / max and buckets are still taken over every
 * list position, so the result is the reference's sorted list restricted to the splats this frame (this rank's strip)
 * draws, and the frame is bit-identical.  This is how the tile-row strips of a multi-GPU draw shard the sort: each rank
 * keys all splats (12 bytes each) but radix-sorts and bins only its own.  Order per frame:
 *     gs_mesh_project(m, cam) -> gs_sorter_sort(s, mvp of the same camera, ...) -> gs_mesh_render(m, cam, sorter = s). */
int gs_sorter_set_visibility_cull(gs_sorter* s, int enable);

/* Test hooks: intermediates of the last sort, positions [0, render_count) (valid in the sorted tail).
 * what: 0 = int32 depth keys (mappedDistances before mapping), 1 = int32 buckets (after), 2 = sorted,
 *       3 = keep bits of the last culled sort, bit (i & 31) of uint32 word i >> 5: per list position i after a frustum-culled
 *           sort, per ORIGINAL splat index i (the bound mesh's visibility mask as the sort consumed it) after a
 *           visibility-culled one. */
int gs_sorter_debug_read(gs_sorter* s, int what, void* dst, uint32_t count);

/* ------------------------------------------------------------------------------------------------ *
 * CULL (feeds the sort seam): the reference's octree and its per-sort frustum cull
 * ------------------------------------------------------------------------------------------------ */
/* SplatMesh.buildSplatTree -> SplatTree.processSplatMesh -> worker createSplatTree
 * (src/splatmesh/SplatMesh.js:231-280, src/splattree/SplatTree.js:132-271, 320-420).
 *   centers      float[3*count], SplatMesh.getSplatCenter of splats first_index .. first_index+count-1
 *   keep         uint8[count] or NULL: the alpha filter `splatColor.w >= minAlpha` (SplatMesh.js:239-244)
 *   max_depth / max_centers_per_node   8 / 1000 in the reference (SplatMesh.js:236)
 * The tree is built on the host with the reference's arithmetic (leaves, their order and their index lists are
 * identical); with ctx != NULL it is mirrored to the device for gs_tree_gather, with ctx == NULL it is host-only. */
int gs_tree_create(gs_context* ctx, const float* centers, const uint8_t* keep, uint32_t count, uint32_t first_index,
                   uint32_t max_depth, uint32_t max_centers_per_node, gs_tree** out);
void gs_tree_destroy(gs_tree* t);

typedef struct gs_tree_info {
    uint32_t leaves;        /* subTree.nodesWithIndexes.length (leaves holding >= 1 index)                    */
    uint32_t all_leaves;    /* countLeaves()                                                                  */
    uint32_t nodes;
    uint32_t splats;        /* sum of the leaves' index counts                                                */
    double scene_min[3], scene_max[3];
} gs_tree_info;
int gs_tree_get_info(gs_tree* t, gs_tree_info* info);
/* Leaves in nodesWithIndexes (depth-first) order; any pointer may be NULL.  bounds double[6*leaves] = min xyz,
 * max xyz; centers double[3*leaves]; depths uint32[leaves]; offsets uint32[leaves+1]; indexes uint32[splats]. */
int gs_tree_read(gs_tree* t, double* bounds, double* centers, uint32_t* depths, uint32_t* offsets, uint32_t* indexes);

/* Viewer.gatherSceneNodesForSort (src/Viewer.js:1969-2077). */
typedef struct gs_gather_params {
    double model_view[16];   /* inverse(camera.matrixWorld) * splatMesh.matrixWorld, fp64, column-major (:1999-2000) */
    double fov_y_deg;        /* camera.fov                                                                    */
    double render_width, render_height;   /* getRenderDimensions                                              */
    uint32_t gather_all;     /* gatherAllNodes                                                                */
    uint32_t pad;
} gs_gather_params;
/* Tests every leaf, orders the kept ones by distance and lays their index lists out far -> near (the nearest leaf
 * ends the buffer), on the device: four small launches plan where every kept leaf's list goes (a count-weighted rank by
 * distance), then the lists are copied - at once, or, when a static sorter is the only reader, by that sorter's next full sort,
 * fused with its key kernel (the list is then in place when gs_sorter_sort_gathered returns).
 *   dst               sorter whose device-side indexesToSort buffer receives the list (then call
 *                     gs_sorter_sort_gathered), or NULL
 *   render_count      out: splatRenderCount (this waits for the device); or NULL = asynchronous: nothing returns to the host,
 *                     the count stays on the device next to the list (needs dst, no host copy).  A following
 *                     gs_sorter_sort_gathered(dst, ..., sort_count >= the tree's splat count) then sorts the whole list
 *                     without ever learning its length on the host, and gs_mesh_render(..., sorter = dst, render_count =
 *                     the tree's splat count) draws it: the frame needs no host round trip at all
 *   indexes_out_host  uint32[tree splats] host copy of the list, or NULL */
int gs_tree_gather(gs_tree* t, const gs_gather_params* params, gs_sorter* dst, uint32_t* render_count,
                   uint32_t* indexes_out_host);

/* ------------------------------------------------------------------------------------------------ *
 * ASSETS (host side, no GPU needed): the reference's file readers up to the arrays the seams consume
 * ------------------------------------------------------------------------------------------------ */
typedef struct gs_asset gs_asset;
#define GS_ASSET_PLY 1u      /* INRIA-v1 .ply, binary little endian: src/loaders/ply/INRIAV1PlyParser.js            */
#define GS_ASSET_KSPLAT 2u   /* .ksplat, compression levels 0/1/2: src/loaders/SplatBuffer.js                        */
/* Parses `data` (the bytes of the file).  max_sh_degree: outSphericalHarmonicsDegree (Viewer option
 * sphericalHarmonicsDegree); splats keep FILE order (the reference's optimizeSplatData:false). */
int gs_asset_open(const void* data, uint64_t bytes, uint32_t format, uint32_t max_sh_degree, gs_asset** out);
void gs_asset_close(gs_asset* a);
typedef struct gs_asset_info {
    uint32_t splat_count;
    uint32_t sh_degree;          /* degree the fill arrays carry                                                */
    uint32_t compression_level;  /* of the splat buffer: 0 (also every PLY), 1, 2                                */
    uint32_t sh_level;           /* getTargetSphericalHarmonicsCompressionLevel: 1 = fp16 output, 2 = uint8      */
    float scene_center[3];
    float sh_min, sh_max;        /* min/maxSphericalHarmonicsCoeff of the file (8-bit SH range)                  */
} gs_asset_info;
int gs_asset_get_info(gs_asset* a, gs_asset_info* info);
/* SplatMesh.fillSplatDataArrays (src/splatmesh/SplatMesh.js:1853-1902) without a scene transform; any pointer may
 * be NULL.  centers float[3n]; cov_f32 float[6n] / cov_f16 half bits[6n] (covariance compression level 0 / 1);
 * rgba uint8[4n] with alpha zeroed below min_alpha; sh_f16 half bits[ncoef*n] (sh_level 1) or sh_u8 uint8[ncoef*n]
 * (sh_level 2), coefficient-major RGB triples; scales float[3n], rotations float[4n] (x,y,z,w) normalised with
 * w >= 0, as fillSplatScaleRotationArray returns them (SplatBuffer.js:349-437). */
int gs_asset_fill(gs_asset* a, uint32_t min_alpha, float* centers, float* cov_f32, uint16_t* cov_f16, uint8_t* rgba,
                  uint16_t* sh_f16, uint8_t* sh_u8, float* scales, float* rotations);

/* ------------------------------------------------------------------------------------------------ *
 * RENDER SEAM
 * ------------------------------------------------------------------------------------------------ */
#define GS_MESH_COV_HALF 1u   /* halfPrecisionCovariancesOnGPU (SplatMesh.js:667-670,735-739)                 */
#define GS_MESH_KEEP_ORDER 4u /* keep splats in upload order on the device (default: each upload is re-ordered along a
                                 Morton curve internally; indexes at this ABI are always the caller's)            */
#define GS_MESH_SH_U8 2u      /* SH stored as uint8 (compression level 2, SplatMesh.js:680-684,789,1064-1066):
                                 sphericalHarmonics8BitMode, dequantised per scene as v/255*(max-min)+min      */
#define GS_SH_F16 0u          /* SH as fp16 (compression level <= 1, SplatMesh.js:1064-1066)                  */

/* SplatMesh.build + setupDataTextures (SplatMesh.js:306-405, 637-898): device SoA planes instead of data
 * textures, so none of the 4096^2-texel limits apply. */
int gs_mesh_create(gs_context* ctx, uint32_t max_splat_count, uint32_t sh_degree, uint32_t flags,
                   gs_mesh** out);
void gs_mesh_destroy(gs_mesh* m);

/* Upload splats [from, from+count) in the formats fillSplatDataArrays emits (SplatMesh.js:1853-1902):
 *   centers float[3*count]
 *   cov     (m00,m01,m02,m11,m12,m22) per splat (SplatBuffer.js:440-486): cov_f32 float[6*count] for a
 *           full-precision mesh, cov_f16 uint16 (IEEE half bits)[6*count] for a GS_MESH_COV_HALF mesh -
 *           the caller narrows exactly as THREE.DataUtils.toHalfFloat does (SplatBuffer.js:469-474);
 *           exactly one of the two is non-NULL
 *   rgba    uint8[4*count]
 *   sh_f16  uint16 (IEEE half bits)[ncoef*count], coefficient-major RGB triples, ncoef = 0/9/24
 *           (SplatBuffer.js:680-728); NULL when the mesh has sh_degree 0. */
int gs_mesh_upload(gs_mesh* m, uint32_t from, uint32_t count, const float* centers, const float* cov_f32,
                   const uint16_t* cov_f16, const uint8_t* rgba, const uint16_t* sh_f16);

/* 8-bit SH of a GS_MESH_SH_U8 mesh: uint8[ncoef*count], same coefficient order as sh_f16 (pass sh_f16 = NULL to
 * gs_mesh_upload for such a mesh and call this for the same range). */
int gs_mesh_upload_sh_u8(gs_mesh* m, uint32_t from, uint32_t count, const uint8_t* sh_u8);

/* Per-splat scene index (sceneIndexesTexture, SplatMesh.js:881-897): needed when more than one scene is loaded. */
int gs_mesh_upload_scene_indexes(gs_mesh* m, uint32_t from, uint32_t count, const uint32_t* scene_indexes);

/* Per-scene uniforms (SplatMesh.updateUniforms :1263-1276, setupDataTextures :868-878). */
typedef struct gs_scene_params {
    uint32_t scene_count;                       /* sceneCount                                                  */
    uint32_t pad;
    float transforms[GS_MAX_SCENES][16];        /* uniforms.transforms (GS_CAM_DYNAMIC)                        */
    float inv_cam_pos[GS_MAX_SCENES][4];        /* inverse(transform) * cameraPosition per scene; the shader
                                                   evaluates GLSL inverse() here (precision implementation-
                                                   defined), the caller passes the fp64 result               */
    float opacity[GS_MAX_SCENES];               /* sceneOpacity, clamped to [0,1]                              */
    uint32_t visible[GS_MAX_SCENES];            /* sceneVisibility                                             */
    float sh8_min[GS_MAX_SCENES];               /* sphericalHarmonics8BitCompressionRangeMin / Max             */
    float sh8_max[GS_MAX_SCENES];
} gs_scene_params;
int gs_mesh_set_scenes(gs_mesh* m, const gs_scene_params* params);

/* Uniforms of one draw: three's modelViewMatrix / projectionMatrix / cameraPosition plus
 * SplatMesh.updateUniforms (SplatMesh.js:1248-1280) as computed by Viewer.updateSplatMesh
 * (src/Viewer.js:651-677). */
typedef struct gs_camera {
    float view[16];          /* modelViewMatrix = viewMatrix * mesh.matrixWorld, column-major                */
    float proj[16];          /* projectionMatrix                                                            */
    float cam_pos[3];        /* cameraPosition                                                              */
    float focal[2];          /* focal.x/.y                                                                  */
    uint32_t width, height;  /* viewport in pixels (renderDimensions * devicePixelRatio)                    */
    float splat_scale;       /* setSplatScale, SplatMesh.js:1282                                            */
    float kernel2d;          /* kernel2DSize, default 0.3                                                   */
    float max_splat_px;      /* maxScreenSpaceSplatSize, default 1024 (src/Viewer.js:147)                   */
    float inv_focal_adj;     /* inverseFocalAdjustment                                                      */
    uint32_t sh_degree;      /* sphericalHarmonicsDegree uniform (<= mesh degree)                           */
    uint32_t flags;          /* GS_CAM_*                                                                    */
    uint32_t tile_row_begin; /* multi-GPU: this rank renders 16-px tile rows [begin,end); 0,0 = all rows    */
    uint32_t tile_row_end;
    /* shader permutations (all zero = static perspective scene) */
    float ortho_zoom;        /* orthoZoom (GS_CAM_ORTHOGRAPHIC)                                             */
    float scene_center[3];   /* sceneCenter (GS_CAM_FADE_IN)                                                */
    float fade_start_radius; /* visibleRegionFadeStartRadius (GS_CAM_FADE_IN)                               */
    float view_matrix[16];   /* viewMatrix (GS_CAM_DYNAMIC: modelView = viewMatrix * transforms[scene])     */
} gs_camera;
#define GS_CAM_ANTIALIASED 1u
#define GS_CAM_POINT_CLOUD 2u
#define GS_CAM_ORTHOGRAPHIC 4u   /* orthographicMode: J = diag(orthoZoom), SplatMaterial3D.js:112-117           */
#define GS_CAM_FADE_IN 8u        /* fadeInComplete == 0: distance fade-in, SplatMaterial.js:347-363             */
#define GS_CAM_SCENE_EFFECTS 16u /* enableOptionalEffects: per-scene opacity / visibility, SplatMaterial.js:129 */
#define GS_CAM_DYNAMIC 32u       /* dynamicMode: per-scene transforms, SplatMaterial.js:140-144,179-183         */
#define GS_TILE 16u
#define GS_DRAW_POOL_EXHAUSTED 1u   /* gs_render_stats.flags */

typedef struct gs_render_stats {
    float device_ms;          /* whole draw; the five times are 0 after a draw made without `stats` and
                                 without GS_CTX_STAGE_TIMING                                                 */
    float project_ms, bin_ms, tile_sort_ms, blend_ms;
    uint32_t visible_splats;  /* splats that survive the vertex-stage rejects                                */
    uint64_t tile_entries;    /* list entries = sum over splats of list bins (list_bin_px) touched           */
    uint32_t entry_capacity;
    uint32_t overflowed;      /* 1 = frame was re-run after growing the entry buffer                         */
    uint64_t tiles16;         /* D of SURVEY.md 8d = sum over splats of 16x16-px tiles touched               */
    uint32_t list_bin_px;     /* edge of a list bin of this draw (32, 128, 256 or 512): the unit of the entry lists and of
                                 gs_mesh_debug_read(what = 2); chosen per mesh from the previous measured draw    */
    uint32_t flags;           /* GS_DRAW_*: bit 0 = the per-bin blend ran out of chunk-partial slots (a list thousands of
                                 splats deep outside the deep pass): the affected quadrants were composited as one long chunk -
                                 still a valid front-to-back composite, but no longer bit-identical to what a strip of another
                                 cut or the deep pass would produce (was `pad`, always 0, before round 4)                    */
    uint64_t entries_scanned; /* list entries the blend read before its pixels saturated (<= tile_entries per 32-px
                                 bin of a list)                                                                  */
    uint64_t splats_walked;   /* (splat, 16x16-px tile) pairs the blend evaluated                                */
    uint64_t halves_evaluated; /* = 2 x splats_walked: both 16x8-px halves of a walked (splat, tile) pair are evaluated (the
                                 per-half skip of ABI 3's first draft was measured slower and removed; the field stays
                                 for layout compatibility)                                                         */
} gs_render_stats;

/* Entry-buffer overflow: a draw that returns statistics or pixels to the host checks and re-runs itself after growing the
 * buffers (gs_render_stats.overflowed).  A draw that returns nothing cannot; the following gs_mesh_render notices it without
 * synchronising, grows the buffers and returns GS_WARN_FRAME_TRUNCATED once.
 *
 * updateRenderIndexes(globalIndexes, renderSplatCount) + renderer.render(splatMesh, camera)
 * (SplatMesh.js:1228-1235, src/Viewer.js:1616).  Draw order = index order = back-to-front.
 *   sorted_host   uint32[render_count] host, or NULL
 *   sorter        take the device-resident result of the last gs_sorter_sort (when sorted_host == NULL)
 *   rgba_out_host uint8[4*W*rows*16...] RGBA8, row 0 = bottom (GL), covering tile rows [begin,end) clipped
 *                 to the viewport; or NULL
 *   rgba_out_dev  same, device pointer (e.g. a torch uint8 tensor); or NULL to use an internal buffer */
int gs_mesh_render(gs_mesh* m, const gs_camera* cam, const uint32_t* sorted_host, gs_sorter* sorter,
                   uint32_t render_count, uint8_t* rgba_out_host, void* rgba_out_dev, gs_render_stats* stats);

/* The DESTINATION the splats are blended into.  The reference draws the other scene geometry first and then the splats with
 * `depthTest: true, depthWrite: false` and NormalBlending (src/splatmesh/SplatMaterial3D.js:72-73; draw order
 * src/Viewer.js:1610-1616; drop-in mode, where the splat mesh is one object of the host's own scene:
 * src/DropInViewer.js:34-42): a splat fragment is kept where its depth passes three's default depthFunc (LessEqualDepth)
 * against the depth the opaque geometry left, and what is kept is blended OVER the colour that geometry left.  The quad of a
 * splat sits at its centre's depth (gl_Position.z = ndcCenter.z, SplatMaterial3D.js:206-210), so the test is per (splat,
 * pixel): window depth of the splat's CENTRE, 0.5 * ndc.z + 0.5 (glDepthRange 0..1), <= depth[pixel].
 *
 *   depth_host / depth_dev   float[height * width], window-space depth in [0, 1] as glReadPixels(DEPTH_COMPONENT, FLOAT) or
 *                            a THREE.DepthTexture(FloatType) holds it, row 0 = bottom; at most one of the two; both NULL = no
 *                            depth test (every fragment passes)
 *   rgba_host / rgba_dev     uint8[4 * height * width] RGBA8, row 0 = bottom: the colour the splats are blended over
 *                            (rgb = C + T * dst.rgb, alpha = 1 - T * (1 - dst.a): what NormalBlending, back to front, leaves);
 *                            both NULL = the reference Viewer's clear colour (0, 0, 0, 0)
 *   width, height            must equal the camera's viewport at every draw while the destination is set
 *   GS_DEST_DEPTH_UNORM24    compare as a 24-bit fixed-point depth buffer does: both sides round(z * (2^24 - 1)) first
 *
 * Host buffers are copied to the device by this call (it returns when they are reusable); device buffers are READ BY EVERY
 * FOLLOWING DRAW until the destination is replaced or cleared (dest == NULL) - the caller keeps them alive and orders its
 * writes to them before the draws (same stream or an event).  A multi-GPU strip reads its own rows of the full-size buffers.
 * Draw modes and outputs are unchanged; frames stay bit-identical across strips and across the deep pass.
 * The call waits for the mesh's draws in flight (they read the previous destination).  A descriptor that is refused (both a host
 * and a device pointer for one plane, unknown flags, a size of 0 or beyond 65536 px) leaves the mesh WITHOUT a destination. */
#define GS_DEST_DEPTH_UNORM24 1u
typedef struct gs_destination {
    const float* depth_host;
    const void* depth_dev;
    const uint8_t* rgba_host;
    const void* rgba_dev;
    uint32_t width, height;
    uint32_t flags;
    uint32_t pad;
} gs_destination;
int gs_mesh_set_destination(gs_mesh* m, const gs_destination* dest);

/* The vertex stage of a draw on its own (the GLSL vertex shader, SplatMaterial.js:112-341 + SplatMaterial3D.js:81-217):
 * projects every uploaded splat for `cam` and leaves records, tile rects and the visibility mask on the device.  The next
 * gs_mesh_render with an identical gs_camera consumes them instead of projecting again (once: the vertex stage runs exactly
 * one time per frame either way); any other camera simply projects afresh.  Needed before a visibility-culled sort. */
int gs_mesh_project(gs_mesh* m, const gs_camera* cam);

/* Intermediates of the last draw (tests, strip load-balancing).  what: 0 = per splat (storage order) the
 * 32-byte vertex-stage record {cx, cy, ax, ay, bx, by, r|g<<16, b|a<<16 (unorm16)}; 1 = per splat the tile
 * rect {x0|y0<<16, x1|y1<<16} (0 and 1 are defined only for splats whose mask bit is set); 2 = per list bin (gs_render_stats.list_bin_px) of the
 * drawn strip the [begin,end) range of its entry list ((~0,0) = untouched); 3 = the visibility mask, 1 bit per
 * splat packed in uint64 words (count = number of words); 4 = per 32x32-px blend bin of the drawn strip (row-major,
 * bins_x = ceil(width / 32)) the pair {list entries scanned, 2 x (splat, 16x16-px quadrant) pairs composited}: the blend's
 * real cost, used to balance multi-GPU strips; 5 = the deep pass of the last draw: {bins it composited, bins over its threshold,
 * chunk partials the per-bin kernel closed itself, 1 if that pool ran out}, then the bin numbers (count = 4 .. 4 + 512 words);
 * 6 = host state (count = 3 words): {visible splats, splats projected} of the last full-frame draw whose verdict has reached the
 * host, and where the last vertex stage ran its block test (1 = a kernel of its own, 0 = in every workgroup, 2 = not at all).
 *
 * The composite (csrc/tile_blend.hip).  Per 16x16-px quadrant, the ordered list entries whose ellipse reaches the quadrant are
 * cut into chunks of 1024; a chunk is the plain front-to-back composite from T = 1, and the chunks are merged near -> far
 * (C = fma(T, C_c, C); T = T * T_c).  Up to 1024 contributing splats per quadrant - every quadrant of the BASELINE
 * configurations - that is exactly the single front-to-back composite; beyond, it differs from it by fp32 rounding, and it lets
 * many waves composite one very deep quadrant at once (the "deep pass", chosen per bin from the previous draw's statistics).
 * The frame does not depend on that choice, on list batching or on how a multi-GPU draw cuts its strips. */
int gs_mesh_debug_read(gs_mesh* m, int what, void* dst, uint32_t count);

/* VERIFICATION of the blend state (SplatMaterial3D.js:65-75): the reference composites back to front into an RGBA8 target,
 * rounding every channel to 8 bits after EVERY splat; gs_mesh_render composites front to back in fp32 and rounds once.  This
 * entry point reproduces the reference's own semantics for a window of <= 65536 pixels of the LAST draw (its lists, records
 * and camera): rgb = a*src + (1-a)*rgb, alpha = a + (1-a)*alpha, then floor(clamp01(v)*255 + 0.5) per channel, splat by
 * splat, farthest first.  (x0, y0) in GL window coordinates (row 0 = bottom), inside the rows the last draw covered;
 * rgba_out_host: uint8[4*width*height], row-major from y0 upwards.  A thread per pixel walks its whole list: not a draw mode. */
int gs_mesh_debug_rop8(gs_mesh* m, uint32_t x0, uint32_t y0, uint32_t width, uint32_t height, uint8_t* rgba_out_host);

/* DRAW MODE of the draws that follow.
 *   GS_DRAW_FP32 (default)  the front-to-back fp32 composite, rounded to RGBA8 once (early termination, chunks, the deep pass):
 *                           <= 0.52 / 255 from the exact composite, 3-4 / 255 from what a browser's RGBA8 target shows on
 *                           translucent content (tests/test_gpu_crops.py gates both).
 *   GS_DRAW_ROP8_FULL       the reference's own blend state as it executes on a GPU (SplatMaterial3D.js:65-75: NormalBlending into
 *                           an RGBA8 target; src/Viewer.js:358-359: cleared to (0,0,0,0), or the destination's colour): back to
 *                           front, rgb = a*src + (1-a)*rgb, alpha = a + (1-a)*alpha, every channel rounded to 8 bits after EVERY
 *                           splat - gs_mesh_debug_rop8's semantics for the whole frame: >= 99.85 % of the channel values equal to
 *                           the ROP-emulating oracle, never more than 1 apart.  Every list is walked whole (no early termination
 *                           is possible back to front): the blend of a 1080p garden frame takes 4.0 ms instead of 0.06.
 *   GS_DRAW_ROP8            the same composite over the splats IN FRONT OF each 16x16 quadrant's saturation depth only: a first
 *                           pass walks the list front to back until every pixel of the quadrant has let <= 1e-6 through (counting
 *                           fragments of alpha >= 1/64: fainter ones hide nothing from an 8-bit target), a second pass blends
 *                           exactly those splats back to front with the per-splat rounding.  Colour: the full walk's gate
 *                           (>= 99.5 % of the r, g, b values equal to the ROP-emulating oracle, never more than 1 apart; on the
 *                           C3T crops colour is identical to the full walk).  Alpha: exact wherever it reaches 255; where it
 *                           stalls below 255 - alpha = q8(a + (1-a) alpha) stops moving once a (255 - alpha) < 0.5 - the value
 *                           it stalls at depends on the whole list and may differ by <= 2 steps.  Garden stand-in at 1080p:
 *                           blend 0.32 ms, i.e. a 0.50 ms frame = 11.6 Gsplats/s with the browser's colours; translucent content
 *                           (nothing saturates) costs what the full walk costs.
 * Strips, destinations (depth test and colour) and device-resident outputs work as in the fp32 mode; the statistics count the pairs
 * the mode walked, and they never schedule a later fp32 draw. */
#define GS_DRAW_FP32 0u
#define GS_DRAW_ROP8 1u
#define GS_DRAW_ROP8_FULL 2u
int gs_mesh_set_draw_mode(gs_mesh* m, uint32_t mode);

/* Scheduling switch, 1 by default (0 also via $GSPLAT_NO_DEEP at gs_mesh_create): whether the draws that follow may composite
 * very deep bins through the deep pass.  The frame is the same either way (see the composite above). */
int gs_mesh_set_deep_pass(gs_mesh* m, int enabled);

/* Measurement hook: summed device duration (HIP events on the stream the kernel is launched on) and number of
 * launches since the last reset.  which: 0 = k_project alone (the sampled untimed draws), 1 = the whole vertex stage (block
 * test + mask reset + k_project) of the timed draws - two clocks, kept apart.  Synchronises the streams.
 * On a single-stream context without GS_CTX_STAGE_TIMING every 8th launch is measured ($GSPLAT_KERNEL_SAMPLE; the two event
 * records cost 1.5 % of a frame): `launches` counts the measured ones. */
int gs_mesh_kernel_time(gs_mesh* m, int which, int reset, double* sum_ms, uint32_t* launches);

/* ------------------------------------------------------------------------------------------------ *
 * MULTI-GPU: tile-row strips, one rank per GPU, strips gathered over RCCL (no counterpart in the reference)
 * ------------------------------------------------------------------------------------------------ */
typedef struct gs_group gs_group;
#define GS_GROUP_ID_BYTES 128
/* ncclGetUniqueId: call on ONE rank and hand the 128 bytes to every rank (any side channel: MPI, a file, a pipe). */
int gs_group_unique_id(uint8_t* id_out);
/* ncclCommInitRank on the context's device.  Collective: every rank of the group calls it with the same id.
 * world_size 1 needs no id and never loads RCCL. */
int gs_group_create(gs_context* ctx, const uint8_t* id, uint32_t world_size, uint32_t rank, gs_group** out);
void gs_group_destroy(gs_group* g);
/* Gatherv of framebuffer strips, enqueued on the context's stream (returns without waiting).  Rank r owns pixel rows
 * [row_begin[r], row_end[r]) of a `width`-pixel RGBA8 frame (GL row order, as gs_mesh_render writes a strip: its first row
 * is row_begin[r]); strip_dev = this rank's rows, full_dev = the whole frame on `root` (ignored elsewhere).  Collective. */
int gs_group_gather_strips(gs_group* g, const void* strip_dev, void* full_dev, uint32_t width, const uint32_t* row_begin,
                           const uint32_t* row_end, uint32_t root);
/* Overlapped gathers (off by default).  The strips of an 8K frame are 133 MB: the root receives 16.6 MB from each of 7 peers
 * (~0.25 ms over one xGMI link each; with 2 ranks 66 MB over ONE link, ~0.9 ms) - as long as a rank's whole frame.  With
 * overlap on, gs_group_gather_strips / gs_group_render_gather enqueue the transfer on a stream of the group: it starts when
 * everything enqueued on the context's stream so far (the draw of this frame) has finished, and runs while the context's
 * stream goes on with the next frame.  Contract: the caller ALTERNATES between two strip buffers and two full-frame buffers
 * (gs_group_render_gather does that itself; its root's frame alternates between two internal buffers too); call k makes the
 * context's stream wait for the transfer of call k-1, so the draw that follows may overwrite the buffers of call k-1's
 * predecessor.  gs_group_wait blocks the host until every transfer issued so far has completed (the root's consumer calls it -
 * or synchronises the device - before it reads a gathered frame).  Every rank of the group must use the same setting. */
int gs_group_set_overlap(gs_group* g, int enabled);
int gs_group_wait(gs_group* g);

/* One rank's whole share of a multi-GPU draw: gs_mesh_render (same sorted_host | sorter choice) of its strip (pixel rows [row_begin[rank], row_end[rank]),
 * whole 16-px tile rows; `cam` describes the full viewport) into the mesh's own device framebuffer, then the gather above.
 * On `root`, rgba_out_host (nullable) receives the full W*H*4 frame (this waits for it); the other ranks only enqueue.
 * With gs_sorter_set_visibility_cull call gs_mesh_project with the same strip in the camera first.  Collective. */
int gs_group_render_gather(gs_group* g, gs_mesh* m, const gs_camera* cam, const uint32_t* sorted_host, gs_sorter* sorter, uint32_t render_count,
                           const uint32_t* row_begin, const uint32_t* row_end, uint32_t root, uint8_t* rgba_out_host);

/* Statistics of the last draw (synchronises the stream). */
int gs_mesh_last_stats(gs_mesh* m, gs_render_stats* stats);
/* Test hook: shrink (or grow) the entry buffers so a small scene can exercise the overflow -> regrow path. */
int gs_mesh_debug_set_entry_capacity(gs_mesh* m, uint32_t capacity);
int gs_sorter_last_stats(gs_sorter* s, gs_sort_stats* stats);

#ifdef __cplusplus
}
#endif
#endif /* GSPLAT_HIP_H */
