// tree.hip — the octree the reference culls with, and the per-sort cull + index gather on the device.
//
//   build   /root/reference/src/splattree/SplatTree.js:132-271 (createSplatTreeWorker: buildSubTree +
//           processSplatTreeNode), driven by SplatMesh.buildSplatTree (src/splatmesh/SplatMesh.js:231-280: depth 8,
//           1000 centres per node, alpha filter).  One-off, host side, native C++ (the reference runs it in a Web
//           Worker).  The arithmetic is the reference's: fp32 centres widened to double, child boxes from
//           min + (max-min)*0.5 in double, INCLUSIVE containment so a point on a split plane enters several children,
//           first leaf in depth-first order keeps it, leaves sorted ascending, leaf order = depth-first.
//   gather  Viewer.gatherSceneNodesForSort (src/Viewer.js:1969-2077): per leaf, centre -> view space (three.js
//           Vector3.applyMatrix4 / normalize, fp64, same operation order), keep unless outside fov-0.6 AND farther than
//           its own diagonal, order kept leaves by distance, lay their index lists out far -> near.  The reference does
//           this in JS on the main thread and memcpy's up to R indexes per sort; here it is three small kernels plus a
//           coalesced copy, and the list lands directly in the sorter's device buffer.
// Built with -ffp-contract=off: every fp64 product and sum rounds once, like the JS engine's.
#include <algorithm>
#include <math.h>

#include "gs_internal.hpp"

// ---------------------------------------------------------------------------------------------------
// host-side build
// ---------------------------------------------------------------------------------------------------
struct TreeLeaf {
    double mn[3], mx[3], center[3];
    uint32_t depth;
    uint32_t begin, count;      // slice of gs_tree::indexes
};

struct gs_tree {
    gs_context* ctx = nullptr;
    uint32_t max_depth = 8, max_centers = 1000;
    uint32_t all_leaves = 0, nodes = 0;
    double scene_min[3] = {0, 0, 0}, scene_max[3] = {0, 0, 0};
    std::vector<TreeLeaf> leaves;          // nodesWithIndexes order
    std::vector<uint32_t> indexes;
    // device mirror
    DevBuf d_center, d_size, d_begin, d_count, d_indexes;
    DevBuf d_key, d_cnt, d_rank, d_sorted_cnt, d_sorted_leaf, d_offset, d_total, d_out;
};

namespace {

struct BuildCtx {
    const float* centers;       // xyz per LOCAL splat
    uint32_t first_index;
    uint32_t max_depth, max_centers;
    std::vector<uint8_t> added;
    gs_tree* tree;
};

struct Box {
    double mn[3], mx[3];
    bool contains(const float* p) const {
        return (double)p[0] >= mn[0] && (double)p[0] <= mx[0] && (double)p[1] >= mn[1] && (double)p[1] <= mx[1] &&
               (double)p[2] >= mn[2] && (double)p[2] <= mx[2];
    }
};

// processSplatTreeNode, SplatTree.js:132-245.  `list` holds LOCAL splat numbers in ascending order.
void process_node(BuildCtx& b, const Box& box, uint32_t depth, std::vector<uint32_t>& list) {
    gs_tree* t = b.tree;
    t->nodes++;
    if (list.size() < b.max_centers || depth > b.max_depth) {                          // :135
        t->all_leaves++;
        TreeLeaf leaf;
        leaf.begin = (uint32_t)t->indexes.size();
        for (uint32_t i : list)                                                        // :136-143 first leaf visited wins
            if (!b.added[i]) {
                b.added[i] = 1;
                t->indexes.push_back(b.first_index + i);
            }
        leaf.count = (uint32_t)t->indexes.size() - leaf.begin;
        std::sort(t->indexes.begin() + leaf.begin, t->indexes.end());                  // :144-147
        if (leaf.count > 0) {                                                          // convertWorkerSubTree :71-76
            for (int k = 0; k < 3; k++) {
                leaf.mn[k] = box.mn[k];
                leaf.mx[k] = box.mx[k];
                leaf.center[k] = (box.mx[k] - box.mn[k]) * 0.5 + box.mn[k];            // WorkerSplatTreeNode :121-123
            }
            leaf.depth = depth;
            t->leaves.push_back(leaf);
        }
        return;
    }
    double dim[3], half[3], c[3];
    for (int k = 0; k < 3; k++) {                                                      // :152-160
        dim[k] = box.mx[k] - box.mn[k];
        half[k] = dim[k] * 0.5;
        c[k] = box.mn[k] + half[k];
    }
    const double x0 = c[0] - half[0], x1 = c[0], x2 = c[0] + half[0];
    const double y0 = c[1] - half[1], y1 = c[1], y2 = c[1] + half[1];
    const double z0 = c[2] - half[2], z1 = c[2], z2 = c[2] + half[2];
    const Box child[8] = {                                                             // :162-182, same order
        {{x0, y1, z0}, {x1, y2, z1}}, {{x1, y1, z0}, {x2, y2, z1}}, {{x1, y1, z1}, {x2, y2, z2}}, {{x0, y1, z1}, {x1, y2, z2}},
        {{x0, y0, z0}, {x1, y1, z1}}, {{x1, y0, z0}, {x2, y1, z1}}, {{x1, y0, z1}, {x2, y1, z2}}, {{x0, y0, z1}, {x1, y1, z2}}};
    std::vector<uint32_t> lists[8];
    for (uint32_t i : list) {                                                          // :191-203
        const float* p = b.centers + 3 * (size_t)i;
        for (int j = 0; j < 8; j++)
            if (child[j].contains(p)) lists[j].push_back(i);
    }
    std::vector<uint32_t>().swap(list);                                                // node.data = {} :213
    for (int j = 0; j < 8; j++) process_node(b, child[j], depth + 1, lists[j]);        // :214-216
}

}  // namespace

// ---------------------------------------------------------------------------------------------------
// device-side gather
// ---------------------------------------------------------------------------------------------------
struct GatherParams {
    double mv[16];              // baseModelView = inverse(camera.matrixWorld) * mesh.matrixWorld
    double thr_x, thr_y;        // cos(fovX/2) - 0.6, cos(fovY/2) - 0.6
    uint32_t gather_all, leaves;
};

// Viewer.js:2010-2035 for one leaf
__global__ __launch_bounds__(256) void k_tree_test(GatherParams p, const double* __restrict__ center,
                                                   const double* __restrict__ size, const uint32_t* __restrict__ count,
                                                   double* __restrict__ key, uint32_t* __restrict__ cnt,
                                                   uint32_t* __restrict__ rank) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= p.leaves) return;
    (void)rank;
    const double x = center[3 * (size_t)i], y = center[3 * (size_t)i + 1], z = center[3 * (size_t)i + 2];
    const double* e = p.mv;
    // Vector3.applyMatrix4 (three r160)
    const double w = 1.0 / (e[3] * x + e[7] * y + e[11] * z + e[15]);
    double vx = (e[0] * x + e[4] * y + e[8] * z + e[12]) * w;
    double vy = (e[1] * x + e[5] * y + e[9] * z + e[13]) * w;
    double vz = (e[2] * x + e[6] * y + e[10] * z + e[14]) * w;
    const double dist = sqrt(vx * vx + vy * vy + vz * vz);                  // tempVector.length()
    {   // tempVector.normalize() = multiplyScalar(1 / (length || 1))
        const double s = 1.0 / (dist != 0.0 ? dist : 1.0);
        vx *= s; vy *= s; vz *= s;
    }
    // tempVectorYZ = (0, vy, vz).normalize(); tempVectorXZ = (vx, 0, vz).normalize(); forward = (0,0,-1)
    const double lyz = sqrt(0.0 * 0.0 + vy * vy + vz * vz);
    const double yz_z = vz * (1.0 / (lyz != 0.0 ? lyz : 1.0));
    const double lxz = sqrt(vx * vx + 0.0 * 0.0 + vz * vz);
    const double xz_z = vz * (1.0 / (lxz != 0.0 ? lxz : 1.0));
    const double dot_xz = -xz_z, dot_yz = -yz_z;                           // forward.dot(v) = -v.z
    const bool out_y = dot_yz < p.thr_y, out_x = dot_xz < p.thr_x;
    const bool skip = !p.gather_all && ((out_x || out_y) && dist > size[i]);
    key[i] = skip ? __longlong_as_double(0x7FF0000000000000ll) : dist;      // +inf sorts behind every kept leaf
    cnt[i] = skip ? 0u : count[i];
}

// rank of leaf i in ascending (distance, leaf number) order: O(n^2) compares, n = a few 10^4 leaves.  Distances are
// non-negative doubles (or +inf), so their bit patterns order like unsigned integers.  blockIdx.y splits the j range;
// the partial counts land in rank[slice][leaf] and are summed by k_tree_place.
constexpr uint32_t RANK_SPLIT = 64;
constexpr uint32_t RANK_PER_THREAD = 4;
// MODE 0: every key of the slice precedes every leaf of the block (j < i): a tie counts
// MODE 1: every key follows (j > i): a tie does not count     MODE 2: slices overlap: ties are broken by leaf number
// Keys are non-negative doubles or +inf, so their bit patterns order like unsigned integers.  64-bit compares run at a
// fraction of the 32-bit rate here (measured: ~30 cycles per wave and key-leaf pair with v_cmp_f64), so the common path
// compares the HIGH words only - one full-rate v_cmp + v_addc per pair - and a batch is redone exactly only when some
// lane saw equal high words (distances within 2^-20 of each other, or a leaf meeting itself).
template <int MODE>
__device__ __forceinline__ uint32_t exact_before(unsigned long long kj, unsigned long long mine, uint32_t j, uint32_t i) {
    if (MODE == 0) return kj <= mine ? 1u : 0u;
    if (MODE == 1) return kj < mine ? 1u : 0u;
    return (kj < mine || (kj == mine && j < i)) ? 1u : 0u;
}

template <int MODE>
__device__ __forceinline__ void rank_slice(const unsigned long long* __restrict__ key, uint32_t j_begin, uint32_t j_end,
                                           uint32_t i0, const unsigned long long (&mine)[RANK_PER_THREAD],
                                           uint32_t (&r)[RANK_PER_THREAD]) {
    constexpr uint32_t B = 16;                             // keys fetched per round with wave-uniform addresses (scalar loads)
    uint32_t mine_hi[RANK_PER_THREAD];
#pragma unroll
    for (uint32_t k = 0; k < RANK_PER_THREAD; k++) mine_hi[k] = (uint32_t)(mine[k] >> 32);
    uint32_t j = j_begin;
    for (; j + B <= j_end; j += B) {
        unsigned long long kk[B];
#pragma unroll
        for (uint32_t u = 0; u < B; u++) kk[u] = key[j + u];
        uint32_t fast[RANK_PER_THREAD] = {0, 0, 0, 0};
        bool tie = false;
#pragma unroll
        for (uint32_t u = 0; u < B; u++) {
            const uint32_t hi = (uint32_t)(kk[u] >> 32);
#pragma unroll
            for (uint32_t k = 0; k < RANK_PER_THREAD; k++) {
                fast[k] += hi < mine_hi[k] ? 1u : 0u;
                tie = tie || (hi == mine_hi[k]);
            }
        }
        if (__any(tie)) {                                  // rare, wave-uniform: redo this batch exactly
#pragma unroll
            for (uint32_t k = 0; k < RANK_PER_THREAD; k++) fast[k] = 0;
#pragma unroll
            for (uint32_t u = 0; u < B; u++)
#pragma unroll
                for (uint32_t k = 0; k < RANK_PER_THREAD; k++) fast[k] += exact_before<MODE>(kk[u], mine[k], j + u, i0 + k);
        }
#pragma unroll
        for (uint32_t k = 0; k < RANK_PER_THREAD; k++) r[k] += fast[k];
    }
    for (; j < j_end; j++) {
        const unsigned long long kj = key[j];
#pragma unroll
        for (uint32_t k = 0; k < RANK_PER_THREAD; k++) r[k] += exact_before<MODE>(kj, mine[k], j, i0 + k);
    }
}

__global__ __launch_bounds__(256) void k_tree_rank(const unsigned long long* __restrict__ key, uint32_t n,
                                                   uint32_t* __restrict__ rank) {
    // every thread ranks RANK_PER_THREAD leaves against a slice of all keys
    const uint32_t block_i0 = blockIdx.x * 256u * RANK_PER_THREAD, block_i1 = block_i0 + 256u * RANK_PER_THREAD;
    const uint32_t i0 = block_i0 + threadIdx.x * RANK_PER_THREAD;
    unsigned long long mine[RANK_PER_THREAD];
    uint32_t r[RANK_PER_THREAD];
#pragma unroll
    for (uint32_t k = 0; k < RANK_PER_THREAD; k++) {
        mine[k] = i0 + k < n ? key[i0 + k] : 0ull;
        r[k] = 0;
    }
    const uint32_t per = (n + RANK_SPLIT - 1) / RANK_SPLIT;
    const uint32_t j_begin = min(n, blockIdx.y * per), j_end = min(n, j_begin + per);
    if (j_end <= block_i0) rank_slice<0>(key, j_begin, j_end, i0, mine, r);
    else if (j_begin >= block_i1) rank_slice<1>(key, j_begin, j_end, i0, mine, r);
    else rank_slice<2>(key, j_begin, j_end, i0, mine, r);
#pragma unroll
    for (uint32_t k = 0; k < RANK_PER_THREAD; k++)
        if (i0 + k < n) rank[(size_t)blockIdx.y * n + i0 + k] = r[k];      // partial count of this key slice
}

__global__ __launch_bounds__(256) void k_tree_place(const uint32_t* __restrict__ rank, const uint32_t* __restrict__ cnt,
                                                    uint32_t n, uint32_t* __restrict__ sorted_cnt,
                                                    uint32_t* __restrict__ sorted_leaf) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    uint32_t r = 0;
#pragma unroll 8
    for (uint32_t y = 0; y < RANK_SPLIT; y++) r += rank[(size_t)y * n + i];
    sorted_cnt[r] = cnt[i];
    sorted_leaf[r] = i;
}

// one workgroup: offset[r] = total - inclusive_prefix(sorted_cnt)[r]  (nearest leaf, r = 0, ends the buffer:
// Viewer.js:2046-2055 copies from the END backwards).  Thread t owns the contiguous ranks [t*per, (t+1)*per).
__global__ __launch_bounds__(1024) void k_tree_offsets(const uint32_t* __restrict__ sorted_cnt, uint32_t n,
                                                       uint32_t* __restrict__ offset, uint32_t* __restrict__ total_out) {
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_carry;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    constexpr uint32_t CH = 16;                            // values per thread and round, all loads in flight together
    // pass 1: total
    uint32_t sum = 0;
    for (uint32_t i = tid; i < n; i += 1024u) sum += sorted_cnt[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if (lane == 0) s_wave[wave] = sum;
    __syncthreads();
    uint32_t total = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) total += s_wave[w];
    if (tid == 0) {
        *total_out = total;
        s_carry = 0;
    }
    // pass 2: rounds of 16384 ranks, thread t owns CH consecutive ones
    for (uint32_t base = 0; base < n; base += 1024u * CH) {
        const uint32_t b = base + tid * CH;
        uint32_t v[CH];
#pragma unroll
        for (uint32_t k = 0; k < CH; k++) v[k] = b + k < n ? sorted_cnt[b + k] : 0u;
        uint32_t mine = 0;
#pragma unroll
        for (uint32_t k = 0; k < CH; k++) mine += v[k];
        uint32_t incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(incl, o, 64);
            if ((int)lane >= o) incl += t;
        }
        __syncthreads();                                   // s_wave / s_carry of the previous round consumed
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        uint32_t wbase = 0, chunk = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) {
            const uint32_t c = s_wave[w];
            wbase += ((uint32_t)w < wave) ? c : 0u;
            chunk += c;
        }
        uint32_t run = s_carry + wbase + incl - mine;
#pragma unroll
        for (uint32_t k = 0; k < CH; k++) {
            run += v[k];
            if (b + k < n) offset[b + k] = total - run;
        }
        __syncthreads();
        if (tid == 0) s_carry += chunk;
    }
}

// one workgroup per rank: coalesced copy of the leaf's index list to its place in indexesToSort
__global__ __launch_bounds__(256) void k_tree_copy(const uint32_t* __restrict__ sorted_leaf, const uint32_t* __restrict__ sorted_cnt,
                                                   const uint32_t* __restrict__ offset, const uint32_t* __restrict__ leaf_begin,
                                                   const uint32_t* __restrict__ leaf_indexes, uint32_t* __restrict__ out) {
    const uint32_t r = blockIdx.x;
    const uint32_t n = sorted_cnt[r];
    if (n == 0) return;
    const uint32_t* src = leaf_indexes + leaf_begin[sorted_leaf[r]];
    uint32_t* dst = out + offset[r];
    for (uint32_t t = threadIdx.x; t < n; t += 256u) dst[t] = src[t];
}

extern "C" {

int gs_tree_create(gs_context* ctx, const float* centers, const uint8_t* keep, uint32_t count, uint32_t first_index,
                   uint32_t max_depth, uint32_t max_centers_per_node, gs_tree** out) {
    GS_REQUIRE(out && (centers || count == 0), "out / centers == NULL");
    *out = nullptr;
    GS_REQUIRE(max_centers_per_node > 0, "max_centers_per_node == 0");
    gs_tree* t = new (std::nothrow) gs_tree();
    if (!t) return GS_ERR_NOMEM;
    t->ctx = ctx;
    t->max_depth = max_depth;
    t->max_centers = max_centers_per_node;
    try {
        // buildSubTree, SplatTree.js:218-246: bounds over the FILTERED centres, root list in upload order
        std::vector<uint32_t> root;
        root.reserve(count);
        bool first = true;
        for (uint32_t i = 0; i < count; i++) {
            if (keep && !keep[i]) continue;                                            // filterFunc, SplatTree.js:332
            const float* p = centers + 3 * (size_t)i;
            for (int k = 0; k < 3; k++) {
                if (first || (double)p[k] < t->scene_min[k]) t->scene_min[k] = p[k];
                if (first || (double)p[k] > t->scene_max[k]) t->scene_max[k] = p[k];
            }
            first = false;
            root.push_back(i);
        }
        BuildCtx b = {centers, first_index, max_depth, max_centers_per_node, std::vector<uint8_t>(count, 0), t};
        Box box;
        for (int k = 0; k < 3; k++) {
            box.mn[k] = t->scene_min[k];
            box.mx[k] = t->scene_max[k];
        }
        process_node(b, box, 0, root);
    } catch (const std::bad_alloc&) {
        delete t;
        gs_set_error("out of host memory while building the splat tree");
        return GS_ERR_NOMEM;
    }
    if (ctx) {
        ScopedDevice sd(ctx->device);
        const size_t L = t->leaves.size();
        std::vector<double> center(3 * L), size(L);
        std::vector<uint32_t> begin(L), cnt(L);
        for (size_t i = 0; i < L; i++) {
            const TreeLeaf& lf = t->leaves[i];
            for (int k = 0; k < 3; k++) center[3 * i + k] = lf.center[k];
            // nodeSize: tempMax.copy(node.max).sub(node.min).length(), Viewer.js:1983-1986
            const double dx = lf.mx[0] - lf.mn[0], dy = lf.mx[1] - lf.mn[1], dz = lf.mx[2] - lf.mn[2];
            size[i] = sqrt(dx * dx + dy * dy + dz * dz);
            begin[i] = lf.begin;
            cnt[i] = lf.count;
        }
        int st = GS_OK;
        auto A = [&](DevBuf& buf, size_t bytes) { if (st == GS_OK) st = buf.alloc(bytes); };
        A(t->d_center, 24 * L + 24); A(t->d_size, 8 * L + 8); A(t->d_begin, 4 * L + 4); A(t->d_count, 4 * L + 4);
        A(t->d_indexes, 4 * t->indexes.size() + 4);
        A(t->d_key, 8 * L + 8); A(t->d_cnt, 4 * L + 4); A(t->d_rank, 4 * L * RANK_SPLIT + 4); A(t->d_sorted_cnt, 4 * L + 4);
        A(t->d_sorted_leaf, 4 * L + 4); A(t->d_offset, 4 * L + 4); A(t->d_total, 16);
        if (st != GS_OK) {
            delete t;
            return st;
        }
        hipStream_t s = ctx->stream;
        hipError_t e = hipSuccess;
        auto UP = [&](DevBuf& buf, const void* src, size_t bytes) {
            if (e == hipSuccess && bytes) e = hipMemcpyAsync(buf.p, src, bytes, hipMemcpyHostToDevice, s);
        };
        UP(t->d_center, center.data(), 24 * L); UP(t->d_size, size.data(), 8 * L); UP(t->d_begin, begin.data(), 4 * L);
        UP(t->d_count, cnt.data(), 4 * L); UP(t->d_indexes, t->indexes.data(), 4 * t->indexes.size());
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) {
            gs_set_error("uploading the splat tree failed: %s", hipGetErrorString(e));
            delete t;
            return GS_ERR_HIP;
        }
    }
    *out = t;
    return GS_OK;
}

void gs_tree_destroy(gs_tree* t) {
    if (!t) return;
    if (t->ctx) {
        ScopedDevice sd(t->ctx->device);
        (void)hipDeviceSynchronize();
        delete t;
    } else {
        delete t;
    }
}

int gs_tree_get_info(gs_tree* t, gs_tree_info* info) {
    GS_REQUIRE(t && info, "tree / info == NULL");
    info->leaves = (uint32_t)t->leaves.size();
    info->all_leaves = t->all_leaves;
    info->nodes = t->nodes;
    info->splats = (uint32_t)t->indexes.size();
    for (int k = 0; k < 3; k++) {
        info->scene_min[k] = t->scene_min[k];
        info->scene_max[k] = t->scene_max[k];
    }
    return GS_OK;
}

int gs_tree_read(gs_tree* t, double* bounds, double* centers, uint32_t* depths, uint32_t* offsets, uint32_t* indexes) {
    GS_REQUIRE(t != nullptr, "tree == NULL");
    const size_t L = t->leaves.size();
    for (size_t i = 0; i < L; i++) {
        const TreeLeaf& lf = t->leaves[i];
        for (int k = 0; k < 3; k++) {
            if (bounds) { bounds[6 * i + k] = lf.mn[k]; bounds[6 * i + 3 + k] = lf.mx[k]; }
            if (centers) centers[3 * i + k] = lf.center[k];
        }
        if (depths) depths[i] = lf.depth;
        if (offsets) offsets[i] = lf.begin;
    }
    if (offsets) offsets[L] = (uint32_t)t->indexes.size();
    if (indexes && !t->indexes.empty()) memcpy(indexes, t->indexes.data(), 4 * t->indexes.size());
    return GS_OK;
}

int gs_tree_gather(gs_tree* t, const gs_gather_params* gp, gs_sorter* dst, uint32_t* render_count, uint32_t* indexes_out_host) {
    GS_REQUIRE(t && gp && render_count, "tree / params / render_count == NULL");
    GS_REQUIRE(t->ctx != nullptr, "host-only tree (created without a context) cannot gather");
    GS_REQUIRE(!dst || dst->ctx == t->ctx, "sorter lives on another context");
    GS_REQUIRE(!dst || dst->max_count >= t->indexes.size(), "sorter is smaller than the tree");
    gs_context* ctx = t->ctx;
    ScopedDevice sd(ctx->device);
    hipStream_t st = dst ? dst->stream : ctx->stream;      // the list is produced where the sort will consume it
    const uint32_t L = (uint32_t)t->leaves.size();
    uint32_t* out_dev = nullptr;
    if (dst) {
        GS_TRY(dst->idx_in.ensure((size_t)dst->max_count * 4));
        out_dev = dst->idx_in.as<uint32_t>();
    } else {
        GS_TRY(t->d_out.ensure(4 * t->indexes.size() + 4));
        out_dev = t->d_out.as<uint32_t>();
    }
    *render_count = 0;
    if (L == 0) {
        if (dst) dst->gathered = 0, dst->has_gathered = true;
        return GS_OK;
    }
    GatherParams p;
    memcpy(p.mv, gp->model_view, sizeof(p.mv));
    // Viewer.js:1990-1996
    const double deg2rad = 3.14159265358979323846 / 180.0;              // THREE.MathUtils.DEG2RAD = Math.PI / 180
    const double focal = (gp->render_height / 2.0) / tan(gp->fov_y_deg / 2.0 * deg2rad);
    const double fov_x2 = atan(gp->render_width / 2.0 / focal), fov_y2 = atan(gp->render_height / 2.0 / focal);
    p.thr_x = cos(fov_x2) - .6;
    p.thr_y = cos(fov_y2) - .6;
    p.gather_all = gp->gather_all ? 1u : 0u;
    p.leaves = L;
    const dim3 g((L + 255u) / 256u), b(256);
    hipLaunchKernelGGL(k_tree_test, g, b, 0, st, p, t->d_center.as<double>(), t->d_size.as<double>(), t->d_count.as<uint32_t>(),
                       t->d_key.as<double>(), t->d_cnt.as<uint32_t>(), t->d_rank.as<uint32_t>());
    hipLaunchKernelGGL(k_tree_rank, dim3((L + 256u * RANK_PER_THREAD - 1u) / (256u * RANK_PER_THREAD), RANK_SPLIT), b, 0, st, t->d_key.as<unsigned long long>(), L,
                       t->d_rank.as<uint32_t>());
    hipLaunchKernelGGL(k_tree_place, g, b, 0, st, t->d_rank.as<uint32_t>(), t->d_cnt.as<uint32_t>(), L,
                       t->d_sorted_cnt.as<uint32_t>(), t->d_sorted_leaf.as<uint32_t>());
    hipLaunchKernelGGL(k_tree_offsets, dim3(1), dim3(1024), 0, st, t->d_sorted_cnt.as<uint32_t>(), L, t->d_offset.as<uint32_t>(),
                       t->d_total.as<uint32_t>());
    hipLaunchKernelGGL(k_tree_copy, dim3(L), b, 0, st, t->d_sorted_leaf.as<uint32_t>(), t->d_sorted_cnt.as<uint32_t>(),
                       t->d_offset.as<uint32_t>(), t->d_begin.as<uint32_t>(), t->d_indexes.as<uint32_t>(), out_dev);
    GS_HIP(hipGetLastError());
    uint32_t total = 0;
    GS_HIP(hipMemcpyAsync(&total, t->d_total.p, 4, hipMemcpyDeviceToHost, st));
    GS_HIP(hipStreamSynchronize(st));                      // the Viewer needs splatRenderCount on the host
    *render_count = total;
    if (indexes_out_host && total) {
        GS_HIP(hipMemcpyAsync(indexes_out_host, out_dev, (size_t)total * 4, hipMemcpyDeviceToHost, st));
        GS_HIP(hipStreamSynchronize(st));
    }
    if (dst) {
        dst->gathered = total;
        dst->has_gathered = true;
    }
    return GS_OK;
}

}  // extern "C"
