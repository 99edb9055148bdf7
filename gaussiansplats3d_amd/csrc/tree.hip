// tree.hip — the octree the reference culls with, and the per-sort cull + index gather on the device.
//
//   build   /root/reference/src/splattree/SplatTree.js:132-271 (createSplatTreeWorker: buildSubTree +
//           processSplatTreeNode), driven by SplatMesh.buildSplatTree (src/splatmesh/SplatMesh.js:231-280: depth 8,
//           1000 centres per node, alpha filter).  One-off (the reference runs it in a Web Worker).  The arithmetic is the
//           reference's: fp32 centres widened to double, child boxes from min + (max-min)*0.5 in double, INCLUSIVE
//           containment so a point on a split plane enters several children, first leaf in depth-first order keeps it,
//           leaves sorted ascending, leaf order = depth-first.  With a context the build runs ON THE DEVICE, level by level
//           (tree_build_device below): the kernels only compare centres with split planes and move memberships, every
//           fp64 box is computed by the same host code as the host builder, so the leaves are the reference's bit for
//           bit; without a context (or under GSPLAT_TREE_HOST_BUILD=1) the recursive host builder runs.
//   gather  Viewer.gatherSceneNodesForSort (src/Viewer.js:1969-2077): per leaf, centre -> view space (three.js
//           Vector3.applyMatrix4 / normalize, fp64, same operation order), keep unless outside fov-0.6 AND farther than
//           its own diagonal, order kept leaves by distance, lay their index lists out far -> near.  The reference does
//           this in JS on the main thread and memcpy's up to R indexes per sort; here it is a handful of small kernels plus a
//           coalesced copy, and the list lands directly in the sorter's device buffer.
// Built with -ffp-contract=off: every fp64 product and sum rounds once, like the JS engine's.
#include <algorithm>
#include <math.h>
#include <stdlib.h>

#include "radix.hpp"

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
    bool built_on_device = false;
    uint64_t uid = 0;               // unique per tree of this process (a sorter's leaf-major caches are keyed on it)
    double prev_scale = 1.0;        // bucket scale of the previous gather (its member counters are reset by the next one)
    uint32_t gather_serial = 0;     // gathers planned so far
    gs_sorter* pending_sorter = nullptr;   // a sorter that has not yet copied the last planned gather's lists (deferred, fused copy)
    double scene_min[3] = {0, 0, 0}, scene_max[3] = {0, 0, 0};
    std::vector<TreeLeaf> leaves;          // nodesWithIndexes order
    std::vector<uint32_t> indexes;
    // device mirror
    DevBuf d_center, d_size, d_begin, d_count, d_indexes;
    DevBuf d_bucket;            // uint64 hist [TREE_BUCKETS] | uint64 start [TREE_BUCKETS] | uint32 fill [TREE_BUCKETS] | uint64 chunk sums
    DevBuf d_key, d_rank, d_offset, d_total, d_out;   // d_total: {splats gathered, leaves kept}
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
// device-side build
// ---------------------------------------------------------------------------------------------------
// processSplatTreeNode is a recursion over NODES whose only per-point work is "which of the 8 child boxes contain this
// centre" (inclusive, so possibly several, possibly none: c - half may round above the parent's min).  Level by level, with
// one MEMBERSHIP (point, node) per list element of the recursion:
//   k_tree_split   every membership of a node that splits at this level is tested against the node's 9 split planes
//                  (x0 x1 x2 | y0 y1 y2 | z0 z1 z2, doubles computed on the host by the reference's own expressions) and
//                  moves to the first child that contains the point; further children get appended memberships, no child
//                  = the membership dies (the reference drops the point from that subtree too).  Children count their
//                  list lengths (duplicates included, as `list.size()` does at SplatTree.js:135).
//   host           reads the 8 * (split nodes) counts, decides leaf / split for the next level (count >= maxCentres and
//                  depth <= maxDepth), computes the next split planes and child boxes.
//   k_tree_claim   `addedIndexes` (first leaf VISITED wins, SplatTree.js:136-143): depth-first visiting order of the leaves is
//                  the order of their child-index paths; every point takes the minimum over its surviving memberships.
//   radix sort     (final leaf number, point index): stable, so every leaf's list is ascending (SplatTree.js:144-147).
constexpr uint32_t TREE_DEAD = 0xFFFFFFFFu;

template <bool LDS_COUNTS>
__global__ __launch_bounds__(256) void k_tree_split(const float* __restrict__ pts, uint32_t* __restrict__ mem_point,
                                                    uint32_t* __restrict__ mem_node, uint32_t M, uint32_t cap,
                                                    uint32_t* __restrict__ tail, uint32_t lvl_base, uint32_t lvl_nodes,
                                                    const uint32_t* __restrict__ lvl_rank, const double* __restrict__ planes,
                                                    uint32_t child_base, uint32_t* __restrict__ child_count, uint32_t n_child,
                                                    uint32_t* __restrict__ overflow) {
    __shared__ uint32_t s_cnt[LDS_COUNTS ? 4096 : 1];
    if (LDS_COUNTS) {
        for (uint32_t k = threadIdx.x; k < n_child; k += 256u) s_cnt[k] = 0u;
        __syncthreads();
    }
    for (uint32_t m = blockIdx.x * 256u + threadIdx.x; m < M; m += gridDim.x * 256u) {
        const uint32_t id = mem_node[m];
        if (id - lvl_base >= lvl_nodes) continue;             // a final leaf of an earlier level, or dead
        const uint32_t r = lvl_rank[id - lvl_base];
        if (r == TREE_DEAD) continue;                         // a leaf of this level
        const uint32_t i = mem_point[m];
        const double px = (double)pts[3 * (size_t)i], py = (double)pts[3 * (size_t)i + 1], pz = (double)pts[3 * (size_t)i + 2];
        const double* pl = planes + 9 * (size_t)r;
        // Box3.containsPoint, inclusive on both sides; false for NaN
        const bool xl = px >= pl[0] && px <= pl[1], xh = px >= pl[1] && px <= pl[2];
        const bool yl = py >= pl[3] && py <= pl[4], yh = py >= pl[4] && py <= pl[5];
        const bool zl = pz >= pl[6] && pz <= pl[7], zh = pz >= pl[7] && pz <= pl[8];
        // child order of SplatTree.js:162-182: (x lo, y hi, z lo) (x hi, y hi, z lo) (x hi, y hi, z hi) (x lo, y hi, z hi)
        //                                       (x lo, y lo, z lo) (x hi, y lo, z lo) (x hi, y lo, z hi) (x lo, y lo, z hi)
        const bool in[8] = {xl && yh && zl, xh && yh && zl, xh && yh && zh, xl && yh && zh,
                            xl && yl && zl, xh && yl && zl, xh && yl && zh, xl && yl && zh};
        bool first = true;
#pragma unroll
        for (uint32_t j = 0; j < 8u; j++) {
            if (!in[j]) continue;
            const uint32_t c = 8u * r + j;
            if (LDS_COUNTS) atomicAdd(&s_cnt[c], 1u);
            else atomicAdd(&child_count[c], 1u);
            if (first) {
                mem_node[m] = child_base + c;
                first = false;
            } else {
                const uint32_t slot = atomicAdd(tail, 1u);
                if (slot < cap) {
                    mem_point[slot] = i;
                    mem_node[slot] = child_base + c;
                } else {
                    *overflow = 1u;
                }
            }
        }
        if (first) mem_node[m] = TREE_DEAD;
    }
    if (LDS_COUNTS) {
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < n_child; k += 256u)
            if (s_cnt[k]) atomicAdd(&child_count[k], s_cnt[k]);
    }
}

__global__ __launch_bounds__(256) void k_tree_root(uint32_t* __restrict__ mem_point, uint32_t* __restrict__ mem_node,
                                                   const uint32_t* __restrict__ root_list, uint32_t M) {
    for (uint32_t m = blockIdx.x * 256u + threadIdx.x; m < M; m += gridDim.x * 256u) {
        mem_point[m] = root_list ? root_list[m] : m;
        mem_node[m] = 0u;
    }
}

__global__ __launch_bounds__(256) void k_tree_claim(const uint32_t* __restrict__ mem_point, const uint32_t* __restrict__ mem_node,
                                                    uint32_t M, const uint32_t* __restrict__ dfs_rank, uint32_t* __restrict__ best) {
    for (uint32_t m = blockIdx.x * 256u + threadIdx.x; m < M; m += gridDim.x * 256u) {
        const uint32_t id = mem_node[m];
        if (id != TREE_DEAD) atomicMin(&best[mem_point[m]], dfs_rank[id]);
    }
}

__global__ __launch_bounds__(256) void k_tree_claimed_count(const uint32_t* __restrict__ best, uint32_t n, uint32_t* __restrict__ rank_count) {
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u)
        if (best[i] != TREE_DEAD) atomicAdd(&rank_count[best[i]], 1u);
}

__global__ __launch_bounds__(256) void k_tree_keys(const uint32_t* __restrict__ best, uint32_t n, const uint32_t* __restrict__ final_of_rank,
                                                   uint32_t n_final, uint32_t first_index, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        keys[i] = best[i] != TREE_DEAD ? final_of_rank[best[i]] : n_final;     // unclaimed points sort behind the last leaf
        vals[i] = first_index + i;
    }
}

namespace {

struct BuildNode {
    double mn[3], mx[3];
    uint32_t depth, count, first_child;        // first_child = global id of child 0, 0 = leaf
};

inline uint32_t grid_of(uint32_t n) {
    uint32_t g = (n + 255u) / 256u;
    return g < 1u ? 1u : (g > 4096u ? 4096u : g);
}

// returns GS_OK and fills t->leaves / indexes / d_indexes; GS_ERR_CAPACITY when the membership buffer overflowed (a degenerate
// scene whose points sit on split planes by the million: the caller falls back to the host builder)
int tree_build_device(gs_tree* t, gs_context* ctx, const float* centers, const std::vector<uint32_t>& root, uint32_t count,
                      uint32_t first_index) {
    ScopedDevice sd(ctx->device);
    hipStream_t st = ctx->stream;
    const uint32_t M0 = (uint32_t)root.size();
    const uint64_t cap64 = 2ull * M0 + (1ull << 21);
    if (cap64 > 0x7FFFFFFFull) return GS_ERR_CAPACITY;
    const uint32_t cap = (uint32_t)cap64;
    DevBuf pts, mem_point, mem_node, scalars, lvl_rank, planes, child_count, rank_tab, best, root_dev;
    GS_TRY(pts.alloc((size_t)count * 12));
    GS_TRY(mem_point.alloc((size_t)cap * 4));
    GS_TRY(mem_node.alloc((size_t)cap * 4));
    GS_TRY(scalars.alloc(64));                                // [0] tail, [1] overflow
    GS_HIP(hipMemcpyAsync(pts.p, centers, (size_t)count * 12, hipMemcpyHostToDevice, st));
    const bool filtered = M0 != count;
    if (filtered) {
        GS_TRY(root_dev.alloc((size_t)M0 * 4));
        GS_HIP(hipMemcpyAsync(root_dev.p, root.data(), (size_t)M0 * 4, hipMemcpyHostToDevice, st));
    }
    hipLaunchKernelGGL(k_tree_root, dim3(grid_of(M0)), dim3(256), 0, st, mem_point.as<uint32_t>(), mem_node.as<uint32_t>(),
                       filtered ? root_dev.as<uint32_t>() : nullptr, M0);
    uint32_t init[2] = {M0, 0u};
    GS_HIP(hipMemcpyAsync(scalars.p, init, 8, hipMemcpyHostToDevice, st));

    std::vector<BuildNode> nodes(1);
    for (int k = 0; k < 3; k++) { nodes[0].mn[k] = t->scene_min[k]; nodes[0].mx[k] = t->scene_max[k]; }
    nodes[0].depth = 0; nodes[0].count = M0; nodes[0].first_child = 0;
    uint32_t lvl_base = 0, lvl_nodes = 1, M = M0;
    std::vector<uint32_t> rank_host, counts_host;
    std::vector<double> planes_host;
    for (;;) {
        // which nodes of this level split (SplatTree.js:135), their planes and children (:152-182)
        rank_host.assign(lvl_nodes, TREE_DEAD);
        planes_host.clear();
        uint32_t n_split = 0;
        const uint32_t child_base = lvl_base + lvl_nodes;
        for (uint32_t k = 0; k < lvl_nodes; k++) {
            BuildNode nd = nodes[lvl_base + k];               // by value: `nodes` grows below
            if (nd.count < t->max_centers || nd.depth > t->max_depth) continue;
            rank_host[k] = n_split;
            double dim[3], half[3], c[3];
            for (int a = 0; a < 3; a++) {
                dim[a] = nd.mx[a] - nd.mn[a];
                half[a] = dim[a] * 0.5;
                c[a] = nd.mn[a] + half[a];
            }
            double pl[9];
            for (int a = 0; a < 3; a++) { pl[3 * a] = c[a] - half[a]; pl[3 * a + 1] = c[a]; pl[3 * a + 2] = c[a] + half[a]; }
            planes_host.insert(planes_host.end(), pl, pl + 9);
            nodes[lvl_base + k].first_child = child_base + 8u * n_split;
            static const int sel[8][3] = {{0, 1, 0}, {1, 1, 0}, {1, 1, 1}, {0, 1, 1}, {0, 0, 0}, {1, 0, 0}, {1, 0, 1}, {0, 0, 1}};
            for (int j = 0; j < 8; j++) {
                BuildNode ch;
                for (int a = 0; a < 3; a++) { ch.mn[a] = pl[3 * a + sel[j][a]]; ch.mx[a] = pl[3 * a + sel[j][a] + 1]; }
                ch.depth = nd.depth + 1; ch.count = 0; ch.first_child = 0;
                nodes.push_back(ch);
            }
            n_split++;
        }
        if (n_split == 0) break;
        const uint32_t n_child = 8u * n_split;
        GS_TRY(lvl_rank.ensure((size_t)lvl_nodes * 4));
        GS_TRY(planes.ensure((size_t)n_split * 72));
        GS_TRY(child_count.ensure((size_t)n_child * 4));
        GS_HIP(hipMemcpyAsync(lvl_rank.p, rank_host.data(), (size_t)lvl_nodes * 4, hipMemcpyHostToDevice, st));
        GS_HIP(hipMemcpyAsync(planes.p, planes_host.data(), (size_t)n_split * 72, hipMemcpyHostToDevice, st));
        GS_HIP(hipMemsetAsync(child_count.p, 0, (size_t)n_child * 4, st));
        if (n_child <= 4096u)      // few, hot counters: per-workgroup LDS histograms (5.8 M atomics on 8 addresses would serialise)
            hipLaunchKernelGGL(k_tree_split<true>, dim3(grid_of(M)), dim3(256), 0, st, pts.as<float>(), mem_point.as<uint32_t>(),
                               mem_node.as<uint32_t>(), M, cap, scalars.as<uint32_t>(), lvl_base, lvl_nodes, lvl_rank.as<uint32_t>(),
                               planes.as<double>(), child_base, child_count.as<uint32_t>(), n_child, scalars.as<uint32_t>() + 1);
        else
            hipLaunchKernelGGL(k_tree_split<false>, dim3(grid_of(M)), dim3(256), 0, st, pts.as<float>(), mem_point.as<uint32_t>(),
                               mem_node.as<uint32_t>(), M, cap, scalars.as<uint32_t>(), lvl_base, lvl_nodes, lvl_rank.as<uint32_t>(),
                               planes.as<double>(), child_base, child_count.as<uint32_t>(), n_child, scalars.as<uint32_t>() + 1);
        GS_HIP(hipGetLastError());
        counts_host.resize(n_child);
        uint32_t sc[2];
        GS_HIP(hipMemcpyAsync(counts_host.data(), child_count.p, (size_t)n_child * 4, hipMemcpyDeviceToHost, st));
        GS_HIP(hipMemcpyAsync(sc, scalars.p, 8, hipMemcpyDeviceToHost, st));
        GS_HIP(hipStreamSynchronize(st));
        if (sc[1]) return GS_ERR_CAPACITY;
        M = sc[0];
        for (uint32_t k = 0; k < n_child; k++) nodes[child_base + k].count = counts_host[k];
        lvl_base = child_base;
        lvl_nodes = n_child;
    }
    // depth-first visiting order of the leaf nodes = processSplatTreeNode's recursion order (children 0..7)
    const uint32_t n_nodes = (uint32_t)nodes.size();
    std::vector<uint32_t> dfs_rank(n_nodes, TREE_DEAD), leaf_ids;
    {
        std::vector<uint32_t> stack(1, 0u);
        while (!stack.empty()) {
            const uint32_t id = stack.back();
            stack.pop_back();
            if (nodes[id].first_child == 0) {
                dfs_rank[id] = (uint32_t)leaf_ids.size();
                leaf_ids.push_back(id);
            } else {
                for (int j = 7; j >= 0; j--) stack.push_back(nodes[id].first_child + (uint32_t)j);
            }
        }
    }
    const uint32_t n_leaf_nodes = (uint32_t)leaf_ids.size();
    t->nodes = n_nodes;
    t->all_leaves = n_leaf_nodes;
    // first visitor wins, then the claimed points per leaf
    GS_TRY(rank_tab.alloc((size_t)n_nodes * 4));
    GS_TRY(best.alloc((size_t)count * 4));
    GS_TRY(child_count.ensure((size_t)n_leaf_nodes * 4));
    GS_HIP(hipMemcpyAsync(rank_tab.p, dfs_rank.data(), (size_t)n_nodes * 4, hipMemcpyHostToDevice, st));
    GS_HIP(hipMemsetAsync(best.p, 0xFF, (size_t)count * 4, st));
    GS_HIP(hipMemsetAsync(child_count.p, 0, (size_t)n_leaf_nodes * 4, st));
    hipLaunchKernelGGL(k_tree_claim, dim3(grid_of(M)), dim3(256), 0, st, mem_point.as<uint32_t>(), mem_node.as<uint32_t>(), M,
                       rank_tab.as<uint32_t>(), best.as<uint32_t>());
    hipLaunchKernelGGL(k_tree_claimed_count, dim3(grid_of(count)), dim3(256), 0, st, best.as<uint32_t>(), count, child_count.as<uint32_t>());
    GS_HIP(hipGetLastError());
    std::vector<uint32_t> claimed(n_leaf_nodes);
    GS_HIP(hipMemcpyAsync(claimed.data(), child_count.p, (size_t)n_leaf_nodes * 4, hipMemcpyDeviceToHost, st));
    GS_HIP(hipStreamSynchronize(st));
    // the leaves that keep at least one index (convertWorkerSubTree), in visiting order
    std::vector<uint32_t> final_of_rank(n_leaf_nodes, 0u);
    t->leaves.clear();
    uint32_t total = 0;
    for (uint32_t r = 0; r < n_leaf_nodes; r++) {
        final_of_rank[r] = (uint32_t)t->leaves.size();
        if (claimed[r] == 0) continue;
        const BuildNode& nd = nodes[leaf_ids[r]];
        TreeLeaf leaf;
        for (int k = 0; k < 3; k++) {
            leaf.mn[k] = nd.mn[k];
            leaf.mx[k] = nd.mx[k];
            leaf.center[k] = (nd.mx[k] - nd.mn[k]) * 0.5 + nd.mn[k];                // WorkerSplatTreeNode :121-123
        }
        leaf.depth = nd.depth;
        leaf.begin = total;
        leaf.count = claimed[r];
        total += claimed[r];
        t->leaves.push_back(leaf);
    }
    const uint32_t n_final = (uint32_t)t->leaves.size();
    t->indexes.assign(total, 0u);
    if (total == 0) return GS_OK;
    // stable sort of (final leaf number, ascending point index): the leaves' index lists, concatenated in leaf order
    DevBuf kA, kB, vA, vB;
    RadixScratch scratch;
    GS_TRY(scratch.init());
    GS_TRY(kA.alloc((size_t)count * 4)); GS_TRY(kB.alloc((size_t)count * 4));
    GS_TRY(vA.alloc((size_t)count * 4)); GS_TRY(vB.alloc((size_t)count * 4));
    GS_HIP(hipMemcpyAsync(rank_tab.p, final_of_rank.data(), (size_t)n_leaf_nodes * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_tree_keys, dim3(grid_of(count)), dim3(256), 0, st, best.as<uint32_t>(), count, rank_tab.as<uint32_t>(), n_final,
                       first_index, kA.as<uint32_t>(), vA.as<uint32_t>());
    GS_HIP(hipMemsetAsync(scratch.digit_total.p, 0, sizeof(uint32_t) * RADIX_TOTAL_WORDS, st));
    uint32_t bits = 1;
    while ((1ull << bits) <= n_final) bits++;                 // keys 0 .. n_final
    const uint32_t passes = (bits + 7) / 8;
    uint32_t* kbuf[2] = {kA.as<uint32_t>(), kB.as<uint32_t>()};
    uint32_t* vbuf[2] = {vA.as<uint32_t>(), vB.as<uint32_t>()};
    const RadixExec ex = {st, &scratch, ctx->lds_atomic_lane_order};
    for (uint32_t pass = 0; pass < passes; pass++) {
        ArrayLoader<uint32_t> al = {kbuf[pass & 1], vbuf[pass & 1], nullptr, count};
        GS_TRY((radix_pass<ArrayLoader<uint32_t>, uint32_t, true>(ex, al, al, count, 8 * (int)pass, (int)pass, kbuf[(pass + 1) & 1],
                                                                 vbuf[(pass + 1) & 1])));
    }
    GS_TRY(t->d_indexes.alloc(4 * (size_t)total + 4));
    GS_HIP(hipMemcpyAsync(t->d_indexes.p, vbuf[passes & 1], (size_t)total * 4, hipMemcpyDeviceToDevice, st));
    GS_HIP(hipMemcpyAsync(t->indexes.data(), vbuf[passes & 1], (size_t)total * 4, hipMemcpyDeviceToHost, st));
    GS_HIP(hipStreamSynchronize(st));
    return GS_OK;
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

// Viewer.js:2010-2035 for one leaf: the sort key (distance, or +inf when the leaf is culled)
__device__ __forceinline__ double tree_leaf_key(const GatherParams& p, double x, double y, double z, double size) {
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
    const bool skip = !p.gather_all && ((out_x || out_y) && dist > size);
    return skip ? __longlong_as_double(0x7FF0000000000000ll) : dist;         // +inf: the leaf takes no part in the ranking
}

// THE PLAN (round 4).  What the copy needs per kept leaf is only WHERE its index list goes: the reference lays the kept leaves
// out far -> near (Viewer.js:2046-2055 copies from the END backwards), so leaf i starts at
//     offset(i) = total - before(i) - count(i),   before(i) = sum of count(j) over the kept leaves j with (key_j, j) < (key_i, i)
// - a count-WEIGHTED rank.  No leaf-by-rank arrays, no second prefix over the ranks: rounds 2-3 built those in one launch of 256
// resident workgroups with five hand-rolled grid barriers (13 us each on MI355X: 67 us per gather, and a spin barrier that
// needs every workgroup resident - ADVICE r03); now four small launches do it (a kernel boundary is ~2 us):
//   k_tree_test     every leaf: its key (fp64 distance, +inf when culled) and ONE 64-bit atomic on its bucket's counter
//                   (members << 40 | splats).  Buckets are LINEAR in the distance: floor(key * scale) with scale from the largest
//                   distance any leaf centre can have (the farthest corner of the scene box, computed on the host for this
//                   modelView); monotonic in the key whatever scale is, so the order below is exact - scale only spreads the
//                   leaves (2^18 buckets over ~26 k leaves: mostly one per bucket; the float-bit buckets of round 3 put the
//                   leaves 10 units away six to a bucket)
//   k_tree_scan     256 workgroups x 1024 buckets: exclusive scan inside the chunk (both fields at once), chunk totals
//   k_tree_fill     chunk bases (every workgroup scans the 256 totals in LDS), then members[base + local + fill++] = leaf
//   k_tree_offsets  before(i) = the scan's splat field + the splats of the members of its own bucket that precede it on the
//                   full fp64 key and the leaf number -> offset(i); totals; the fused cull's keep words go back to zero
// (the counters are handed back zeroed: hist by k_tree_fill, fill by the NEXT gather's k_tree_test)
// Deterministic (integer sums; the order of a bucket's members is arbitrary but only their set matters).
constexpr uint32_t TREE_BUCKET_BITS = 18;
constexpr uint32_t TREE_BUCKETS = 1u << TREE_BUCKET_BITS;
constexpr uint32_t PLAN_GRID = 256, PLAN_THREADS = 256;
constexpr uint32_t PLAN_CHUNK = TREE_BUCKETS / PLAN_GRID;          // buckets per workgroup of the scan
constexpr uint32_t PLAN_PER_THREAD = PLAN_CHUNK / PLAN_THREADS;    // ... and per thread: consecutive ones
static_assert(PLAN_PER_THREAD * PLAN_THREADS * PLAN_GRID == TREE_BUCKETS && PLAN_PER_THREAD % 2 == 0, "plan geometry");
constexpr unsigned long long TREE_INF = 0x7FF0000000000000ull;
constexpr unsigned long long TREE_W_MASK = (1ull << 40) - 1ull;    // low 40 bits: splats; above: members
constexpr uint32_t TREE_CULLED = 0xFFFFFFFFu;

__device__ __forceinline__ uint32_t tree_bucket(double key, double scale) {
    const double b = key * scale;                                   // key >= 0 and finite here
    return b < (double)(TREE_BUCKETS - 1u) ? (uint32_t)b : TREE_BUCKETS - 1u;
}

struct PlanBuffers {
    const double* center; const double* size; const uint32_t* count;      // per leaf
    unsigned long long* key;                                              // per leaf: the bits of the distance (sortable: >= 0)
    unsigned long long* hist;                                             // per bucket: members << 40 | splats; zero between gathers
    unsigned long long* start;                                            // per bucket: the same two fields, exclusive inside its chunk
    uint32_t* fill;                                                       // per bucket: members placed so far; zero between gathers
    unsigned long long* chunk_sum;                                        // [PLAN_GRID]
    uint32_t* members;                                                    // kept leaves grouped by bucket
    uint32_t* offset;                                                     // per leaf: first slot of its list, TREE_CULLED when culled
    uint32_t* totals; uint32_t* count_out;                                // {splats gathered, leaves kept}; the sorter's copy of the first
    unsigned long long* keep_zero; uint32_t keep_words;                   // the sorter's keep mask of a fused per-splat cull (nullable)
};

__global__ __launch_bounds__(PLAN_THREADS) void k_tree_test(GatherParams p, PlanBuffers B, double scale, double prev_scale) {
    const uint32_t i = blockIdx.x * PLAN_THREADS + threadIdx.x;
    if (i >= p.leaves) return;
    // the member counters of the PREVIOUS gather go back to zero first (its keys are still here; +inf before the first gather)
    const unsigned long long old = B.key[i];
    if (old != TREE_INF) B.fill[tree_bucket(__longlong_as_double((long long)old), prev_scale)] = 0u;
    const double k = tree_leaf_key(p, B.center[3 * (size_t)i], B.center[3 * (size_t)i + 1], B.center[3 * (size_t)i + 2], B.size[i]);
    const unsigned long long kb = (unsigned long long)__double_as_longlong(k);       // non-negative doubles order like their bits
    B.key[i] = kb;
    if (kb != TREE_INF) atomicAdd(&B.hist[tree_bucket(k, scale)], (1ull << 40) | (unsigned long long)B.count[i]);
}

__global__ __launch_bounds__(PLAN_THREADS) void k_tree_scan(PlanBuffers B) {
    __shared__ unsigned long long s_w[4];
    const uint32_t tid = threadIdx.x, wg = blockIdx.x, lane = tid & 63u, wave = tid >> 6;
    ulonglong2* hp = reinterpret_cast<ulonglong2*>(B.hist + (size_t)wg * PLAN_CHUNK + tid * PLAN_PER_THREAD);
    ulonglong2* sp = reinterpret_cast<ulonglong2*>(B.start + (size_t)wg * PLAN_CHUNK + tid * PLAN_PER_THREAD);
    unsigned long long v[PLAN_PER_THREAD], sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < PLAN_PER_THREAD / 2; k++) {
        const ulonglong2 q = hp[k];
        v[2 * k] = q.x; v[2 * k + 1] = q.y;
        sum += q.x + q.y;
    }
    unsigned long long incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long t = __shfl_up(incl, o, 64);
        if ((int)lane >= o) incl += t;
    }
    if (lane == 63u) s_w[wave] = incl;
    __syncthreads();
    unsigned long long run = incl - sum, total = 0;
#pragma unroll
    for (uint32_t w = 0; w < 4; w++) {
        const unsigned long long c = s_w[w];
        run += w < wave ? c : 0ull;
        total += c;
    }
#pragma unroll
    for (uint32_t k = 0; k < PLAN_PER_THREAD / 2; k++) {
        ulonglong2 o;
        o.x = run; run += v[2 * k];
        o.y = run; run += v[2 * k + 1];
        sp[k] = o;
    }
    if (tid == 0) B.chunk_sum[wg] = total;
}

// exclusive prefix of the PLAN_GRID chunk totals into LDS (thread t owns chunk t); returns the grand total
__device__ __forceinline__ unsigned long long plan_chunk_bases(const unsigned long long* chunk_sum, unsigned long long* s_base, unsigned long long* s_w) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const unsigned long long c = chunk_sum[tid];
    unsigned long long incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long t = __shfl_up(incl, o, 64);
        if ((int)lane >= o) incl += t;
    }
    if (lane == 63u) s_w[wave] = incl;
    __syncthreads();
    unsigned long long base = incl - c, total = 0;
#pragma unroll
    for (uint32_t w = 0; w < 4; w++) {
        const unsigned long long t = s_w[w];
        base += w < wave ? t : 0ull;
        total += t;
    }
    s_base[tid] = base;
    __syncthreads();
    return total;
}

__global__ __launch_bounds__(PLAN_THREADS) void k_tree_fill(GatherParams p, PlanBuffers B, double scale) {
    __shared__ unsigned long long s_base[PLAN_GRID], s_w[4];
    (void)plan_chunk_bases(B.chunk_sum, s_base, s_w);
    const uint32_t i = blockIdx.x * PLAN_THREADS + threadIdx.x;
    if (i >= p.leaves) return;
    const unsigned long long kb = B.key[i];
    if (kb == TREE_INF) return;
    const uint32_t b = tree_bucket(__longlong_as_double((long long)kb), scale);
    const uint32_t first = (uint32_t)((s_base[b / PLAN_CHUNK] + B.start[b]) >> 40);
    B.members[first + atomicAdd(&B.fill[b], 1u)] = i;                 // the order inside a bucket is arbitrary
    B.hist[b] = 0ull;                                                  // consumed by the scan: zero for the next gather
}

__global__ __launch_bounds__(PLAN_THREADS) void k_tree_offsets(GatherParams p, PlanBuffers B, double scale) {
    __shared__ unsigned long long s_base[PLAN_GRID], s_w[4];
    const unsigned long long all = plan_chunk_bases(B.chunk_sum, s_base, s_w);
    const uint32_t total = (uint32_t)(all & TREE_W_MASK);
    const uint32_t gid = blockIdx.x * PLAN_THREADS + threadIdx.x;
    if (gid == 0) {
        B.totals[0] = total;
        B.totals[1] = (uint32_t)(all >> 40);
        if (B.count_out) *B.count_out = total;
    }
    // the keep mask of a fused per-splat cull is filled with atomicOr by the copy: zero it here
    for (uint32_t w = gid; w < B.keep_words; w += gridDim.x * PLAN_THREADS) B.keep_zero[w] = 0ull;
    if (gid >= p.leaves) return;
    const uint32_t i = gid;
    const unsigned long long mine = B.key[i];
    if (mine == TREE_INF) {
        B.offset[i] = TREE_CULLED;
        return;
    }
    const uint32_t b = tree_bucket(__longlong_as_double((long long)mine), scale);
    const unsigned long long at = s_base[b / PLAN_CHUNK] + B.start[b];
    const uint32_t lo = (uint32_t)(at >> 40), n = B.fill[b];         // (complete: k_tree_fill has finished)
    uint32_t before = (uint32_t)(at & TREE_W_MASK);
    for (uint32_t q = 0; q < n; q += 8u) {
        uint32_t m[8], c[8];
        unsigned long long k[8];
#pragma unroll
        for (uint32_t j = 0; j < 8u; j++) m[j] = q + j < n ? B.members[lo + q + j] : i;      // padding = the leaf itself: adds 0
#pragma unroll
        for (uint32_t j = 0; j < 8u; j++) { k[j] = B.key[m[j]]; c[j] = B.count[m[j]]; }
#pragma unroll
        for (uint32_t j = 0; j < 8u; j++) before += (k[j] < mine || (k[j] == mine && m[j] < i)) ? c[j] : 0u;
    }
    B.offset[i] = total - before - B.count[i];
}

// Coalesced copy of the kept leaves' index lists (<= ~1000 indexes each) to their places in indexesToSort: one wave per
// leaf, four leaves per workgroup.  (A sorter that sorts the whole list of a static scene does this copy itself, fused with
// its key kernel: sorter.hip, k_tree_copy_keys.)
constexpr uint32_t COPY_WAVES = 4;
__global__ __launch_bounds__(64 * COPY_WAVES) void k_tree_copy(uint32_t leaves, const uint32_t* __restrict__ leaf_offset,
                                                                const uint32_t* __restrict__ leaf_count,
                                                                const uint32_t* __restrict__ leaf_begin,
                                                                const uint32_t* __restrict__ leaf_indexes, uint32_t* __restrict__ out) {
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t i = blockIdx.x * COPY_WAVES + (threadIdx.x >> 6); i < leaves; i += gridDim.x * COPY_WAVES) {
        const uint32_t off = leaf_offset[i];
        if (off == TREE_CULLED) continue;
        const uint32_t n = leaf_count[i];
        const uint32_t* src = leaf_indexes + leaf_begin[i];
        uint32_t* dst = out + off;
        for (uint32_t t = lane; t < n; t += 64u) dst[t] = src[t];
    }
}

static uint64_t g_tree_uid = 0;

extern "C" {

int gs_tree_create(gs_context* ctx, const float* centers, const uint8_t* keep, uint32_t count, uint32_t first_index,
                   uint32_t max_depth, uint32_t max_centers_per_node, gs_tree** out) {
    GS_REQUIRE(out && (centers || count == 0), "out / centers == NULL");
    *out = nullptr;
    GS_REQUIRE(max_centers_per_node > 0, "max_centers_per_node == 0");
    gs_tree* t = new (std::nothrow) gs_tree();
    if (!t) return GS_ERR_NOMEM;
    t->ctx = ctx;
    t->uid = ++g_tree_uid;
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
        // with a context: level-synchronous build on the device (same leaves, see tree_build_device); the recursive host
        // builder serves host-only trees, empty inputs and the degenerate case of a membership buffer that overflows
        int dev = GS_ERR_CAPACITY;
        if (ctx && !root.empty() && !getenv("GSPLAT_TREE_HOST_BUILD")) {
            dev = tree_build_device(t, ctx, centers, root, count, first_index);
            if (dev < 0 && dev != GS_ERR_CAPACITY) {
                delete t;
                return dev;
            }
        }
        t->built_on_device = dev == GS_OK;
        if (!t->built_on_device) {
            t->leaves.clear(); t->indexes.clear();
            t->nodes = t->all_leaves = 0;
            BuildCtx b = {centers, first_index, max_depth, max_centers_per_node, std::vector<uint8_t>(count, 0), t};
            Box box;
            for (int k = 0; k < 3; k++) {
                box.mn[k] = t->scene_min[k];
                box.mx[k] = t->scene_max[k];
            }
            process_node(b, box, 0, root);
        }
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
        if (!t->built_on_device) A(t->d_indexes, 4 * t->indexes.size() + 4);      // the device build left them there
        A(t->d_key, 8 * L + 8); A(t->d_rank, 4 * L + 4); A(t->d_offset, 4 * L + 4);
        A(t->d_bucket, (size_t)TREE_BUCKETS * (8 + 8 + 4) + 8 * PLAN_GRID);
        A(t->d_total, 64);                        // {splats gathered, kept leaves}
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
        UP(t->d_count, cnt.data(), 4 * L);
        if (!t->built_on_device) UP(t->d_indexes, t->indexes.data(), 4 * t->indexes.size());
        if (e == hipSuccess) e = hipMemsetAsync(t->d_total.p, 0, 64, s);
        if (e == hipSuccess) e = hipMemsetAsync(t->d_bucket.p, 0, (size_t)TREE_BUCKETS * (8 + 8 + 4) + 8 * PLAN_GRID, s);   // the gather kernels keep hist / fill zero
        std::vector<unsigned long long> inf(L + 1, TREE_INF);          // "no previous gather": nothing for k_tree_test to reset
        UP(t->d_key, inf.data(), 8 * L);
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
    if (t->pending_sorter) {                               // a sorter still waits to copy this tree's last gather: it has no list
        t->pending_sorter->pending_tree = nullptr;
        t->pending_sorter->has_gathered = false;
        t->pending_sorter = nullptr;
    }
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
    GS_REQUIRE(t && gp, "tree / params == NULL");
    GS_REQUIRE(render_count || (dst && !indexes_out_host), "render_count == NULL (asynchronous gather) needs a sorter and no host copy");
    GS_REQUIRE(t->ctx != nullptr, "host-only tree (created without a context) cannot gather");
    GS_REQUIRE(!dst || dst->ctx == t->ctx, "sorter lives on another context");
    GS_REQUIRE(!dst || dst->max_count >= t->indexes.size(), "sorter is smaller than the tree");
    gs_context* ctx = t->ctx;
    ScopedDevice sd(ctx->device);
    hipStream_t st = dst ? dst->stream : ctx->stream;      // the list is produced where the sort will consume it
    const uint32_t L = (uint32_t)t->leaves.size();
    // another sorter has not yet copied the previous gather, whose offsets are about to be overwritten: it copies now
    if (t->pending_sorter && t->pending_sorter != dst) {
        gs_sorter* o = t->pending_sorter;
        GS_TRY(o->idx_in.ensure((size_t)o->max_count * 4));
        GS_TRY(gs_tree_copy_plain(t, o->idx_in.as<uint32_t>(), o->stream));
        GS_HIP(hipStreamSynchronize(o->stream));
        gs_tree_forget_sorter(t, o);
    }
    if (dst && dst->pending_tree && dst->pending_tree != t) gs_tree_forget_sorter(dst->pending_tree, dst);   // superseded
    uint32_t* out_dev = nullptr;
    if (dst) {
        GS_TRY(dst->idx_in.ensure((size_t)dst->max_count * 4));
        out_dev = dst->idx_in.as<uint32_t>();
    } else {
        GS_TRY(t->d_out.ensure(4 * t->indexes.size() + 4));
        out_dev = t->d_out.as<uint32_t>();
    }
    if (render_count) *render_count = 0;
    if (L == 0) {
        if (dst) dst->gathered = 0, dst->gathered_on_device = false, dst->has_gathered = true;
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
    // bucket scale: the farthest a leaf centre can be from the eye = the farthest corner of the scene box under this modelView
    // (the distance is convex in the point); the order never depends on it (tree_bucket clamps), only how the leaves spread
    double far = 0.0;
    for (int c = 0; c < 8; c++) {
        const double x = (c & 1) ? t->scene_max[0] : t->scene_min[0], y = (c & 2) ? t->scene_max[1] : t->scene_min[1],
                     z = (c & 4) ? t->scene_max[2] : t->scene_min[2];
        const double* e = p.mv;
        double w = e[3] * x + e[7] * y + e[11] * z + e[15];
        w = w != 0.0 ? 1.0 / w : 1.0;
        const double vx = (e[0] * x + e[4] * y + e[8] * z + e[12]) * w, vy = (e[1] * x + e[5] * y + e[9] * z + e[13]) * w,
                     vz = (e[2] * x + e[6] * y + e[10] * z + e[14]) * w;
        const double d = sqrt(vx * vx + vy * vy + vz * vz);
        if (d > far) far = d;
    }
    const double scale = (far > 0.0 && far < 1e300) ? (double)TREE_BUCKETS / (far * 1.0001) : 1.0;
    // the sorter gets the list's length on the device as well
    uint32_t* count_dev = nullptr;
    if (dst) {
        GS_TRY(dst->gathered_dev.ensure(16));
        count_dev = dst->gathered_dev.as<uint32_t>();
    }
    // The copy is DEFERRED to the sorter when nothing but the sorter will read the list: a full sort of a static scene then
    // copies the lists, keys them (and tests them against the frustum) in one kernel that streams the centres in leaf order
    // (sorter.hip, k_tree_copy_keys); any other sort copies first (gs_tree_copy_plain) and goes on as before.
    const bool deferred = dst && !indexes_out_host && !(dst->flags & GS_SORT_DYNAMIC) && !getenv("GSPLAT_TREE_NO_DEFER");
    PlanBuffers pb;
    pb.center = t->d_center.as<double>(); pb.size = t->d_size.as<double>(); pb.count = t->d_count.as<uint32_t>();
    pb.key = t->d_key.as<unsigned long long>();
    pb.hist = t->d_bucket.as<unsigned long long>();
    pb.start = pb.hist + TREE_BUCKETS;
    pb.chunk_sum = pb.start + TREE_BUCKETS;
    pb.fill = reinterpret_cast<uint32_t*>(pb.chunk_sum + PLAN_GRID);
    pb.members = t->d_rank.as<uint32_t>(); pb.offset = t->d_offset.as<uint32_t>();
    pb.totals = t->d_total.as<uint32_t>(); pb.count_out = count_dev;
    pb.keep_zero = nullptr; pb.keep_words = 0;
    if (deferred && dst->frustum_cull) {                   // the fused copy ORs its keep bits into a zeroed mask
        GS_TRY(dst->keep_mask.ensure((((size_t)dst->max_count + 63) / 64 + 8) * 8));
        pb.keep_zero = dst->keep_mask.as<unsigned long long>();
        pb.keep_words = (uint32_t)(((size_t)t->indexes.size() + 63) / 64 + 2);
    }
    const uint32_t lgrid = (L + PLAN_THREADS - 1u) / PLAN_THREADS;
    hipLaunchKernelGGL(k_tree_test, dim3(lgrid), dim3(PLAN_THREADS), 0, st, p, pb, scale, t->prev_scale);
    hipLaunchKernelGGL(k_tree_scan, dim3(PLAN_GRID), dim3(PLAN_THREADS), 0, st, pb);
    hipLaunchKernelGGL(k_tree_fill, dim3(lgrid), dim3(PLAN_THREADS), 0, st, p, pb, scale);
    hipLaunchKernelGGL(k_tree_offsets, dim3(lgrid), dim3(PLAN_THREADS), 0, st, p, pb, scale);
    GS_HIP(hipGetLastError());
    t->prev_scale = scale;
    t->gather_serial++;
    if (deferred) {
        dst->pending_tree = t;
        dst->pending_keep_zeroed = pb.keep_zero != nullptr;
        t->pending_sorter = dst;
    } else {
        GS_TRY(gs_tree_copy_plain(t, out_dev, st));
        if (dst && dst->pending_tree == t) gs_tree_forget_sorter(t, dst);
    }
    if (!render_count) {
        // asynchronous: nothing returns to the host, splatRenderCount stays on the device next to the list; the sorter takes
        // both from there (gs_sorter_sort_gathered), the draw takes the sorted list's length from the sorter
        dst->gathered = (uint32_t)t->indexes.size();          // upper bound: sizes grids and buffers
        dst->gathered_on_device = true;
        dst->has_gathered = true;
        return GS_OK;
    }
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
        dst->gathered_on_device = false;
        dst->has_gathered = true;
    }
    return GS_OK;
}

}  // extern "C"

void gs_tree_view(gs_tree* t, TreeGatherView* v) {
    v->tree_uid = t->uid;
    v->leaves = (uint32_t)t->leaves.size();
    v->tree_splats = (uint32_t)t->indexes.size();
    v->leaf_offset = t->d_offset.as<uint32_t>();
    v->leaf_count = t->d_count.as<uint32_t>();
    v->leaf_begin = t->d_begin.as<uint32_t>();
    v->leaf_indexes = t->d_indexes.as<uint32_t>();
    v->totals = t->d_total.as<uint32_t>();
}

int gs_tree_copy_plain(gs_tree* t, uint32_t* out_dev, hipStream_t st) {
    const uint32_t L = (uint32_t)t->leaves.size();
    if (L == 0) return GS_OK;
    const uint32_t copy_grid = (L + COPY_WAVES - 1u) / COPY_WAVES;
    hipLaunchKernelGGL(k_tree_copy, dim3(copy_grid < 8192u ? copy_grid : 8192u), dim3(64 * COPY_WAVES), 0, st, L, t->d_offset.as<uint32_t>(),
                       t->d_count.as<uint32_t>(), t->d_begin.as<uint32_t>(), t->d_indexes.as<uint32_t>(), out_dev);
    GS_HIP(hipGetLastError());
    return GS_OK;
}

void gs_tree_forget_sorter(gs_tree* t, gs_sorter* s) {
    if (t && t->pending_sorter == s) t->pending_sorter = nullptr;
    if (s && s->pending_tree == t) s->pending_tree = nullptr;
}
