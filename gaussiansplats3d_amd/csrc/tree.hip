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
    uint32_t barrier_epoch = 0;     // launches of k_tree_plan so far (its grid barriers count arrivals on one monotonic word)
    double scene_min[3] = {0, 0, 0}, scene_max[3] = {0, 0, 0};
    std::vector<TreeLeaf> leaves;          // nodesWithIndexes order
    std::vector<uint32_t> indexes;
    // device mirror
    DevBuf d_center, d_size, d_begin, d_count, d_indexes;
    DevBuf d_bucket;            // uint32 [hist 65536 | fill 65536 | start 65537]; hist / fill are zero between gathers
    DevBuf d_key, d_rank, d_sorted_cnt, d_sorted_leaf, d_offset, d_total, d_out;   // d_total: {splats gathered, leaves kept}
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

// bucket of a distance key: top 20 bits of (float)key (8 exponent + 11 mantissa bits: 0.05 % wide), monotonic in key; any bit
// pattern stays below TREE_BUCKETS.  (r03g: with 16-bit buckets - 0.8 % wide - the leaves 10 units from the camera share a
// bucket by the hundred and the slowest thread's exact ranking, a chain of dependent loads, set the plan's length: 113 us.)
constexpr uint32_t TREE_BUCKET_BITS = 20;
constexpr uint32_t TREE_BUCKETS = 1u << TREE_BUCKET_BITS;
__device__ __forceinline__ uint32_t tree_bucket(double key) { return __float_as_uint((float)key) >> (32u - TREE_BUCKET_BITS); }

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

// Rank of a leaf in ascending (distance, leaf number) order.  An all-pairs count is O(leaves^2); instead the leaves are
// bucketed by the top 20 bits of (float)distance (monotonic in the distance: 8 exponent + 11 mantissa bits, i.e. 0.05 % wide
// buckets), the buckets are scanned, and a leaf is ranked exactly - on the full fp64 key and its number - only against the
// members of its own bucket.  The result is the same total order; only the work is smaller.
//
// ONE launch plans the whole gather: PLAN_GRID workgroups (one per CU, all resident) pass through six phases separated by
// grid-wide barriers (a monotonic arrival counter in global memory; no workgroup waits for anything but "everybody has
// arrived", and every workgroup is resident, so there is no circular wait).  The phases are a handful of microseconds of
// latency-bound work each; as separate launches they cost 8-60 us apiece (r03f profile: test 12, a one-workgroup scan of
// the 65536 counters 44, fill 13, rank 31, a one-workgroup offset scan 62 = 181 us per gather with the copy), and as ONE
// workgroup of 1024 threads the ranking alone was 1.3 ms (26 k leaves x ~55 bucket members = a serial chain of dependent L2
// loads per thread).
//   1 test     every leaf: distance key (or +inf when culled), one atomic on its bucket's counter
//   2 scan     workgroup g owns a chunk of 4096 buckets: exclusive scan inside the chunk -> start_local, chunk total;
//              the histogram is handed back zeroed
//   3 fill     chunk bases = scan of the 256 chunk totals (every workgroup, in LDS); members[base + start_local + fill++]
//   4 rank     exact rank inside the bucket (its members read eight at a time) -> leaf and count at their rank
//   5 sums     workgroup g owns a contiguous run of ranks: its count total
//   6 offsets  offset[r] = total - inclusive_prefix(counts by rank)[r]: the nearest leaf, r = 0, ends the buffer
//              (Viewer.js:2046-2055 copies from the END backwards); totals = {splats gathered, leaves kept}; count_out =
//              the sorter's own copy of splatRenderCount (nullable); the fill counters are handed back zeroed
constexpr uint32_t PLAN_GRID = 256, PLAN_THREADS = 256;
constexpr uint32_t PLAN_BARRIERS = 5;      // grid barriers per launch: the arrival counter advances by exactly PLAN_BARRIERS * PLAN_GRID
constexpr uint32_t PLAN_CHUNK = TREE_BUCKETS / PLAN_GRID;          // buckets per workgroup of the plan
constexpr uint32_t PLAN_PER_THREAD = PLAN_CHUNK / PLAN_THREADS;    // ... and per thread: 16 consecutive ones
static_assert(PLAN_PER_THREAD * PLAN_THREADS * PLAN_GRID == TREE_BUCKETS && PLAN_PER_THREAD % 4 == 0, "plan geometry");

// arrivals count on `counter`; the last one to arrive publishes the target in `flag` (another cache line), which is what the
// others poll - 255 pollers on the counter's own line would queue in front of the remaining arrivals' atomics
__device__ __forceinline__ void plan_grid_barrier(uint32_t* counter, uint32_t* flag, uint32_t target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();                                    // this workgroup's writes are visible before it arrives
        const uint32_t prev = atomicAdd(counter, 1u);
        if (prev + 1u == target) {
            __hip_atomic_store(flag, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while ((int32_t)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) __builtin_amdgcn_s_sleep(8);
        }
        __threadfence();                                    // ... and the others' writes before anybody here reads them
    }
    __syncthreads();
}

struct PlanBuffers {
    const double* center; const double* size; const uint32_t* count;      // per leaf
    unsigned long long* key;                                              // per leaf: sortable bits of the distance
    uint32_t* hist; uint32_t* fill; uint32_t* start_local;                // per bucket
    uint32_t* chunk_sum;                                                  // [2][PLAN_GRID]: bucket chunks | rank chunks
    uint32_t* members; uint32_t* sorted_cnt; uint32_t* sorted_leaf; uint32_t* offset;   // per kept leaf / rank
    uint32_t* totals; uint32_t* count_out; uint32_t* barrier;
};

__global__ __launch_bounds__(PLAN_THREADS) void k_tree_plan(GatherParams p, PlanBuffers B, uint32_t barrier_base) {
    __shared__ uint32_t s_tmp[4];
    __shared__ uint32_t s_base[PLAN_GRID];
    const uint32_t tid = threadIdx.x, wg = blockIdx.x, L = p.leaves;
    const uint32_t gid = wg * PLAN_THREADS + tid, stride = PLAN_GRID * PLAN_THREADS;
    const unsigned long long INF = 0x7FF0000000000000ull;
    uint32_t phase = 0;
    auto barrier = [&]() { plan_grid_barrier(B.barrier, B.barrier + 32, barrier_base + (++phase) * PLAN_GRID); };
    // 1. test
    for (uint32_t i = gid; i < L; i += stride) {
        const double k = tree_leaf_key(p, B.center[3 * (size_t)i], B.center[3 * (size_t)i + 1], B.center[3 * (size_t)i + 2], B.size[i]);
        const unsigned long long kb = (unsigned long long)__double_as_longlong(k);   // non-negative doubles order like their bits
        B.key[i] = kb;
        if (kb != INF) atomicAdd(&B.hist[tree_bucket(k)], 1u);                      // culled leaves take no part in the ranking
    }
    barrier();
    // 2. scan inside this workgroup's chunk of buckets (thread t owns PLAN_PER_THREAD consecutive ones, 16-byte accesses)
    {
        uint4* hp = reinterpret_cast<uint4*>(B.hist + (size_t)wg * PLAN_CHUNK + tid * PLAN_PER_THREAD);
        uint4* sp = reinterpret_cast<uint4*>(B.start_local + (size_t)wg * PLAN_CHUNK + tid * PLAN_PER_THREAD);
        uint32_t v[PLAN_PER_THREAD], sum = 0;
#pragma unroll
        for (uint32_t k = 0; k < PLAN_PER_THREAD / 4; k++) {
            const uint4 q = hp[k];
            v[4 * k] = q.x; v[4 * k + 1] = q.y; v[4 * k + 2] = q.z; v[4 * k + 3] = q.w;
            sum += q.x + q.y + q.z + q.w;
            if (q.x | q.y | q.z | q.w) hp[k] = make_uint4(0u, 0u, 0u, 0u);
        }
        uint32_t total;
        uint32_t run = block_excl_scan_256(sum, s_tmp, &total);
#pragma unroll
        for (uint32_t k = 0; k < PLAN_PER_THREAD / 4; k++) {
            uint4 o;
            o.x = run; run += v[4 * k];
            o.y = run; run += v[4 * k + 1];
            o.z = run; run += v[4 * k + 2];
            o.w = run; run += v[4 * k + 3];
            sp[k] = o;
        }
        if (tid == 0) B.chunk_sum[wg] = total;
    }
    barrier();
    // 3. chunk bases (every workgroup scans the 256 chunk totals), then the members of every bucket
    uint32_t K;
    {
        const uint32_t c = B.chunk_sum[tid];
        const uint32_t excl = block_excl_scan_256(c, s_tmp, &K);       // K = kept leaves
        s_base[tid] = excl;
        __syncthreads();
    }
    for (uint32_t i = gid; i < L; i += stride) {
        const unsigned long long kb = B.key[i];
        if (kb == INF) continue;
        const uint32_t b = tree_bucket(__longlong_as_double((long long)kb));
        B.members[s_base[b / PLAN_CHUNK] + B.start_local[b] + atomicAdd(&B.fill[b], 1u)] = i;   // order inside a bucket is arbitrary
    }
    barrier();
    // 4. exact rank inside the bucket -> the leaf and its count at their rank
    for (uint32_t i = gid; i < L; i += stride) {
        const unsigned long long mine = B.key[i];
        if (mine == INF) continue;                             // culled: has no rank
        const uint32_t b = tree_bucket(__longlong_as_double((long long)mine));
        const uint32_t lo = s_base[b / PLAN_CHUNK] + B.start_local[b], hi = lo + B.fill[b];
        uint32_t r = lo;
        for (uint32_t q = lo; q < hi; q += 8u) {
            uint32_t m[8];
            unsigned long long k[8];
#pragma unroll
            for (uint32_t j = 0; j < 8u; j++) m[j] = q + j < hi ? B.members[q + j] : i;   // padding = the leaf itself: counts 0
#pragma unroll
            for (uint32_t j = 0; j < 8u; j++) k[j] = B.key[m[j]];
#pragma unroll
            for (uint32_t j = 0; j < 8u; j++) r += (k[j] < mine || (k[j] == mine && m[j] < i)) ? 1u : 0u;
        }
        B.sorted_cnt[r] = B.count[i];
        B.sorted_leaf[r] = i;
    }
    barrier();
    // 5. this workgroup's run of ranks [r0, r1): its total
    const uint32_t per = (K + PLAN_GRID - 1u) / PLAN_GRID;
    const uint32_t r0 = min(wg * per, K), r1 = min(r0 + per, K);
    {
        uint32_t sum = 0;
        for (uint32_t r = r0 + tid; r < r1; r += PLAN_THREADS) sum += B.sorted_cnt[r];
        uint32_t total;
        (void)block_excl_scan_256(sum, s_tmp, &total);
        if (tid == 0) B.chunk_sum[PLAN_GRID + wg] = total;
    }
    barrier();
    // (phase == PLAN_BARRIERS here)
    // 6. offsets of this workgroup's ranks; the fill counters go back to zero (nobody reads them any more)
    {
        const uint32_t c = B.chunk_sum[PLAN_GRID + tid];
        uint32_t total;
        const uint32_t excl = block_excl_scan_256(c, s_tmp, &total);
        __syncthreads();
        s_base[tid] = excl;
        __syncthreads();
        uint32_t run = s_base[wg];
        if (gid == 0) {
            B.totals[0] = total;
            B.totals[1] = K;
            if (B.count_out) *B.count_out = total;
        }
        for (uint32_t rb = r0; rb < r1; rb += PLAN_THREADS) {
            const uint32_t r = rb + tid;
            const uint32_t v = r < r1 ? B.sorted_cnt[r] : 0u;
            uint32_t round_total;
            const uint32_t ex = block_excl_scan_256(v, s_tmp, &round_total);
            if (r < r1) B.offset[r] = total - (run + ex + v);
            run += round_total;
        }
        for (uint32_t i = gid; i < L; i += stride) {
            const unsigned long long kb = B.key[i];
            if (kb != INF) B.fill[tree_bucket(__longlong_as_double((long long)kb))] = 0u;    // same value from every member
        }
    }
}

// Coalesced copy of the kept leaves' index lists (<= ~1000 indexes each) to their places in indexesToSort: one wave per
// rank, four ranks per workgroup; ranks >= the kept count (read on the device) have nothing to copy.
constexpr uint32_t COPY_WAVES = 4;
__global__ __launch_bounds__(64 * COPY_WAVES) void k_tree_copy(const uint32_t* __restrict__ totals, const uint32_t* __restrict__ sorted_leaf,
                                                                const uint32_t* __restrict__ sorted_cnt, const uint32_t* __restrict__ offset,
                                                                const uint32_t* __restrict__ leaf_begin,
                                                                const uint32_t* __restrict__ leaf_indexes, uint32_t* __restrict__ out) {
    const uint32_t K = totals[1], lane = threadIdx.x & 63u;
    for (uint32_t r = blockIdx.x * COPY_WAVES + (threadIdx.x >> 6); r < K; r += gridDim.x * COPY_WAVES) {
        const uint32_t n = sorted_cnt[r];
        const uint32_t* src = leaf_indexes + leaf_begin[sorted_leaf[r]];
        uint32_t* dst = out + offset[r];
        for (uint32_t t = lane; t < n; t += 64u) dst[t] = src[t];
    }
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
        A(t->d_key, 8 * L + 8); A(t->d_rank, 4 * L + 4); A(t->d_bucket, 4 * (2 * (size_t)TREE_BUCKETS + TREE_BUCKETS + 4)); A(t->d_sorted_cnt, 4 * L + 4);
        A(t->d_sorted_leaf, 4 * L + 4); A(t->d_offset, 4 * L + 4);
        A(t->d_total, 16 + 8 * PLAN_GRID + 512);  // {splats, kept leaves, pad} + the plan's two chunk-total tables + its barrier words
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
        if (e == hipSuccess) e = hipMemsetAsync(t->d_total.p, 0, 16 + 8 * PLAN_GRID + 512, s);
        if (e == hipSuccess) e = hipMemsetAsync(t->d_bucket.p, 0, sizeof(uint32_t) * 2 * TREE_BUCKETS, s);   // the gather kernels keep them zero
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
    GS_REQUIRE(t && gp, "tree / params == NULL");
    GS_REQUIRE(render_count || (dst && !indexes_out_host), "render_count == NULL (asynchronous gather) needs a sorter and no host copy");
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
    uint32_t* bhist = t->d_bucket.as<uint32_t>();
    uint32_t* bfill = bhist + TREE_BUCKETS;
    uint32_t* bstart = bfill + TREE_BUCKETS;
    // the sorter gets the list's length on the device as well
    uint32_t* count_dev = nullptr;
    if (dst) {
        GS_TRY(dst->gathered_dev.ensure(16));
        count_dev = dst->gathered_dev.as<uint32_t>();
    }
    PlanBuffers pb;
    pb.center = t->d_center.as<double>(); pb.size = t->d_size.as<double>(); pb.count = t->d_count.as<uint32_t>();
    pb.key = t->d_key.as<unsigned long long>();
    pb.hist = bhist; pb.fill = bfill; pb.start_local = bstart;
    pb.chunk_sum = t->d_total.as<uint32_t>() + 4;
    pb.members = t->d_rank.as<uint32_t>(); pb.sorted_cnt = t->d_sorted_cnt.as<uint32_t>();
    pb.sorted_leaf = t->d_sorted_leaf.as<uint32_t>(); pb.offset = t->d_offset.as<uint32_t>();
    pb.totals = t->d_total.as<uint32_t>(); pb.count_out = count_dev;
    pb.barrier = t->d_total.as<uint32_t>() + 4 + 2 * PLAN_GRID + 32;      // arrival counter; its release flag 128 bytes further
    // PLAN_BARRIERS grid barriers per launch on one monotonic counter (wrap-safe: compared by difference)
    hipLaunchKernelGGL(k_tree_plan, dim3(PLAN_GRID), dim3(PLAN_THREADS), 0, st, p, pb, t->barrier_epoch * PLAN_BARRIERS * PLAN_GRID);
    t->barrier_epoch++;
    const uint32_t copy_grid = (L + COPY_WAVES - 1u) / COPY_WAVES;
    hipLaunchKernelGGL(k_tree_copy, dim3(copy_grid < 8192u ? copy_grid : 8192u), dim3(64 * COPY_WAVES), 0, st, t->d_total.as<uint32_t>(),
                       t->d_sorted_leaf.as<uint32_t>(), t->d_sorted_cnt.as<uint32_t>(), t->d_offset.as<uint32_t>(),
                       t->d_begin.as<uint32_t>(), t->d_indexes.as<uint32_t>(), out_dev);
    GS_HIP(hipGetLastError());
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
