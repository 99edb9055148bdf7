"""oracle/tree_oracle.py — TEST INFRASTRUCTURE.  Pure-Python restatements (IEEE doubles, no FMA) of

* the reference's octree build: /root/reference/src/splattree/SplatTree.js:132-271 (createSplatTreeWorker) and
  the leaf filtering of SplatSubTree.convertWorkerSubTree (:55-79).  PINNED: tests/golden/tree_kat.json holds the
  leaves the reference's own code produces for tests/tree_cases.py (recorded by oracle/make_golden_tree.py, which
  evaluates the worker function cut out of the reference's source text under Node);
* Viewer.gatherSceneNodesForSort (/root/reference/src/Viewer.js:1969-2077) with three r160's Vector3.applyMatrix4 /
  normalize / length spelled out.  PINNED since round 2: tests/golden/gather_kat.json is recorded by
  oracle/make_golden_gather.py, which evaluates the method's own text (cut out of Viewer.js) under Node over trees built by
  the reference's worker, with THREE = oracle/three_min.mjs (3 trees x 7 cameras: render count, index list, modelView).

Small cases only (plain Python loops).
"""
import math

import numpy as np


def build_tree(centers, keep=None, max_depth=8, max_centers=1000, first_index=0):
    """Returns leaves = [dict(min, max, center, depth, indexes)] in nodesWithIndexes order, all_leaf_count."""
    c = np.ascontiguousarray(centers, dtype=np.float32).reshape(-1, 3).astype(np.float64)   # Float32Array -> number
    n = c.shape[0]
    root = [i for i in range(n) if keep is None or keep[i]]
    if root:
        scene_min = [float(c[root, k].min()) for k in range(3)]
        scene_max = [float(c[root, k].max()) for k in range(3)]
    else:
        scene_min, scene_max = [0.0, 0.0, 0.0], [0.0, 0.0, 0.0]
    added = set()
    leaves = []
    all_leaves = [0]

    def process(mn, mx, depth, idx):
        if len(idx) < max_centers or depth > max_depth:
            all_leaves[0] += 1
            new = []
            for i in idx:
                if i not in added:
                    added.add(i)
                    new.append(i + first_index)
            new.sort()
            if new:
                center = [(mx[k] - mn[k]) * 0.5 + mn[k] for k in range(3)]
                leaves.append(dict(min=list(mn), max=list(mx), center=center, depth=depth, indexes=new))
            return
        dim = [mx[k] - mn[k] for k in range(3)]
        half = [d * 0.5 for d in dim]
        nc = [mn[k] + half[k] for k in range(3)]
        x0, x1, x2 = nc[0] - half[0], nc[0], nc[0] + half[0]
        y0, y1, y2 = nc[1] - half[1], nc[1], nc[1] + half[1]
        z0, z1, z2 = nc[2] - half[2], nc[2], nc[2] + half[2]
        boxes = [([x0, y1, z0], [x1, y2, z1]), ([x1, y1, z0], [x2, y2, z1]), ([x1, y1, z1], [x2, y2, z2]), ([x0, y1, z1], [x1, y2, z2]),
                 ([x0, y0, z0], [x1, y1, z1]), ([x1, y0, z0], [x2, y1, z1]), ([x1, y0, z1], [x2, y1, z2]), ([x0, y0, z1], [x1, y1, z2])]
        lists = [[] for _ in range(8)]
        for i in idx:
            p = c[i]
            for j, (bmn, bmx) in enumerate(boxes):
                if bmn[0] <= p[0] <= bmx[0] and bmn[1] <= p[1] <= bmx[1] and bmn[2] <= p[2] <= bmx[2]:
                    lists[j].append(i)
        for j in range(8):
            process(boxes[j][0], boxes[j][1], depth + 1, lists[j])

    process(scene_min, scene_max, 0, root)
    return leaves, all_leaves[0]


def gather(leaves, model_view, fov_y_deg, render_w, render_h, gather_all=False):
    """Viewer.gatherSceneNodesForSort for one sub-tree; returns uint32 indexesToSort[0:splatRenderCount]."""
    e = [float(v) for v in np.asarray(model_view, dtype=np.float64).reshape(16)]
    focal = (render_h / 2.0) / math.tan(fov_y_deg / 2.0 * (math.pi / 180.0))
    cos_x = math.cos(math.atan(render_w / 2.0 / focal))
    cos_y = math.cos(math.atan(render_h / 2.0 / focal))
    kept = []
    for order, leaf in enumerate(leaves):
        x, y, z = leaf["center"]
        w = 1.0 / (e[3] * x + e[7] * y + e[11] * z + e[15])
        vx = (e[0] * x + e[4] * y + e[8] * z + e[12]) * w
        vy = (e[1] * x + e[5] * y + e[9] * z + e[13]) * w
        vz = (e[2] * x + e[6] * y + e[10] * z + e[14]) * w
        dist = math.sqrt(vx * vx + vy * vy + vz * vz)
        s = 1.0 / (dist or 1.0)
        vx, vy, vz = vx * s, vy * s, vz * s
        lyz = math.sqrt(0.0 * 0.0 + vy * vy + vz * vz)
        yz_z = vz * (1.0 / (lyz or 1.0))
        lxz = math.sqrt(vx * vx + 0.0 * 0.0 + vz * vz)
        xz_z = vz * (1.0 / (lxz or 1.0))
        dx, dy, dz = (leaf["max"][k] - leaf["min"][k] for k in range(3))
        ns = math.sqrt(dx * dx + dy * dy + dz * dz)
        out_y = -yz_z < (cos_y - .6)
        out_x = -xz_z < (cos_x - .6)
        if not gather_all and ((out_x or out_y) and dist > ns):
            continue
        kept.append((dist, order, leaf))
    kept.sort(key=lambda t: (t[0], t[1]))                 # ascending distance; ties: see DESIGN.md (JS comparator is non-strict)
    out = []
    for _, _, leaf in reversed(kept):                     # the nearest leaf ends the buffer
        out.extend(leaf["indexes"])
    return np.array(out, dtype=np.uint32)
