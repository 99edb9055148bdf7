"""Multi-GPU sharding of the render seam: one process per GPU, screen tile rows split into strips, strips gathered
to rank 0 with grouped send/recv over RCCL (torch.distributed backend "nccl" on ROCm) — SURVEY.md §8(e).

The reference is single-GPU (one WebGL context); the scene and the sort are replicated on every rank, the only
exchange step is the framebuffer gather (W*H*4 bytes per frame in total)."""
import numpy as np

GS_TILE = 16


def balanced_row_strips(row_cost, world_size, align=1):
    """Split tile rows [0, len(row_cost)) into `world_size` contiguous strips of near-equal cost.
    row_cost: per-tile-row work estimate (tile entries).  Returns [(begin, end)] * world_size, covering all rows;
    strips may be empty when there are fewer rows than ranks.  align: cut only at multiples of `align` tile rows (2 = the
    32-px blend bins: a bin cut in half is drawn by two ranks, each with half of its waves idle)."""
    rows = len(row_cost)
    if align > 1 and rows >= 2 * align * world_size:
        grouped = np.add.reduceat(np.asarray(row_cost, dtype=np.float64), np.arange(0, rows, align))
        return [(min(b * align, rows), min(e * align, rows)) for b, e in balanced_row_strips(grouped, world_size)]
    cost = np.asarray(row_cost, dtype=np.float64) + 1.0          # +1: empty rows still cost a launch slot
    csum = np.concatenate([[0.0], np.cumsum(cost)])
    total = csum[-1]
    cuts = [0]
    for r in range(1, world_size):
        target = total * r / world_size
        k = int(np.searchsorted(csum, target, side="left"))
        if k > 0 and abs(csum[k - 1] - target) <= abs(csum[min(k, rows)] - target):
            k -= 1
        k = min(max(k, cuts[-1]), rows)
        cuts.append(k)
    cuts.append(rows)
    return [(cuts[i], cuts[i + 1]) for i in range(world_size)]


def transfer_balanced_row_strips(row_cost, world_size, width, ms_per_cost, fixed_ms=0.0, link_GBps=77.0, root=0, align=2,
                                 overlapped=True):
    """Strips balanced for DRAWING AND TRANSFER.  The root draws its own strip and receives every other strip over that peer's
    own point-to-point xGMI link (`link_GBps` per direction, per link; transfers from different peers run in parallel), so with
    few ranks the strips are long and a peer's transfer, not its draw, sets the frame rate: at 8K, 2 ranks, the peer's 66 MB take
    0.87 ms at 77 GB/s against a 0.55 ms draw.  The root therefore takes more rows than an equal-cost cut gives
    it.  Model per rank r with tile rows [b, e):
        draw_r = fixed_ms + ms_per_cost * sum(row_cost[b:e])          xfer_r = (e - b) * 16 * width * 4 bytes / link rate  (r != root)
    overlapped (the gather of frame k runs beside the draw of frame k + 1): minimise max_r max(draw_r, xfer_r);
    serial: minimise max_r draw_r + max_r xfer_r (here: by the same cut, which bounds it).
    Contiguous strips in rank order, cut at multiples of `align` tile rows; a binary search on the frame time with a greedy
    front-to-back feasibility test (maximal strips are optimal for a monotone cost).  Returns [(begin, end)] * world_size."""
    rows = len(row_cost)
    if world_size <= 1:
        return [(0, rows)]
    cost = np.asarray(row_cost, dtype=np.float64) + 1.0
    groups = list(range(0, rows, align)) + [rows]                       # cut positions
    csum = np.concatenate([[0.0], np.cumsum(cost)])
    row_ms = GS_TILE * width * 4 / (link_GBps * 1e9) * 1e3               # transfer of one tile row

    def draw(b, e):
        return fixed_ms + ms_per_cost * (csum[e] - csum[b]) if e > b else 0.0

    def fits(T):
        cuts, g = [0], 0
        for r in range(world_size):
            k = g
            while k + 1 < len(groups):
                e = groups[k + 1]
                if draw(groups[g], e) > T or (r != root and (e - groups[g]) * row_ms > T):
                    break
                k += 1
            g = k
            cuts.append(groups[g])
        return cuts if cuts[-1] == rows else None

    lo, hi = 0.0, fixed_ms + ms_per_cost * csum[-1] + rows * row_ms + 1e-9
    best = fits(hi)
    for _ in range(60):
        mid = 0.5 * (lo + hi)
        c = fits(mid)
        if c is None:
            lo = mid
        else:
            hi, best = mid, c
    return [(best[i], best[i + 1]) for i in range(world_size)]


def strip_frame_model(strips, row_cost, width, ms_per_cost, fixed_ms=0.0, link_GBps=77.0, root=0):
    """(max draw ms, max transfer ms) of a strip assignment under transfer_balanced_row_strips' model."""
    cost = np.asarray(row_cost, dtype=np.float64) + 1.0
    row_ms = GS_TILE * width * 4 / (link_GBps * 1e9) * 1e3
    draws = [fixed_ms + ms_per_cost * cost[b:e].sum() if e > b else 0.0 for b, e in strips]
    xfers = [(e - b) * row_ms if r != root else 0.0 for r, (b, e) in enumerate(strips)]
    return max(draws), max(xfers)


def equal_row_strips(rows, world_size):
    return balanced_row_strips(np.zeros(rows), world_size)


def strip_pixel_rows(strip, height):
    y0 = strip[0] * GS_TILE
    y1 = min(strip[1] * GS_TILE, height)
    return y0, max(y1, y0)


def gather_strips(local_strip, strips, full, rank, world_size, dist, dst=0):
    """Grouped send/recv (a gatherv): every rank sends its uint8 [h_r, W, 4] strip to `dst`, which receives each
    one directly into its slice of `full` (uint8 [H, W, 4]).  All tensors live on the calling rank's device
    (or CPU with the gloo backend).  Returns `full` on dst, None elsewhere."""
    height = full.shape[0] if full is not None else None
    ops = []
    if rank == dst:
        for r, s in enumerate(strips):
            y0, y1 = strip_pixel_rows(s, height)
            if y1 <= y0:
                continue
            if r == dst:
                full[y0:y1].copy_(local_strip)
            else:
                ops.append(dist.P2POp(dist.irecv, full[y0:y1], r))
    elif local_strip is not None and local_strip.numel() > 0:
        ops.append(dist.P2POp(dist.isend, local_strip, dst))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return full if rank == dst else None


# ---------------------------------------------------------------------------------------------------------------------------
# SORT-MIDDLE split: the design that removes the per-rank work which does not shrink with the strip.
# With tile-row strips alone every rank still runs the vertex stage and the key / min-max pass over ALL N splats.  Here rank r
# OWNS the splats with original indexes [r N / W, (r + 1) N / W): it projects and keys only those, the ranks all-reduce the
# key range (8 bytes: the reference's buckets need the global min / max, sorter.cpp:142-143), and every survivor travels - as
# {original index, depth key, 32-byte record, 8-byte tile rect} = 48 bytes - to each rank whose strip its tile rect touches
# (an all-to-all-v: one grouped ncclSend / ncclRecv per peer pair, ~V_strip * 48 bytes received per rank).  A source lists its
# survivors in ASCENDING original index; the destination concatenates what it receives in source-rank order, which - sources
# owning ascending index ranges - is again ascending original index: exactly the list the per-rank visibility-culled sort
# compacts today, so the same two stable radix passes over it reproduce the reference's order (descending bucket, ties in
# reverse input order) bit for bit, and binning / blending of the strip go on unchanged.
# What follows is the exchange's index logic on the host (numpy + torch.distributed point-to-point, the primitive RCCL runs
# too); tests/test_dist_gloo.py drives it with world 2 and 3.  The device path is not built: tools/strip_scaling.py costs the
# design from measured parts.
def owner_range(rank, world_size, n):
    """Original-index range [begin, end) rank `rank` projects and keys in the sort-middle split."""
    per = (n + world_size - 1) // world_size
    return min(rank * per, n), min((rank + 1) * per, n)


def sort_middle_destinations(tile_rows, strips):
    """tile_rows: int [m, 2] inclusive tile-row span (ty0, ty1) of m visible splats.  -> bool [m, world]: splat x strip touched."""
    t = np.asarray(tile_rows).reshape(-1, 2)
    out = np.zeros((t.shape[0], len(strips)), dtype=bool)
    for d, (b, e) in enumerate(strips):
        out[:, d] = (t[:, 0] < e) & (t[:, 1] >= b) & (e > b)
    return out


def sort_middle_exchange(ids, payload, tile_rows, strips, rank, world_size, dist):
    """One frame's all-to-all-v.  ids: int64 [m] ascending original indexes of this rank's visible splats; payload: uint8 [m, B]
    their records; tile_rows as above.  Returns (ids_for_my_strip ascending, payload rows in the same order)."""
    import torch
    ids = np.asarray(ids, dtype=np.int64)
    payload = np.ascontiguousarray(payload, dtype=np.uint8).reshape(ids.shape[0], -1)
    B = payload.shape[1]
    touch = sort_middle_destinations(tile_rows, strips)
    send_sel = [np.nonzero(touch[:, d])[0] for d in range(world_size)]
    counts = torch.tensor([len(sel) for sel in send_sel], dtype=torch.int64)
    table = [torch.zeros(world_size, dtype=torch.int64) for _ in range(world_size)]
    dist.all_gather(table, counts)                                   # table[s][d] = survivors source s sends to strip d
    recv_counts = [int(table[s][rank]) for s in range(world_size)]
    recv_ids = [torch.empty(c, dtype=torch.int64) for c in recv_counts]
    recv_pay = [torch.empty((c, B), dtype=torch.uint8) for c in recv_counts]
    ops = []
    for peer in range(world_size):
        if peer == rank:
            recv_ids[rank].copy_(torch.from_numpy(ids[send_sel[rank]]))
            recv_pay[rank].copy_(torch.from_numpy(payload[send_sel[rank]]))
            continue
        if len(send_sel[peer]):
            ops.append(dist.P2POp(dist.isend, torch.from_numpy(np.ascontiguousarray(ids[send_sel[peer]])), peer))
            ops.append(dist.P2POp(dist.isend, torch.from_numpy(np.ascontiguousarray(payload[send_sel[peer]])), peer))
        if recv_counts[peer]:
            ops.append(dist.P2POp(dist.irecv, recv_ids[peer], peer))
            ops.append(dist.P2POp(dist.irecv, recv_pay[peer], peer))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return torch.cat(recv_ids).numpy(), torch.cat(recv_pay).numpy()    # source-rank order = ascending original index


class StripGroup:
    """The strip gather behind the C ABI (gs_group_*): one grouped ncclSend / ncclRecv per frame on the context's stream,
    no Python per strip.  `dist` is only the side channel that carries rank 0's ncclUniqueId to the other ranks."""

    def __init__(self, context, rank, world_size, dist=None):
        import ctypes as C

        from . import _lib as L
        self.lib, self.rank, self.world = context.lib, int(rank), int(world_size)
        self.handle = C.c_void_p()
        ident = np.zeros(128, dtype=np.uint8)
        if self.world > 1:
            import torch
            if self.rank == 0:
                L.check(self.lib.gs_group_unique_id(ident.ctypes.data))
            t = torch.from_numpy(ident)
            dev = None
            if dist.get_backend() == "nccl":
                dev = torch.device("cuda", torch.cuda.current_device())
                t = t.to(dev)
            dist.broadcast(t, src=0)
            ident = t.cpu().numpy().copy()
        L.check(self.lib.gs_group_create(context.handle, ident.ctypes.data, self.world, self.rank, C.byref(self.handle)))
        context._adopt(self)

    def gather_strips(self, local_strip_ptr, full_ptr, width, strips, height, dst=0):
        """Every rank sends its strip (device pointer) to `dst`, which receives strip r at its rows of the full frame."""
        import ctypes as C

        from . import _lib as L
        rows = np.array([strip_pixel_rows(s, height) for s in strips], dtype=np.uint32)
        b, e = np.ascontiguousarray(rows[:, 0]), np.ascontiguousarray(rows[:, 1])
        L.check(self.lib.gs_group_gather_strips(self.handle, C.c_void_p(local_strip_ptr or None), C.c_void_p(full_ptr or None),
                                                int(width), b.ctypes.data, e.ctypes.data, int(dst)))

    def set_overlap(self, enabled):
        """Transfers on the group's own stream, beside the next frame (the caller alternates two strip / frame buffers)."""
        from . import _lib as L
        L.check(self.lib.gs_group_set_overlap(self.handle, 1 if enabled else 0))

    def wait(self):
        from . import _lib as L
        L.check(self.lib.gs_group_wait(self.handle))

    def close(self):
        if self.handle:
            self.lib.gs_group_destroy(self.handle)
            import ctypes as C
            self.handle = C.c_void_p()

    dispose = terminate = close
