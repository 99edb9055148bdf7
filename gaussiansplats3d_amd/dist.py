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
