"""CPU tier, world_size 2 over gloo: the strip gather (grouped send/recv) that the N>1 bench path runs over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gaussiansplats3d_amd import dist as gdist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, H, W, cost, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rows = (H + 15) // 16
    strips = gdist.balanced_row_strips(np.asarray(cost[:rows]), world)
    y0, y1 = gdist.strip_pixel_rows(strips[rank], H)
    # every rank "renders" a deterministic pattern for its strip
    yy, xx = np.meshgrid(np.arange(y0, y1), np.arange(W), indexing="ij")
    local = np.stack([yy % 251, xx % 253, (yy + xx) % 255, np.full_like(yy, 255)], axis=-1).astype(np.uint8)
    strip = torch.from_numpy(local)
    full = torch.zeros((H, W, 4), dtype=torch.uint8) if rank == 0 else None
    for _ in range(2):                                   # twice: the bench calls it every frame
        res = gdist.gather_strips(strip, strips, full, rank, world, dist)
    if rank == 0:
        np.save(out_path, res.numpy())
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,H,W", [(2, 170, 40), (2, 16, 8), (3, 100, 24)])
def test_gather_strips_reassembles_the_frame(tmp_path, world, H, W):
    port = _free_port()
    out = str(tmp_path / "full.npy")
    cost = (np.arange(64) % 7 + 1).tolist()
    mp.spawn(_worker, args=(world, port, H, W, cost, out), nprocs=world, join=True)
    got = np.load(out)
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    exp = np.stack([yy % 251, xx % 253, (yy + xx) % 255, np.full_like(yy, 255)], axis=-1).astype(np.uint8)
    np.testing.assert_array_equal(got, exp)
