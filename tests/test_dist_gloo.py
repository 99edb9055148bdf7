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


def _sm_worker(rank, world, port, n, rows, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(1234)                       # every rank draws the SAME scene
    visible = rng.uniform(size=n) < 0.3
    ty0 = rng.integers(0, rows, size=n)
    ty1 = np.minimum(ty0 + rng.geometric(0.4, size=n) - 1, rows - 1)
    payload = rng.integers(0, 256, size=(n, 48), dtype=np.uint8)
    cost = rng.uniform(1.0, 5.0, size=rows)
    strips = gdist.balanced_row_strips(cost, world)
    b, e = gdist.owner_range(rank, world, n)
    mine = np.nonzero(visible[b:e])[0] + b                  # ascending original indexes of this rank's survivors
    ids, pay = gdist.sort_middle_exchange(mine, payload[mine], np.stack([ty0[mine], ty1[mine]], axis=1), strips, rank, world, dist)
    # what the strips-only path computes on every rank from its own full-N vertex stage
    sb, se = strips[rank]
    want = np.nonzero(visible & (ty0 < se) & (ty1 >= sb) & (se > sb))[0]
    np.save(os.path.join(out_dir, f"ok{rank}.npy"), np.array([np.array_equal(ids, want) and np.array_equal(pay, payload[want]), len(want)]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sort_middle_exchange_delivers_each_strip_its_ascending_survivor_list(tmp_path, world):
    """The sort-middle split (gaussiansplats3d_amd/dist.py): every rank projects only the splats it owns and sends each survivor
    to the ranks whose strips its tile rect touches; what a rank receives, concatenated in source order, must be exactly the
    ascending list of ALL survivors touching its strip - the list the per-rank visibility-culled sort starts from today - with the
    payloads in the same order (a splat spanning two strips is delivered to both)."""
    port = _free_port()
    mp.spawn(_sm_worker, args=(world, port, 20000, 68, str(tmp_path)), nprocs=world, join=True)
    total = 0
    for r in range(world):
        ok, cnt = np.load(tmp_path / f"ok{r}.npy")
        assert ok == 1, f"rank {r}"
        total += int(cnt)
    assert total > 0


def test_transfer_aware_strips_give_the_root_more_rows_when_the_link_is_the_bottleneck():
    """gdist.transfer_balanced_row_strips: contiguous, covering, aligned cuts; with 2 ranks at 8K (a peer's strip takes longer
    to cross one xGMI link than to draw) the root draws more than half of the cost and the modelled frame time (max of the slowest
    draw and the slowest transfer) is lower than under the cost-only cut; with a fast link both cuts agree to within a cut step."""
    rng = np.random.default_rng(3)
    rows, W = 270, 7680
    cost = np.abs(rng.normal(1000.0, 300.0, rows))
    k, fixed = 0.58 / cost.sum(), 0.18
    for world in (2, 4, 8):
        a = gdist.balanced_row_strips(cost, world, align=2)
        b = gdist.transfer_balanced_row_strips(cost, world, W, k, fixed, link_GBps=77.0, align=2)
        assert b[0][0] == 0 and b[-1][1] == rows and all(b[i][1] == b[i + 1][0] for i in range(world - 1))
        assert all(x % 2 == 0 for s in b[:-1] for x in s)
        ta, tb = max(gdist.strip_frame_model(a, cost, W, k, fixed)), max(gdist.strip_frame_model(b, cost, W, k, fixed))
        assert tb <= ta + 1e-9
        if world == 2:
            assert b[0][1] > a[0][1] and tb < 0.75 * ta
    fast = gdist.transfer_balanced_row_strips(cost, 4, W, k, fixed, link_GBps=1e6, align=2)
    slow = gdist.balanced_row_strips(cost, 4, align=2)
    assert max(abs(f[1] - s[1]) for f, s in zip(fast, slow)) <= 4
