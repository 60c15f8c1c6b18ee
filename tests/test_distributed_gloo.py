"""N>1 path on CPU: world-size-2 (and 3) `gloo` runs of the partition + gather + assembly logic.
No pixels are traced here (that needs the GPU); every rank fills its tile-major slots with the GLOBAL pixel index the
library's own layout assigns to it, the ranks gather to rank 0 exactly as bench.py does, and rank 0 un-tiles with the
spec's mapping: the result must be the identity image — every pixel owned exactly once, by the right rank."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, w, h, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pathtracing_amd as P
    from pathtracing_amd.distributed import gather_tiles, slot_to_pixel
    lay = P.tile_layout(P.make_params(w, h, rank=rank, nranks=world))
    n_slots = lay.tiles_per_rank * 4096
    per_rank = n_slots * 4
    x, y, tile = slot_to_pixel(np.arange(n_slots), rank, world, lay.tiles_x)
    valid = (tile < lay.n_tiles) & (x < w) & (y < h)
    buf = np.full((n_slots, 4), -1.0, np.float32)
    buf[valid, 0] = (y * w + x)[valid]      # "radiance" = global pixel index
    buf[valid, 1] = rank
    buf[valid, 3] = 1.0
    assert int(valid.sum()) == sum(min(64, w - (t % lay.tiles_x) * 64) * min(64, h - (t // lay.tiles_x) * 64)
                                   for t in range(rank, lay.n_tiles, world))
    got = gather_tiles(torch.from_numpy(buf.reshape(-1)), per_rank, rank, world, dist)
    if rank == 0:
        g = got.numpy().reshape(world, n_slots, 4)
        img = np.full((h, w, 4), -2.0, np.float32)
        for r in range(world):
            xx, yy, tt = slot_to_pixel(np.arange(n_slots), r, world, lay.tiles_x)
            ok = (tt < lay.n_tiles) & (xx < w) & (yy < h)
            assert (img[yy[ok], xx[ok], 0] == -2.0).all()          # nobody wrote these pixels before
            img[yy[ok], xx[ok]] = g[r][ok]
        want = np.arange(w * h, dtype=np.float32).reshape(h, w)
        tiles = (np.arange(h)[:, None] // 64) * lay.tiles_x + (np.arange(w)[None, :] // 64)
        q.put((bool((img[..., 0] == want).all()), bool((img[..., 1] == tiles % world).all()), bool((img[..., 3] == 1).all())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,w,h", [(2, 200, 131), (3, 130, 70), (2, 64, 64)])
def test_gloo_gather_and_assembly(world, w, h):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() + world * 7 + w) % 300
    procs = [ctx.Process(target=_worker, args=(r, world, port, w, h, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) == (True, True, True)
