"""Image-space partition across ranks (docs/SPEC.md §6): gather of per-rank tile radiance to rank 0.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI on ROCm; "gloo" on CPU for tests). The data
path has exactly one collective per frame: `gather` of `tiles_per_rank * 4096` float4 per rank. The reference is
single-GPU (GraphicsDevice.cs:176-183) — nothing here mirrors reference code.
"""
import numpy as np

TILE = 64


def slot_to_pixel(slot, rank, nranks, tiles_x):
    """Vectorised docs/SPEC.md §6 mapping: slot of `rank` -> (x, y, tile). Pure index arithmetic (host logic)."""
    slot = np.asarray(slot, np.int64)
    tl, inner = slot >> 12, slot & 4095
    tile = rank + nranks * tl
    tx, ty = tile % tiles_x, tile // tiles_x
    blk, ln = inner >> 6, inner & 63
    x = (tx << 6) + ((blk & 7) << 3) + (ln & 7)
    y = (ty << 6) + ((blk >> 3) << 3) + (ln >> 3)
    return x, y, tile


def gather_tiles(local_tiles, per_rank_floats, rank, world, dist=None, dst=0, out=None):
    """Gather every rank's tile-major buffer (1-D float32 tensor of per_rank_floats) to `dst`.
    Returns the concatenated [world * per_rank_floats] tensor on dst, None elsewhere. `out`: a receive buffer of
    world * per_rank_floats elements to reuse on dst (a frame loop allocates it once)."""
    import torch
    if world == 1:
        return local_tiles
    assert local_tiles.numel() == per_rank_floats
    if rank == dst:
        if out is None:
            out = torch.empty(world * per_rank_floats, dtype=local_tiles.dtype, device=local_tiles.device)
        assert out.numel() == world * per_rank_floats and out.device == local_tiles.device and out.dtype == local_tiles.dtype
        glist = list(out.split(per_rank_floats))
    else:
        out, glist = None, None
    dist.gather(local_tiles, glist, dst=dst)
    return out
