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


class PipelinedGather:
    """The frame loop's exchange for N > 1 on RCCL, overlapped with rendering: frame i's tiles are copied out of the library's
    buffer (a 4-33 MB device copy, the only thing the next frame waits for), gathered to `dst` on a side stream while frame i+1
    renders, and un-tiled on `dst` right after frame i+1 returns. Two staging / receive buffers alternate; every buffer is reused
    only after the collective that read or wrote it has completed (events). `finish()` completes the last frame.

    Why: a rank's share of the 1080p frame at N = 8 is ~2.9 ms; a gather + two host synchronisations + the un-tiling kernel in line
    with it are ~0.3 ms, the difference between ~6.0x and ~6.7x. Ownership rule of pt_tiles_device_ptr: the library rewrites its
    tile buffer at the END of the next pt_render, so the copy must be complete before that call — `submit` waits for it."""

    def __init__(self, renderer, per_rank_floats, rank, world, dist, dst=0):
        import torch
        self.r, self.n, self.rank, self.world, self.dist, self.dst = renderer, per_rank_floats, rank, world, dist, dst
        self.stream = torch.cuda.Stream()
        self.stage = [torch.empty(per_rank_floats, dtype=torch.float32, device="cuda") for _ in range(2)]
        self.recv = [torch.empty(world * per_rank_floats, dtype=torch.float32, device="cuda") for _ in range(2)] if rank == dst else [None, None]
        self.copied = [torch.cuda.Event() for _ in range(2)]
        self.gathered = [torch.cuda.Event() for _ in range(2)]
        self.used = [False, False]
        self.frames = 0
        self._mine = None

    def _tiles(self):
        import torch
        v = self.r.TilesDevice()
        ptr = v.__cuda_array_interface__["data"][0]
        if self._mine is None or self._mine.data_ptr() != ptr or self._mine.numel() != self.n:
            self._mine = torch.as_tensor(v, device="cuda")  # aliases the library's buffer (no copy)
        return self._mine

    def submit(self):
        """Call right after Render() returned. Returns once the library's tile buffer may be rewritten."""
        import torch
        b = self.frames & 1
        if self.used[b]:
            self.gathered[b].synchronize()  # frame i-2's collective is done with stage[b] / recv[b]
        mine = self._tiles()
        with torch.cuda.stream(self.stream):
            self.stage[b].copy_(mine, non_blocking=True)
            self.copied[b].record()
            if self.rank == self.dst:
                self.dist.gather(self.stage[b], list(self.recv[b].split(self.n)), dst=self.dst)
            else:
                self.dist.gather(self.stage[b], None, dst=self.dst)
            self.gathered[b].record()
        self.used[b] = True
        self.copied[b].synchronize()
        if self.rank == self.dst and self.frames > 0:
            self._assemble(1 - b)  # the previous frame: its gather ran while this frame rendered
        self.frames += 1

    def _assemble(self, b):
        self.gathered[b].synchronize()
        self.r.AssembleTiles(self.recv[b].data_ptr(), self.recv[b].numel())

    def finish(self):
        """Completes the last submitted frame (un-tiled on dst; sends drained elsewhere)."""
        if self.frames == 0:
            return
        b = (self.frames - 1) & 1
        if self.rank == self.dst:
            self._assemble(b)
        else:
            self.gathered[b].synchronize()
