"""Multi-GPU partitioning of a frame, one process per GPU over torch.distributed (RCCL on GPUs, gloo on CPU).

Pixels are independent units (own RNG stream, own texel); the samples of one pixel are not (the RNG state runs
on from sample to sample, reference src/raytrace.wgsl:210,221). Two well-defined partitions:

* "tiles": the image is cut into 8x8 pixel tiles numbered row-major and tile k belongs to rank k % world
  (PtFrame.part_index / part_count). Every rank writes zeros outside its tiles, so one all-reduce(sum) of the
  buffers assembles the frame. The result is bit-identical to the single-GPU frame.
* "virtual_frames": every rank renders all pixels as an independent frame with its own `time` (the reference
  de-correlates successive frames exactly this way: the seed is multiplied by time + 1) and the radiance is
  averaged: all-reduce(sum) then divide by world. world x spp samples per pixel per step.

`render_fn(frame) -> tensor (H, W, 4) f32` is supplied by the caller: the HIP path on a GPU, anything else in
tests. The collective is the only exchange step of the path.
"""
from __future__ import annotations

import numpy as np

TILE = 8


def virtual_frame_time(rank: int, base_time: float = 0.0) -> float:
    """`time` of rank's virtual frame. Rank 0 keeps the caller's time; the others add rank + 0.5, which keeps
    (x + 0.5) * (time + 1) away from integer boundaries when base_time is 0 (seed = trunc of that product)."""
    return float(base_time) if rank == 0 else float(base_time) + float(rank) + 0.5


def tile_owner_mask(width: int, height: int, part_index: int, part_count: int) -> np.ndarray:
    """(H, W) bool: pixels of the tiles k with k % part_count == part_index (the partition PtFrame describes)."""
    tiles_x = (width + TILE - 1) // TILE
    ty, tx = np.meshgrid(np.arange(height) // TILE, np.arange(width) // TILE, indexing="ij")
    return ((ty * tiles_x + tx) % max(part_count, 1)) == part_index


class DistributedRenderer:
    def __init__(self, render_fn, rank: int, world: int, mode: str = "virtual_frames", group=None):
        if mode not in ("virtual_frames", "tiles"):
            raise ValueError("mode must be 'virtual_frames' or 'tiles'")
        self.render_fn, self.rank, self.world, self.mode, self.group = render_fn, rank, world, mode, group

    def frame_kwargs(self, base_time: float = 0.0) -> dict:
        """Keyword arguments for Scene.frame(...) / abi.make_frame(...) of this rank."""
        if self.mode == "tiles":
            return dict(time=base_time, part_index=self.rank, part_count=self.world)
        return dict(time=virtual_frame_time(self.rank, base_time), part_index=0, part_count=1)

    def render(self, make_frame, base_time: float = 0.0):
        """make_frame(**frame_kwargs) -> PtFrame. Returns the assembled (H, W, 4) tensor on every rank."""
        import torch.distributed as dist
        img = self.render_fn(make_frame(**self.frame_kwargs(base_time)))
        if self.world > 1:
            dist.all_reduce(img, op=dist.ReduceOp.SUM, group=self.group)
            if self.mode == "virtual_frames":
                img.mul_(1.0 / self.world)   # alpha: world * 1 / world = 1
        return img
