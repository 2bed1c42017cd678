"""Multi-GPU plumbing of the tile-partitioned render: one process per GPU, torch.distributed only.

The path shards by image tile -- `block_width x block_height` tiles in the reference's NaiveSpiral order
(queue.h:68-127), tile k owned by rank k mod world -- with no communication while rendering; the single exchange is a
sum-reduce of the RGBA framebuffer onto rank 0 (RCCL over xGMI when the backend is "nccl").  Because tile ownership is
disjoint, every pixel receives exactly one non-zero contribution, so the reduced image is bit-identical to a
single-GPU render for any world size.  No rendering happens in this module.
"""
from __future__ import annotations

from typing import List, Tuple

from . import spiral_tiles


def tiles_for_rank(width: int, height: int, block_w: int, block_h: int, rank: int, world: int) -> List[Tuple[int, int, int, int]]:
    """Tile rects (x0, y0, x1, y1) owned by `rank`: NaiveSpiral order, round-robin (balances the cheap border tiles)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    return spiral_tiles(width, height, block_w, block_h)[rank::world]


def reduce_framebuffer(fb, dst: int = 0):
    """Sum-reduce the framebuffer tensor (H, W, 4 float32, device or CPU) onto `dst`; the one collective of the path."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(fb, dst=dst, op=dist.ReduceOp.SUM)
    return fb
