"""Multi-GPU plumbing of the tile-partitioned render: one process per GPU, torch.distributed only.

The path shards by image tile -- `block_width x block_height` tiles in the reference's NaiveSpiral order
(queue.h:68-127), tile k owned by rank k mod world -- with no communication while rendering; the single exchange is a
sum-reduce of the RGBA framebuffer onto rank 0 (RCCL over xGMI when the backend is "nccl").  Because tile ownership is
disjoint, every pixel receives exactly one non-zero contribution, so the reduced image is bit-identical to a
single-GPU render for any world size and any ownership map.  No rendering happens in this module.

k mod world balances pixels but not work: the spiral's rings beat against the world size and the Cornell box's interior
tiles cost ~10x the background ones (rays per rank 6 % / 7.5 % above the mean at 4 / 8 ranks at 1080p).  When per-tile
costs are known (`measure_tile_costs`: one sample per pixel, counted in the reference's own unit, World::hit queries,
integrator.h:192,247) ownership is the longest-processing-time greedy over them instead: every rank measures the same
integers, so every rank derives the same map without talking to the others.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

from . import spiral_tiles


def balanced_owners(costs: Sequence[int], world: int) -> List[int]:
    """Owner rank of every tile: longest-processing-time greedy (costliest tile first, ties by spiral index, to the least
    loaded rank, ties to the lowest rank).  Pure integer arithmetic: identical on every rank for identical costs."""
    if world < 1:
        raise ValueError("world must be >= 1")
    costs = [int(c) for c in costs]
    if any(c < 0 for c in costs):
        raise ValueError("tile costs must be non-negative")
    load = [0] * world
    owner = [0] * len(costs)
    for k in sorted(range(len(costs)), key=lambda k: (-costs[k], k)):
        r = min(range(world), key=lambda r: (load[r], r))
        owner[k] = r
        load[r] += costs[k]
    return owner


def snake_owners(costs: Sequence[int], world: int) -> List[int]:
    """Owner rank of every tile: tiles sorted by cost (ties by spiral index) are dealt out in snake order -- ranks
    0 .. world-1, then world-1 .. 0, and so on -- so that every rank receives one tile of every cost stratum: equal tile
    counts (within one) and the same MIX of cheap and costly tiles, not only equal sums."""
    if world < 1:
        raise ValueError("world must be >= 1")
    order = sorted(range(len(costs)), key=lambda k: (-int(costs[k]), k))
    owner = [0] * len(costs)
    for i, k in enumerate(order):
        rnd, pos = divmod(i, world)
        owner[k] = pos if rnd % 2 == 0 else world - 1 - pos
    return owner


def tiles_for_rank(width: int, height: int, block_w: int, block_h: int, rank: int, world: int,
                   costs: Optional[Sequence[int]] = None, strategy: str = "lpt") -> List[Tuple[int, int, int, int]]:
    """Tile rects (x0, y0, x1, y1) owned by `rank`, in NaiveSpiral order.  Without costs: tile k -> rank k mod world;
    with one cost per spiral tile: the cost-balanced map of `balanced_owners` ("lpt") or `snake_owners` ("snake")."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    tiles = spiral_tiles(width, height, block_w, block_h)
    if costs is None:
        return tiles[rank::world]
    if len(costs) != len(tiles):
        raise ValueError(f"{len(costs)} costs for {len(tiles)} tiles")
    owner = snake_owners(costs, world) if strategy == "snake" else balanced_owners(costs, world)
    return [t for t, o in zip(tiles, owner) if o == rank]


def measure_tile_costs(renderer, tiles) -> List[int]:
    """Work per tile in ray-equivalents: the reference's ray count for one sample per pixel (extension rays + light_samples shadow
    rays per hit, traced or not: a tile's time follows its hits) plus one unit per camera sample (ray generation + framebuffer accumulation).  `renderer` is a pathtrace_amd.Renderer; its framebuffer and counters are
    cleared afterwards.  Deterministic (the RNG is keyed by pixel and sample), so all ranks agree.  One pass over all
    tiles (pt_measure_tile_costs: a tally kernel attributes every bounce's rays to the tile of the path's pixel), not
    one render and one blocking counter read per tile."""
    rays = renderer.measure_tile_costs(tiles, 1)
    return [r + (x1 - x0) * (y1 - y0) for r, (x0, y0, x1, y1) in zip(rays, tiles)]


def tile_pixel_index(tiles, width: int, device=None):
    """Flat pixel indices (row-major over the film, int64 tensor) of the pixels of `tiles`, tile after tile, row-major inside
    a tile: the packing order of `gather_owned_tiles`."""
    import torch

    parts = []
    for (x0, y0, x1, y1) in tiles:
        ys = torch.arange(y0, y1, dtype=torch.int64).unsqueeze(1) * width
        xs = torch.arange(x0, x1, dtype=torch.int64).unsqueeze(0)
        parts.append((ys + xs).reshape(-1))
    idx = torch.cat(parts) if parts else torch.zeros(0, dtype=torch.int64)
    return idx.to(device) if device is not None else idx


class OwnedTileExchange:
    """The one exchange of the tile-partitioned render, moving only what a rank owns: every rank packs the pixels of ITS
    tiles (1/N of the frame) and rank `dst` receives them with one `gather` and writes them into its framebuffer -- (N-1)/N
    of one frame over the links in total, where a sum-reduce of whole frames moves N-1 of them (33 MB each at 1080p).
    Ownership is disjoint, so writing is adding: the assembled image is the reduce's, bit for bit.  The index tensors and the
    staging buffers are built once, outside the timed region; `run` is the collective.  `tiles_by_rank[r]` must be the same
    list on every rank (it is: ownership is a pure function of the tile costs)."""

    def __init__(self, tiles_by_rank, width: int, height: int, rank: int, world: int, device, dst: int = 0):
        import torch

        self.rank, self.world, self.dst = rank, world, dst
        self.counts = [sum((x1 - x0) * (y1 - y0) for (x0, y0, x1, y1) in tl) for tl in tiles_by_rank]
        self.pad = max(self.counts) if self.counts else 0   # gather wants equal shapes: padded to the largest share
        self.mine = tile_pixel_index(tiles_by_rank[rank], width, device)
        self.send = torch.zeros((self.pad, 4), dtype=torch.float32, device=device)
        self.recv = None
        self.index = None
        if rank == dst:
            self.recv = [torch.zeros((self.pad, 4), dtype=torch.float32, device=device) for _ in range(world)]
            self.index = [tile_pixel_index(tl, width, device) for tl in tiles_by_rank]

    def bytes_moved(self) -> int:
        return sum(c for r, c in enumerate(self.counts) if r != self.dst) * 16

    def run(self, fb):
        """fb: (H, W, 4) float32 on this exchange's device.  After the call rank `dst` holds the whole image."""
        import torch.distributed as dist

        flat = fb.view(-1, 4)
        n = self.counts[self.rank]
        self.send[:n] = flat[self.mine]
        dist.gather(self.send, self.recv if self.rank == self.dst else None, dst=self.dst)
        if self.rank == self.dst:
            for r in range(self.world):
                if r != self.dst:
                    flat[self.index[r]] = self.recv[r][: self.counts[r]]
        return fb


def reduce_framebuffer(fb, dst: int = 0):
    """Sum-reduce the framebuffer tensor (H, W, 4 float32, device or CPU) onto `dst`; the one collective of the path."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(fb, dst=dst, op=dist.ReduceOp.SUM)
    return fb
