"""Multi-GPU sharding of one image: which samples / tiles a rank renders, and the one exchange at the end.

Pixels and samples are independent (the RNG is a pure function of seed, pixel, sample and path node), so a frame is
split either by sample range (every rank renders samples [b, e) of every pixel) or by image tile (square tiles dealt
round-robin to ranks).  Each rank accumulates into its own full-frame sum buffer; the buffers are summed once with
``torch.distributed`` (backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU).  Tile shards touch disjoint pixels,
so their sum is bit-exact; sample-range shards differ from the single-GPU result only by float summation order.
"""
from __future__ import annotations

from dataclasses import dataclass


@dataclass(frozen=True)
class ShardPlan:
    mode: str  # "spp" | "tile"
    rank: int
    world: int
    spp_begin: int
    spp_end: int
    tile_size: int  # 0 in spp mode

    def apply(self, params):
        """Writes the shard into an rt_params structure (include/rt_api.h)."""
        params.spp_begin, params.spp_end = self.spp_begin, self.spp_end
        params.tile_size = self.tile_size
        params.shard_rank = self.rank if self.tile_size else 0
        params.shard_count = self.world if self.tile_size else 1
        return params


def plan_shards(mode: str, rank: int, world: int, spp: int, tile_size: int = 64) -> ShardPlan:
    """spp mode: samples [rank*spp/world, (rank+1)*spp/world) of every pixel.
    tile mode: all samples of the tiles t with t % world == rank."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    if mode == "spp":
        b = (rank * spp) // world
        e = ((rank + 1) * spp) // world
        return ShardPlan("spp", rank, world, b, e, 0)
    if mode == "tile":
        if tile_size <= 0:
            raise ValueError("tile_size must be positive")
        return ShardPlan("tile", rank, world, 0, spp, tile_size if world > 1 else 0)
    raise ValueError(f"unknown shard mode {mode!r}")


def reduce_accum(accum, dst: int | None = 0):
    """Sums the per-rank accumulation buffers (a torch tensor, device or CPU) in place.

    dst = rank that needs the image (``reduce``), or None for every rank (``all_reduce``).  No-op when
    torch.distributed is not initialised or the world is a single rank."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return accum
    if dst is None:
        dist.all_reduce(accum, op=dist.ReduceOp.SUM)
    else:
        dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM)
    return accum
