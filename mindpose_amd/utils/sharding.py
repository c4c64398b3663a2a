"""Crop sharding for one-process-per-GPU inference.

Mirrors the reference's dataset sharding ``num_shards=device_num, shard_id=rank_id``
(mindpose/data/data_factory.py:59-66): the path partitions into independent crops, so ranks exchange
nothing on the data path; results can be gathered for the evaluator afterwards.
"""
from typing import Tuple


def shard_range(total: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous [begin, end) slice of ``total`` crops owned by ``rank`` (sizes differ by at most 1)."""
    if world_size < 1 or not (0 <= rank < world_size) or total < 0:
        raise ValueError("bad sharding arguments")
    base, rem = divmod(total, world_size)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)
