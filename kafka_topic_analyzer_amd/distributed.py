"""Partition-sharded multi-GPU plumbing (one process per GPU, torch.distributed; backend "nccl" is
RCCL over xGMI on ROCm, "gloo" on CPU in the tests).

The reference is single-process (src/kafka.rs:92-135), but every per-partition counter depends only
on its own partition's records (src/metric.rs:74-100) and the globals are min/max/sum, so Kafka
partitions shard across ranks with no data-path collective and ONE exchange step at the end:

    C1  all-reduce SUM  over vec[0 : P*7 + 4]      (counters + SUM-type globals; i64 wrap == u64 wrap)
    C2  all-reduce MAX  over vec[P*7 + 4 : P*7+8]  (~min ts, max ts, ~smallest, largest; signed)

The alive-key set is global and order dependent (src/metric.rs:262-264, 289-304): each rank keeps the
last-writer table for its partitions (entry = ((seq+1)<<1)|alive with seq the GLOBAL consumption
index), and an element-wise MAX across ranks is exactly "the last writer in consumption order wins".
"""
from __future__ import annotations

import copy

from . import _native as N


def shard_spec(spec, rank: int, world: int):
    """The synthetic-topic shard of `rank`: partitions p with p % world == rank."""
    s = copy.copy(spec)
    s.shard_index, s.shard_count = rank, world
    return s


def partition_owner(partition: int, world: int) -> int:
    return partition % world


def sum_prefix_len(n_partitions: int) -> int:
    return n_partitions * N.KTA_NCOUNTERS + N.KTA_NSUM_GLOBALS


def allreduce_counter_vector(vec, n_partitions: int, group=None) -> None:
    """In-place exchange step over an int64 view of the counter vector (device or CPU tensor)."""
    import torch.distributed as dist
    k = sum_prefix_len(n_partitions)
    assert vec.numel() == n_partitions * N.KTA_NCOUNTERS + N.KTA_NGLOBALS
    dist.all_reduce(vec[:k], op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(vec[k:], op=dist.ReduceOp.MAX, group=group)


def allreduce_alive_table(table, group=None, chunk_elems: int = 1 << 28) -> None:
    """Element-wise MAX of the last-writer tables, in place, chunked (2 GiB of int64 per call) so
    RCCL's staging stays bounded.  Values are < 2^63, so signed MAX == unsigned MAX."""
    import torch.distributed as dist
    n = table.numel()
    for lo in range(0, n, chunk_elems):
        dist.all_reduce(table[lo:min(lo + chunk_elems, n)], op=dist.ReduceOp.MAX, group=group)
