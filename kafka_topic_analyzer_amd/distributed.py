"""Partition-sharded multi-GPU plumbing over torch.distributed (backend "nccl" is RCCL over xGMI on ROCm,
"gloo" on CPU in the tests).  The PRODUCT path of the exchange is native — kta_comm_create / kta_exchange
(csrc/kta_comm.hip, include/kta_hip.h) — and needs no torch; this module is the same exchange written on
torch collectives, kept as the harness the world-size-2/3 gloo tests drive on CPU and as a cross-check of
the native one.

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


def allreduce_counter_vector(vec, n_partitions: int, group=None, alive_keys: str = "share") -> None:
    """In-place exchange step over an int64 view of the SNAPSHOT counter vector (kta_result_vector: device
    tensor, or a CPU tensor in the tests) — the live accumulator is never reduced.

    KTA_G_ALIVE_KEYS sits in the SUM prefix, so what every rank contributes must add up to the job's count:
    "share"  — each rank's word is the count of a DISJOINT share of the slots (the hash range it owns after
               exchange_alive_by_hash_range; what kta_exchange does natively); the default;
    "merged" — the tables were fully merged first (exchange_alive_entries / allreduce_alive_table), every rank
               holds the global count: all ranks but 0 contribute zero."""
    import torch.distributed as dist
    k = sum_prefix_len(n_partitions)
    assert vec.numel() == n_partitions * N.KTA_NCOUNTERS + N.KTA_NGLOBALS
    assert alive_keys in ("share", "merged")
    if alive_keys == "merged" and dist.get_rank(group) != 0:
        vec[n_partitions * N.KTA_NCOUNTERS + N.KTA_G_ALIVE_KEYS] = 0
    dist.all_reduce(vec[:k], op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(vec[k:], op=dist.ReduceOp.MAX, group=group)


def allreduce_alive_table(table, group=None, chunk_elems: int = 1 << 28) -> None:
    """Element-wise MAX of the last-writer tables, in place, chunked (2 GiB of int64 per call) so
    RCCL's staging stays bounded.  Values are < 2^63, so signed MAX == unsigned MAX."""
    import torch.distributed as dist
    n = table.numel()
    for lo in range(0, n, chunk_elems):
        dist.all_reduce(table[lo:min(lo + chunk_elems, n)], op=dist.ReduceOp.MAX, group=group)


def gather_entries(slots, vals, group=None):
    """all_gather of variable-length (slot u32-as-i32, value u64-as-i64) entry lists -> per-rank lists.
    Works on device tensors (RCCL) and CPU tensors (gloo).  Shorter lists are zero padded: value 0 means
    "never written" and is ignored by the import kernel."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    n = torch.tensor([slots.numel()], dtype=torch.int64, device=slots.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    m = max(max(sizes), 1)
    ps = torch.zeros(m, dtype=slots.dtype, device=slots.device)
    pv = torch.zeros(m, dtype=vals.dtype, device=vals.device)
    ps[:slots.numel()] = slots
    pv[:vals.numel()] = vals
    gs = [torch.empty_like(ps) for _ in range(world)]
    gv = [torch.empty_like(pv) for _ in range(world)]
    dist.all_gather(gs, ps, group=group)
    dist.all_gather(gv, pv, group=group)
    return gs, gv, sizes


class _DevArray:
    """Zero-copy torch view of library-owned device memory."""

    def __init__(self, ptr: int, n: int, typestr: str):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


def exchange_alive_entries(handler, device, group=None) -> int:
    """Merge the alive-key state of all ranks into every rank, compactly: each rank exports the table
    entries it ever wrote (<= one per distinct key hash: 12 B each), the lists are all-gathered, and every
    rank imports the foreign entries with atomicMax — the element-wise MAX of the tables without moving
    32 GiB per rank.  Returns the number of foreign entries imported."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank(group)
    ps, pv, n = handler.alive_export_entries()
    dev = torch.device("cuda", device)
    if n:
        slots = torch.as_tensor(_DevArray(ps, n, "<i4"), device=dev)
        vals = torch.as_tensor(_DevArray(pv, n, "<i8"), device=dev)
    else:
        slots = torch.zeros(0, dtype=torch.int32, device=dev)
        vals = torch.zeros(0, dtype=torch.int64, device=dev)
    gs, gv, sizes = gather_entries(slots, vals, group)
    torch.cuda.synchronize(dev)            # gathered lists complete before the library's stream reads them
    imported = 0
    for r, k in enumerate(sizes):
        if r != rank and k:
            handler.alive_import_entries(gs[r].data_ptr(), gv[r].data_ptr(), k)
            imported += k
    handler.sync()                         # the gathered tensors may be freed after this
    return imported


# ---- hash-range exchange (option ii) ------------------------------------------------
# exchange_alive_entries leaves the WHOLE merged table on every rank: every rank receives and imports
# every other rank's entries (traffic and atomics grow with the world size).  When only the number of
# alive keys is wanted — which is all the report prints (src/main.rs:141-147) — a hash range per rank
# suffices: rank r owns the slots [ceil(r * 2^32 / world), ceil((r+1) * 2^32 / world)), receives only the
# entries of its range (all-to-all: every link carries 1/world of the entries, all 7 xGMI links in
# parallel), merges them by MAX, counts its range and the counts are summed.

def hash_range(rank: int, world: int):
    """Slots owned by `rank`: owner(slot) == (slot * world) >> 32."""
    lo = -((-rank * (1 << 32)) // world)
    hi = -((-(rank + 1) * (1 << 32)) // world)
    return lo, hi


def route_entries_by_hash_range(slots, vals, group=None):
    """all-to-all of (slot u32-as-i32, value u64-as-i64) entries: each entry goes to the rank that owns
    its slot.  Returns (slots, vals) received by this rank (its own entries of its range included).
    Works on device tensors (RCCL) and CPU tensors (gloo)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    owner = ((slots.to(torch.int64) & 0xFFFFFFFF) * world) >> 32
    order = torch.argsort(owner, stable=True)
    send_slots, send_vals = slots[order].contiguous(), vals[order].contiguous()
    send_counts = torch.bincount(owner, minlength=world).to(torch.int64)
    recv_counts = torch.empty_like(send_counts)
    dist.all_to_all_single(recv_counts, send_counts, group=group)
    ss, rs = send_counts.tolist(), recv_counts.tolist()
    recv_slots = torch.empty(sum(rs), dtype=slots.dtype, device=slots.device)
    recv_vals = torch.empty(sum(rs), dtype=vals.dtype, device=vals.device)
    dist.all_to_all_single(recv_slots, send_slots, rs, ss, group=group)
    dist.all_to_all_single(recv_vals, send_vals, rs, ss, group=group)
    return recv_slots, recv_vals


def exchange_alive_by_hash_range(handler, device, group=None) -> int:
    """The global number of alive keys of a partition-sharded run (every rank gets the same number).
    Afterwards a rank's table is merged for its own hash range only."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    ps, pv, n = handler.alive_export_entries()
    dev = torch.device("cuda", device)
    if n:
        slots = torch.as_tensor(_DevArray(ps, n, "<i4"), device=dev)
        vals = torch.as_tensor(_DevArray(pv, n, "<i8"), device=dev)
    else:
        slots = torch.zeros(0, dtype=torch.int32, device=dev)
        vals = torch.zeros(0, dtype=torch.int64, device=dev)
    rslots, rvals = route_entries_by_hash_range(slots, vals, group)
    torch.cuda.synchronize(dev)            # received lists complete before the library's stream reads them
    if rslots.numel():                     # own entries are among them: atomicMax makes that a no-op
        handler.alive_import_entries(rslots.data_ptr(), rvals.data_ptr(), rslots.numel())
    lo, hi = hash_range(rank, world)
    mine = handler.alive_count_range(lo, hi)   # synchronous: the received tensors may be freed after this
    total = torch.tensor([mine], dtype=torch.int64, device=dev)
    dist.all_reduce(total, op=dist.ReduceOp.SUM, group=group)
    return int(total.item())
