"""world_size-2 ... 8 gloo tests (CPU) of the N > 1 path: the partition->rank map and the exchange step
(one all-reduce SUM over the counter prefix + one all-reduce MAX over the four extrema) reproduce the
unsharded result; the alive-table MAX merge reproduces sequential last-writer-wins.  Shard-local
results are produced by the oracle here (no GPU); the reduction code is the product's."""
import ctypes as C
import os
import socket

import numpy as np
import pytest

import kafka_topic_analyzer_amd as kta
from kafka_topic_analyzer_amd import _native as N
from kafka_topic_analyzer_amd import distributed as D
from helpers import NOW, random_cols
from oracle_c import Oracle

I64_MAX, I64_MIN = np.iinfo(np.int64).max, np.iinfo(np.int64).min


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def oracle_vector(cols, P, now=NOW):
    """Encode the oracle's result for `cols` in the device counter-vector layout (kta_hip.h)."""
    o = Oracle(now)
    n = len(cols["partition"])
    if n:
        o.run_soa(cols)
    v = np.zeros(P * 7 + 8, dtype=np.int64)
    v[:P * 7] = o.counters(P).astype(np.int64).ravel()
    g = v[P * 7:]
    ts = np.where(cols["ts_ms"] == -1, 0, cols["ts_ms"]) if n else np.zeros(0, np.int64)
    live = cols["val_len"] >= 0 if n else np.zeros(0, bool)
    sizes = (np.maximum(cols["key_len"], 0).astype(np.int64) + np.maximum(cols["val_len"], 0))[live] if n else []
    g[N.KTA_G_RECORDS] = n
    g[N.KTA_G_NOT_MIN_TS_MS] = ~(int(ts.min()) if n else I64_MAX)
    g[N.KTA_G_MAX_TS_MS] = int(ts.max()) if n else I64_MIN
    g[N.KTA_G_NOT_SMALLEST] = ~(int(min(sizes)) if len(sizes) else I64_MAX)
    g[N.KTA_G_LARGEST] = int(max(sizes)) if len(sizes) else 0
    return v, o


def _worker(rank, world, port, P, seed, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(seed)
        cols = random_cols(rng, 20000, P, key_space=300)  # same global topic on every rank
        mine = np.array([D.partition_owner(int(p), world) == rank for p in cols["partition"]])
        shard = {k: (v[mine] if k != "key_bytes" else v) for k, v in cols.items()}
        vec, _ = oracle_vector(shard, P)
        t = torch.from_numpy(vec.copy())
        D.allreduce_counter_vector(t, P)
        # alive table merge on a 2^16-slot stand-in (slot = h & 0xffff), seq = global index
        tbl = np.zeros(1 << 16, dtype=np.int64)
        idx = np.nonzero(mine & (cols["key_len"] >= 0))[0]
        kb = cols["key_bytes"].tobytes()
        from oracle_c import fnv32
        for i in idx:
            k = kb[int(cols["key_off"][i]):int(cols["key_off"][i]) + int(cols["key_len"][i])]
            slot = fnv32(k) & 0xFFFF
            val = ((int(i) + 1) << 1) | (1 if cols["val_len"][i] >= 0 else 0)
            tbl[slot] = max(tbl[slot], val)
        tt = torch.from_numpy(tbl)
        D.allreduce_alive_table(tt, chunk_elems=1 << 14)
        # KTA_G_ALIVE_KEYS through the SUM all-reduce: after a full merge every rank holds the global count
        # ("merged": only rank 0 contributes); with hash-range ownership each rank counts its own share
        ak = P * N.KTA_NCOUNTERS + N.KTA_G_ALIVE_KEYS
        merged_count = int((tt & 1).sum())
        t_merged = torch.from_numpy(vec.copy())
        t_merged[ak] = merged_count
        D.allreduce_counter_vector(t_merged, P, alive_keys="merged")
        lo, hi = (1 << 16) * rank // world, (1 << 16) * (rank + 1) // world
        t_share = torch.from_numpy(vec.copy())
        t_share[ak] = int((tt[lo:hi] & 1).sum())
        D.allreduce_counter_vector(t_share, P, alive_keys="share")
        assert int(t_merged[ak]) == int(t_share[ak]) == merged_count, (int(t_merged[ak]), int(t_share[ak]), merged_count)
        t_merged[ak] = 0
        assert torch.equal(t_merged, t)                      # everything else is untouched by the mode
        q.put((rank, t.numpy().copy(), tt.numpy().copy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,P,seed", [(2, 8, 1), (2, 5, 2), (8, 256, 3), (8, 5, 4)])
def test_exchange_equals_unsharded(world, P, seed):
    """Two ranks, and the target machine's eight: config 4's 256 partitions (32 per rank, p % 8) and a topic with
    fewer partitions than ranks (three ranks own nothing and still end with the job's result)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, P, seed, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=120) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = np.random.default_rng(seed)
    cols = random_cols(rng, 20000, P, key_space=300)
    whole, o = oracle_vector(cols, P)
    lib = N.load()
    for _, vec, tbl in outs:  # every rank ends with the global result
        assert np.array_equal(vec, whole)
        res = N.KtaResult()
        c = np.zeros((P, 7), np.uint64)
        assert lib.kta_decode_vector(vec.ctypes.data, P, 0, C.byref(res), c.ctypes.data) == N.KTA_OK
        assert np.array_equal(c, o.counters(P))
        mm = kta.MessageMetrics(res, c, NOW)
        assert mm.earliest_message() == o.earliest() and mm.latest_message() == o.latest()
        assert mm.smallest_message() == o.get("smallest_message")
        assert mm.largest_message() == o.get("largest_message")
        assert mm.overall_size() == o.get("overall_size")
    # sequential last-writer-wins on the same 2^16-slot stand-in
    from oracle_c import fnv32
    want = {}
    kb = cols["key_bytes"].tobytes()
    for i in range(len(cols["partition"])):
        if cols["key_len"][i] >= 0:
            k = kb[int(cols["key_off"][i]):int(cols["key_off"][i]) + int(cols["key_len"][i])]
            want[fnv32(k) & 0xFFFF] = cols["val_len"][i] >= 0
    for _, _, tbl in outs:
        alive = {int(s) for s in np.nonzero(tbl & 1)[0]}
        assert alive == {s for s, a in want.items() if a}
        assert set(np.nonzero(tbl)[0].tolist()) == set(want.keys())


def test_host_merge_matches_collective_semantics():
    """kta_merge_vectors (the host statement of the reduction) == SUM prefix + MAX suffix."""
    lib = N.load()
    rng = np.random.default_rng(3)
    P = 6
    cols = random_cols(rng, 5000, P)
    half = len(cols["partition"]) // 2
    a, _ = oracle_vector({k: (v[:half] if k != "key_bytes" else v) for k, v in cols.items()}, P)
    b, _ = oracle_vector({k: (v[half:] if k != "key_bytes" else v) for k, v in cols.items()}, P)
    whole, _ = oracle_vector(cols, P)
    k = D.sum_prefix_len(P)
    manual = np.concatenate([a[:k] + b[:k], np.maximum(a[k:], b[k:])])
    assert np.array_equal(manual, whole)
    acc = a.copy()
    assert lib.kta_merge_vectors(acc.ctypes.data, b.ctypes.data, P) == N.KTA_OK
    assert np.array_equal(acc, whole)


def test_shard_spec_is_rank_disjoint():
    sp, _ = kta.synth_preset("c4")
    a = kta.synth_fill_host(D.shard_spec(sp, 0, 2), 0, 5000)
    b = kta.synth_fill_host(D.shard_spec(sp, 1, 2), 0, 5000)
    assert (a["partition"] % 2 == 0).all() and (b["partition"] % 2 == 1).all()
    assert sp.shard_count == 1  # the caller's spec is untouched


def _gather_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = [5, 0, 3][rank]
        slots = torch.arange(n, dtype=torch.int32) + 100 * rank
        vals = (torch.arange(n, dtype=torch.int64) + 1) * (rank + 1)
        gs, gv, sizes = D.gather_entries(slots, vals)
        q.put((rank, sizes, [g[:k].tolist() for g, k in zip(gs, sizes)], [g[:k].tolist() for g, k in zip(gv, sizes)],
               [int(g[k:].abs().sum()) for g, k in zip(gv, sizes)]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_variable_length_entry_gather_three_ranks():
    """The compact alive-table exchange: ragged (slot, value) lists incl. an empty rank, zero padded."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, sizes, slots, vals, pad in outs:
        assert sizes == [5, 0, 3]
        assert slots == [[0, 1, 2, 3, 4], [], [200, 201, 202]]
        assert vals == [[1, 2, 3, 4, 5], [], [3, 6, 9]]
        assert pad == [0, 0, 0]  # padding is value 0 == "never written"


def _route_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # every rank's "table": slots all over the 32-bit range, values ((seq+1)<<1)|alive with global seq
        rng = np.random.default_rng(100 + rank)
        n = [4000, 0, 2500, 1200, 0, 3100, 700, 1][rank]                  # (ranks with nothing to send take part all the same)
        pool = np.unique(np.concatenate([np.random.default_rng(7).integers(0, 1 << 32, 6000, dtype=np.uint64),
                                         np.array([0, (1 << 32) - 1, 1431655765, 1431655766, 2863311530, 2863311531],
                                                  np.uint64)]))   # shared by the ranks: slots collide across shards
        slots_u = rng.choice(pool, n, replace=False) if n else np.zeros(0, np.uint64)
        seq = rng.permutation(20000)[:n].astype(np.uint64) * np.uint64(world) + np.uint64(rank)   # globally unique
        vals_u = ((seq + np.uint64(1)) << np.uint64(1)) | rng.integers(0, 2, n, dtype=np.uint64)
        slots = torch.from_numpy(slots_u.astype(np.uint32).view(np.int32).copy())
        vals = torch.from_numpy(vals_u.view(np.int64).copy())
        rs, rv = D.route_entries_by_hash_range(slots, vals)
        lo, hi = D.hash_range(rank, world)
        got_slots = rs.numpy().view(np.uint32).astype(np.uint64)
        assert ((got_slots >= lo) & (got_slots < hi)).all()           # only my range arrives
        merged = {}
        for sl, v in zip(got_slots.tolist(), rv.numpy().view(np.uint64).tolist()):
            merged[sl] = max(merged.get(sl, 0), v)                    # atomicMax import
        mine = sum(v & 1 for v in merged.values())
        total = torch.tensor([mine], dtype=torch.int64)
        dist.all_reduce(total)
        q.put((rank, slots_u.tolist(), vals_u.tolist(), int(total.item()), len(got_slots)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [3, 8])
def test_hash_range_exchange_counts_the_merged_alive_set(world):
    """SURVEY section 8(e) option ii: entries travel to the owner of their hash range (all-to-all), the owner
    merges by MAX and counts; the sum equals the alive count of the element-wise MAX of all tables.  Three ranks, and
    the target machine's eight."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_route_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    merged = {}
    for _, slots, vals, _, _ in outs:
        for sl, v in zip(slots, vals):
            merged[sl] = max(merged.get(sl, 0), v)
    want = sum(v & 1 for v in merged.values())
    assert [o[3] for o in outs] == [want] * world
    assert sum(o[4] for o in outs) == sum(len(o[1]) for o in outs)   # every entry went to exactly one owner
    for w in (1, 2, 3, 7, 8):                                         # ranges tile [0, 2^32) exactly
        rs = [D.hash_range(r, w) for r in range(w)]
        assert rs[0][0] == 0 and rs[-1][1] == 1 << 32 and all(rs[i][1] == rs[i + 1][0] for i in range(w - 1))
        assert all((lo * w) >> 32 == r and ((hi - 1) * w) >> 32 == r for r, (lo, hi) in enumerate(rs))
