"""GPU tests of the exchange step on real RCCL with a 1-rank process group (the build container
reaches one GPU): the collectives run on the library's own device memory."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_with_forced_collectives_one_rank():
    env = dict(os.environ, KTA_BENCH_FORCE_COLLECTIVES="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", "29517", os.path.join(ROOT, "bench.py"),
           "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-alive", "--no-decode", "--no-hostfed",
           "--records-per-gpu", str(1 << 24)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["config"]["forced_collectives"] is True and line["n_gpus"] == 1
    assert line["value"] > 1e9 and line["roofline"]["frac"] > 0.1


_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import torch, torch.distributed as dist
import numpy as np
import kafka_topic_analyzer_amd as kta
from kafka_topic_analyzer_amd import distributed as D
from helpers import random_cols, NOW
from oracle_c import Oracle
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29518")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
class V:
    def __init__(s, p, n): s.__cuda_array_interface__ = {"shape": (n,), "typestr": "<i8", "data": (p, False), "version": 2}
rng = np.random.default_rng(9)
cols = random_cols(rng, 100000, 8, key_space=2000, tomb=0.4)
o = Oracle(NOW, True); o.run_soa(cols)
h = kta.HipMetricHandler(8, count_alive_keys=True, now=NOW, alive_table=True)
h.submit_columns(**cols); h.finish_device(); h.sync()
p, n = h.result_vector()
vec = torch.as_tensor(V(p, n), device="cuda:0")
before = vec.clone()
D.allreduce_counter_vector(vec, 8)
torch.cuda.synchronize()
assert torch.equal(vec, before)                      # 1 rank: the reduction is the identity
tp, tn = h.alive_table()
table = torch.as_tensor(V(tp, tn), device="cuda:0")
D.allreduce_alive_table(table)                       # 32 GiB, chunked MAX all-reduce on RCCL
torch.cuda.synchronize()
h.alive_table_modified()
res, c = h.finish()
assert res.alive_keys == o.alive_keys(), (res.alive_keys, o.alive_keys())
assert np.array_equal(c, o.counters(8))
assert D.exchange_alive_entries(h, 0) == 0            # compact exchange on RCCL: nothing foreign with one rank
res2, _ = h.finish()
assert res2.alive_keys == o.alive_keys()
ps, pv, ne = h.alive_export_entries()
assert ne >= res2.alive_keys
assert D.exchange_alive_by_hash_range(h, 0) == o.alive_keys()   # hash-range owner exchange (all-to-all) on RCCL
h.close(); dist.destroy_process_group(); print("OK")
'''


def test_alive_table_allreduce_on_rccl_one_rank(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(_WORKER)
    r = subprocess.run([sys.executable, str(script), ROOT], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "OK" in r.stdout, r.stderr[-3000:]


# ---- the native exchange (csrc/kta_comm.hip: RCCL behind the C ABI, no torch in the data path) ---------------
_NATIVE_WORKER = r'''
import os, sys, time
root, rank, nranks, idfile = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import kafka_topic_analyzer_amd as kta
from helpers import random_cols, NOW
from oracle_c import Oracle
P = 12
rng = np.random.default_rng(77)
cols = random_cols(rng, 300000, P, key_space=5000, tomb=0.4)       # the whole topic, identical on every rank
o = Oracle(NOW, True); o.run_soa(cols)
n = len(cols["partition"])
seq = np.arange(n, dtype=np.uint64)                                # GLOBAL consumption order
idx = np.nonzero(cols["partition"] % nranks == rank)[0]            # this rank's partitions
kl = np.maximum(cols["key_len"][idx], 0).astype(np.int64)
off = np.zeros(len(idx), np.int64); off[1:] = np.cumsum(kl)[:-1]
kb = np.zeros(max(int(kl.sum()), 1), np.uint8)
src = cols["key_off"][idx].astype(np.int64)
pos = np.repeat(src, kl) + (np.arange(int(kl.sum())) - np.repeat(off, kl))
kb[:int(kl.sum())] = cols["key_bytes"][pos]
shard = {"partition": cols["partition"][idx], "key_len": cols["key_len"][idx], "val_len": cols["val_len"][idx],
         "ts_ms": cols["ts_ms"][idx], "key_off": off.astype(np.uint32), "key_bytes": kb[:int(kl.sum())], "seq": seq[idx]}
h = kta.HipMetricHandler(P, count_alive_keys=True, now=NOW, alive_table=True)
if nranks > 1:
    if rank == 0:
        uid = kta.HipMetricHandler.comm_unique_id()
        with open(idfile + ".tmp", "wb") as f: f.write(uid)
        os.rename(idfile + ".tmp", idfile)
    else:
        t0 = time.time()
        while not os.path.exists(idfile):
            assert time.time() - t0 < 120
            time.sleep(0.05)
        uid = open(idfile, "rb").read()
else:
    uid = None
try:
    h.comm_create(nranks, rank, uid)
except kta.KtaError as e:
    print("COMM_CREATE_FAILED", e); sys.exit(3)
b, nb = h.upload_batch(shard, with_keys=True)
h.submit_device(b, nb, 0)
h.exchange()
res, c = h.exchange_result()
assert res.alive_keys == o.alive_keys(), (rank, res.alive_keys, o.alive_keys())
assert np.array_equal(c, o.counters(P))
assert res.overall_count == n
# the snapshot was reduced, the accumulator was not: a second exchange gives the same answer
h.exchange()
res2, c2 = h.exchange_result()
assert res2.alive_keys == o.alive_keys() and np.array_equal(c2, c)
ends = np.zeros(P, np.int64); ends[rank::nranks] = 1000 + np.arange(P)[rank::nranks]
got = h.comm_allreduce_i64(ends)
assert np.array_equal(got, 1000 + np.arange(P))
assert np.array_equal(h.comm_allreduce_i64(np.array([rank, -rank], np.int64), op_max=True), [nranks - 1, 0])
nr, rk, sent, recv = h.comm_info()
assert (nr, rk) == (nranks, rank) and (nranks == 1 or sent + recv > 0)
h.device_batch_free(b); h.comm_destroy(); h.close(); print("OK", rank, sent, recv)
'''


@pytest.mark.parametrize("nranks", [1, 2])
def test_native_exchange_over_rccl(tmp_path, nranks):
    """kta_comm_create / kta_exchange: partition-sharded ranks with global sequence numbers, hash-range
    exchange of the alive entries + grouped SUM / MAX all-reduce of the snapshot vector, every rank ends with
    the unsharded oracle's result.  Two ranks share the one reachable GPU (RCCL permitting)."""
    script = tmp_path / "w.py"
    script.write_text(_NATIVE_WORKER)
    idfile = str(tmp_path / "rccl_id")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG="WARN")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(r), str(nranks), idfile], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True, env=env) for r in range(nranks)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=600) + (p.returncode,))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    if nranks > 1 and any(rc == 3 for _, _, rc in outs):
        pytest.skip("RCCL refused two ranks on one GPU: " + " | ".join(o[-300:] for o, _, _ in outs))
    for out, err, rc in outs:
        assert rc == 0 and "OK" in out, (out[-2000:], err[-3000:])


_THREADED_WORKER = r'''
import os, sys, threading
root, nranks, P, alive = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4] == "1"
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import kafka_topic_analyzer_amd as kta
from helpers import random_cols, NOW
from oracle_c import Oracle
rng = np.random.default_rng(5 + nranks)
cols = random_cols(rng, 400000, P, key_space=30000, tomb=0.35)
o = Oracle(NOW, alive); o.run_soa(cols)
n = len(cols["partition"])
seq = np.arange(n, dtype=np.uint64)
def shard_of(rank):
    idx = np.nonzero(cols["partition"] % nranks == rank)[0]
    kl = np.maximum(cols["key_len"][idx], 0).astype(np.int64)
    off = np.zeros(len(idx), np.int64); off[1:] = np.cumsum(kl)[:-1]
    tot = int(kl.sum())
    kb = np.zeros(max(tot, 1), np.uint8)
    pos = np.repeat(cols["key_off"][idx].astype(np.int64), kl) + (np.arange(tot) - np.repeat(off, kl))
    kb[:tot] = cols["key_bytes"][pos]
    return {"partition": cols["partition"][idx], "key_len": cols["key_len"][idx], "val_len": cols["val_len"][idx],
            "ts_ms": cols["ts_ms"][idx], "key_off": off.astype(np.uint32), "key_bytes": kb[:tot], "seq": seq[idx]}
uid = kta.HipMetricHandler.comm_unique_id()
errors, stats = [], [None] * nranks
def run(rank):
    try:
        h = kta.HipMetricHandler(P, count_alive_keys=alive, now=NOW, alive_table=alive)
        h.comm_create(nranks, rank, uid)
        b, nb = h.upload_batch(shard_of(rank), with_keys=alive)
        h.submit_device(b, nb, 0)
        for _ in range(2):                     # the second exchange finds every owner's range already merged
            h.exchange()
            res, c = h.exchange_result()
            assert not alive or res.alive_keys == o.alive_keys(), (rank, res.alive_keys, o.alive_keys())
            assert np.array_equal(c, o.counters(P)) and res.overall_count == n
            assert res.smallest_message == o.get("smallest_message") and res.largest_message == o.get("largest_message")
        if alive:
            lo, hi = -((-rank * (1 << 32)) // nranks), -((-(rank + 1) * (1 << 32)) // nranks)
            words = h.export_alive_bitmap()    # this rank's table is the merged one on its own hash range
            want = o.alive_words()
            wl, wh = (lo + 31) // 32, hi // 32
            assert np.array_equal(words[wl:wh], want[wl:wh]), rank
        ends = np.zeros(P, np.int64); ends[rank::nranks] = 7
        assert np.array_equal(h.comm_allreduce_i64(ends), np.full(P, 7))
        stats[rank] = h.comm_info()
        h.device_batch_free(b); h.comm_destroy(); h.close()
    except BaseException as e:
        errors.append((rank, repr(e)))
        os._exit(2)                            # the other ranks would wait at the next rendezvous for ever
ts = [threading.Thread(target=run, args=(r,)) for r in range(nranks)]
[t.start() for t in ts]; [t.join() for t in ts]
assert not errors, errors
assert sum(s[2] for s in stats) == sum(s[3] for s in stats)          # every entry sent was received
assert not alive or sum(s[2] for s in stats) > 0
print("OK", stats)
'''


def _build_mock_rccl(tmp_path):
    lib = tmp_path / "libmock_rccl.so"
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O1", "-shared", "-fPIC", "-std=c++17", os.path.join(ROOT, "tests", "mock_rccl.cpp"),
                        "-o", str(lib), "-lrt", "-lpthread"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return lib


@pytest.mark.parametrize("nranks,P,alive", [(2, 10, True), (3, 10, True), (4, 256, True), (8, 256, False)])
def test_native_exchange_logic_with_rccl_test_double(tmp_path, nranks, P, alive):
    """The multi-rank paths of csrc/kta_comm.hip (per-owner export, count all-gather, grouped send / recv,
    owner merge + range count, SUM / MAX all-reduce) with 2, 3, 4 and 8 ranks as threads of one process on the one
    reachable GPU, RCCL replaced by tests/mock_rccl.cpp: every rank ends with the unsharded oracle's counters
    and alive count, and with the oracle's BitSet on the hash range it owns.  The target machine has eight GPUs:
    config 4's sharding (256 partitions, 32 per rank, p % 8) runs with all eight communicator ranks; with -c every
    rank keeps a 32 GiB table, of which four fit the one GPU here and eight do not (8 x 32 GiB + workspaces > 288 GB),
    so the hash-range exchange runs with four owners (hash_range(r, 4))."""
    lib = _build_mock_rccl(tmp_path)
    script = tmp_path / "w.py"
    script.write_text(_THREADED_WORKER)
    env = dict(os.environ, KTA_RCCL_LIBRARY=str(lib))
    r = subprocess.run([sys.executable, str(script), ROOT, str(nranks), str(P), "1" if alive else "0"], capture_output=True,
                       text=True, timeout=900, env=env)
    assert r.returncode == 0 and "OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


# ---- the driver's own launch line: one bench.py process per rank ------------------------------------------------
def _run_bench_ranks(tmp_path, nproc, port, extra, timeout=900):
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...` as the driver launches it,
    on the one reachable GPU: every rank process on device 0 (KTA_BENCH_SHARE_DEVICE=1), the library's RCCL replaced by the
    process-capable test double (tests/mock_rccl.cpp: shared-memory rendezvous keyed by the unique id).  Returns rank 0's
    JSON line."""
    lib = _build_mock_rccl(tmp_path)
    env = dict(os.environ, KTA_RCCL_LIBRARY=str(lib), KTA_RCCL_ONLY_ENV="1", KTA_BENCH_SHARE_DEVICE="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", str(nproc), "--steps", "3", "--warmup", "1", "--preroll", "2", "--no-cpu-baseline", "--no-alive",
           "--no-decode", "--no-hostfed"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-4000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                 # ONE JSON line, from rank 0
    return json.loads(lines[0])


@pytest.mark.parametrize("mode", ["weak", "strong"])
def test_bench_two_rank_processes_config_4(tmp_path, mode):
    """bench.py's N > 1 path with two rank PROCESSES (gloo rendezvous + barriers, ONE RCCL per process: the library's):
    config 4's sharding p -> rank p % 2, weak and strong scaling.  bench.py itself asserts that the exchanged result holds
    the whole job's record count and every partition; here the line's own bookkeeping is checked."""
    n = 1 << 23
    line = _run_bench_ranks(tmp_path, 2, 29531 if mode == "weak" else 29532, ["--records-per-gpu", str(n), "--scaling", mode])
    per_rank = n if mode == "weak" else n // 2
    assert line["n_gpus"] == 2 and line["scaling"] == mode and line["steps"] == 3
    assert line["config"]["records_per_gpu"] == per_rank and line["config"]["total_records_per_step"] == 2 * per_rank
    assert line["config"]["shared_device"] is True and line["config"]["forced_collectives"] is False
    assert "gloo" in line["config"]["control_plane"] and "kta_exchange" in line["config"]["exchange"]
    assert abs(line["value"] - 2 * per_rank * 3 / (line["ms_per_step"] * 3e-3)) < 1e-3 * line["value"]
    assert line["roofline"]["launches"] == 3 and line["roofline"]["frac"] > 0.02


def test_bench_two_rank_processes_config_5(tmp_path):
    """--config c5 with two rank processes: both handlers per step (the fused table pass) and the WHOLE exchange — alive
    entries to their hash-range owners (count all-gather, grouped send / recv, owner merge), then the grouped all-reduces.
    The alive count of the exchanged result against the C oracle: every pass submits the same records with later sequence
    numbers, rank 0's before rank 1's, so what is left is rank 0's batch followed by rank 1's."""
    import kafka_topic_analyzer_amd as kta
    from kafka_topic_analyzer_amd import distributed as D
    from oracle_c import Oracle
    n = 1 << 21
    line = _run_bench_ranks(tmp_path, 2, 29533, ["--records-per-gpu", str(n), "--config", "c5"])
    assert line["n_gpus"] == 2 and line["config"]["records_per_gpu"] == n
    spec, _ = kta.synth_preset("c5")
    o = Oracle(count_alive_keys=True)
    for rank in range(2):
        o.run_soa(kta.synth_fill_host(D.shard_spec(spec, rank, 2), rank * n, n, with_keys=True))
    assert line["alive_pass"]["alive_keys"] == o.alive_keys() > 0
    o.close()


def test_bench_eight_rank_processes_config_4(tmp_path):
    """The target machine's eight ranks as eight bench.py processes (without -c: eight 32 GiB tables do not fit one GPU):
    config 4's sharding, 32 partitions per rank, one grouped all-reduce per step."""
    n = 1 << 21
    line = _run_bench_ranks(tmp_path, 8, 29534, ["--records-per-gpu", str(n)], timeout=1200)
    assert line["n_gpus"] == 8 and line["config"]["total_records_per_step"] == 8 * n
    assert line["config"]["parallelism"] == "partition-sharded x8"
