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
           "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-alive",
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
h = kta.HipMetricHandler(8, count_alive_keys=True, now=NOW)
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
