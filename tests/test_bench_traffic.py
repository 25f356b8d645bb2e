"""bench.py replays HBM traffic from profiles/traffic.json (the PMC counters need their own rocprofv3 passes); a
kernel whose source file changed since those passes must not inherit their number."""
import hashlib
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(tmp_root):
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    m.ROOT = str(tmp_root)
    return m


def test_traffic_is_null_once_the_kernel_source_changes(tmp_path):
    (tmp_path / "profiles").mkdir()
    (tmp_path / "csrc").mkdir()
    src = tmp_path / "csrc" / "k.hip"
    src.write_text("__global__ void k() {}\n")
    h = hashlib.sha256(src.read_bytes()).hexdigest()[:16]
    table = {"k": {"records_per_launch": 1000, "hbm_bytes_per_launch": 20480.0, "source_file": "csrc/k.hip", "source_sha256_16": h},
             "old": {"records_per_launch": 1000, "hbm_bytes_per_launch": 1.0}}      # an entry from before the hashes
    (tmp_path / "profiles" / "traffic.json").write_text(json.dumps(table))
    b = _bench(tmp_path)
    assert b._traffic("k", 1000) == 20480.0
    assert b._traffic("k", 999) is None                      # another launch size
    assert b._traffic("old", 1000) is None                   # no hash recorded: not trusted
    assert b._traffic(["k", "old"], 1000) is None
    src.write_text("__global__ void k() { /* touched */ }\n")
    assert b._traffic("k", 1000) is None                     # the kernel is no longer the measured one


def test_a_kernel_in_several_files_is_stale_when_any_of_them_changes(tmp_path):
    (tmp_path / "profiles").mkdir()
    (tmp_path / "csrc").mkdir()
    files = [tmp_path / "csrc" / n for n in ("k.hip", "k_body.h")]
    files[0].write_text('#include "k_body.h"\n')
    files[1].write_text("__global__ void k() {}\n")
    digests = "".join(hashlib.sha256(f.read_bytes()).hexdigest() for f in files)
    table = {"k": {"records_per_launch": 10, "hbm_bytes_per_launch": 5.0, "source_file": "csrc/k.hip+csrc/k_body.h",
                   "source_sha256_16": hashlib.sha256(digests.encode()).hexdigest()[:16]}}
    (tmp_path / "profiles" / "traffic.json").write_text(json.dumps(table))
    b = _bench(tmp_path)
    assert b._traffic("k", 10) == 5.0
    files[1].write_text("__global__ void k() { /* touched */ }\n")
    assert b._traffic("k", 10) is None


def test_committed_traffic_file_matches_the_tree_or_yields_null():
    """Whatever profiles/traffic.json holds, every entry either carries the hash of its source file as committed,
    or bench.py reports null for it."""
    b = _bench(ROOT)
    table = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    for name, e in table.items():
        if not isinstance(e, dict) or "records_per_launch" not in e:
            continue
        got = b._traffic(name, e["records_per_launch"])
        fresh = e.get("source_sha256_16") is not None and e["source_sha256_16"] == b._source_hash(e.get("source_file", ""))
        assert (got is not None) == fresh, name
