"""The drop-in surface around the hot path: the report printer (src/main.rs:123-178) against the golden
text produced by the independent Python restatement, and the CLI's flag handling (main.rs:32-92).
CPU tests; the end-to-end CLI run on a topic dump is a gpu test at the bottom."""
import ctypes as C
import os
import re
import struct
import subprocess

import numpy as np
import pytest

import kafka_topic_analyzer_amd as kta
from kafka_topic_analyzer_amd import _native as N
from helpers import GOLDEN, load_golden, records_to_cols, scenario_records
from test_dist import oracle_vector

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "kafka_topic_analyzer_amd", "kta-analyzer")


def render(topic, secs, vec, P, alive, now, start=None, end=None):
    lib = N.load()
    n = C.c_size_t()
    buf = C.create_string_buffer(1 << 20)
    so = None if start is None else np.asarray(start, np.int64).ctypes.data
    eo = None if end is None else np.asarray(end, np.int64).ctypes.data
    rc = lib.kta_render_report(topic.encode(), secs, vec.ctypes.data, P, alive, now[0], now[1], so, eo, buf,
                               len(buf), C.byref(n))
    return rc, buf.value.decode()


@pytest.mark.parametrize("with_c", [False, True])
def test_report_matches_golden_text(with_c):
    g = load_golden("scenarios.json")
    P = g["n_partitions"]
    sc = g["scenarios"]["mixed_400"]
    cols = records_to_cols(scenario_records(sc))
    vec, o = oracle_vector(cols, P, tuple(g["now"]))
    vec[P * 7 + N.KTA_G_ALIVE_KEYS] = sc["expect"]["alive_keys"]
    rc, text = render("synthetic.mixed_400", 3, vec, P, 1 if with_c else 0, tuple(g["now"]))
    assert rc == N.KTA_OK
    want = open(os.path.join(GOLDEN, "report_mixed_400_%s.txt" % ("with_c" if with_c else "without_c"))).read()
    assert text == want


def test_report_sentinels_and_panic():
    g = load_golden("scenarios.json")
    P, now = g["n_partitions"], tuple(g["now"])
    # only tombstones: "Smallest Message: 0", earliest/latest from the record timestamps
    cols = records_to_cols(scenario_records(g["scenarios"]["only_tombstones"]))
    vec, _ = oracle_vector(cols, P, now)
    rc, text = render("t", 0, vec, P, 0, now)
    assert rc == N.KTA_OK
    assert "Smallest Message: 0 bytes\n" in text and "Largest Message: 0 bytes\n" in text
    assert "Earliest Message: 1970-01-01 00:00:01 UTC\n" in text
    assert "Estimated Msg/s: 2\n" in text  # overall_count / max(secs, 1)
    # timestamp later than "now": the Utc::now() sentinel (with fractional seconds) is printed
    cols = records_to_cols(scenario_records(g["scenarios"]["ts_future_beyond_now"]))
    vec, _ = oracle_vector(cols, P, now)
    rc, text = render("t", 1, vec, P, 0, now)
    assert "Earliest Message: 2100-01-01 00:00:00.123456789 UTC\n" in text
    assert "Latest Message: 2100-01-01 00:00:10 UTC\n" in text
    # keyed tombstones only: the reference panics in key_size_avg (metric.rs:135, main.rs:154)
    cols = records_to_cols(scenario_records(g["scenarios"]["keyed_tombstones_only_panics"]))
    vec, _ = oracle_vector(cols, P, now)
    rc, _ = render("t", 1, vec, P, 0, now)
    assert rc == N.KTA_ERR_DIV_BY_ZERO


def test_report_dirty_ratio_formatting_matches_golden_rows():
    g = load_golden("scenarios.json")
    P, now = g["n_partitions"], tuple(g["now"])
    sc = g["scenarios"]["mixed_400"]
    vec, _ = oracle_vector(records_to_cols(scenario_records(sc)), P, now)
    _, text = render("t", 1, vec, P, 0, now)
    rows = [l for l in text.split("\n") if re.match(r"^\| \d", l)]
    for p, row in enumerate(rows):
        cells = [c.strip() for c in row.strip("|").split("|")]
        assert cells[6] == sc["expect"]["partitions"][p]["dirty_ratio_4"]
        assert cells[0] == str(p) and cells[1] == "0" and cells[2] == cells[3]


# ------------------------------------------------------------------------------------------ CLI flags
def run_cli(*args):
    return subprocess.run([CLI, *args], capture_output=True, text=True, timeout=120)


def test_cli_version_and_help():
    r = run_cli("--version")
    assert r.returncode == 0 and r.stdout == "Kafka Topic Analyzer 0.4.1\n"  # main.rs:35 (not Cargo's 0.5.0)
    r = run_cli("-V")
    assert r.stdout == "Kafka Topic Analyzer 0.4.1\n"
    r = run_cli("--help")
    assert r.returncode == 0
    for flag in ("-t, --topic <TOPIC>", "-b, --bootstrap-server <BOOTSTRAP_SERVER>", "--librdkafka <LIBRDKAFKA>",
                 "-c, --count-alive-keys", "-h, --help", "-V, --version"):
        assert flag in r.stdout


def test_cli_required_arguments():
    r = run_cli("-t", "x")
    assert r.returncode == 1 and "--bootstrap-server <BOOTSTRAP_SERVER>" in r.stderr
    r = run_cli("-b", "synthetic://c1")
    assert r.returncode == 1 and "--topic <TOPIC>" in r.stderr
    r = run_cli("--nope")
    assert r.returncode == 1 and "wasn't expected" in r.stderr


def test_cli_kta_gpus_is_parsed_strictly():
    for bad in ("two", "-1", "0", "2x", ""):
        r = run_cli("-t", "x", "-b", "synthetic://c1", "--librdkafka", "kta.gpus=" + bad)
        assert r.returncode == 2 and "kta.gpus=" in r.stderr and "expected a number" in r.stderr, (bad, r.stderr)


def test_cli_librdkafka_pair_without_equals_panics():
    r = run_cli("-t", "x", "-b", "synthetic://c1", "--librdkafka", "a=b,broken")
    assert r.returncode == 101 and "panicked" in r.stderr and "src/main.rs:89" in r.stderr


def test_cli_refuses_real_brokers_and_unknown_topics():
    r = run_cli("-t", "x", "-b", "localhost:9092")
    assert r.returncode == 101 and "librdkafka" in r.stderr
    r = run_cli("-t", "x", "-b", "synthetic://c9")
    assert r.returncode == 101 and "unknown synthetic topic" in r.stderr
    r = run_cli("-t", "x", "-b", "dump:///nonexistent/file")
    assert r.returncode == 101


def write_dump(path, cols, P):
    n = len(cols["partition"])
    totals = np.bincount(cols["partition"], minlength=P).astype(np.int64)
    with open(path, "wb") as f:
        f.write(b"KTADUMP1" + struct.pack("<IIQQ", 1, P, n, 1))
        f.write(np.zeros(P, np.int64).tobytes() + totals.tobytes())
        kb = np.ascontiguousarray(cols["key_bytes"], np.uint8)
        f.write(struct.pack("<QQ", n, len(kb)))
        for name, dt in (("partition", np.int32), ("key_len", np.int32), ("val_len", np.int32), ("ts_ms", np.int64),
                         ("key_off", np.uint32)):
            b = np.ascontiguousarray(cols[name], dt).tobytes()
            f.write(b + b"\0" * ((-len(b)) % 8))
        b = kb.tobytes()
        f.write(b + b"\0" * ((-len(b)) % 8))


def _normalise(text):
    text = re.sub(r"Scanning took: \d+ seconds", "Scanning took: 3 seconds", text)
    return re.sub(r"Estimated Msg/s: \d+", "Estimated Msg/s: 133", text)


@pytest.mark.gpu
@pytest.mark.parametrize("with_c", [False, True])
def test_cli_end_to_end_on_topic_dump_matches_golden_report(tmp_path, with_c):
    g = load_golden("scenarios.json")
    cols = records_to_cols(scenario_records(g["scenarios"]["mixed_400"]))
    path = str(tmp_path / "mixed_400.ktadump")
    write_dump(path, cols, g["n_partitions"])
    args = ["-t", "synthetic.mixed_400", "-b", "dump://" + path, "--librdkafka", "kta.batch=128"]
    if with_c:
        args.append("-c")
    r = run_cli(*args)
    assert r.returncode == 0, r.stderr
    head = "Subscribing to synthetic.mixed_400\nStarting message consumption...\n"
    assert r.stdout.startswith(head)
    want = open(os.path.join(GOLDEN, "report_mixed_400_%s.txt" % ("with_c" if with_c else "without_c"))).read()
    assert _normalise(r.stdout[len(head):]) == want


def test_cli_rejects_a_repeated_flag_like_clap():
    """-c is not `multiple(true)` (main.rs:60-66): clap 2 refuses the second occurrence before main.rs:77-80
    could ever see occurrences_of != 1."""
    r = run_cli("-t", "x", "-b", "synthetic://c1", "-c", "--count-alive-keys")
    assert r.returncode == 1 and "provided more than once" in r.stderr and "USAGE" in r.stderr


@pytest.mark.gpu
def test_cli_refuses_a_topic_dump_whose_keys_point_outside_the_batch(tmp_path):
    """An untrusted KTADUMP1 file: key_off + key_len beyond the batch's key bytes, or a length below -1, must
    not be copied from (a heap over-read) — the batch is refused like a truncated file (warn, go on)."""
    g = load_golden("scenarios.json")
    cols = records_to_cols(scenario_records(g["scenarios"]["mixed_400"]))
    for mutate in ("off", "len"):
        bad = {k: np.array(v, copy=True) for k, v in cols.items()}
        keyed = np.nonzero(bad["key_len"] > 0)[0]
        if mutate == "off":
            bad["key_off"][keyed[3]] = np.uint32(len(bad["key_bytes"]) + 1000)
        else:
            bad["key_len"][keyed[3]] = -7
        path = str(tmp_path / ("bad_%s.ktadump" % mutate))
        write_dump(path, bad, g["n_partitions"])
        r = run_cli("-t", "bad", "-b", "dump://" + path, "-c")
        assert "truncated topic dump" in r.stderr and r.returncode in (0, 101), (r.returncode, r.stderr[-500:])


@pytest.mark.gpu
def test_cli_per_message_handler_path_equals_column_path():
    """kta.per_message=1: one MetricHandler::handle_message call per record through the C++ mirror of
    the reference's handlers == the batched column path, byte for byte (incl. Alive keys)."""
    a = run_cli("-t", "c2", "-b", "synthetic://c2?records=200000", "-c", "--librdkafka", "kta.per_message=1,kta.batch=4096")
    b = run_cli("-t", "c2", "-b", "synthetic://c2?records=200000", "-c")
    assert a.returncode == 0 and b.returncode == 0, a.stderr + b.stderr
    assert _normalise(a.stdout) == _normalise(b.stdout)
    assert "Alive keys: " in a.stdout


@pytest.mark.gpu
def test_cli_synthetic_topic_and_dump_round_trip(tmp_path):
    path = str(tmp_path / "c2.ktadump")
    a = run_cli("-t", "c2", "-b", "synthetic://c2?records=300000", "-c", "--librdkafka", "kta.write_dump=" + path)
    assert a.returncode == 0, a.stderr
    b = run_cli("-t", "c2", "-b", "dump://" + path, "-c")
    assert b.returncode == 0, b.stderr
    assert _normalise(a.stdout) == _normalise(b.stdout)
    assert "Alive keys: " in a.stdout and a.stdout.count("\n| ") >= 8


@pytest.mark.gpu
@pytest.mark.parametrize("with_c", [False, True])
def test_cli_on_raw_kafka_log_segments(tmp_path, with_c):
    """segment://: broker `*.log` files (record-batch v2) decoded on the GPU; the printed report must be
    the one the Python restatement prints for the records the independent encoder put into the files."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as OP
    from kafka_cases import random_record_set
    rng = np.random.default_rng(5)
    files, records, starts, ends = [], [], {}, {}
    for p in range(3):
        blob, (part, klen, vlen, ts, keys), _ = random_record_set(rng, 30, partition=p, key_space=40, with_noise=False)
        path = tmp_path / ("0000000000000000000%d.log" % p)
        path.write_bytes(blob)
        files.append(str(path))
        records += [(p, ts[i], keys[i], None if vlen[i] < 0 else vlen[i]) for i in range(len(part))]
        base = int.from_bytes(blob[0:8], "big")
        starts[p], ends[p] = base, base + len(part)
    now = (4102444800, 0)
    mm, lc = OP.run(records, now, with_c)
    want = OP.report("seg", 0, mm, lc, [0, 1, 2], starts, ends)
    args = ["-t", "seg", "-b", "segment://" + ",".join(files)] + (["-c"] if with_c else [])
    r = run_cli(*args)
    assert r.returncode == 0, r.stderr
    got = r.stdout.split("Starting message consumption...\n", 1)[1]
    norm = lambda t: re.sub(r"Estimated Msg/s: \d+", "Estimated Msg/s: X", re.sub(r"Scanning took: \d+ seconds", "Scanning took: 0 seconds", t))
    # earliest message: the CLI's Utc::now() sentinel never wins here (all timestamps are in 2020)
    assert norm(got) == norm(want)
    if with_c:
        return
    # check.crcs=true forwarded like a librdkafka option: a flipped value byte makes that batch undeliverable
    # (warned about, not counted); without the option the same file is analysed as if nothing happened
    blob = bytearray(open(files[1], "rb").read())
    total0 = 12 + int.from_bytes(blob[8:12], "big")
    n0 = int.from_bytes(blob[57:61], "big")
    blob[total0 - 1] ^= 0x01                      # last byte of the first batch (inside a value)
    open(files[1], "wb").write(bytes(blob))
    r1 = run_cli(*args, "--librdkafka", "check.crcs=true")
    assert r1.returncode == 0 and "CRC failure" in r1.stderr and "%d record(s)" % n0 in r1.stderr
    r2 = run_cli(*args)
    assert r2.returncode == 0 and "CRC" not in r2.stderr
    tot = lambda out: sum(int(l.split("|")[4]) for l in out.split("\n") if re.match(r"^\| \d", l))
    assert tot(r2.stdout) == len(records) and tot(r1.stdout) == len(records) - n0


@pytest.fixture(scope="module")
def mock_rccl(tmp_path_factory):
    lib = tmp_path_factory.mktemp("mock") / "libmock_rccl.so"
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O1", "-shared", "-fPIC", "-std=c++17", os.path.join(ROOT, "tests", "mock_rccl.cpp"),
                        "-o", str(lib), "-lrt", "-lpthread"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return str(lib)


@pytest.mark.gpu
def test_cli_sharded_over_eight_ranks_prints_the_single_gpu_report(mock_rccl):
    """kta.gpus=8 on config 4's sharding (256 partitions, 32 per rank): the eight communicator ranks of the target
    machine, here as eight threads on the one reachable GPU with tests/mock_rccl.cpp.  Without -c (eight 32 GiB
    tables do not fit one GPU); with -c on four ranks."""
    env = dict(os.environ, KTA_RCCL_LIBRARY=mock_rccl)
    one = run_cli("-t", "c4", "-b", "synthetic://c4?records=400000")
    many = subprocess.run([CLI, "-t", "c4", "-b", "synthetic://c4?records=400000", "--librdkafka",
                           "kta.gpus=8,kta.batch=16384,kta.oversubscribe=1"], capture_output=True, text=True, timeout=300, env=env)
    assert one.returncode == 0 and many.returncode == 0, one.stderr + many.stderr
    assert _normalise(many.stdout) == _normalise(one.stdout)
    one = run_cli("-t", "c3", "-b", "synthetic://c3?records=300000", "-c")
    many = subprocess.run([CLI, "-t", "c3", "-b", "synthetic://c3?records=300000", "-c", "--librdkafka",
                           "kta.gpus=4,kta.batch=16384,kta.oversubscribe=1"], capture_output=True, text=True, timeout=300, env=env)
    assert one.returncode == 0 and many.returncode == 0, one.stderr + many.stderr
    assert "Alive keys: " in one.stdout and _normalise(many.stdout) == _normalise(one.stdout)


@pytest.mark.gpu
@pytest.mark.parametrize("gpus", [2, 3])
def test_cli_sharded_over_several_ranks_prints_the_single_gpu_report(tmp_path, mock_rccl, gpus):
    """kta.gpus=N: partition p on rank p % N (one thread + context + communicator rank each; here all on the one
    reachable GPU, RCCL replaced by tests/mock_rccl.cpp), global sequence numbers, one kta_exchange — the
    report, alive keys included, is the single-GPU run's byte for byte (synthetic:// and segment://)."""
    env = dict(os.environ, KTA_RCCL_LIBRARY=mock_rccl)
    one = run_cli("-t", "c2", "-b", "synthetic://c2?records=250000", "-c")
    many = subprocess.run([CLI, "-t", "c2", "-b", "synthetic://c2?records=250000", "-c", "--librdkafka",
                           "kta.gpus=%d,kta.batch=32768,kta.oversubscribe=1" % gpus], capture_output=True, text=True, timeout=300, env=env)
    assert one.returncode == 0 and many.returncode == 0, one.stderr + many.stderr
    assert "Alive keys: " in one.stdout and _normalise(many.stdout) == _normalise(one.stdout)
    from kafka_cases import random_record_set
    rng = np.random.default_rng(17)
    files = []
    for p in range(5):
        blob, _, _ = random_record_set(rng, 25, partition=p, key_space=30, with_noise=False, snappy=(p % 2 == 0))
        path = tmp_path / ("%020d.log" % p)
        path.write_bytes(blob)
        files.append(str(path))
    src = "segment://" + ",".join(files)
    one = run_cli("-t", "seg", "-b", src, "-c")
    many = subprocess.run([CLI, "-t", "seg", "-b", src, "-c", "--librdkafka", "kta.gpus=%d,kta.oversubscribe=1" % gpus],
                          capture_output=True, text=True, timeout=300, env=env)
    assert one.returncode == 0 and many.returncode == 0, one.stderr + many.stderr
    assert _normalise(many.stdout) == _normalise(one.stdout) and "Alive keys: " in many.stdout
    r = subprocess.run([CLI, "-t", "x", "-b", "dump:///nonexistent", "--librdkafka", "kta.gpus=2"], capture_output=True, text=True,
                       timeout=60, env=env)
    assert r.returncode != 0
    # more ranks than devices is refused before any thread starts (one rank per GPU) ...
    r = subprocess.run([CLI, "-t", "c2", "-b", "synthetic://c2?records=1000", "--librdkafka", "kta.gpus=999"],
                       capture_output=True, text=True, timeout=60, env=env)
    assert r.returncode == 2 and "HIP device(s) visible" in r.stderr
