"""The reference's consume loop (src/kafka.rs:23-137) over librdkafka's C API, bound at run time
(csrc/host/rdkafka_source.cpp), against a test double of the library that serves the synthetic topic
(tests/mock_rdkafka.cpp).  CPU tests drive the loop with a recording handler; the gpu test runs the
whole CLI against the mock "cluster" and expects the `synthetic://` report, byte for byte."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import kafka_topic_analyzer_amd as kta
from kafka_topic_analyzer_amd import _native as N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "kafka_topic_analyzer_amd", "csrc")
CLI = os.path.join(ROOT, "kafka_topic_analyzer_amd", "kta-analyzer")


@pytest.fixture(scope="module")
def built(tmp_path_factory):
    d = tmp_path_factory.mktemp("rdkafka")
    mock = str(d / "libmockrdkafka.so")
    drv = str(d / "loop_driver")
    inc = ["-I", os.path.join(ROOT, "include"), "-I", os.path.join(CSRC, "host"), "-I", CSRC]
    subprocess.run(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", *inc, os.path.join(ROOT, "tests", "mock_rdkafka.cpp"),
                    "-o", mock], check=True)
    subprocess.run(["g++", "-O1", "-std=c++17", *inc, os.path.join(ROOT, "tests", "native", "loop_driver.cpp"),
                    os.path.join(CSRC, "host", "rdkafka_source.cpp"), "-o", drv, "-ldl"], check=True)
    return {"mock": mock, "driver": drv, "dir": d}


def spec_file(d, preset, n_records, name="spec.bin"):
    spec, _ = kta.synth_preset(preset)
    path = str(d / name)
    with open(path, "wb") as f:
        f.write(bytes(spec))
        f.write(np.uint64(n_records).tobytes())
    return spec, path


def env_for(built, spec_path, log=None, **extra):
    env = dict(os.environ, KTA_RDKAFKA_LIB=built["mock"], MOCK_RDKAFKA_SPEC=spec_path, USER="tester")
    if log:
        env["MOCK_RDKAFKA_LOG"] = log
    env.update({k: str(v) for k, v in extra.items()})
    return env


def parse(out):
    msgs = [tuple(int(x) for x in l.split()[1:]) for l in out.splitlines() if l.startswith("M ")]
    offs = {int(l.split()[1]): (int(l.split()[2]), int(l.split()[3])) for l in out.splitlines() if l.startswith("O ")}
    summ = [tuple(int(x) for x in l.split()[1:]) for l in out.splitlines() if l.startswith("S ")]
    return msgs, offs, summ


def key_digest(key):
    h = 0x811C9DC5
    for b in key:
        h = ((h ^ b) * 0x01000193) & 0xFFFFFFFF
    return h


def expected_messages(spec, n, start=0):
    cols = kta.synth_fill_host(spec, 0, n, with_keys=True)
    per = {}
    out = []
    for i in range(n):
        p, kl, vl, ts = int(cols["partition"][i]), int(cols["key_len"][i]), int(cols["val_len"][i]), int(cols["ts_ms"][i])
        off = start + per.get(p, 0)
        per[p] = per.get(p, 0) + 1
        if kl >= 0:
            o = int(cols["key_off"][i])
            dig = key_digest(bytes(cols["key_bytes"][o:o + kl]))
        else:
            dig = 0
        out.append((p, off, ts, kl, vl, dig))
    return out, per


@pytest.mark.parametrize("preset,n,extra", [("c2", 3000, {}), ("c4", 4000, {"MOCK_RDKAFKA_START": 100}),
                                           ("c3", 2500, {"MOCK_RDKAFKA_ERR_EVERY": 7, "MOCK_RDKAFKA_NULL_EVERY": 5})])
def test_loop_delivers_every_record_to_every_handler_and_stops_at_the_end_offsets(built, preset, n, extra):
    spec, sp = spec_file(built["dir"], preset, n, f"{preset}.bin")
    log = str(built["dir"] / f"{preset}.log")
    if os.path.exists(log):
        os.remove(log)
    r = subprocess.run([built["driver"], "broker-a:9092,broker-b:9092", "topic.x", "fetch.min.bytes=1,kta.device=3,client.id=mine"],
                       env=env_for(built, sp, log, **extra), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    msgs, offs, summ = parse(r.stdout)
    start = int(extra.get("MOCK_RDKAFKA_START", 0))
    want, per = expected_messages(spec, n, start)
    # two handlers registered: each message is printed twice in a row (kafka.rs:107-109, registration order)
    assert msgs[0::2] == want and msgs[1::2] == want
    assert summ == [(n, n, n)]                                  # seq == messages handled; loop ended by itself
    assert offs == {p: (start, start + c) for p, c in per.items()}   # kafka.rs:66-70
    assert r.stdout.startswith("O ") and "Subscribing to topic.x\nStarting message consumption...\n" in r.stdout
    assert r.stderr.endswith("done\n")
    if "MOCK_RDKAFKA_ERR_EVERY" in extra:   # kafka.rs:95-97: warn and go on
        assert r.stderr.count("[WARN] Kafka error: Message consumption error: Local: Broker transport failure") > n // 8
    lines = open(log).read().splitlines()
    conf = [l[len("conf_set "):] for l in lines if l.startswith("conf_set ")]
    assert re.fullmatch(r"group\.id=topic-analyzer--tester-[0-9a-f]{8}-[0-9a-f]{4}-4[0-9a-f]{3}-[89ab][0-9a-f]{3}-[0-9a-f]{12}",
                        conf[0])                                # kafka.rs:27
    assert conf[1:9] == ["bootstrap.servers=broker-a:9092,broker-b:9092", "enable.partition.eof=false",
                         "auto.offset.reset=earliest", "enable.auto.commit=false", "api.version.request=true",
                         "enable.auto.offset.store=false", "client.id=topic-analyzer",
                         "queue.buffering.max.ms=1000"]           # kafka.rs:28-36
    assert sorted(conf[9:]) == ["client.id=mine", "fetch.min.bytes=1"]   # user pairs on top; kta.* stay here
    assert "set_log_level 6" in lines and "poll_set_consumer" in lines
    assert "metadata all_topics=0 timeout_ms=10000" in lines       # kafka.rs:61
    assert f"watermarks p=0 timeout_ms=1000" in lines              # kafka.rs:67
    assert "subscribe topic.x partition=-1" in lines               # kafka.rs:89
    assert lines[-2].startswith(f"consumer_close delivered={n} ") and lines[-1] == "destroy"


def test_loop_failure_modes_panic_like_the_reference(built):
    spec, sp = spec_file(built["dir"], "c2", 100, "f.bin")
    run = lambda *a, **e: subprocess.run([built["driver"], *a], env=env_for(built, sp, **e), capture_output=True,
                                         text=True, timeout=60)
    r = run("b:9092", "absent-topic")
    assert r.returncode == 101 and "panicked at 'Topic not found!', src/kafka.rs:62" in r.stderr
    r = run("b:9092", "t", "mock.reject=1")
    assert r.returncode == 101 and "Consumer creation failed: No such configuration property" in r.stderr
    r = run("b:9092", "t", KTA_RDKAFKA_LIB="/nonexistent/librdkafka.so")
    assert r.returncode == 101 and "Consumer creation failed: librdkafka could not be loaded" in r.stderr
    # kafka.rs:103-105: the progress line's NaiveDateTime::from_timestamp(ts / 1000, 0) panics outside chrono's
    # range, before any handler sees the record — 40 records delivered, the 41st ends the run; the bound passes
    hi = N.KTA_CHRONO_MAX_SEC
    r = run("b:9092", "t", MOCK_RDKAFKA_TS_AT="40:%d" % ((hi + 1) * 1000))
    assert r.returncode == 101 and "panicked at 'invalid or out-of-range datetime'" in r.stderr
    assert "src/kafka.rs:104" in r.stderr and len(parse(r.stdout)[0]) == 2 * 40    # two handlers, 40 records each
    r = run("b:9092", "t", MOCK_RDKAFKA_TS_AT="40:%d" % (hi * 1000 + 999))
    assert r.returncode == 0 and parse(r.stdout)[2] == [(100, 100, 100)]


def _normalise(text):
    text = re.sub(r"Scanning took: \d+ seconds", "Scanning took: 0 seconds", text)
    return re.sub(r"Estimated Msg/s: \d+", "Estimated Msg/s: 1", text)


@pytest.mark.gpu
@pytest.mark.parametrize("with_c", [False, True])
def test_cli_against_a_mock_cluster_prints_the_synthetic_report(built, with_c):
    """--bootstrap-server host:port: reference loop -> HipMetricHandler::handle_message per message ->
    pinned staging -> kernels; same report as the column-fed synthetic:// source, incl. Alive keys."""
    n = 150000
    spec, sp = spec_file(built["dir"], "c2", n, "cli.bin")
    extra = ["-c"] if with_c else []
    a = subprocess.run([CLI, "-t", "c2", "-b", "mock-broker:9092", "--librdkafka", "kta.batch=8192", *extra],
                       env=env_for(built, sp, MOCK_RDKAFKA_ERR_EVERY=1000), capture_output=True, text=True, timeout=300)
    b = subprocess.run([CLI, "-t", "c2", "-b", f"synthetic://c2?records={n}", *extra], capture_output=True, text=True,
                       timeout=300)
    assert a.returncode == 0 and b.returncode == 0, a.stderr + b.stderr
    assert _normalise(a.stdout) == _normalise(b.stdout)
    assert ("Alive keys: " in a.stdout) == with_c
    assert "[WARN] Kafka error" in a.stderr and a.stderr.endswith("done\n")
