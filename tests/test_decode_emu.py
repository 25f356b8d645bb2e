"""kafka_decode_coop<G, W, R> — the kernel's own source (csrc/kta_decode_coop.h) — run on the CPU: tests/native/wave_emu.h
makes the 64 lanes of a workgroup fibers that meet at __syncthreads / __any / __shfl_xor, and between two meeting
points runs them in ascending, descending or shuffled order.  So the window loads, the LDS hand-over from the leader
to the lanes, the round accounting and the tail of a reported batch are executed here as the GPU executes them, and
compared with the oracle (oracle/kta_kafka_oracle.c), the encoder's expectations and kta_kafka_decode_rounds_host.
The GPU tests of tests/test_kafka_decode.py stay the parity gate: code generation and hardware are not emulated.

Memory safety of the kernel's global accesses (the blob's 64 readable bytes behind its end, the output columns):
    KTA_EMU_ASAN=1 ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD="$(g++ -print-file-name=libasan.so) $(g++ -print-file-name=libubsan.so)" \
        python -m pytest tests/test_decode_emu.py
builds the emulated kernel with AddressSanitizer + UBSan (the blob copy is a heap block of exactly that size)."""
import ctypes as C
import functools
import os
import subprocess

import numpy as np
import pytest

from kafka_topic_analyzer_amd import _native as N
import kafka_format as K
from kafka_cases import assert_columns, random_record_set
from oracle_c import kafka_decode
from test_kafka_decode import GEOMETRY_OF_VARIANT, _forged_count_blob, index_host
import test_decode_rounds as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "kafka_topic_analyzer_amd", "csrc")
NATIVE = os.path.join(ROOT, "tests", "native")

# every geometry tests/native/decode_coop_emu.cpp instantiates: the dispatcher's (kta_kafka.hip) and two small ones
GEOMETRIES = sorted(set(GEOMETRY_OF_VARIANT.values()) | {(8, 1024, 16), (8, 256, 8), (4, 64, 4)})
ORDERS = [(0, 0), (1, 0), (2, 7)]            # (lane order between meeting points, seed)


@pytest.fixture(scope="module")
def emu(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("emu") / "libkta_decode_emu.so")
    sanitize = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all"] if os.environ.get("KTA_EMU_ASAN") else []
    r = subprocess.run(["g++", "-O2", "-g", "-std=c++17", "-shared", "-fPIC", "-Wall", "-Wno-unknown-pragmas", *sanitize,
                        "-I", CSRC, "-I", NATIVE, os.path.join(NATIVE, "decode_coop_emu.cpp"), "-o", so],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    lib = C.CDLL(so)
    lib.kta_emu_last_error.restype = C.c_char_p
    lib.kta_emu_decode_coop.restype = C.c_int
    lib.kta_emu_decode_coop.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_uint32, C.c_uint8, C.c_char_p,
                                        C.c_uint64, C.POINTER(N.KtaKafkaBatchDesc), C.c_uint64, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                        C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.kta_emu_selftest_handover.argtypes = [C.c_int, C.c_uint32, C.c_int, C.c_uint32, C.c_void_p]
    return lib


def run_kernel(lib, blob, partition, geometry, order=(0, 0), poison=0xEE, with_keys=True, with_seq=False,
               seq_base=0, descs=None, st=None, want_key_bytes=True, prefetch=False):
    if descs is None:
        rc, descs, st = index_host(blob, partition)
        assert rc == N.KTA_OK
    n = int(st.n_records)
    cols = {"partition": np.full(n, -7, np.int32), "key_len": np.full(n, -7, np.int32),
            "val_len": np.full(n, -7, np.int32), "ts_ms": np.full(n, -7, np.int64), "key_off": np.full(n, 0xFFFFFFF7, np.uint32)}
    seq = np.zeros(n, np.uint64) if with_seq else None
    kb, bad = C.c_uint64(), C.c_uint64()
    lanes, window, per_round = geometry
    rc = lib.kta_emu_decode_coop(lanes, window, per_round, int(prefetch), order[0], order[1], poison, blob, len(blob), descs,
                                 st.n_batches, cols["partition"].ctypes.data, cols["key_len"].ctypes.data,
                                 cols["val_len"].ctypes.data, cols["ts_ms"].ctypes.data,
                                 cols["key_off"].ctypes.data if with_keys else None,
                                 seq.ctypes.data if with_seq else None, seq_base,
                                 C.byref(kb) if want_key_bytes else None, C.byref(bad))
    assert rc == 0, (rc, lib.kta_emu_last_error())
    if with_keys:
        cols["key_bytes"] = np.frombuffer(blob, dtype=np.uint8)
    else:
        del cols["key_off"]
    cols["n_key_bytes"] = kb.value
    if with_seq:
        cols["seq"] = seq
    return cols, bad.value


def test_the_emulator_runs_lanes_in_the_order_asked_and_sees_a_missing_barrier(emu):
    out = np.zeros(3 * 64, np.uint32)
    for order in ORDERS + [(2, 1), (2, 2)]:
        assert emu.kta_emu_selftest_handover(order[0], order[1], 1, 3, out.ctypes.data) == 0
        want = np.repeat(42 + np.arange(3, dtype=np.uint32), 64) + (2016 << 16)
        assert np.array_equal(out, want), order                      # with the barrier: every lane, every order
    seen = set()
    for order in ORDERS + [(2, 1), (2, 2)]:
        assert emu.kta_emu_selftest_handover(order[0], order[1], 0, 1, out.ctypes.data) == 0
        stale = int(((out[:64] & 0xFFFF) == 0).sum())                # lanes that ran before lane 17 wrote
        seen.add(stale)
        assert stale == {0: 17, 1: 46}.get(order[0], stale)
    assert len(seen) >= 3                                            # the shuffled orders differ from both
    assert emu.kta_emu_selftest_divergent() == -2
    assert b"different points" in emu.kta_emu_last_error()


_random_case = R._random_case          # one record set per seed, shared with the host statement's tests


@pytest.mark.parametrize("geometry", GEOMETRIES)
@pytest.mark.parametrize("seed,with_keys,max_records", [(1, True, 40), (2, False, 40), (3, True, 700), (5, True, 300),
                                                        (6, True, 3000)])
def test_kernel_source_matches_encoder_and_oracle(emu, seed, with_keys, max_records, geometry):
    """The cases of test_device_decode_matches_encoder_and_oracle (tests/test_kafka_decode.py), lane orders in turn."""
    blob, expected, want = _random_case(seed, max_records)
    host, _, _, _ = R.rounds_host(blob, 3, geometry, with_keys)
    order = ORDERS[(seed + geometry[1] // 64) % len(ORDERS)]
    cols, bad = run_kernel(emu, blob, 3, geometry, order, with_keys=with_keys, with_seq=(seed % 2 == 1), seq_base=10**12)
    assert bad == 0
    assert_columns(cols, expected)
    for k in ("partition", "key_len", "val_len", "ts_ms"):
        assert np.array_equal(cols[k], want[k]), k
    if with_keys:
        assert np.array_equal(cols["key_off"], host["key_off"])
    assert cols["n_key_bytes"] == int(np.maximum(want["key_len"], 0).sum()) == host["n_key_bytes"]
    if "seq" in cols:
        assert np.array_equal(cols["seq"], 10**12 + np.arange(len(cols["seq"]), dtype=np.uint64))


@functools.lru_cache(maxsize=None)
def _awkward_blobs():
    """The blobs of test_device_rounds_equal_their_host_statement_bit_for_bit (tests/test_kafka_decode.py)."""
    blobs = []
    filler = [(i, b"key-%d" % i, b"x" * (37 * i % 400)) for i in range(40)]
    for recs in R.UNUSUAL:
        raw = b"".join(R.record(r[0], r[1], r[2], r[3], offset_delta=i) for i, r in enumerate(recs))
        blobs.append(K.encode_batch(0, filler, 1000) + K.encode_batch(40, recs, 10**12, raw_records=raw) +
                     K.encode_batch(50, filler, 2000))
    for what in sorted(R.MALFORMED):
        for before in (0, 3, 60):
            good = [(i, b"key-%d" % i, b"x" * (53 * i % 300)) for i in range(before)]
            raw = b"".join(K.encode_record(i, *r) for i, r in enumerate(good)) + R.MALFORMED[what]()
            blobs.append(K.encode_batch(0, filler, 1000) +
                         K.encode_batch(20, good + [(0, b"?", b"?")], 5000, raw_records=raw) +
                         K.encode_batch(100, filler, 2000))
    rng = np.random.default_rng(77)
    clean, _, _ = random_record_set(rng, 8, max_records=120, with_noise=False, big=True)
    _, descs, st = index_host(clean, 1)
    for _ in range(40):
        hurt = bytearray(clean)
        for _ in range(int(rng.integers(1, 4))):
            d = descs[int(rng.integers(0, st.n_batches))]
            at = int(rng.integers(d.payload_off, d.payload_end))
            hurt[at] = int(rng.integers(0, 256))
        blobs.append(bytes(hurt))
    good = K.encode_batch(0, [(0, b"a", b"b"), (1, b"c", None)], 1000)
    blobs.append(good + K.encode_batch(10, [(0, b"a", b"b"), (1, b"c", None)], 1000, count=3) + good)
    blobs.append(_forged_count_blob()[0])
    return blobs


@pytest.mark.parametrize("geometry", GEOMETRIES)
def test_kernel_source_equals_its_host_statement_bit_for_bit(emu, geometry):
    """Unusual encodings, every malformed record, randomly damaged record sets, a short and a forged record count:
    identical columns — including WHICH records of a reported batch are withheld — whatever the lane order and
    whatever lies behind the blob's end."""
    reported = 0
    for n, blob in enumerate(_awkward_blobs()):
        want, _, _, want_bad = R.rounds_host(blob, 1, geometry)
        for order, poison in ((ORDERS[n % 3], 0xEE), (ORDERS[(n + 1) % 3], 0x00 if n % 2 else 0xFF)):
            cols, bad = run_kernel(emu, blob, 1, geometry, order, poison)
            assert bad == want_bad, (n, order)
            for k in ("partition", "key_len", "val_len", "ts_ms", "key_off"):
                assert np.array_equal(cols[k], want[k]), (n, order, k)
            assert cols["n_key_bytes"] == want["n_key_bytes"], (n, order)
        reported += want_bad
    assert reported > 25


def test_kernel_source_honours_batch_status_append_time_and_record_base(emu):
    """What the kernel takes from the descriptor: a failed check (status) withholds the batch, LogAppendTime stamps every
    record with the batch's maximum, record_base places the batch's records — here in reverse order of the batches."""
    rng = np.random.default_rng(9)
    blob, _, _ = random_record_set(rng, 12, max_records=90, with_noise=False)
    rc, descs, st = index_host(blob, 6)
    assert rc == N.KTA_OK and st.n_batches >= 12
    nb, n = int(st.n_batches), int(st.n_records)
    plain, _ = run_kernel(emu, blob, 6, (16, 3072, 16), descs=descs, st=st)
    base, counts = [descs[i].record_base for i in range(nb)], [descs[i].n_records for i in range(nb)]
    at = n
    for i in range(nb):                                   # batch i's records now end where batch i - 1's begin
        at -= counts[i]
        descs[i].record_base = at
    descs[3].status = 1
    descs[5].flags |= 1                       # KTA_KB_LOG_APPEND_TIME (include/kta_kafka.h)
    descs[5].max_ts_ms = 777
    for geometry in ((16, 3072, 16), (8, 1024, 16), (32, 8192, 32), (64, 8192, 256)):
        cols, bad = run_kernel(emu, blob, 6, geometry, ORDERS[2], descs=descs, st=st, want_key_bytes=False)
        assert bad == 1 and cols["n_key_bytes"] == 0
        for i in range(nb):
            mine = slice(descs[i].record_base, descs[i].record_base + counts[i])
            was = slice(base[i], base[i] + counts[i])
            if i == 3:
                for k in ("partition", "key_len", "val_len", "ts_ms"):
                    assert (cols[k][mine] == -1).all()
                assert (cols["key_off"][mine] == 0).all()
                continue
            for k in ("partition", "key_len", "val_len", "key_off"):
                assert np.array_equal(cols[k][mine], plain[k][was]), (geometry, i, k)
            if i == 5:
                assert (cols["ts_ms"][mine] == 777).all()
            else:
                assert np.array_equal(cols["ts_ms"][mine], plain["ts_ms"][was])
