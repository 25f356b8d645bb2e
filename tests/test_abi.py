"""CPU tests of the drop-in boundary: libkta_hip.so loads and exports every symbol the headers
declare, host-side helpers (vector decode/merge, synthetic generator) behave, and — without a
GPU — creating a context fails loudly instead of falling back to a CPU path."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import kafka_topic_analyzer_amd as kta
from kafka_topic_analyzer_amd import _native as N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    names = set()
    for h in ("kta_hip.h", "kta_synth.h", "kta_kafka.h"):
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        # prototypes only (skip the static inline generator definitions)
        for m in re.finditer(r"^(?:int|void|uint32_t|int64_t|const char \*)\s*\*?\s*(kta_\w+)\s*\(", text, flags=re.M):
            names.add(m.group(1))
    return names


def test_library_exports_every_declared_symbol():
    lib = N.load()
    declared = _declared_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), f"libkta_hip.so does not export {name}"
    assert declared == set(N.SIGNATURES.keys())
    assert lib.kta_abi_version() == N.KTA_ABI_VERSION == int(re.search(r"#define KTA_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "kta_hip.h")).read()).group(1))


def test_library_is_gfx950_code_object():
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", N.LIB_PATH],
                         capture_output=True, text=True)
    if out.returncode == 0 and out.stdout.strip():
        assert "gfx950" in out.stdout


def test_product_does_not_reference_oracle():
    """The product path must not import, link or call anything under oracle/."""
    pkg = os.path.join(ROOT, "kafka_topic_analyzer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".hpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                if f == "build.py":
                    continue  # builds the checker (allowed), does not use it
                assert "oracle_py" not in text and "kta_oracle" not in text and "libkta_oracle" not in text, f
    ldd = subprocess.run(["ldd", N.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in ldd


def _no_gpu():
    try:
        h = kta.HipMetricHandler(1)
        h.close()
        return False
    except kta.KtaError:
        return True


def test_create_fails_loudly_without_gpu():
    if not _no_gpu():
        pytest.skip("a GPU is present")
    with pytest.raises(kta.KtaError) as e:
        kta.HipMetricHandler(4)
    assert e.value.code in (N.KTA_ERR_NO_DEVICE, N.KTA_ERR_HIP)
    assert "fallback" in str(e.value) or "device" in str(e.value).lower()


def test_create_rejects_bad_arguments():
    lib = N.load()
    ctx = C.c_void_p()
    assert lib.kta_create(None, C.byref(ctx)) == N.KTA_ERR_INVALID
    cfg = N.KtaConfig(0, 0, 0, 0, 0, 0)
    assert lib.kta_create(C.byref(cfg), C.byref(ctx)) == N.KTA_ERR_INVALID
    assert b"n_partitions" in lib.kta_last_error(None)


I64_MAX, I64_MIN = np.iinfo(np.int64).max, np.iinfo(np.int64).min


def _u(x):
    return np.int64(x).astype(np.uint64)


def _vec(P):
    v = np.zeros(P * 7 + 8, dtype=np.uint64)
    g = v[P * 7:]
    g[N.KTA_G_NOT_MIN_TS_MS] = _u(~I64_MAX)
    g[N.KTA_G_MAX_TS_MS] = _u(I64_MIN)
    g[N.KTA_G_NOT_SMALLEST] = _u(~I64_MAX)
    return v


def test_decode_vector_empty_and_truncating_division():
    lib = N.load()
    P = 3
    v = _vec(P)
    res = N.KtaResult()
    c = np.zeros((P, 7), np.uint64)
    assert lib.kta_decode_vector(v.ctypes.data, P, 0, C.byref(res), c.ctypes.data) == N.KTA_OK
    assert res.any_records == 0 and res.smallest_message == 0xFFFFFFFFFFFFFFFF and res.largest_message == 0
    # one live record in partition 1: ts -1500 ms -> -1 s (toward zero), 1999 ms -> 1 s
    v[1 * 7 + N.KTA_C_TOTAL] = 1
    v[1 * 7 + N.KTA_C_ALIVE] = 1
    v[1 * 7 + N.KTA_C_KEY_NON_NULL] = 1
    v[1 * 7 + N.KTA_C_KEY_SIZE_SUM] = 3
    v[1 * 7 + N.KTA_C_VALUE_SIZE_SUM] = 7
    g = v[P * 7:]
    g[N.KTA_G_NOT_MIN_TS_MS] = _u(~np.int64(-1500))
    g[N.KTA_G_MAX_TS_MS] = np.uint64(1999)
    g[N.KTA_G_NOT_SMALLEST] = _u(~np.int64(10))
    g[N.KTA_G_LARGEST] = 10
    assert lib.kta_decode_vector(v.ctypes.data, P, 0, C.byref(res), c.ctypes.data) == N.KTA_OK
    assert (res.min_ts_sec, res.max_ts_sec) == (-1, 1)
    assert (res.overall_count, res.overall_size, res.smallest_message, res.largest_message) == (1, 10, 10, 10)
    g[N.KTA_G_BAD_PARTITION] = 2
    assert lib.kta_decode_vector(v.ctypes.data, P, 0, C.byref(res), None) == N.KTA_ERR_BAD_PARTITION
    assert res.bad_partition_records == 2
    # where the reference panics in NaiveDateTime::from_timestamp (metric.rs:210, kafka.rs:104; chrono 0.4.19's
    # years [-262144, 262143]): the bounds pass — with up to 999 ms beyond them, the division truncates — one
    # second more does not, and that verdict comes before the bad-partition one
    lo, hi = N.KTA_CHRONO_MIN_SEC, N.KTA_CHRONO_MAX_SEC
    for min_ms, max_ms, want in ((lo * 1000 - 999, hi * 1000 + 999, N.KTA_ERR_BAD_PARTITION),
                                 ((lo - 1) * 1000, 1999, N.KTA_ERR_TIMESTAMP_RANGE),
                                 (-1500, (hi + 1) * 1000, N.KTA_ERR_TIMESTAMP_RANGE),
                                 (-2**63, 2**63 - 1, N.KTA_ERR_TIMESTAMP_RANGE)):
        g[N.KTA_G_NOT_MIN_TS_MS] = _u(~np.int64(min_ms))
        g[N.KTA_G_MAX_TS_MS] = _u(np.int64(max_ms))
        assert lib.kta_decode_vector(v.ctypes.data, P, 0, C.byref(res), None) == want
        assert res.overall_count == 1 and res.bad_partition_records == 2          # filled in either way
    g[N.KTA_G_BAD_PARTITION] = 0
    out_len = C.c_size_t()
    buf = C.create_string_buffer(1 << 14)
    assert lib.kta_render_report(b"t", 1, v.ctypes.data, P, 0, 4102444800, 0, None, None, buf, len(buf),
                                 C.byref(out_len)) == N.KTA_ERR_TIMESTAMP_RANGE   # no report: the reference died before


def test_merge_vectors_operators():
    lib = N.load()
    P = 2
    a, b = _vec(P), _vec(P)
    a[:P * 7] = np.arange(P * 7, dtype=np.uint64)
    b[:P * 7] = 100
    ga, gb = a[P * 7:], b[P * 7:]
    ga[N.KTA_G_NOT_MIN_TS_MS], gb[N.KTA_G_NOT_MIN_TS_MS] = _u(~np.int64(5000)), _u(~np.int64(-7000))
    ga[N.KTA_G_MAX_TS_MS], gb[N.KTA_G_MAX_TS_MS] = np.uint64(9000), np.uint64(8000)
    ga[N.KTA_G_NOT_SMALLEST] = _u(~np.int64(12))  # b: none seen
    ga[N.KTA_G_LARGEST], gb[N.KTA_G_LARGEST] = 40, 77
    ga[N.KTA_G_ALIVE_KEYS], gb[N.KTA_G_ALIVE_KEYS] = 3, 4
    assert lib.kta_merge_vectors(a.ctypes.data, b.ctypes.data, P) == N.KTA_OK
    assert list(a[:P * 7]) == [i + 100 for i in range(P * 7)]
    g = a[P * 7:]
    assert ~np.int64(g[N.KTA_G_NOT_MIN_TS_MS]) == -7000 and g[N.KTA_G_MAX_TS_MS] == 9000
    assert ~np.int64(g[N.KTA_G_NOT_SMALLEST]) == 12 and g[N.KTA_G_LARGEST] == 77 and g[N.KTA_G_ALIVE_KEYS] == 7
    # the SUM prefix / MAX suffix split the collectives rely on
    assert N.KTA_NSUM_GLOBALS == 4 and N.KTA_G_NOT_MIN_TS_MS == 4


def test_synth_presets_and_host_generator():
    for name, P in (("c1", 1), ("c2", 8), ("c3", 64), ("c4", 256), ("c5", 256)):
        sp, n = kta.synth_preset(name)
        assert sp.n_partitions == P and n >= 10**6
    with pytest.raises(kta.KtaError):
        kta.synth_preset("nope")
    sp, _ = kta.synth_preset("c2")
    a = kta.synth_fill_host(sp, 1000, 5000, with_keys=True)
    b = kta.synth_fill_host(sp, 1000, 5000, with_keys=True)
    for k in ("partition", "key_len", "val_len", "ts_ms", "key_off", "key_bytes"):
        assert np.array_equal(a[k], b[k])
    # record i is a pure function of (spec, i): a shifted window agrees on the overlap
    c = kta.synth_fill_host(sp, 3000, 1000)
    assert np.array_equal(a["partition"][2000:3000], c["partition"])
    assert np.array_equal(a["val_len"][2000:3000], c["val_len"])
    assert np.array_equal(a["ts_ms"][2000:3000], c["ts_ms"])
    assert set(np.unique(a["partition"])) <= set(range(8))
    kl = a["key_len"]
    assert (kl == -1).any() and set(np.unique(kl)) <= {-1, 0, 8, 16, 36, 64, 200}
    assert (a["val_len"] == -1).any() and (a["ts_ms"] == -1).sum() >= 0
    # keys are packed in record order
    off = np.zeros(len(kl), np.int64)
    off[1:] = np.cumsum(np.maximum(kl, 0))[:-1]
    assert np.array_equal(off, a["key_off"].astype(np.int64))
    assert a["n_key_bytes"] == int(np.maximum(kl, 0).sum())


def test_synth_mean_record_size_is_about_256_bytes():
    sp, _ = kta.synth_preset("c4")
    a = kta.synth_fill_host(sp, 0, 200000)
    mean = (np.maximum(a["key_len"], 0) + np.maximum(a["val_len"], 0)).mean()
    assert 230 < mean < 285, mean


def test_synth_sharding_partitions_are_disjoint_and_cover():
    sp, _ = kta.synth_preset("c4")
    seen = set()
    for r in range(8):
        sp.shard_index, sp.shard_count = r, 8
        a = kta.synth_fill_host(sp, 0, 20000)
        parts = set(np.unique(a["partition"]).tolist())
        assert all(p % 8 == r for p in parts)
        assert not (parts & seen)
        seen |= parts
    assert seen == set(range(256))


def test_missing_rccl_is_an_error_not_a_crash(tmp_path):
    """kta_comm_unique_id with no loadable RCCL: KTA_ERR_COMM (the loader once built its message from two
    dlerror() calls, the second of which returns NULL).  Own process: the binding is made once per process."""
    code = ("import os, ctypes as C\n"
            "from kafka_topic_analyzer_amd import _native as N\n"
            "lib = N.load()\n"
            "buf = C.create_string_buffer(128)\n"
            "rc = lib.kta_comm_unique_id(buf)\n"
            "print('rc', rc, N.KTA_ERR_COMM)\n"
            "assert rc == N.KTA_ERR_COMM\n")
    env = dict(os.environ, KTA_RCCL_LIBRARY=str(tmp_path / "no_such_librccl.so"), KTA_RCCL_ONLY_ENV="1",
               PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
