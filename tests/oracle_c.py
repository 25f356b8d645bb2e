"""ctypes binding of the C oracle (oracle/libkta_oracle.so).  Test infrastructure only."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(ROOT, "oracle", "libkta_oracle.so")
        if not os.path.exists(path):
            import subprocess
            subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True)
        L = C.CDLL(path)
        L.kto_fnv32.restype = C.c_uint32
        L.kto_fnv32.argtypes = [C.c_char_p, C.c_size_t]
        L.kto_metrics_new.restype = C.c_void_p
        L.kto_metrics_new.argtypes = [C.c_int64, C.c_uint32]
        L.kto_metrics_free.argtypes = [C.c_void_p]
        L.kto_metrics_handle_message.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_int, C.c_int64, C.c_int64]
        for f in ("total", "tombstones", "alive", "key_null", "key_non_null", "key_size_sum", "value_size_sum"):
            fn = getattr(L, "kto_" + f)
            fn.restype = C.c_uint64
            fn.argtypes = [C.c_void_p, C.c_int32]
        for f in ("key_size_avg", "value_size_avg", "message_size_avg"):
            fn = getattr(L, "kto_" + f)
            fn.restype = C.c_int
            fn.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_uint64)]
        L.kto_dirty_ratio.restype = C.c_float
        L.kto_dirty_ratio.argtypes = [C.c_void_p, C.c_int32]
        for f in ("latest_message", "earliest_message"):
            fn = getattr(L, "kto_" + f)
            fn.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_uint32)]
        for f in ("smallest_message", "largest_message", "overall_count", "overall_size"):
            fn = getattr(L, "kto_" + f)
            fn.restype = C.c_uint64
            fn.argtypes = [C.c_void_p]
        L.kto_metrics_panicked.restype = C.c_int
        L.kto_metrics_panicked.argtypes = [C.c_void_p]
        L.kto_lc_new.restype = C.c_void_p
        L.kto_lc_free.argtypes = [C.c_void_p]
        L.kto_lc_handle_message.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, C.c_int64]
        L.kto_lc_sum_all_alive.restype = C.c_uint64
        L.kto_lc_sum_all_alive.argtypes = [C.c_void_p]
        L.kto_lc_contains.restype = C.c_int
        L.kto_lc_contains.argtypes = [C.c_void_p, C.c_uint32]
        L.kto_lc_nbits.restype = C.c_uint64
        L.kto_lc_nbits.argtypes = [C.c_void_p]
        L.kto_lc_export_words.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        L.kto_run_soa.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64] + [C.c_void_p] * 6
        L.kto_lc_run_soa_slot_range.argtypes = [C.c_void_p, C.c_uint64] + [C.c_void_p] * 5 + [C.c_uint64, C.c_uint64]
        L.kto_fnv1a_soa.argtypes = [C.c_uint64, C.c_uint64] + [C.c_void_p] * 4
        L.kto_export_counters.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        _LIB = L
    return _LIB


def fnv32(b: bytes) -> int:
    return lib().kto_fnv32(b, len(b))


class Oracle:
    """MessageMetrics (+ optional LogCompactionInMemoryMetrics) of the C oracle."""

    def __init__(self, now=(4102444800, 123456789), count_alive_keys=False):
        self.L = lib()
        self.m = self.L.kto_metrics_new(now[0], now[1])
        self.lc = self.L.kto_lc_new() if count_alive_keys else None

    def close(self):
        if self.m:
            self.L.kto_metrics_free(self.m)
            self.m = None
        if self.lc:
            self.L.kto_lc_free(self.lc)
            self.lc = None

    def __del__(self):
        self.close()

    def handle(self, part, ts, key, vlen):
        """ts None == timestamp not available; key None == key None; vlen None == payload None."""
        self.L.kto_metrics_handle_message(self.m, part, 0 if ts is None else ts, 0 if ts is None else 1,
                                          -1 if key is None else len(key), -1 if vlen is None else vlen)
        if self.lc and not self.panicked():     # (the process died in the first handler)
            self.L.kto_lc_handle_message(self.lc, key, -1 if key is None else len(key), -1 if vlen is None else vlen)

    def run_soa(self, cols):
        n = len(cols["partition"])
        a = {k: np.ascontiguousarray(v) for k, v in cols.items() if isinstance(v, np.ndarray)}
        assert a["partition"].dtype == np.int32 and a["ts_ms"].dtype == np.int64
        koff = a["key_off"].ctypes.data if "key_off" in a else None
        kb = a["key_bytes"].ctypes.data if "key_bytes" in a and len(a["key_bytes"]) else None
        if self.lc and kb is None and "key_bytes" in a:
            kb = np.zeros(16, np.uint8).ctypes.data
        self.L.kto_run_soa(self.m, self.lc, n, a["partition"].ctypes.data, a["key_len"].ctypes.data,
                           a["val_len"].ctypes.data, a["ts_ms"].ctypes.data, koff, kb)

    def panicked(self):
        """A record's timestamp was outside chrono's range (metric.rs:210): the reference is gone."""
        return bool(self.L.kto_metrics_panicked(self.m))

    def counters(self, P):
        out = np.zeros((P, 7), dtype=np.uint64)
        self.L.kto_export_counters(self.m, P, out.ctypes.data)
        return out

    def get(self, name, p=None):
        fn = getattr(self.L, "kto_" + name)
        return fn(self.m) if p is None else fn(self.m, p)

    def avg(self, name, p):
        out = C.c_uint64()
        rc = getattr(self.L, "kto_" + name)(self.m, p, C.byref(out))
        return None if rc != 0 else out.value

    def earliest(self):
        s, ns = C.c_int64(), C.c_uint32()
        self.L.kto_earliest_message(self.m, C.byref(s), C.byref(ns))
        return (s.value, ns.value)

    def latest(self):
        s, ns = C.c_int64(), C.c_uint32()
        self.L.kto_latest_message(self.m, C.byref(s), C.byref(ns))
        return (s.value, ns.value)

    def alive_keys(self):
        return self.L.kto_lc_sum_all_alive(self.lc)

    def alive_words(self, n_words=1 << 27):
        out = np.zeros(n_words, dtype=np.uint32)
        self.L.kto_lc_export_words(self.lc, out.ctypes.data, n_words)
        return out


class SlotRangeAliveOracle:
    """The LogCompaction oracle on K threads: K instances over disjoint ranges of the 2^32 slots, each fed the WHOLE topic in
    consumption order and applying only the records whose key hashes into its range (kto_lc_run_soa_slot_range, a harness
    function of the oracle: a record touches one bit, so every slot still sees its own records in order).  Between them the
    instances hold exactly the single instance's BitSet."""

    def __init__(self, k=16):
        from concurrent.futures import ThreadPoolExecutor
        assert k & (k - 1) == 0 and k <= 64
        self.L = lib()
        self.k = k
        self.lcs = [self.L.kto_lc_new() for _ in range(k)]
        self.pool = ThreadPoolExecutor(k)

    def run_soa(self, cols):
        n = len(cols["key_len"])
        a = {f: np.ascontiguousarray(cols[f]) for f in ("key_len", "val_len", "key_off", "key_bytes")}
        step = (1 << 32) // self.k
        slots = np.empty(n, np.uint32)               # every key hashed ONCE (by the oracle's own fnv1a), a share per thread
        per = (n + self.k - 1) // self.k

        def hash_share(j):                           # (ctypes calls release the GIL)
            lo = min(j * per, n)
            self.L.kto_fnv1a_soa(lo, min(per, n - lo), a["key_len"].ctypes.data, a["key_off"].ctypes.data,
                                 a["key_bytes"].ctypes.data, slots.ctypes.data)
        list(self.pool.map(hash_share, range(self.k)))

        def one(j):
            self.L.kto_lc_run_soa_slot_range(self.lcs[j], n, a["key_len"].ctypes.data, a["val_len"].ctypes.data,
                                             a["key_off"].ctypes.data, a["key_bytes"].ctypes.data, slots.ctypes.data,
                                             j * step, (j + 1) * step)
        list(self.pool.map(one, range(self.k)))

    def alive_keys(self):
        return sum(int(self.L.kto_lc_sum_all_alive(lc)) for lc in self.lcs)

    def alive_words(self, n_words=1 << 27):
        out = np.zeros(n_words, dtype=np.uint32)
        per = n_words // self.k
        for j, lc in enumerate(self.lcs):
            tmp = np.zeros((j + 1) * per, dtype=np.uint32)
            self.L.kto_lc_export_words(lc, tmp.ctypes.data, (j + 1) * per)
            assert not tmp[:j * per].any()           # nothing outside its range
            out[j * per:(j + 1) * per] = tmp[j * per:]
        return out

    def close(self):
        for lc in self.lcs:
            self.L.kto_lc_free(lc)
        self.lcs = []
        self.pool.shutdown()


def analytics(cols, P):
    """The oracle's restatement of the additive analytics (no reference counterpart)."""
    L = lib()
    L.kto_analytics_new.restype = C.c_void_p
    L.kto_analytics_new.argtypes = [C.c_int32]
    L.kto_analytics_free.argtypes = [C.c_void_p]
    L.kto_analytics_run_soa.argtypes = [C.c_void_p, C.c_uint64] + [C.c_void_p] * 4
    L.kto_analytics_export.argtypes = [C.c_void_p] + [C.c_void_p] * 6
    a = L.kto_analytics_new(P)
    c = {k: np.ascontiguousarray(v) for k, v in cols.items() if isinstance(v, np.ndarray)}
    L.kto_analytics_run_soa(a, len(c["partition"]), c["partition"].ctypes.data, c["key_len"].ctypes.data,
                            c["val_len"].ctypes.data, c["ts_ms"].ctypes.data)
    out = {"key_size_hist": np.zeros(34, np.uint64), "value_size_hist": np.zeros(34, np.uint64),
           "part_min_ts_sec": np.zeros(P, np.int64), "part_max_ts_sec": np.zeros(P, np.int64),
           "part_smallest": np.zeros(P, np.uint64), "part_largest": np.zeros(P, np.uint64)}
    L.kto_analytics_export(a, *[out[k].ctypes.data for k in ("key_size_hist", "value_size_hist", "part_min_ts_sec",
                                                              "part_max_ts_sec", "part_smallest", "part_largest")])
    L.kto_analytics_free(a)
    return out


class KafkaStats(C.Structure):
    _fields_ = [("control_batches", C.c_uint64), ("compressed_batches", C.c_uint64), ("old_magic_batches", C.c_uint64),
                ("trailing_bytes", C.c_uint64), ("bad_batches", C.c_uint64), ("batches", C.c_uint64)]


def _zstd_batches_to_plain(blob: bytes) -> bytes:
    """zstd batches (codec 4) inflated by libzstd itself (through pyarrow) and re-framed as uncompressed
    batches, so that the C oracle — which has no zstd of its own: no zstd.h in the image — decodes them.
    A batch libzstd refuses becomes a batch whose records overrun it (the oracle reports it as corrupt,
    like the product).  Without pyarrow the blob is returned unchanged (zstd batches count as skipped)."""
    try:
        import pyarrow as pa
    except ImportError:
        return blob
    L = lib()
    L.kto_crc32c.restype = C.c_uint32
    L.kto_crc32c.argtypes = [C.c_char_p, C.c_uint64]
    out, pos = bytearray(), 0
    while pos + 12 <= len(blob):
        length = int.from_bytes(blob[pos + 8:pos + 12], "big", signed=True)
        total = 12 + length
        if length < 49 or pos + total > len(blob):
            break
        b = blob[pos:pos + total]
        attrs = int.from_bytes(b[21:23], "big")
        if b[16] == 2 and not attrs & 0x20 and attrs & 7 == 4:
            try:
                recs = pa.input_stream(pa.BufferReader(b[61:]), compression="zstd").read()
            except Exception:
                recs = b""
            after_crc = (attrs & ~7).to_bytes(2, "big") + b[23:61] + recs
            crc = L.kto_crc32c(after_crc, len(after_crc))
            b = b[0:8] + (49 + len(recs)).to_bytes(4, "big") + b[12:17] + crc.to_bytes(4, "big") + after_crc
        out += b
        pos += total
    return bytes(out) + blob[pos:]


def kafka_decode(blob: bytes, partition: int, transcode_zstd: bool = True):
    """The oracle's sequential decode of one Kafka v2 record set -> (columns dict, stats).
    transcode_zstd=False skips the Python walk that hands zstd batches to libzstd (bench.py times the C
    decoder alone on a blob it knows to be uncompressed)."""
    L = lib()
    if transcode_zstd:
        blob = _zstd_batches_to_plain(blob)
    L.kto_kafka_decode.restype = C.c_int64
    L.kto_kafka_decode.argtypes = [C.c_char_p, C.c_uint64, C.c_int32] + [C.c_void_p] * 7 + [C.POINTER(C.c_uint64),
                                                                                          C.POINTER(KafkaStats)]
    st, kb = KafkaStats(), C.c_uint64()
    n = L.kto_kafka_decode(blob, len(blob), partition, None, None, None, None, None, None, None, C.byref(kb), C.byref(st))
    cols = {"partition": np.zeros(n, np.int32), "key_len": np.zeros(n, np.int32), "val_len": np.zeros(n, np.int32),
            "ts_ms": np.zeros(n, np.int64), "offset": np.zeros(n, np.int64), "key_off": np.zeros(n, np.uint32),
            "key_bytes": np.zeros(max(kb.value, 1), np.uint8)}
    n2 = L.kto_kafka_decode(blob, len(blob), partition, cols["partition"].ctypes.data, cols["key_len"].ctypes.data,
                            cols["val_len"].ctypes.data, cols["ts_ms"].ctypes.data, cols["offset"].ctypes.data,
                            cols["key_off"].ctypes.data, cols["key_bytes"].ctypes.data, C.byref(kb), C.byref(st))
    assert n2 == n
    cols["key_bytes"] = cols["key_bytes"][:kb.value]
    return cols, st
