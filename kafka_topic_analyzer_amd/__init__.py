"""kafka_topic_analyzer_amd — MI355X (gfx950) implementation of kafka-topic-analyzer's per-record
metric-accumulation hot path (reference: /root/reference/src/metric.rs, src/fnv32.rs).

This Python layer is a thin ctypes mirror of the C ABI (include/kta_hip.h) used by the tests and
by bench.py.  Names follow the reference:

    HipMetricHandler                 one `MetricHandler` (kafka.rs:18-20) that feeds both reference
                                     handlers' state on the GPU
    MessageMetrics                   accessor view == metric.rs:104-195
    LogCompactionInMemoryMetrics     accessor view == metric.rs:282-284

All arithmetic happens in libkta_hip.so's HIP kernels; nothing here (or anywhere in this package)
computes metrics on the CPU, and importing works without a GPU but creating a handler does not.
"""
from __future__ import annotations

import ctypes as C
import time
from typing import NamedTuple, Optional, Tuple

import numpy as np

from . import _native as N
from ._native import KtaBatch, KtaConfig, KtaResult, KtaSynthSpec  # noqa: F401

__all__ = ["HipMetricHandler", "MessageMetrics", "LogCompactionInMemoryMetrics", "Message", "KtaError",
           "DivideByZeroPanic", "DateTimeRangePanic", "synth_preset", "synth_fill_host", "fnv_reference_kats"]

U64_MAX = 0xFFFFFFFFFFFFFFFF


class KtaError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libkta_hip error {code}: {msg}")
        self.code = code


class DivideByZeroPanic(ZeroDivisionError):
    """Where the reference panics with 'attempt to divide by zero' (metric.rs:135,144,153)."""


class DateTimeRangePanic(KtaError):
    """Where the reference panics with 'invalid or out-of-range datetime' (NaiveDateTime::from_timestamp,
    metric.rs:210 / kafka.rs:104): a record's ts / 1000 outside chrono 0.4.19's range.  The library has counted
    the record; kta_finish reports it with KTA_ERR_TIMESTAMP_RANGE."""


class Message(NamedTuple):
    """What the reference handlers read from a BorrowedMessage (metric.rs:208-209,218,233)."""
    partition: int
    timestamp_ms: Optional[int]   # None == Timestamp::to_millis() None
    key: Optional[bytes]          # None == m.key() None
    payload_len: Optional[int]    # None == m.payload() None; the bytes are never read


def _np_ptr(a: np.ndarray) -> C.c_void_p:
    return C.c_void_p(a.ctypes.data)


class HipMetricHandler:
    """Owns one `kta_ctx`.  `handle_message` == MetricHandler::handle_message (kafka.rs:18-20)."""

    def __init__(self, n_partitions: int, count_alive_keys: bool = False, device: int = 0,
                 batch_capacity: int = 0, key_bytes_capacity: int = 0, n_staging: int = 0,
                 now: Optional[Tuple[int, int]] = None, analytics: bool = False, alive_table: bool = False,
                 seq_column: bool = False):
        """alive_table: keep the alive set as the sequence-numbered table (KTA_FLAG_ALIVE_TABLE: batches / shards in
        any order, needed by a rank of a sharded run) instead of the reference's bit set (submission order);
        seq_column: the staging batches carry every record's global sequence number (KTA_FLAG_SEQ_COLUMN)."""
        self._lib = N.load()
        self._ctx = C.c_void_p()
        self.n_partitions = int(n_partitions)
        self.count_alive_keys = bool(count_alive_keys)
        cfg = KtaConfig(device, n_partitions, 1 if count_alive_keys else 0, n_staging, batch_capacity,
                        key_bytes_capacity, (N.KTA_FLAG_ANALYTICS if analytics else 0) |
                        (N.KTA_FLAG_ALIVE_TABLE if alive_table else 0) | (N.KTA_FLAG_SEQ_COLUMN if seq_column else 0), 0)
        rc = self._lib.kta_create(C.byref(cfg), C.byref(self._ctx))
        if rc != N.KTA_OK:
            raise KtaError(rc, self._lib.kta_last_error(None).decode())
        if now is None:  # Utc::now() at MessageMetrics::new (metric.rs:39)
            t = time.time_ns()
            now = (t // 1_000_000_000, t % 1_000_000_000)
        self.now = now
        self._next_seq = 0

    # ------------------------------------------------------------------ plumbing
    def _check(self, rc: int, allow=()):
        if rc != N.KTA_OK and rc not in allow:
            kind = DateTimeRangePanic if rc == N.KTA_ERR_TIMESTAMP_RANGE else KtaError
            raise kind(rc, self._lib.kta_last_error(self._ctx).decode())
        return rc

    def close(self):
        if self._ctx:
            self._lib.kta_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ record entry points
    def handle_message(self, m: Message) -> None:
        ts = -1 if m.timestamp_ms is None else int(m.timestamp_ms)
        if m.key is None:
            kptr, klen = None, -1
        else:
            kbuf = C.create_string_buffer(m.key, max(len(m.key), 1))
            kptr, klen = C.cast(kbuf, C.c_void_p), len(m.key)
        vlen = -1 if m.payload_len is None else int(m.payload_len)
        self._check(self._lib.kta_handle_message(self._ctx, m.partition, ts, kptr, klen, vlen))

    def flush(self) -> None:
        self._check(self._lib.kta_flush(self._ctx))

    def replay_messages(self, cols: dict, n: Optional[int] = None) -> None:
        """kta_handle_message for every record of host (numpy) columns, as a native loop (kta_replay_messages)."""
        b = KtaBatch()
        keep = []
        for name, dt in (("partition", np.int32), ("key_len", np.int32), ("val_len", np.int32), ("ts_ms", np.int64)):
            a = np.ascontiguousarray(cols[name], dtype=dt)
            keep.append(a)
            setattr(b, name, a.ctypes.data)
        if "key_off" in cols and "key_bytes" in cols:
            ko = np.ascontiguousarray(cols["key_off"], dtype=np.uint32)
            kb = np.ascontiguousarray(cols["key_bytes"], dtype=np.uint8)
            keep += [ko, kb]
            b.key_off, b.key_bytes = ko.ctypes.data, kb.ctypes.data
        m = len(keep[0]) if n is None else n
        self._check(self._lib.kta_replay_messages(self._ctx, C.byref(b), m))

    def handle_message_stats(self) -> dict:
        out = (C.c_uint64 * 4)()
        self._check(self._lib.kta_handle_message_stats(self._ctx, C.byref(out)))
        return {"messages": int(out[0]), "batches": int(out[1]), "submit_ns": int(out[2]), "ring_wait_ns": int(out[3])}

    def submit_columns(self, partition, key_len, val_len, ts_ms, key_off=None, key_bytes=None,
                       base_seq: Optional[int] = None) -> None:
        """Feed a struct-of-arrays batch through the pinned staging ring (chunked to its capacity)."""
        partition = np.ascontiguousarray(partition, dtype=np.int32)
        key_len = np.ascontiguousarray(key_len, dtype=np.int32)
        val_len = np.ascontiguousarray(val_len, dtype=np.int32)
        ts_ms = np.ascontiguousarray(ts_ms, dtype=np.int64)
        n = len(partition)
        if self.count_alive_keys:
            key_off = np.ascontiguousarray(key_off, dtype=np.uint32)
            key_bytes = np.ascontiguousarray(key_bytes, dtype=np.uint8)
        seq0 = self._next_seq if base_seq is None else base_seq
        i = 0
        while i < n:
            b = KtaBatch()
            self._check(self._lib.kta_batch_acquire(self._ctx, C.byref(b)))
            m = min(n - i, b.capacity)
            kb = 0
            if self.count_alive_keys:
                # largest prefix whose (packed, monotone) key bytes fit the staging key capacity
                kl = np.maximum(key_len[i:i + m], 0).astype(np.int64)
                ends = np.cumsum(kl)
                fit = int(np.searchsorted(ends, b.key_bytes_capacity, side="right"))
                if fit == 0:
                    raise KtaError(N.KTA_ERR_CAPACITY, "a single key exceeds key_bytes_capacity")
                m = min(m, fit)
                kb = int(ends[m - 1])
                # re-pack this chunk's keys contiguously (input offsets may be arbitrary)
                starts = ends[:m] - kl[:m]
                dst_off = np.frombuffer((C.c_uint32 * m).from_address(b.key_off), dtype=np.uint32)
                dst_off[:] = starts.astype(np.uint32)
                if kb:
                    dst_kb = np.frombuffer((C.c_uint8 * kb).from_address(b.key_bytes), dtype=np.uint8)
                    src_off = key_off[i:i + m].astype(np.int64)
                    if np.array_equal(src_off - src_off[0], starts):  # already packed: one memcpy
                        s0 = int(src_off[0])
                        dst_kb[:] = key_bytes[s0:s0 + kb]
                    else:
                        for j in np.nonzero(kl[:m])[0]:
                            dst_kb[starts[j]:starts[j] + kl[j]] = key_bytes[src_off[j]:src_off[j] + kl[j]]
            C.memmove(b.partition, partition.ctypes.data + 4 * i, 4 * m)
            C.memmove(b.key_len, key_len.ctypes.data + 4 * i, 4 * m)
            C.memmove(b.val_len, val_len.ctypes.data + 4 * i, 4 * m)
            C.memmove(b.ts_ms, ts_ms.ctypes.data + 8 * i, 8 * m)
            self._check(self._lib.kta_batch_submit(self._ctx, m, kb, seq0 + i))
            i += m
        self._next_seq = seq0 + n

    # ------------------------------------------------------------------ device-resident batches
    def device_batch_alloc(self, capacity: int, key_bytes_capacity: int = 0, with_seq: bool = False) -> KtaBatch:
        b = KtaBatch()
        self._check(self._lib.kta_device_batch_alloc(self._ctx, capacity, key_bytes_capacity,
                                                     1 if with_seq else 0, C.byref(b)))
        return b

    def device_batch_free(self, b: KtaBatch) -> None:
        self._check(self._lib.kta_device_batch_free(self._ctx, C.byref(b)))

    def synth_fill_device(self, spec: KtaSynthSpec, first: int, n: int, b: KtaBatch) -> int:
        kb = C.c_uint64(0)
        self._check(self._lib.kta_synth_fill_device(self._ctx, C.byref(spec), first, n, C.byref(b), C.byref(kb)))
        return kb.value

    def submit_device(self, b: KtaBatch, n: int, base_seq: int = 0, which: int = 3) -> None:
        self._check(self._lib.kta_submit_device_ex(self._ctx, C.byref(b), n, base_seq, which))

    def download_batch(self, b: KtaBatch, n: int, n_key_bytes: int = 0):
        """Copy a device batch's columns to numpy arrays (tests)."""
        out = {}
        for name, dt in (("partition", np.int32), ("key_len", np.int32), ("val_len", np.int32), ("ts_ms", np.int64)):
            a = np.empty(n, dtype=dt)
            if n:
                self._check(self._lib.kta_copy_to_host(self._ctx, _np_ptr(a), getattr(b, name), a.nbytes))
            out[name] = a
        if b.key_off:
            a = np.empty(n, dtype=np.uint32)
            if n:
                self._check(self._lib.kta_copy_to_host(self._ctx, _np_ptr(a), b.key_off, a.nbytes))
            out["key_off"] = a
            kbs = np.empty(n_key_bytes, dtype=np.uint8)
            if n_key_bytes:
                self._check(self._lib.kta_copy_to_host(self._ctx, _np_ptr(kbs), b.key_bytes, kbs.nbytes))
            out["key_bytes"] = kbs
        if b.seq:
            a = np.empty(n, dtype=np.uint64)
            if n:
                self._check(self._lib.kta_copy_to_host(self._ctx, _np_ptr(a), b.seq, a.nbytes))
            out["seq"] = a
        return out

    def upload_batch(self, cols: dict, with_keys: bool = False) -> Tuple[KtaBatch, int]:
        """numpy columns -> a new device batch (tests: bypasses the staging ring)."""
        n = len(cols["partition"])
        kbytes = np.ascontiguousarray(cols.get("key_bytes", np.zeros(0, np.uint8)), dtype=np.uint8)
        b = self.device_batch_alloc(max(n, 1), max(len(kbytes), 1) if with_keys else 0, "seq" in cols)
        for name, dt in (("partition", np.int32), ("key_len", np.int32), ("val_len", np.int32), ("ts_ms", np.int64)):
            a = np.ascontiguousarray(cols[name], dtype=dt)
            if n:
                self._check(self._lib.kta_copy_to_device(self._ctx, getattr(b, name), _np_ptr(a), a.nbytes))
        if with_keys:
            a = np.ascontiguousarray(cols["key_off"], dtype=np.uint32)
            if n:
                self._check(self._lib.kta_copy_to_device(self._ctx, b.key_off, _np_ptr(a), a.nbytes))
            if len(kbytes):
                self._check(self._lib.kta_copy_to_device(self._ctx, b.key_bytes, _np_ptr(kbytes), kbytes.nbytes))
        if "seq" in cols:
            a = np.ascontiguousarray(cols["seq"], dtype=np.uint64)
            if n:
                self._check(self._lib.kta_copy_to_device(self._ctx, b.seq, _np_ptr(a), a.nbytes))
        return b, n

    # ------------------------------------------------------------------ results
    def use_stream(self, hip_stream: Optional[int]) -> None:
        """Run on a caller-owned, CREATED HIP stream (e.g. torch.cuda.Stream().cuda_stream; torch's default
        stream is the null stream, handle 0, which means "restore" here); None restores."""
        self._check(self._lib.kta_set_compute_stream(self._ctx, C.c_void_p(hip_stream) if hip_stream else None))

    def sync(self) -> None:
        self._check(self._lib.kta_sync(self._ctx))

    def reset(self) -> None:
        self._check(self._lib.kta_reset(self._ctx))
        self._next_seq = 0

    def finish(self, allow_bad_partition: bool = False):
        """-> (KtaResult, counters[P,7] uint64)."""
        res = KtaResult()
        counters = np.zeros((self.n_partitions, N.KTA_NCOUNTERS), dtype=np.uint64)
        allow = (N.KTA_ERR_BAD_PARTITION,) if allow_bad_partition else ()
        self._check(self._lib.kta_finish(self._ctx, C.byref(res), _np_ptr(counters)), allow)
        return res, counters

    def finish_device(self) -> None:
        self._check(self._lib.kta_finish_device(self._ctx))

    # ------------------------------------------------------------------ multi-GPU exchange (RCCL, native)
    @staticmethod
    def comm_unique_id() -> bytes:
        """On one rank; hand the bytes to the other ranks out of band."""
        buf = C.create_string_buffer(N.KTA_COMM_ID_BYTES)
        rc = N.load().kta_comm_unique_id(buf)
        if rc != N.KTA_OK:
            raise KtaError(rc, "kta_comm_unique_id (RCCL not loadable?)")
        return buf.raw

    def comm_create(self, nranks: int, rank: int, unique_id: Optional[bytes] = None) -> None:
        self._check(self._lib.kta_comm_create(self._ctx, nranks, rank, unique_id))

    def comm_destroy(self) -> None:
        self._check(self._lib.kta_comm_destroy(self._ctx))

    def exchange(self) -> None:
        """The one exchange step of a partition-sharded run (asynchronous on the compute stream but for the
        two count read-backs of a -c run)."""
        self._check(self._lib.kta_exchange(self._ctx))

    def exchange_result(self, allow_bad_partition: bool = False):
        """-> (KtaResult, counters[P,7]) of the snapshot vector: after exchange(), the whole job's."""
        res = KtaResult()
        counters = np.zeros((self.n_partitions, N.KTA_NCOUNTERS), dtype=np.uint64)
        allow = (N.KTA_ERR_BAD_PARTITION,) if allow_bad_partition else ()
        self._check(self._lib.kta_exchange_result(self._ctx, C.byref(res), _np_ptr(counters)), allow)
        return res, counters

    def comm_allreduce_i64(self, values: np.ndarray, op_max: bool = False) -> np.ndarray:
        a = np.ascontiguousarray(values, dtype=np.int64).copy()
        self._check(self._lib.kta_comm_allreduce_i64(self._ctx, _np_ptr(a), a.size, 1 if op_max else 0))
        return a

    def comm_info(self):
        nr, rk, se, re = C.c_int(), C.c_int(), C.c_uint64(), C.c_uint64()
        self._check(self._lib.kta_comm_info(self._ctx, C.byref(nr), C.byref(rk), C.byref(se), C.byref(re)))
        return nr.value, rk.value, se.value, re.value

    def result_vector(self) -> Tuple[int, int]:
        """(device pointer, length in u64) of the counter vector (for collectives)."""
        p, n = C.c_void_p(), C.c_size_t()
        self._check(self._lib.kta_result_vector(self._ctx, C.byref(p), C.byref(n)))
        return p.value, n.value

    def result_vector_host(self) -> np.ndarray:
        p, n = self.result_vector()
        a = np.empty(n, dtype=np.uint64)
        self._check(self._lib.kta_copy_to_host(self._ctx, _np_ptr(a), C.c_void_p(p), a.nbytes))
        return a

    def alive_table(self) -> Tuple[int, int]:
        p, n = C.c_void_p(), C.c_size_t()
        self._check(self._lib.kta_alive_table(self._ctx, C.byref(p), C.byref(n)))
        return p.value, n.value

    def analytics(self) -> dict:
        """Additive analytics (not in the reference): size histograms + per-partition extrema."""
        a = N.KtaAnalytics()
        P = self.n_partitions
        mn, mx = np.zeros(P, np.int64), np.zeros(P, np.int64)
        sm, lg = np.zeros(P, np.uint64), np.zeros(P, np.uint64)
        self._check(self._lib.kta_get_analytics(self._ctx, C.byref(a), _np_ptr(mn), _np_ptr(mx), _np_ptr(sm),
                                                _np_ptr(lg)))
        return {"key_size_hist": np.array(a.key_size_hist[:], dtype=np.uint64),
                "value_size_hist": np.array(a.value_size_hist[:], dtype=np.uint64),
                "part_min_ts_sec": mn, "part_max_ts_sec": mx, "part_smallest": sm, "part_largest": lg}

    def alive_export_entries(self) -> Tuple[int, int, int]:
        """(device ptr slots u32[n], device ptr values u64[n], n): the entries ever written."""
        ps, pv, n = C.c_void_p(), C.c_void_p(), C.c_uint64()
        self._check(self._lib.kta_alive_export_entries(self._ctx, C.byref(ps), C.byref(pv), C.byref(n)))
        return ps.value or 0, pv.value or 0, n.value

    def alive_export_entries_host(self) -> Tuple[np.ndarray, np.ndarray]:
        ps, pv, n = self.alive_export_entries()
        slots, vals = np.empty(n, np.uint32), np.empty(n, np.uint64)
        if n:
            self._check(self._lib.kta_copy_to_host(self._ctx, _np_ptr(slots), C.c_void_p(ps), slots.nbytes))
            self._check(self._lib.kta_copy_to_host(self._ctx, _np_ptr(vals), C.c_void_p(pv), vals.nbytes))
        return slots, vals

    def alive_import_entries(self, d_slots: int, d_vals: int, n: int) -> None:
        self._check(self._lib.kta_alive_import_entries(self._ctx, C.c_void_p(d_slots), C.c_void_p(d_vals), n))

    def alive_count_range(self, slot_lo: int, slot_hi: int) -> int:
        """Alive keys whose hash slot is in [slot_lo, slot_hi) (the owner's share of a hash range)."""
        n = C.c_uint64()
        self._check(self._lib.kta_alive_count_range(self._ctx, slot_lo, slot_hi, C.byref(n)))
        return n.value

    def alive_table_modified(self) -> None:
        self._check(self._lib.kta_alive_table_modified(self._ctx))

    def export_alive_bitmap(self) -> np.ndarray:
        """The alive set as a 2^32-bit bitmap: uint32[2^27], bit h%32 of word h//32."""
        bm = np.empty(1 << 27, dtype=np.uint32)
        self._check(self._lib.kta_export_alive_bitmap(self._ctx, _np_ptr(bm)))
        return bm

    def fnv32(self, keys) -> np.ndarray:
        """Hash a list of byte strings on the device with the reference FNV variant."""
        lens = np.array([len(k) for k in keys], dtype=np.int32)
        offs = np.zeros(len(keys), dtype=np.uint32)
        if len(keys):
            offs[1:] = np.cumsum(lens[:-1])
        blob = np.frombuffer(b"".join(keys), dtype=np.uint8).copy()
        out = np.zeros(len(keys), dtype=np.uint32)
        self._check(self._lib.kta_fnv32_device(self._ctx, _np_ptr(blob) if len(blob) else None, _np_ptr(offs),
                                               _np_ptr(lens), len(keys), len(blob), _np_ptr(out)))
        return out

    def set_timing(self, on: bool) -> None:
        self._check(self._lib.kta_set_timing(self._ctx, 1 if on else 0))

    def kernel_time_stats(self):
        """-> ([avg ms scan, fold, alive], [launch counts]) since the previous call (syncs)."""
        a, c = (C.c_float * 3)(), (C.c_uint64 * 3)()
        self._check(self._lib.kta_kernel_time_stats(self._ctx, C.byref(a), C.byref(c)))
        return list(a), list(c)

    def last_kernel_ms(self):
        return self.kernel_time_stats()[0]

    def set_tuning(self, scan_workgroups=0, scan_variant=16, alive_workgroups=0, alive_variant=3) -> None:
        self._check(self._lib.kta_set_tuning(self._ctx, scan_workgroups, scan_variant, alive_workgroups,
                                             alive_variant))

    # views with the reference's accessor names
    def set_fuse(self, on: bool) -> None:
        """Both handlers of a batch in one pass where possible (default) or always as two passes; same results."""
        self._check(self._lib.kta_set_fuse(self._ctx, 1 if on else 0))

    def alive_pass_info(self) -> dict:
        """Host-side counters of the partitioned alive-key pass since create / reset (kta_alive_pass_info)."""
        out = (C.c_uint64 * 6)()
        self._check(self._lib.kta_alive_pass_info(self._ctx, C.byref(out)))
        return {"slice": int(out[0]), "slices": int(out[1]), "fused": int(out[2]), "scanned": int(out[3]),
                "failed_buckets": int(out[4]), "fuse": bool(out[5])}

    def metrics(self) -> "MessageMetrics":
        res, counters = self.finish()
        return MessageMetrics(res, counters, self.now)

    def log_compaction(self) -> "LogCompactionInMemoryMetrics":
        res, _ = self.finish()
        return LogCompactionInMemoryMetrics(res)


class MessageMetrics:
    """Accessors of metric.rs:104-195 over a finished counter vector."""

    def __init__(self, res: KtaResult, counters: np.ndarray, now: Tuple[int, int]):
        self._c = counters
        self._res = res
        # metric.rs:39-40 sentinels merged with the device extrema exactly as
        # cmp_and_set_message_timestamp would have (metric.rs:65-72)
        self._earliest = now
        self._latest = (0, 0)
        if res.any_records:
            if self._earliest > (res.min_ts_sec, 0):
                self._earliest = (res.min_ts_sec, 0)
            if self._latest < (res.max_ts_sec, 0):
                self._latest = (res.max_ts_sec, 0)

    def _m(self, p: int, c: int) -> int:  # metric.rs:198-203
        return int(self._c[p, c]) if 0 <= p < self._c.shape[0] else 0

    def total(self, p): return self._m(p, N.KTA_C_TOTAL)
    def tombstones(self, p): return self._m(p, N.KTA_C_TOMBSTONES)
    def alive(self, p): return self._m(p, N.KTA_C_ALIVE)
    def key_null(self, p): return self._m(p, N.KTA_C_KEY_NULL)
    def key_non_null(self, p): return self._m(p, N.KTA_C_KEY_NON_NULL)
    def key_size_sum(self, p): return self._m(p, N.KTA_C_KEY_SIZE_SUM)
    def value_size_sum(self, p): return self._m(p, N.KTA_C_VALUE_SIZE_SUM)

    def _avg(self, s: int, p: int) -> int:  # metric.rs:132-157
        if s > 0:
            if self.alive(p) == 0:
                raise DivideByZeroPanic("attempt to divide by zero")
            return s // self.alive(p)
        return 0

    def key_size_avg(self, p): return self._avg(self.key_size_sum(p), p)
    def value_size_avg(self, p): return self._avg(self.value_size_sum(p), p)
    def message_size_avg(self, p): return self._avg(self.key_size_sum(p) + self.value_size_sum(p), p)

    def dirty_ratio(self, p) -> float:  # metric.rs:159-167, f32 with two roundings
        t, tm = self.tombstones(p), self.total(p)
        if tm > 0 and t > 0:
            return float(np.float32(t) / (np.float32(tm) / np.float32(100.0)))
        return 0.0

    def earliest_message(self): return self._earliest
    def latest_message(self): return self._latest

    def smallest_message(self) -> int:  # metric.rs:177-183
        s = int(self._res.smallest_message)
        return 0 if s == U64_MAX else s

    def largest_message(self) -> int: return int(self._res.largest_message)
    def overall_count(self) -> int: return int(self._res.overall_count)
    def overall_size(self) -> int: return int(self._res.overall_size)


class LogCompactionInMemoryMetrics:
    def __init__(self, res: KtaResult):
        self._res = res

    def sum_all_alive(self) -> int:  # metric.rs:282-284
        return int(self._res.alive_keys)


# ---------------------------------------------------------------------- synthetic topic helpers
def synth_preset(name: str) -> Tuple[KtaSynthSpec, int]:
    lib = N.load()
    sp, n = KtaSynthSpec(), C.c_uint64()
    rc = lib.kta_synth_preset(name.encode(), C.byref(sp), C.byref(n))
    if rc != N.KTA_OK:
        raise KtaError(rc, f"unknown preset {name!r}")
    return sp, n.value


def synth_fill_host(spec: KtaSynthSpec, first: int, n: int, with_keys: bool = False, with_seq: bool = False,
                    key_bytes_capacity: Optional[int] = None) -> dict:
    """Generate records [first, first+n) of the synthetic topic on the host (no GPU needed)."""
    lib = N.load()
    cols = {"partition": np.empty(n, np.int32), "key_len": np.empty(n, np.int32),
            "val_len": np.empty(n, np.int32), "ts_ms": np.empty(n, np.int64)}
    b = KtaBatch()
    b.partition, b.key_len = cols["partition"].ctypes.data, cols["key_len"].ctypes.data
    b.val_len, b.ts_ms = cols["val_len"].ctypes.data, cols["ts_ms"].ctypes.data
    b.capacity = n
    if with_seq:
        cols["seq"] = np.empty(n, np.uint64)
        b.seq = cols["seq"].ctypes.data
    kb = C.c_uint64(0)
    if with_keys:
        if key_bytes_capacity is None:  # first pass: count
            rc = lib.kta_synth_fill_host(C.byref(spec), first, n, C.byref(b), C.byref(kb))
            if rc != N.KTA_OK:
                raise KtaError(rc, "kta_synth_fill_host")
            key_bytes_capacity = kb.value
        cols["key_off"] = np.empty(n, np.uint32)
        cols["key_bytes"] = np.empty(max(key_bytes_capacity, 1), np.uint8)
        b.key_off, b.key_bytes = cols["key_off"].ctypes.data, cols["key_bytes"].ctypes.data
        b.key_bytes_capacity = key_bytes_capacity
    rc = lib.kta_synth_fill_host(C.byref(spec), first, n, C.byref(b), C.byref(kb))
    if rc != N.KTA_OK:
        raise KtaError(rc, "kta_synth_fill_host")
    if with_keys:
        cols["key_bytes"] = cols["key_bytes"][:kb.value]
    cols["n_key_bytes"] = kb.value
    return cols


def fnv_reference_kats():
    """Known-answer vectors of the reference FNV variant (fnv32.rs:92-101), SURVEY.md §8c."""
    return [(b"", 0x811C9DC5), (b"a", 0xC9A2E334), (b"b", 0x4CF8BC83), (b"foobar", 0xFFF67B86),
            (b"\x00", 0x6E533999), (b"\xff", 0x53C98FA2), (b"key-0", 0x7ECF789B), (b"key-1", 0xFDB2DAD6),
            (bytes(range(64)), 0x626AD045), (b"k" * 256, 0x7975FBC5)]
