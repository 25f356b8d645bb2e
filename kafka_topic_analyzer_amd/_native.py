"""ctypes binding of libkta_hip.so (include/kta_hip.h, include/kta_synth.h).

The library is built in-tree by kafka_topic_analyzer_amd/build.py.  There is deliberately no
fallback: if the shared object is missing, loading raises, and if no gfx950 device is visible
`kta_create` fails with KTA_ERR_NO_DEVICE.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KTA_LIB_PATH") or os.path.join(HERE, "libkta_hip.so")   # (KTA_LIB_PATH: another build of the library, tools/build_variant.sh)

KTA_OK = 0
KTA_ERR_INVALID = -1
KTA_ERR_HIP = -2
KTA_ERR_NOMEM = -3
KTA_ERR_NO_DEVICE = -4
KTA_ERR_BAD_PARTITION = -5
KTA_ERR_CAPACITY = -6
KTA_ERR_DIV_BY_ZERO = -7
KTA_ERR_COMM = -8
KTA_ERR_TIMESTAMP_RANGE = -9
KTA_CHRONO_MIN_SEC = -8334632851200
KTA_CHRONO_MAX_SEC = 8210298412799
KTA_COMM_ID_BYTES = 128

KTA_NCOUNTERS = 7
KTA_NGLOBALS = 8
(KTA_C_TOTAL, KTA_C_TOMBSTONES, KTA_C_ALIVE, KTA_C_KEY_NULL, KTA_C_KEY_NON_NULL,
 KTA_C_KEY_SIZE_SUM, KTA_C_VALUE_SIZE_SUM) = range(7)
(KTA_G_BAD_PARTITION, KTA_G_ALIVE_KEYS, KTA_G_RECORDS, KTA_G_RESERVED, KTA_G_NOT_MIN_TS_MS,
 KTA_G_MAX_TS_MS, KTA_G_NOT_SMALLEST, KTA_G_LARGEST) = range(8)
KTA_NSUM_GLOBALS = 4

KTA_PART_RANDOM, KTA_PART_KEY_AFFINE, KTA_PART_RUNS = 0, 1, 2
KTA_VAL_FIXED, KTA_VAL_EXP = 0, 1


class KtaConfig(C.Structure):
    _fields_ = [("device_id", C.c_int32), ("n_partitions", C.c_int32),
                ("count_alive_keys", C.c_int32), ("n_staging", C.c_int32),
                ("batch_capacity", C.c_uint64), ("key_bytes_capacity", C.c_uint64),
                ("flags", C.c_uint32), ("reserved", C.c_uint32)]


KTA_FLAG_ANALYTICS = 1
KTA_FLAG_SEQ_COLUMN = 2
KTA_FLAG_ALIVE_TABLE = 4
KTA_HIST_BUCKETS = 34


class KtaAnalytics(C.Structure):
    _fields_ = [("key_size_hist", C.c_uint64 * 34), ("value_size_hist", C.c_uint64 * 34)]


class KtaBatch(C.Structure):
    _fields_ = [("partition", C.c_void_p), ("key_len", C.c_void_p), ("val_len", C.c_void_p),
                ("ts_ms", C.c_void_p), ("key_off", C.c_void_p), ("key_bytes", C.c_void_p),
                ("seq", C.c_void_p), ("capacity", C.c_uint64), ("key_bytes_capacity", C.c_uint64)]


class KtaResult(C.Structure):
    _fields_ = [("n_partitions", C.c_uint32), ("any_records", C.c_uint32), ("any_live", C.c_uint32),
                ("count_alive_keys", C.c_uint32), ("min_ts_sec", C.c_int64), ("max_ts_sec", C.c_int64),
                ("smallest_message", C.c_uint64), ("largest_message", C.c_uint64),
                ("overall_count", C.c_uint64), ("overall_size", C.c_uint64), ("alive_keys", C.c_uint64),
                ("bad_partition_records", C.c_uint64)]


class KtaSynthSpec(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("n_partitions", C.c_uint32), ("shard_index", C.c_uint32),
                ("shard_count", C.c_uint32), ("part_mode", C.c_uint32), ("part_run_len", C.c_uint32),
                ("key_null_permille", C.c_uint32), ("key_empty_permille", C.c_uint32),
                ("n_key_lens", C.c_uint32), ("key_lens", C.c_uint32 * 8), ("n_distinct_keys", C.c_uint64),
                ("tombstone_permille", C.c_uint32), ("val_empty_permille", C.c_uint32),
                ("val_mode", C.c_uint32), ("val_mean", C.c_uint32), ("val_cap", C.c_uint32),
                ("ts_missing_permille", C.c_uint32), ("ts_base_ms", C.c_int64), ("ts_step_us", C.c_uint32),
                ("ts_jitter_ms", C.c_uint32)]


class KtaKafkaBatchDesc(C.Structure):
    _fields_ = [("byte_off", C.c_uint64), ("record_base", C.c_uint64), ("crc", C.c_uint32), ("status", C.c_uint32),
                ("payload_off", C.c_uint64), ("payload_end", C.c_uint64),
                ("base_offset", C.c_int64), ("base_ts_ms", C.c_int64), ("max_ts_ms", C.c_int64),
                ("batch_bytes", C.c_uint32), ("partition", C.c_int32), ("n_records", C.c_int32),
                ("flags", C.c_uint32), ("scratch_end", C.c_uint64)]


class KtaKafkaIndexStats(C.Structure):
    _fields_ = [("n_batches", C.c_uint64), ("n_records", C.c_uint64), ("n_control_batches", C.c_uint64),
                ("n_compressed", C.c_uint64), ("n_snappy", C.c_uint64), ("n_lz4", C.c_uint64),
                ("inflate_bytes", C.c_uint64),
                ("n_old_magic", C.c_uint64), ("trailing_bytes", C.c_uint64),
                ("bytes_consumed", C.c_uint64), ("n_gzip", C.c_uint64), ("n_zstd", C.c_uint64),
                ("first_offset", C.c_int64), ("next_offset", C.c_int64), ("any_offsets", C.c_uint64)]


# every symbol include/kta_hip.h, kta_synth.h and kta_kafka.h declare: (restype, argtypes)
_P = C.c_void_p
SIGNATURES = {
    "kta_abi_version": (C.c_int, []),
    "kta_create": (C.c_int, [C.POINTER(KtaConfig), C.POINTER(_P)]),
    "kta_destroy": (None, [_P]),
    "kta_last_error": (C.c_char_p, [_P]),
    "kta_reset": (C.c_int, [_P]),
    "kta_handle_message": (C.c_int, [_P, C.c_int32, C.c_int64, C.c_void_p, C.c_int64, C.c_int64]),
    "kta_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "kta_flush": (C.c_int, [_P]),
    "kta_replay_messages": (C.c_int, [_P, C.c_void_p, C.c_uint64]),
    "kta_handle_message_stats": (C.c_int, [_P, C.POINTER(C.c_uint64 * 4)]),
    "kta_seek_seq": (C.c_int, [_P, C.c_uint64]),
    "kta_batch_acquire": (C.c_int, [_P, C.POINTER(KtaBatch)]),
    "kta_batch_submit": (C.c_int, [_P, C.c_uint64, C.c_uint64, C.c_uint64]),
    "kta_submit_device": (C.c_int, [_P, C.POINTER(KtaBatch), C.c_uint64, C.c_uint64]),
    "kta_submit_device_ex": (C.c_int, [_P, C.POINTER(KtaBatch), C.c_uint64, C.c_uint64, C.c_int]),
    "kta_device_batch_alloc": (C.c_int, [_P, C.c_uint64, C.c_uint64, C.c_int, C.POINTER(KtaBatch)]),
    "kta_device_batch_free": (C.c_int, [_P, C.POINTER(KtaBatch)]),
    "kta_copy_to_device": (C.c_int, [_P, C.c_void_p, C.c_void_p, C.c_size_t]),
    "kta_copy_to_host": (C.c_int, [_P, C.c_void_p, C.c_void_p, C.c_size_t]),
    "kta_set_compute_stream": (C.c_int, [_P, C.c_void_p]),
    "kta_sync": (C.c_int, [_P]),
    "kta_finish": (C.c_int, [_P, C.POINTER(KtaResult), C.c_void_p]),
    "kta_result_vector": (C.c_int, [_P, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "kta_finish_device": (C.c_int, [_P]),
    "kta_comm_unique_id": (C.c_int, [C.c_char_p]),
    "kta_comm_create": (C.c_int, [_P, C.c_int, C.c_int, C.c_char_p]),
    "kta_comm_destroy": (C.c_int, [_P]),
    "kta_exchange": (C.c_int, [_P]),
    "kta_exchange_result": (C.c_int, [_P, C.POINTER(KtaResult), C.c_void_p]),
    "kta_comm_allreduce_i64": (C.c_int, [_P, C.c_void_p, C.c_size_t, C.c_int]),
    "kta_comm_info": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "kta_decode_vector": (C.c_int, [C.c_void_p, C.c_uint32, C.c_int, C.POINTER(KtaResult), C.c_void_p]),
    "kta_merge_vectors": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32]),
    "kta_get_analytics": (C.c_int, [_P, C.POINTER(KtaAnalytics), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "kta_analytics_vector": (C.c_int, [_P, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "kta_export_alive_bitmap": (C.c_int, [_P, C.c_void_p]),
    "kta_alive_table": (C.c_int, [_P, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "kta_alive_export_entries": (C.c_int, [_P, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
    "kta_alive_import_entries": (C.c_int, [_P, C.c_void_p, C.c_void_p, C.c_uint64]),
    "kta_alive_count_range": (C.c_int, [_P, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]),
    "kta_alive_table_modified": (C.c_int, [_P]),
    "kta_fnv32_device": (C.c_int, [_P, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]),
    "kta_render_report": (C.c_int, [C.c_char_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_int, C.c_int64, C.c_uint32,
                                    C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "kta_set_timing": (C.c_int, [_P, C.c_int]),
    "kta_kernel_time_stats": (C.c_int, [_P, C.POINTER(C.c_float * 3), C.POINTER(C.c_uint64 * 3)]),
    "kta_set_tuning": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int]),
    "kta_set_fuse": (C.c_int, [_P, C.c_int]),
    "kta_alive_pass_info": (C.c_int, [_P, C.POINTER(C.c_uint64 * 6)]),
    "kta_synth_fill_host": (C.c_int, [C.POINTER(KtaSynthSpec), C.c_uint64, C.c_uint64, C.POINTER(KtaBatch),
                                      C.POINTER(C.c_uint64)]),
    "kta_synth_fill_device": (C.c_int, [_P, C.POINTER(KtaSynthSpec), C.c_uint64, C.c_uint64,
                                        C.POINTER(KtaBatch), C.POINTER(C.c_uint64)]),
    "kta_synth_preset": (C.c_int, [C.c_char_p, C.POINTER(KtaSynthSpec), C.POINTER(C.c_uint64)]),
    "kta_kafka_index_host": (C.c_int, [C.c_char_p, C.c_uint64, C.c_int32, C.c_uint64, C.c_uint64, C.c_uint64,
                                       C.POINTER(KtaKafkaBatchDesc), C.c_uint64, C.POINTER(KtaKafkaIndexStats)]),
    "kta_zstd_inflate_host": (C.c_int64, [C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64]),
    "kta_zstd_inflate_host_small": (C.c_int64, [C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64]),
    "kta_gzip_inflate_host": (C.c_int64, [C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64]),
    "kta_gzip_inflate_lane_host": (C.c_int64, [C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64]),
    "kta_lz4_inflate_host": (C.c_int64, [C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64]),
    "kta_snappy_inflate_host": (C.c_int64, [C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64]),
    "kta_kafka_decode_device": (C.c_int, [_P, C.c_void_p, C.c_uint64, C.POINTER(KtaKafkaBatchDesc), C.c_uint64,
                                          C.c_uint64, C.POINTER(KtaBatch), C.POINTER(C.c_uint64),
                                          C.POINTER(C.c_uint64)]),
    "kta_kafka_decode_rounds_host": (C.c_int, [C.c_char_p, C.c_uint64, C.POINTER(KtaKafkaBatchDesc), C.c_uint64,
                                               C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "kta_kafka_configure": (C.c_int, [_P, C.c_uint64, C.c_int]),
    "kta_kafka_blob_acquire": (C.c_int, [_P, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
    "kta_kafka_blob_submit": (C.c_int, [_P, C.c_uint64, C.c_int32, C.POINTER(KtaKafkaIndexStats)]),
    "kta_kafka_consume": (C.c_int, [_P, C.c_char_p, C.c_uint64, C.c_int32, C.POINTER(KtaKafkaIndexStats)]),
    "kta_kafka_encode_synth_host": (C.c_int, [C.POINTER(KtaSynthSpec), C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p,
                                              C.c_uint64, C.POINTER(C.c_uint64)]),
    "kta_kafka_encode_synth_host_ex": (C.c_int, [C.POINTER(KtaSynthSpec), C.c_uint64, C.c_uint64, C.c_uint32, C.c_int,
                                                 C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]),
    "kta_kafka_set_variant": (C.c_int, [_P, C.c_int]),
    "kta_kafka_set_inflate_limit": (C.c_int, [_P, C.c_uint64]),
    "kta_kafka_set_check_crcs": (C.c_int, [_P, C.c_int]),
    "kta_kafka_crc_errors": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "kta_crc32c_host": (C.c_uint32, [C.c_char_p, C.c_uint64]),
    "kta_kafka_time_stats": (C.c_int, [_P, C.POINTER(C.c_float * 2), C.POINTER(C.c_uint64 * 2)]),
}

_lib = None


KTA_ABI_VERSION = 6  # include/kta_hip.h


def load() -> C.CDLL:
    """Load libkta_hip.so; raise loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: run `python -m kafka_topic_analyzer_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.kta_abi_version() != KTA_ABI_VERSION:   # struct layouts below must match the library's
        raise ImportError(f"libkta_hip.so has ABI version {lib.kta_abi_version()}, this package expects "
                          f"{KTA_ABI_VERSION}: rebuild with `python -m kafka_topic_analyzer_amd.build`")
    _lib = lib
    return lib
