"""Regenerate the golden fixtures in this directory.

    python tests/golden/make_golden.py

The reference (/root/reference, Rust) cannot be executed in the build image and has no tests or
fixtures of its own, so these vectors come from the pure-Python restatement of the
reference source (oracle/oracle_py.py; every function there cites the lines it follows).  The first
ten FNV vectors were additionally derived by hand in SURVEY.md §8c.  The C oracle and the HIP path
are both tested against these files; **parity with the Rust binary itself remains unpinned**.
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_py as O  # noqa: E402

NOW = (4102444800, 123456789)  # 2100-01-01T00:00:00.123456789Z: a fixed stand-in for Utc::now()


def fnv_vectors():
    keys = [b"", b"a", b"b", b"foobar", b"\x00", b"\xff", b"key-0", b"key-1", bytes(range(64)), b"k" * 256]
    rnd = random.Random(20260921)
    for n in list(range(0, 40)) + [63, 64, 65, 127, 128, 129, 200, 255, 256, 257, 1000]:
        keys.append(bytes(rnd.randrange(256) for _ in range(n)))
    return [{"key_hex": k.hex(), "hash": O.fnv1a(k), "standard_fnv1a": O.fnv1a_standard(k)} for k in keys]


def find_collision():
    """two different short keys with the same reference hash (birthday search)."""
    seen = {}
    i = 0
    while True:
        k = b"c%d" % i
        h = O.fnv1a(k)
        if h in seen:
            return seen[h], k
        seen[h] = k
        i += 1


def scenarios():
    ka, kb = find_collision()
    S = {}
    # (1) null key + value (metric.rs:227-240)
    S["null_key_with_value"] = [[0, 1000, None, 10]]
    # (2) empty key Some(&[]) (metric.rs:219-225, 294-296): hashed, slot 0x811c9dc5
    S["empty_key"] = [[0, 1000, "", 5]]
    # (3) empty value Some(&[]) is alive, contributes len(k)+0 to min/max (metric.rs:234-240, 249-251)
    S["empty_value"] = [[0, 1000, "6b", 0], [0, 2000, "6b6b", 7]]
    # (4) tombstone: no min/max size update but timestamp update (metric.rs:241-251)
    S["tombstone_updates_ts_only"] = [[0, 5000, "6b", 100], [0, 9000, "6b", None]]
    # (5) un-keyed tombstone: ignored by the alive set (metric.rs:302)
    S["unkeyed_tombstone"] = [[1, 1000, None, None], [1, 2000, "61", 3]]
    # (6) A D A => alive; A D => dead; D on absent => no-op (metric.rs:273-280)
    S["alive_dead_alive"] = [[0, 1, "61", 1], [0, 2, "61", None], [0, 3, "61", 1],
                             [0, 4, "62", 1], [0, 5, "62", None], [0, 6, "63", None]]
    # (7) two different keys with equal hash: the later record decides the shared slot
    S["hash_collision_later_wins_dead"] = [[0, 1, ka.hex(), 1], [1, 2, kb.hex(), None]]
    S["hash_collision_later_wins_alive"] = [[0, 1, ka.hex(), None], [1, 2, kb.hex(), 4]]
    # (8) timestamps: n/a -> 0; -1 -> 0; 1999 ms -> 1 s; -1500 ms -> -1 s (truncation toward zero)
    S["ts_not_available"] = [[0, None, "61", 1]]
    S["ts_minus_one"] = [[0, -1, "61", 1], [0, 1999, "61", 1]]
    S["ts_negative_truncation"] = [[0, -1500, "61", 1], [0, -999, "61", 1]]
    S["ts_future_beyond_now"] = [[0, (NOW[0] + 10) * 1000, "61", 1]]
    # (9) no non-tombstone at all => Smallest Message: 0 (metric.rs:177-183)
    S["only_tombstones"] = [[0, 1000, None, None], [0, 2000, None, None]]
    # (10) keyed tombstones only => divide-by-zero panic in key_size_avg (metric.rs:135)
    S["keyed_tombstones_only_panics"] = [[0, 1000, "6b6579", None]]
    # mixed multi-partition stream
    rnd = random.Random(7)
    recs = []
    for i in range(400):
        key = None if rnd.random() < 0.1 else bytes(rnd.randrange(97, 101) for _ in range(rnd.randrange(0, 4)))
        val = None if rnd.random() < 0.3 else rnd.randrange(0, 5000)
        ts = rnd.choice([None, -1, rnd.randrange(-10**7, 2 * 10**12)])
        recs.append([rnd.randrange(0, 5), ts, None if key is None else key.hex(), val])
    S["mixed_400"] = recs
    return S


def run_scenario(recs, n_partitions):
    records = [(p, ts, None if k is None else bytes.fromhex(k), v) for p, ts, k, v in recs]
    mm, lc = O.run(records, NOW, True)
    per = []
    for p in range(n_partitions):
        row = {"counters": [mm.total(p), mm.tombstones(p), mm.alive(p), mm.key_null(p), mm.key_non_null(p),
                            mm.key_size_sum(p), mm.value_size_sum(p)],
               "dirty_ratio_4": O.format_f32_4(mm.dirty_ratio(p))}
        for name in ("key_size_avg", "value_size_avg", "message_size_avg"):
            try:
                row[name] = getattr(mm, name)(p)
            except O.DivideByZeroPanic:
                row[name] = "panic"
        per.append(row)
    return {"partitions": per, "earliest": list(mm.earliest_message), "latest": list(mm.latest_message),
            "earliest_display": O.format_datetime_utc(*mm.earliest_message),
            "latest_display": O.format_datetime_utc(*mm.latest_message),
            "smallest": mm.smallest_message(), "largest": mm.largest_message(),
            "overall_count": mm.overall_count(), "overall_size": mm.overall_size(),
            "alive_keys": lc.sum_all_alive(), "alive_slots": sorted(lc.store)}


def report_golden():
    recs = scenarios()["mixed_400"]
    records = [(p, ts, None if k is None else bytes.fromhex(k), v) for p, ts, k, v in recs]
    out = {}
    for flag in (False, True):
        mm, lc = O.run(records, NOW, flag)
        parts = list(range(5))
        start = {p: 0 for p in parts}
        end = {p: mm.total(p) for p in parts}
        out["with_c" if flag else "without_c"] = O.report("synthetic.mixed_400", 3, mm, lc, parts, start, end)
    return out


def kafka_fixture():
    """A small byte-level Kafka v2 record set with every feature the decoder distinguishes."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import kafka_format as K
    b1 = [(0, b"k", b"v"), (5, None, None), (-3, b"", b""), (70000, b"key-\xff", b"x" * 200, [(b"h", b"1"), (b"n", None)])]
    b2 = [(1, b"second-batch", None)]
    blob = K.encode_batch(100, b1, 1600000000000)
    blob += K.encode_batch(104, [(0, b"ctl", b"ctl")], 1600000000100, attributes=0x30)          # control batch
    blob += K.encode_batch(105, b2, 1600000001000, attributes=0x08, max_ts=1600000009999)       # LogAppendTime
    blob += K.encode_batch(106, [(0, b"z", b"z")], 1600000002000, attributes=0x05)              # unknown codec: skipped
    blob += K.encode_batch(107, [(0, b"old", b"old")], 1600000003000, magic=1)                  # magic 1: skipped
    b3 = [(0, b"snappy-key-a", b"abcabcabcabcabcabcabcabc" * 8), (7, b"snappy-key-a", None), (9, None, b"x" * 300)]
    blob += K.encode_batch(108, b3, 1600000005000, compression="snappy")                        # Snappy, bare block
    blob += K.encode_batch(111, b3, 1600000006000, compression="snappy-xerial")                 # Snappy, xerial framing
    blob += K.encode_batch(114, b3, 1600000007000, compression="lz4")                           # LZ4 frame, linked blocks
    blob += K.encode_batch(117, b3, 1600000008000, compression="lz4-indep")                     # LZ4, all optional fields
    blob += K.encode_batch(120, b3, 1600000009000, compression="gzip")                          # gzip member (zlib, level 6)
    blob += K.encode_batch(123, b3, 1600000010000, compression="zstd")                          # zstd frame (libzstd, level 3)
    blob += K.encode_batch(126, [(0, b"tail", b"tail")], 1600000004000)[:30]                    # partial tail
    keys = [b"k", b"", b"key-\xff", b"second-batch"] + [b"snappy-key-a"] * 12
    expect = {"partition": [9] * 23, "key_len": [1, -1, 0, 5, 12] + [12, 12, -1] * 6,
              "val_len": [1, -1, 0, 200, -1] + [192, -1, 300] * 6,
              "ts_ms": [1600000000000, 1600000000005, 1599999999997, 1600000070000, 1600000009999] +
                       [1600000005000 + 1000 * b + d for b in range(6) for d in (0, 7, 9)],
              "offset": [100, 101, 102, 103, 105] + list(range(108, 126)), "key_bytes_hex": b"".join(keys).hex(),
              "control_batches": 1, "compressed_batches": 1, "old_magic_batches": 1, "trailing_bytes": 30}
    return {"partition": 9, "blob_hex": blob.hex(), "expect": expect}


def main():
    with open(os.path.join(HERE, "kafka_v2_recordset.json"), "w") as f:
        json.dump(kafka_fixture(), f, indent=0)
    with open(os.path.join(HERE, "fnv32_kats.json"), "w") as f:
        json.dump(fnv_vectors(), f, indent=0)
    S = scenarios()
    out = {"now": list(NOW), "n_partitions": 5, "scenarios": {}}
    for name, recs in S.items():
        out["scenarios"][name] = {"records": recs, "expect": run_scenario(recs, 5)}
    with open(os.path.join(HERE, "scenarios.json"), "w") as f:
        json.dump(out, f, indent=0)
    rep = report_golden()
    for k, text in rep.items():
        with open(os.path.join(HERE, f"report_mixed_400_{k}.txt"), "w") as f:
            f.write(text)
    print("wrote", len(fnv_vectors()), "fnv vectors,", len(S), "scenarios")


if __name__ == "__main__":
    main()
