"""Shared helpers for the parity tests: random record streams as struct-of-arrays columns."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
NOW = (4102444800, 123456789)


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def records_to_cols(records):
    """[(part, ts|None, key|None, vlen|None)] -> SoA columns (ts None -> -1 == not available)."""
    n = len(records)
    cols = {"partition": np.zeros(n, np.int32), "key_len": np.zeros(n, np.int32),
            "val_len": np.zeros(n, np.int32), "ts_ms": np.zeros(n, np.int64),
            "key_off": np.zeros(n, np.uint32)}
    blob = bytearray()
    for i, (p, ts, k, v) in enumerate(records):
        cols["partition"][i] = p
        cols["ts_ms"][i] = -1 if ts is None else ts
        cols["key_len"][i] = -1 if k is None else len(k)
        cols["val_len"][i] = -1 if v is None else v
        cols["key_off"][i] = len(blob)
        if k:
            blob += k
    cols["key_bytes"] = np.frombuffer(bytes(blob), dtype=np.uint8).copy()
    return cols


def scenario_records(sc):
    return [(p, ts, None if k is None else bytes.fromhex(k), v) for p, ts, k, v in sc["records"]]


def random_cols(rng, n, P, key_space=1000, null_key=0.1, empty_key=0.02, tomb=0.2, empty_val=0.02,
                max_key=40, max_val=5000, ts_missing=0.01, runs=False, big_sizes=False):
    """A seeded random topic with every edge the reference distinguishes."""
    if runs:
        part = np.repeat(rng.integers(0, P, size=n // 97 + 1), 97)[:n].astype(np.int32)
    else:
        part = rng.integers(0, P, size=n).astype(np.int32)
    key_id = rng.integers(0, key_space, size=n)
    klen_of = rng.integers(1, max_key + 1, size=key_space)
    klen_of[rng.random(key_space) < empty_key] = 0
    key_len = klen_of[key_id].astype(np.int32)
    key_len[rng.random(n) < null_key] = -1
    val_len = rng.integers(1, max_val + 1, size=n).astype(np.int32)
    if big_sizes:
        big = rng.random(n) < 0.01
        val_len[big] = rng.integers(1 << 20, (1 << 31) - 1, size=int(big.sum())).astype(np.int32)
    val_len[rng.random(n) < empty_val] = 0
    val_len[rng.random(n) < tomb] = -1
    ts = (1_600_000_000_000 + rng.integers(-10**9, 10**9, size=n)).astype(np.int64)
    ts[rng.random(n) < ts_missing] = -1
    kl = np.maximum(key_len, 0).astype(np.int64)
    off = np.zeros(n, np.int64)
    off[1:] = np.cumsum(kl)[:-1]
    total = int(kl.sum())
    # key bytes are a function of the key id (same id -> same bytes)
    key_seed = rng.integers(0, 256, size=(key_space, max_key), dtype=np.uint8)
    blob = np.zeros(max(total, 1), np.uint8)
    step = 1 << 18  # records per gather (bounds the temporaries)
    for lo in range(0, n, step):
        k = kl[lo:lo + step]
        if not k.any():
            continue
        rec = np.repeat(np.arange(lo, lo + len(k)), k)
        within = np.arange(len(rec)) - np.repeat(off[lo:lo + len(k)] - off[lo], k)
        blob[off[lo]:off[lo] + len(rec)] = key_seed[key_id[rec], within]
    return {"partition": part, "key_len": key_len, "val_len": val_len, "ts_ms": ts,
            "key_off": off.astype(np.uint32), "key_bytes": blob[:total]}


def cols_to_records(cols):
    out = []
    kb = cols["key_bytes"].tobytes() if "key_bytes" in cols else b""
    for i in range(len(cols["partition"])):
        kl = int(cols["key_len"][i])
        k = None if kl < 0 else kb[int(cols["key_off"][i]):int(cols["key_off"][i]) + kl]
        ts = int(cols["ts_ms"][i])
        vl = int(cols["val_len"][i])
        out.append((int(cols["partition"][i]), ts, k, None if vl < 0 else vl))
    return out
