"""The wave decoder of zstd's Huffman-coded literals — the kernel's own source (csrc/kta_zstd_huf_wave.h) — run on the CPU:
tests/native/wave_emu.h makes the 64 lanes fibers that meet at barriers, ballots, shuffles and readlanes.  Literals sections
are cut out of frames libzstd wrote (through pyarrow) and decoded twice: by the format's own statement (kta_zstd.h:
zs_huf_stream, one stream after the other, one symbol after the other) and by the wave — 64 segments per 2 KiB of stream
from speculative starts, confirmed lane by lane.  The GPU tests stay the parity gate (tests/test_kafka_decode.py).

    KTA_EMU_ASAN=1 ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD="$(g++ -print-file-name=libasan.so) $(g++ -print-file-name=libubsan.so)" \
        python -m pytest tests/test_zstd_huf_emu.py"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

pa = pytest.importorskip("pyarrow")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "kafka_topic_analyzer_amd", "csrc")
NATIVE = os.path.join(ROOT, "tests", "native")
ORDERS = [(0, 0), (1, 0), (2, 7)]


@pytest.fixture(scope="module")
def emu(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("emu") / "libkta_zstd_huf_emu.so")
    sanitize = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all"] if os.environ.get("KTA_EMU_ASAN") else []
    r = subprocess.run(["g++", "-O2", "-g", "-std=c++17", "-shared", "-fPIC", "-Wall", "-Wno-unknown-pragmas", *sanitize,
                        "-I", CSRC, "-I", NATIVE, os.path.join(NATIVE, "zstd_huf_emu.cpp"), "-o", so],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    lib = C.CDLL(so)
    lib.kta_emu_zstd_huf.restype = C.c_int64
    lib.kta_emu_zstd_huf.argtypes = [C.c_char_p, C.c_uint64, C.c_uint32, C.c_uint8, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int,
                                     C.c_uint32, C.c_int, C.c_char_p, C.c_uint64]
    return lib


def compressed_blocks(frame):
    """the compressed blocks (type 2) of the frames in `frame`: their bytes"""
    pos = 0
    while pos + 6 <= len(frame):
        fhd = frame[pos + 4]
        fcs, single, did = fhd >> 6, (fhd >> 5) & 1, fhd & 3
        pos += 5 + (0 if single else 1) + (0, 1, 2, 4)[did] + ((1, 2, 4, 8)[fcs] if (fcs or single) else 0)
        while True:
            h = int.from_bytes(frame[pos:pos + 3], "little")
            pos += 3
            btype, size = (h >> 1) & 3, h >> 3
            if btype == 2:
                yield frame[pos:pos + size]
            pos += 1 if btype == 1 else size
            if h & 1:
                break
        pos += 4 if fhd & 4 else 0


def run(lib, sec, shift=0, order=(0, 0), poison=0xEE, cap=1 << 17):
    want = (C.c_uint8 * cap)()
    got = (C.c_uint8 * (cap + 16))()
    C.memset(got, 0x5A, cap + 16)
    err = C.create_string_buffer(320)
    r = lib.kta_emu_zstd_huf(sec, len(sec), shift, poison, want, got, cap, order[0], order[1], 0, err, 320)
    assert r != -2, err.value
    if r >= 0:
        assert bytes(got[:r]) == bytes(want[:r])
        assert bytes(got[r:r + 16]) == b"\x5A" * 16
    return r


def _texts():
    rng = np.random.default_rng(41)
    words = [b"user", b"action", b"click", b"view", b"purchase", b"session", b"timestamp", b"amount", b"currency", b"EUR", b"status",
             b"ok", b"error", b"region", b"eu-west", b"device", b"mobile", b"payload", b"items", b"price", b"true", b"false", b"null"]
    json_like = b"".join(words[int(i)] + bytes([b'":, {}_'[int(j)]]) for i, j in zip(rng.integers(0, len(words), 60000), rng.integers(0, 7, 60000)))
    skew = bytes(rng.choice(np.arange(256, dtype=np.uint8), size=200000, p=np.r_[[0.3, 0.2, 0.1], np.full(253, 0.4 / 253)]))
    few = bytes(rng.choice(np.frombuffer(b"abcdefgh", np.uint8), size=50000, p=[0.4, 0.2, 0.1, 0.1, 0.05, 0.05, 0.05, 0.05]))
    return {"json": json_like, "json16k": json_like[:16384], "skew": skew, "few": few, "short": json_like[:700]}


def test_wave_decodes_what_the_streams_hold(emu):
    """Sections with one stream and with four, from a few hundred bytes to 128 KiB of literals (streams of several 2 KiB
    windows), trees of few and of many symbols, every lane order, every alignment of the section in the buffer."""
    done = four = long_streams = 0
    for name, d in _texts().items():
        for level in (1, 3, 19):
            for blk in compressed_blocks(pa.Codec("zstd", compression_level=level).compress(d, asbytes=True)):
                if blk[0] & 3 != 2:
                    continue                              # raw / RLE literals, or the tree of the block before
                r = run(emu, blk, shift=done % 16, order=ORDERS[done % 3])
                assert r > 0, (name, level, r)
                done += 1
                four += (blk[0] >> 2) & 3 != 0
                long_streams += r > 40000
    assert done > 12 and four > 8 and long_streams > 3, (done, four, long_streams)


def test_damaged_streams_get_the_host_statements_verdict(emu):
    """Flipped bits — in the tree's description, the streams' sizes, the streams —: the wave accepts exactly what zs_huf_stream
    accepts, with the same literals (a flipped bit inside a stream usually turns one symbol into another of the same length:
    accepted by both); nothing is written behind them."""
    rng = np.random.default_rng(3)
    d = _texts()["json16k"]
    blk = next(b for b in compressed_blocks(pa.Codec("zstd", compression_level=3).compress(d, asbytes=True)) if b[0] & 3 == 2)
    accepted = refused = 0
    for t in range(240):
        bad = bytearray(blk)
        at = int(rng.integers(3, 200)) if t % 3 == 0 else int(rng.integers(200, len(blk) // 2))
        bad[at] ^= 1 << int(rng.integers(0, 8))
        r = run(emu, bytes(bad), shift=t % 16, order=ORDERS[t % 3])
        assert r != -3                                     # the two never disagree
        accepted += r >= 0
        refused += r == -1
    assert refused > 20 and accepted > 100, (accepted, refused)
