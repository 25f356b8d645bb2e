"""kafka_gzip_tokenize_wave — the kernel's own source (csrc/kta_gzip_wave.h) — run on the CPU: tests/native/wave_emu.h makes
the 64 lanes of the wave fibers that meet at __syncthreads / ballots / shuffles / readlanes and runs them, between two
meeting points, in ascending, descending or shuffled order.  So the speculative decode of a block's 64 segments, the loop
that confirms them, the prefix sums and the writing decode are executed here as the GPU executes them, and what they
leave — literals in place + tokens, applied by kta::gz_apply_tokens — is held against zlib's output; on damaged streams
against the verdict of the lane tokenizer's text (kta_gzip_inflate_host), to which the kernel leaves what it does not
finish.  The GPU tests of tests/test_kafka_decode.py stay the parity gate: code generation and hardware are not emulated.

    KTA_EMU_ASAN=1 ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD="$(g++ -print-file-name=libasan.so) $(g++ -print-file-name=libubsan.so)" \
        python -m pytest tests/test_gzip_wave_emu.py
builds the emulated kernel with AddressSanitizer + UBSan (fetch buffer, output and token area are heap blocks of exactly
their sizes)."""
import ctypes as C
import os
import subprocess
import zlib

import numpy as np
import pytest

from kafka_topic_analyzer_amd import _native as N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "kafka_topic_analyzer_amd", "csrc")
NATIVE = os.path.join(ROOT, "tests", "native")
ORDERS = [(0, 0), (1, 0), (2, 7)]            # (lane order between meeting points, seed)
LEFT, DIVERGED, BAD_TOKENS = -1, -2, -3


@pytest.fixture(scope="module")
def emu(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("emu") / "libkta_gzip_wave_emu.so")
    sanitize = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all"] if os.environ.get("KTA_EMU_ASAN") else []
    r = subprocess.run(["g++", "-O2", "-g", "-std=c++17", "-shared", "-fPIC", "-Wall", "-Wno-unknown-pragmas", *sanitize,
                        "-I", CSRC, "-I", NATIVE, os.path.join(NATIVE, "gzip_wave_emu.cpp"), "-o", so],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    lib = C.CDLL(so)
    lib.kta_emu_gzip_wave.restype = C.c_int64
    lib.kta_emu_gzip_wave.argtypes = [C.c_char_p, C.c_uint64, C.c_uint32, C.c_uint8, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int,
                                      C.c_uint32, C.c_char_p, C.c_uint64]
    lib.kta_emu_gzip_wave_stats.argtypes = [C.POINTER(C.c_uint32), C.c_int]
    return lib


def gz(d, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, flush_at=(), flush=zlib.Z_FULL_FLUSH, mem_level=8):
    co = zlib.compressobj(level, zlib.DEFLATED, 31, mem_level, strategy)
    out, at = b"", 0
    for f in flush_at:
        out += co.compress(d[at:f]) + co.flush(flush)
        at = f
    return out + co.compress(d[at:]) + co.flush()


def token_room(cap, clen):
    # what the host index reserves (kta_kafka.hip: gz_token_bound + gz_closing_tokens)
    return cap // 3 + cap // 255 + 1 + 64 * (2 + clen // 2048)


def run(lib, comp, cap, shift=0, order=(0, 0), poison=0xEE, tok_cap=None):
    """-> (tokens or LEFT, output bytes); asserts that nothing behind the output was touched"""
    dst = (C.c_uint8 * (cap + 16))()
    C.memset(dst, 0x5A, cap + 16)
    err = C.create_string_buffer(320)
    r = lib.kta_emu_gzip_wave(comp, len(comp), shift, poison, dst, cap, token_room(cap, len(comp)) if tok_cap is None else tok_cap,
                              order[0], order[1], err, 320)
    assert r != DIVERGED, err.value
    assert r != BAD_TOKENS
    assert bytes(dst[cap:cap + 16]) == b"\x5A" * 16
    return r, bytes(dst[:cap])


def _cases():
    rng = np.random.default_rng(23)
    text = b"".join(b"user-%05d|%s|balance=%d;" % (i % 513, b"x" * (i % 37), i * 7919 % 100003) for i in range(6000))
    skew = bytes(rng.choice(np.arange(256, dtype=np.uint8), size=60000, p=np.r_[[0.3, 0.2, 0.1], np.full(253, 0.4 / 253)]))
    pattern = b"".join(bytes(rng.integers(0, 256, 24, dtype=np.uint8)) * 9 for _ in range(300))
    return {"text": text, "text16k": text[:16384], "tiny": b"hello hello hello world", "one": b"a", "zeros": b"\0" * 300000,
            "period7": b"abcdefg" * 9000, "skew": skew, "pattern": pattern, "two": b"ab",
            "runs": b"".join(bytes([int(x)]) * int(y) for x, y in zip(rng.integers(0, 256, 2000), rng.integers(1, 300, 2000)))}


def test_wave_tokenizer_against_zlib(emu):
    """Members written by zlib — levels 1 / 6 / 9, the default, fixed-code, Huffman-only, RLE and filtered strategies, small
    memLevels (many short blocks) —: the wave finishes every one of them (none of these has a stored block), in every lane
    order, at every alignment of the stream in the fetch buffer, and the tokens applied give zlib's input back."""
    done = 0
    st = (C.c_uint32 * 4)()
    emu.kta_emu_gzip_wave_stats(st, 1)
    for name, d in _cases().items():
        variants = [gz(d, lv) for lv in (1, 6, 9)] + [gz(d, 6, s) for s in (zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED)]
        variants += [gz(d, 6, mem_level=1), gz(d, 9, mem_level=2)]
        for k, comp in enumerate(variants):
            order = ORDERS[(done + k) % 3]
            r, out = run(emu, comp, len(d), shift=(done * 5 + k) % 16, order=order)
            assert r >= 0 and out == d, (name, k, r)
            done += 1
    emu.kta_emu_gzip_wave_stats(st, 1)
    regions, reps, again, lanes = st[0], st[1], st[2], st[3]
    assert regions > done and lanes > 20 * regions       # the members above are decoded by many lanes at once ...
    assert reps < 4 * regions, (regions, reps)          # ... which fall into step: few repetitions of the confirming loop


def test_members_of_several_blocks_and_windows(emu):
    """Blocks that end inside a window, windows that end inside a block: full and sync flushes every few hundred bytes (each
    closes a block; a sync flush adds an empty stored block, which the kernel leaves to the lane kernel), and members
    several times the 7 KiB window."""
    c = _cases()
    d = c["text"]
    # Z_FULL_FLUSH / Z_SYNC_FLUSH end with an empty STORED block
    for k, flush in enumerate((zlib.Z_FULL_FLUSH, zlib.Z_SYNC_FLUSH)):
        comp = gz(d, 6, flush_at=range(700, len(d), 9001), flush=flush)
        r, out = run(emu, comp, len(d), shift=3 + k, order=ORDERS[k])
        assert r >= 0 and out == d
    # Z_BLOCK closes the block without one
    for order in ORDERS:
        comp = gz(d, 6, flush_at=range(300, len(d), 4099), flush=zlib.Z_BLOCK)
        r, out = run(emu, comp, len(d), shift=9, order=order)
        assert r >= 0 and out == d
    big = c["skew"] * 6                                  # ~ 300 KB compressed: dozens of windows
    comp = gz(big, 6)
    assert len(comp) > 20 * 7168
    r, out = run(emu, comp, len(big), shift=15, order=ORDERS[2])
    assert r >= 0 and out == big


def test_what_the_wave_leaves_to_the_lane_kernel(emu):
    """A token area without room for the closing tokens, an output slice of the wrong size, a damaged stored block: left, not
    mis-decoded — and nothing written behind the output.  (Stored blocks themselves — incompressible values, level 0, several
    of 64 KiB in a row, between coded blocks — are this kernel's.)"""
    rng = np.random.default_rng(5)
    noise = bytes(rng.integers(0, 256, 200000, dtype=np.uint8))
    for k, (d, level) in enumerate([(noise[:5000], 6), (noise[:5000], 0), (noise, 0), (noise[:70000] + b"abc" * 9000 + noise[:3000], 6), (b"", 6)]):
        comp = gz(d, level)
        r, out = run(emu, comp, len(d), shift=k * 3, order=ORDERS[k % 3])
        assert r >= 0 and out == d, k
    comp = bytearray(gz(noise[:5000], 0))
    comp[10 + 3] ^= 1                                                # LEN and ~LEN disagree
    assert run(emu, bytes(comp), 5000)[0] == LEFT
    d = _cases()["text16k"]
    comp = gz(d, 6)
    r, out = run(emu, comp, len(d))
    assert r > 64 and out == d
    assert run(emu, comp, len(d), tok_cap=r - 1)[0] == LEFT          # one token short
    assert run(emu, comp, len(d), tok_cap=r)[0] == r
    assert run(emu, comp, len(d) - 1)[0] == LEFT                     # the slice is a byte short / long (the trailer lied)
    assert run(emu, comp, len(d) + 1)[0] == LEFT
    assert run(emu, comp[:-9] + comp[-8:], len(d))[0] == LEFT        # the stream is a byte short
    assert run(emu, b"\x1f\x8c" + comp[2:], len(d))[0] == LEFT       # not a gzip member


def test_damaged_streams_never_disagree_with_the_lane_tokenizer(emu):
    """Flipped bits: whatever the wave FINISHES, the lane tokenizer's text (kta_gzip_inflate_host) accepts too, with the
    same bytes; the rest it leaves (most damaged streams: an invalid code, a match before the output's first byte, a
    size that does not add up).  Nothing faults, nothing is written behind the output."""
    lib = N.load()
    rng = np.random.default_rng(77)
    c = _cases()
    finished = left = left_though_fine = 0
    for name in ("text16k", "pattern", "skew"):
        d = c[name][:30000]
        good = gz(d, 6)
        for t in range(120):
            bad = bytearray(good)
            for _ in range(int(rng.integers(1, 3))):
                bad[int(rng.integers(10, len(good) - 8))] ^= 1 << int(rng.integers(0, 8))
            bad = bytes(bad)
            r, out = run(emu, bad, len(d), shift=t % 16, order=ORDERS[t % 3])
            ref = C.create_string_buffer(len(d) + 1)
            want = lib.kta_gzip_inflate_host(bad, len(bad), ref, len(d))
            if r >= 0:
                assert want == len(d) and out == ref.raw[:len(d)], (name, t)
                finished += 1
            else:
                left += 1
                left_though_fine += want == len(d)
    # (a flipped bit of a literal's code often yields another literal: the member still adds up — the CRC-32 of the trailer is
    # not checked, the batch's CRC-32C covers the compressed bytes — and both decoders give the same changed bytes)
    assert left > 100 and finished > 100 and left_though_fine <= 2, (left, finished, left_though_fine)
