"""Memory safety of the inflaters the device kernels run (csrc/kta_snappy.h, kta_lz4.h, kta_gzip.h, kta_zstd.h):
compiled for the host with AddressSanitizer and driven with thousands of mutated streams from the
real libraries (zlib, and Google snappy / liblz4 through pyarrow when present) and from the test
compressors.  A corrupt batch must be refused or decoded to *something* inside its slice — never read
or write out of bounds (on the GPU that would be a device fault)."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

import lz4_py
import snappy_py

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "kafka_topic_analyzer_amd", "csrc")


@pytest.fixture(scope="module")
def fuzzer(tmp_path_factory):
    d = tmp_path_factory.mktemp("fuzz")
    exe = str(d / "inflate_fuzz")
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                        "-I", CSRC, os.path.join(ROOT, "tests", "native", "inflate_fuzz.cpp"), "-o", exe],
                       capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("no AddressSanitizer toolchain: " + r.stderr[-300:])
    return exe, d


def _payloads():
    rng = np.random.default_rng(77)
    text = b"".join(b"user-%05d|%s|balance=%d;" % (i % 513, b"x" * (i % 37), i * 7919 % 100003) for i in range(3000))
    return [b"a", b"abcd" * 500, bytes(rng.integers(0, 256, size=3000, dtype=np.uint8)), b"\0" * 20000,
            bytes(rng.integers(0, 4, size=9000, dtype=np.uint8)), text,
            text[:20000] + bytes(rng.integers(0, 256, size=40000, dtype=np.uint8)) + text[:20000]]


def _write(d, name, data, stream):
    p = str(d / name)
    with open(p, "wb") as f:
        f.write(struct.pack("<Q", len(data)) + stream)
    return p


@pytest.mark.parametrize("codec", ["snappy", "lz4", "gzip", "gzip2", "zstd"])
def test_inflaters_are_memory_safe_on_mutated_streams(fuzzer, codec):
    exe, d = fuzzer
    try:
        import pyarrow as pa
    except ImportError:
        pa = None
    seeds = []
    for i, data in enumerate(_payloads()):
        if codec == "snappy":
            seeds.append(_write(d, f"s{i}a", data, snappy_py.compress_block(data)))
            seeds.append(_write(d, f"s{i}b", data, snappy_py.compress_xerial(data, 4096)))
            if pa:
                seeds.append(_write(d, f"s{i}c", data, pa.compress(data, codec="snappy", asbytes=True)))
        elif codec == "lz4":
            seeds.append(_write(d, f"l{i}a", data, lz4_py.compress_frame(data)))
            seeds.append(_write(d, f"l{i}b", data, lz4_py.compress_frame(data, linked=False, content_size=True,
                                                                         block_checksum=True, content_checksum=True)))
            if pa:
                seeds.append(_write(d, f"l{i}c", data, pa.compress(data, codec="lz4", asbytes=True)))
        elif codec == "zstd":
            if not pa:
                pytest.skip("zstd seeds come from libzstd through pyarrow")
            for level in (1, 3, 19):
                seeds.append(_write(d, f"z{i}{level}", data, pa.Codec("zstd", compression_level=level).compress(data, asbytes=True)))
            sink = pa.BufferOutputStream()
            with pa.CompressedOutputStream(sink, "zstd") as o:
                o.write(data)
            seeds.append(_write(d, f"z{i}s", data, sink.getvalue().to_pybytes()))
        else:
            for j, (level, strategy) in enumerate([(6, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_FIXED), (0, zlib.Z_DEFAULT_STRATEGY),
                                                   (9, zlib.Z_HUFFMAN_ONLY)]):
                co = zlib.compressobj(level, zlib.DEFLATED, 15 + 16, 8, strategy)
                seeds.append(_write(d, f"g{i}{j}", data, co.compress(data) + co.flush()))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe, codec, *seeds], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    ok, refused = (int(x.split("=")[1]) for x in r.stdout.split()[1:3])
    assert ok + refused >= 1400 * len(seeds) and refused > ok // 20    # most mutations are noticed, none faults
