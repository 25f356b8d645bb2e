"""Build the native artefacts in-tree (no JIT cache, so the .so files travel with the repo snapshot).

    libkta_hip.so   kafka_topic_analyzer_amd/csrc/*.hip   hipcc --offload-arch=gfx950
    kta-analyzer    kafka_topic_analyzer_amd/csrc/host/*  C++ CLI with the reference's flags/report
    oracle/libkta_oracle.so  (test infrastructure)        gcc

hipcc cross-compiles for gfx950 without a GPU present.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libkta_hip.so")
CLI = os.path.join(HERE, "kta-analyzer")
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "libkta_oracle.so")

HIP_SOURCES = ["kta_kernels.hip", "kta_alive.hip", "kta_api.hip", "kta_comm.hip", "kta_synth.hip", "kta_kafka.hip"]
LIB_HOST_SOURCES = ["host/metric.cpp", "host/report.cpp", "host/kafka_encode.cpp"]  # C++ host mirror, inside libkta_hip.so
HOST_SOURCES = ["host/main.cpp", "host/rdkafka_source.cpp"]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall",
               "-Wno-unused-result"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libkta_hip.so cannot be built (there is no CPU fallback)")


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def _run(cmd) -> None:
    print("+", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)


def _deps(sources):
    deps = [os.path.join(CSRC, s) for s in sources]
    for d in (CSRC, os.path.join(CSRC, "host"), os.path.join(ROOT, "include")):
        if os.path.isdir(d):
            deps += [os.path.join(d, f) for f in os.listdir(d) if f.endswith((".h", ".hpp"))]
    deps.append(os.path.abspath(__file__))
    return deps


def build_lib(force: bool = False) -> str:
    deps = _deps(HIP_SOURCES + LIB_HOST_SOURCES)
    if not force and _newer(LIB, deps):
        return LIB
    hipcc = _hipcc()
    objs = []
    for s in HIP_SOURCES + LIB_HOST_SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.splitext(src)[0] + ".o"
        if force or not _newer(obj, deps):
            arch = ["--offload-arch=gfx950"] if s.endswith(".hip") else []
            flags = [f for f in HIPCC_FLAGS if not f.startswith("--offload-arch")]
            _run([hipcc, *arch, *flags, "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-c", src, "-o", obj])
        objs.append(obj)
    _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, "-ldl",
          "-Wl,-rpath,/opt/rocm/lib"])
    return LIB


def build_cli(force: bool = False):
    srcs = [os.path.join(CSRC, s) for s in HOST_SOURCES]
    if not all(os.path.exists(s) for s in srcs):
        return None
    deps = _deps(HOST_SOURCES) + [LIB]
    if not force and _newer(CLI, deps):
        return CLI
    _run([_hipcc(), "-O2", "-std=c++17", "-ffp-contract=off", "-Wall", "-I", os.path.join(ROOT, "include"),
          "-I", CSRC, *srcs, "-o", CLI, "-L", HERE, "-lkta_hip", "-ldl", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib"])
    return CLI


def build_oracle(force: bool = False) -> str:
    deps = [os.path.join(ORACLE_DIR, f) for f in ("kta_oracle.c", "kta_oracle.h", "kta_kafka_oracle.c", "Makefile")]
    if not force and _newer(ORACLE_LIB, deps):
        return ORACLE_LIB
    _run(["make", "-C", ORACLE_DIR] + (["-B"] if force else []))
    return ORACLE_LIB


def build_all(force: bool = False) -> None:
    build_lib(force)
    build_cli(force)
    build_oracle(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
