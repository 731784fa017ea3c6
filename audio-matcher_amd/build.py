"""Builds libaudiomatch_amd.so (HIP, gfx950 only) in-tree with hipcc.

Used by __graft_entry__.build(), tests/conftest.py and bench.py.  hipcc
cross-compiles without a GPU; the resulting .so is git-ignored but travels to
the GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libaudiomatch_amd.so")
SOURCES = ["am_fft.hip", "am_peaks.hip", "am_api.hip"]
HEADERS = [os.path.join(CSRC, "am_kernels.h"),
           os.path.join(HERE, "..", "include", "audiomatch.h")]
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-ffp-contract=fast", "-fno-slp-vectorize",
         "-Wall", "-Wno-unused-result"]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP extension cannot be built")


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_library(force: bool = False, verbose: bool = False) -> str:
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    if not force and not _stale(LIB, srcs + HEADERS + [os.path.abspath(__file__)]):
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()

    def compile_one(src: str) -> str:
        obj = os.path.join(OBJ, os.path.basename(src) + ".o")
        if force or _stale(obj, [src] + HEADERS + [os.path.abspath(__file__)]):
            cmd = [hipcc, *FLAGS, "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=len(srcs)) as ex:
        objs = list(ex.map(compile_one, srcs))
    cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


def build_variant(name: str, defines: list[str]) -> str:
    """A/B builds for tools/ab.sh: the library with extra -D flags, as build/variants/<name>.so."""
    vdir = os.path.join(OBJ, "variants")
    odir = os.path.join(OBJ, "variants_obj", name)
    os.makedirs(vdir, exist_ok=True)
    os.makedirs(odir, exist_ok=True)
    hipcc = _hipcc()
    objs = []
    for s_ in SOURCES:
        obj = os.path.join(odir, s_ + ".o")
        subprocess.check_call([hipcc, *FLAGS, *[f"-D{d}" for d in defines], "-c", os.path.join(CSRC, s_), "-o", obj])
        objs.append(obj)
    out = os.path.join(vdir, name + ".so")
    subprocess.check_call([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", out, *objs])
    return out


CLI = os.path.join(HERE, "bin", "audiomatch")
HOST = os.path.join(HERE, "host")


def build_cli(force: bool = False) -> str:
    """The C++ host CLI (matcher::run, src/matcher/mod.rs:17-104) linked against the library."""
    lib = build_library()
    srcs = [os.path.join(HOST, "audiomatch_cli.cpp"), os.path.join(HOST, "am_host.hpp")]
    if not force and not _stale(CLI, srcs + [lib]):
        return CLI
    os.makedirs(os.path.dirname(CLI), exist_ok=True)
    cmd = [_hipcc(), "-O2", "-std=c++17", "-x", "c++", srcs[0], "-o", CLI,
           f"-L{HERE}", "-laudiomatch_amd", "-Wl,-rpath,$ORIGIN/.."]
    subprocess.check_call(cmd)
    return CLI


PUSHBENCH = os.path.join(HERE, "bin", "pushbench")


def build_pushbench(force: bool = False) -> str:
    """host/pushbench.cpp: what feeding the library from host memory costs (C ABI only)."""
    lib = build_library()
    src = os.path.join(HOST, "pushbench.cpp")
    if not force and not _stale(PUSHBENCH, [src, lib]):
        return PUSHBENCH
    os.makedirs(os.path.dirname(PUSHBENCH), exist_ok=True)
    subprocess.check_call([_hipcc(), "-O2", "-std=c++17", "-x", "c++", src, "-o", PUSHBENCH,
                           f"-L{HERE}", "-laudiomatch_amd", "-Wl,-rpath,$ORIGIN/.."])
    return PUSHBENCH


def build_selftest() -> str:
    """CPU-only self test of the host-side rows (g++ only)."""
    out = os.path.join(OBJ, "am_host_selftest")
    os.makedirs(OBJ, exist_ok=True)
    src = os.path.join(HOST, "selftest.cpp")
    if _stale(out, [src, os.path.join(HOST, "am_host.hpp")]):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(HERE, "..", "include"), "-o", out, src])
    return out


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
    print(build_cli(force="--force" in sys.argv))
    print(build_pushbench(force="--force" in sys.argv))
