"""Builds the in-tree native libraries with hipcc for gfx950 (cross-compiles without a GPU).

    python -m fqtk_amd.build            # build if stale
    python -m fqtk_amd.build --force
    python -m fqtk_amd.build --sanitize=address   # ASan + UBSan build of the host side (also: --sanitize=thread)
"""
from __future__ import annotations

import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

# name -> (sources, extra flags)
TARGETS = {
    "libfqtk_match.so": (["fqtk_match.hip", "fqtk_bgzf.hip", "fqtk_inflate.hip", "fqtk_demux.hip"], []),
    "libfqtk_synth.so": (["synth.hip"], []),
}


def _stale(out: str, deps) -> bool:
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


HOST = os.path.join(CSRC, "host")
BINDIR = os.path.join(HERE, "bin")
CXX = os.environ.get("CXX", "g++")


SANITIZERS = {"address": ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"], "thread": ["-fsanitize=thread"]}


def sanitized_paths(kind: str):
    """(shim, binary) of a sanitizer build: fqtk_amd/lib/san_<kind>/libfqtk_host.so, fqtk_amd/bin/fqtk.<kind>."""
    return os.path.join(LIBDIR, f"san_{kind}", "libfqtk_host.so"), os.path.join(BINDIR, f"fqtk.{kind}")


def build_sanitized(kind: str, force: bool = False, verbose: bool = False) -> None:
    """ASan + UBSan or TSan builds of the host side: the ctypes shim the CPU tests load under
    LD_PRELOAD=libasan/libtsan (tests/test_sanitizers.py) and the `fqtk` binary (its threads -- cutters, count assistants,
    copy helpers, collector, writers, unmapper, gunzip decoders -- run under TSan on a GPU box: tools/soak_cli.py --exe)."""
    flags = SANITIZERS[kind]
    shim, exe = sanitized_paths(kind)
    os.makedirs(os.path.dirname(shim), exist_ok=True)
    os.makedirs(BINDIR, exist_ok=True)
    deps = (glob.glob(os.path.join(HOST, "*.hpp")) + glob.glob(os.path.join(INCLUDE, "*.h")) + glob.glob(os.path.join(CSRC, "*.hpp")))
    common = [CXX, "-O1", "-g", "-fno-omit-frame-pointer", "-std=c++17", "-Wall", "-pthread"] + flags
    src = os.path.join(HOST, "host_capi.cpp")
    if force or _stale(shim, [src] + deps):
        cmd = common + ["-shared", "-fPIC", "-o", shim, src, "-lz", "-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    src = os.path.join(HOST, "demux.cpp")
    if os.path.exists(os.path.join(LIBDIR, "libfqtk_match.so")) and (force or _stale(exe, [src] + deps)):
        cmd = common + ["-o", exe, src, "-L", LIBDIR, "-lfqtk_match", "-lz", "-ldl", "-Wl,-rpath,$ORIGIN/../lib", "-Wl,-rpath,/opt/rocm/lib"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)


def build_host(force: bool = False, verbose: bool = False) -> None:
    """The C++ host side above the C ABI: `fqtk demux` binary + a ctypes shim for the CPU tests."""
    os.makedirs(BINDIR, exist_ok=True)
    deps = (glob.glob(os.path.join(HOST, "*.hpp")) + glob.glob(os.path.join(INCLUDE, "*.h")) +
            glob.glob(os.path.join(CSRC, "*.hpp")))   # host_capi.cpp uses csrc/lds_memo_plan.hpp + memo_hash.hpp
    shim = os.path.join(LIBDIR, "libfqtk_host.so")
    src = os.path.join(HOST, "host_capi.cpp")
    if force or _stale(shim, [src] + deps):
        cmd = [CXX, "-O2", "-std=c++17", "-shared", "-fPIC", "-Wall", "-pthread", "-o", shim, src, "-lz", "-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    exe = os.path.join(BINDIR, "fqtk")
    src = os.path.join(HOST, "demux.cpp")
    if force or _stale(exe, [src] + deps + [os.path.join(LIBDIR, "libfqtk_match.so")]):
        cmd = [CXX, "-O2", "-std=c++17", "-Wall", "-pthread", "-o", exe, src, "-L", LIBDIR, "-lfqtk_match",
               "-lz", "-ldl", "-Wl,-rpath,$ORIGIN/../lib", "-Wl,-rpath,/opt/rocm/lib"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)


def build(force: bool = False, verbose: bool = False) -> None:
    """Every .hip source is compiled to its own object (side by side: the three of libfqtk_match.so take a minute
    one after another), then linked.  FQTK_EXTRA_DEFS="-DX=1 ..." adds defines (developer A/B builds)."""
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    extra_defs = os.environ.get("FQTK_EXTRA_DEFS", "").split()
    stamp = os.path.join(objdir, "defs.txt")
    if (open(stamp).read() if os.path.exists(stamp) else "") != " ".join(extra_defs):
        force = True
    deps_common = (glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.hpp")) +
                   glob.glob(os.path.join(INCLUDE, "*.h")))
    jobs, links = [], []
    for name, (srcs, extra) in TARGETS.items():
        srcs_abs = [os.path.join(CSRC, s) for s in srcs]
        if not all(os.path.exists(s) for s in srcs_abs):
            continue
        objs = []
        for src in srcs_abs:
            obj = os.path.join(objdir, os.path.basename(src) + ".o")
            objs.append(obj)
            if force or _stale(obj, [src] + deps_common):
                # (-fvisibility-inlines-hidden: the library's copies of inline C++ functions -- std::unique_lock::unlock ... -- are its own; a
                #  C-ABI library neither exports them nor takes the executable's, which under TSan are instrumented while the lock they pair
                #  with was taken in here, unseen: "unlock of an unlocked mutex")
                jobs.append([HIPCC, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-fvisibility-inlines-hidden", "-I", INCLUDE, "-c", "-o", obj] + extra + extra_defs + [src])
        links.append((os.path.join(LIBDIR, name), objs))

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max(1, len(jobs))) as ex:
        list(ex.map(run, jobs))
    for out, objs in links:
        if force or _stale(out, objs):
            run([HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-Wl,-Bsymbolic-functions", "-o", out] + objs)
    open(stamp, "w").write(" ".join(extra_defs))
    build_host(force=force, verbose=verbose)


if __name__ == "__main__":
    kinds = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--sanitize=")]
    for k in kinds:
        if k not in SANITIZERS:
            sys.exit(f"--sanitize={k}: one of {sorted(SANITIZERS)}")
    if kinds:
        for k in kinds:
            build_sanitized(k, force="--force" in sys.argv, verbose=True)
    else:
        build(force="--force" in sys.argv, verbose=True)
