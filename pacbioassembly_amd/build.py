"""Build libpba.so (the HIP kernels + C ABI) in-tree for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting
pacbioassembly_amd/lib/libpba.so is git-ignored but travels to the GPU box with the snapshot.
Every translation unit is compiled to its own object (in parallel, only when it or a header changed) and the
objects are linked into one shared library.
"""
from __future__ import annotations

import concurrent.futures
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libpba.so")

SOURCES = ["pba_core.hip", "pba_align.hip", "pba_drivers.hip", "pba_overlap.hip", "pba_cons.hip", "pba_codec.cpp",
           "pba_synth.cpp"]
# the exchange over RCCL for C / C++ hosts (include/pba_dist.h): host code only, its own small library next to libpba.so
DIST_SOURCE = "pba_dist.hip"
DIST_LIB = os.path.join(LIBDIR, "libpba_dist.so")


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libpba.so cannot be built (there is no CPU fallback)")


def _headers():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join(ROOT, "include", "pba.h")]


def _flags():
    return ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
            "-ffp-contract=off",            # i*R and len*(1-R) must round exactly like the reference's FP64
            "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
            "-I", os.path.join(ROOT, "include"), "-I", CSRC] + os.environ.get("PBA_EXTRA_CFLAGS", "").split()


def source_digest() -> str:
    """Digest of everything the library is built from (bench.py ties committed profile figures to it)."""
    h = hashlib.sha256()
    for p in [os.path.join(CSRC, s) for s in SOURCES] + _headers():
        h.update(os.path.basename(p).encode())
        h.update(open(p, "rb").read())
    h.update(" ".join(_flags()[:8]).encode())
    return h.hexdigest()[:16]


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP/C++ source and link one shared library; returns its path."""
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= max(os.path.getmtime(p) for p in srcs + _headers()) \
            and not os.environ.get("PBA_EXTRA_CFLAGS"):
        build_dist()
        return LIB                      # (the objects do not travel to the GPU box; the library does)
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc, flags = _hipcc(), _flags()
    newest_hdr = max(os.path.getmtime(h) for h in _headers())
    stamp = os.path.join(OBJDIR, "flags.txt")
    flag_text = " ".join(flags)
    if not os.path.exists(stamp) or open(stamp).read() != flag_text:
        force = True
    jobs = []
    for s in SOURCES:
        src, obj = os.path.join(CSRC, s), os.path.join(OBJDIR, s + ".o")
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), newest_hdr):
            jobs.append([hipcc] + flags + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True, cwd=CSRC)

    if jobs:
        with concurrent.futures.ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            list(ex.map(run, jobs))
    objs = [os.path.join(OBJDIR, s + ".o") for s in SOURCES]
    if jobs or not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(o) for o in objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc", "-o", LIB] + objs + ["-lpthread"])
        open(stamp, "w").write(flag_text)
    build_dist(verbose=verbose)
    return LIB


def build_dist(force: bool = False, verbose: bool = False) -> str:
    """libpba_dist.so: pba_dist.hip against libpba.so and librccl.so (found beside libpba.so through $ORIGIN)."""
    src = os.path.join(CSRC, DIST_SOURCE)
    deps = [src, LIB] + _headers() + [os.path.join(ROOT, "include", "pba_dist.h")]
    if not force and os.path.exists(DIST_LIB) and os.path.getmtime(DIST_LIB) >= max(os.path.getmtime(p) for p in deps):
        return DIST_LIB
    rocm_lib = os.path.join(os.path.dirname(os.path.dirname(_hipcc())), "lib")
    cmd = [_hipcc()] + _flags() + ["-shared", "-o", DIST_LIB, src, "-L", LIBDIR, "-lpba", "-L", rocm_lib, "-lrccl",
                                   "-Wl,-rpath,$ORIGIN", "-lpthread"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True, cwd=CSRC)
    return DIST_LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
