"""Build libpba.so (the HIP kernels + C ABI) in-tree for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting
pacbioassembly_amd/lib/libpba.so is git-ignored but travels to the GPU box with the snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libpba.so")

SOURCES = ["pba_device.hip", "pba_codec.cpp", "pba_synth.cpp"]
HEADERS = ["dev_common.h", "align_rowsweep.h", "align_bitvec.h", "align_bvtrace.h", "prefilter.h", "consensus.h", "seed_index.h",
           "overlap.h", "pba_internal.h"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libpba.so cannot be built (there is no CPU fallback)")


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.join(ROOT, "include", "pba.h")]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP/C++ source into one shared library; returns its path."""
    if not force and not _stale():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    cmd = [
        _hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
        "-ffp-contract=off",            # i*R and len*(1-R) must round exactly like the reference's FP64
        "-fgpu-rdc" if False else "-fno-gpu-rdc",
        "-Wall", "-Wno-unused-function",
        "-I", os.path.join(ROOT, "include"), "-I", CSRC,
        "-o", LIB,
    ] + os.environ.get("PBA_EXTRA_CFLAGS", "").split() + [os.path.join(CSRC, s) for s in SOURCES] + ["-lpthread"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
