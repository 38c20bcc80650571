#!/usr/bin/env python3
"""Build a tuning variant of libpba.so next to the real one: tools/build_variant.py NAME -DFOO=1 ...  ->
pacbioassembly_amd/lib/variants/NAME/libpba.so (git-ignored, travels to the GPU box); run anything with
PBA_LIB_PATH=<that path> to use it.  PBA_VARIANT_CSRC=<dir> compiles another copy of csrc/ (e.g. `git worktree` of an
older commit) so that two source states can be timed on the same box."""
import os, subprocess, sys, concurrent.futures
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pacbioassembly_amd import build as b
name, extra = sys.argv[1], sys.argv[2:]
csrc = os.environ.get("PBA_VARIANT_CSRC", b.CSRC)
out = os.path.join(b.LIBDIR, "variants", name)
os.makedirs(out, exist_ok=True)
def cc(s):
    o = os.path.join(out, s + ".o")
    subprocess.run([b._hipcc()] + [csrc if f == b.CSRC else f for f in b._flags()] + extra + ["-c", os.path.join(csrc, s), "-o", o],
                   check=True, cwd=csrc)
    return o
with concurrent.futures.ThreadPoolExecutor(8) as ex:
    objs = list(ex.map(cc, b.SOURCES))
lib = os.path.join(out, "libpba.so")
subprocess.run([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc", "-o", lib] + objs + ["-lpthread"], check=True)
for o in objs:
    os.remove(o)
print(lib)
