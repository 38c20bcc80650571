#!/usr/bin/env python3
"""Build a tuning variant of libpba.so next to the real one: tools/build_variant.py NAME -DFOO=1 ...  ->
pacbioassembly_amd/lib/variants/NAME/libpba.so (git-ignored, travels to the GPU box); run anything with
PBA_LIB_PATH=<that path> to use it."""
import os, subprocess, sys, concurrent.futures
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pacbioassembly_amd import build as b
name, extra = sys.argv[1], sys.argv[2:]
out = os.path.join(b.LIBDIR, "variants", name)
os.makedirs(out, exist_ok=True)
def cc(s):
    o = os.path.join(out, s + ".o")
    subprocess.run([b._hipcc()] + b._flags() + extra + ["-c", os.path.join(b.CSRC, s), "-o", o], check=True, cwd=b.CSRC)
    return o
with concurrent.futures.ThreadPoolExecutor(8) as ex:
    objs = list(ex.map(cc, b.SOURCES))
lib = os.path.join(out, "libpba.so")
subprocess.run([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc", "-o", lib] + objs + ["-lpthread"], check=True)
for o in objs:
    os.remove(o)
print(lib)
