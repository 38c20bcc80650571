#!/usr/bin/env python3
"""Where a kernel's scratch (spill) instructions sit, by loop depth, from the assembly hipcc -S wrote:
tools/scratch_by_depth.py /tmp/drivers.s _Z8k_locateILi2EE"""
import re, sys
from collections import Counter
lines = open(sys.argv[1]).read().split("\n")
start = [i for i, l in enumerate(lines) if l.startswith(sys.argv[2])][0]
end = [i for i, l in enumerate(lines) if i > start and ".amdhsa_kernel" in l][0]
cur, out = 0, []
for i, l in enumerate(lines[start:end]):
    if l.startswith(".LBB"):
        m = re.search(r"Depth=(\d+)", l)
        cur = int(m.group(1)) if m else 0
    if "scratch_" in l:
        out.append((i, cur, l.strip()[:90]))
print(sorted(Counter((d, "store" if "store" in t else "load") for _, d, t in out).items()))
for i, d, t in out:
    if d >= int(sys.argv[3]) if len(sys.argv) > 3 else 99:
        print(i, d, t)
