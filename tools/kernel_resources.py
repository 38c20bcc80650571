#!/usr/bin/env python3
"""Register / LDS / spill / occupancy figures of every kernel of one translation unit, as the compiler reports them
(-Rpass-analysis=kernel-resource-usage).  usage: tools/kernel_resources.py pba_drivers.hip [filter]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pacbioassembly_amd import build as b
src = os.path.join(b.CSRC, sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
p = subprocess.run([b._hipcc()] + b._flags() + ["-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"],
                   capture_output=True, text=True, cwd=b.CSRC)
cur = None
rows = {}
for ln in p.stderr.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", ln)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = subprocess.run(["c++filt", t.split(":", 1)[1].strip()], capture_output=True, text=True).stdout.strip().split("(")[0]
        rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1)
        rows[cur][k.strip()] = v.strip()
print(f"{'kernel':44s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'spillV':>6s} {'spillS':>6s} {'scratch':>8s} {'occ':>4s} {'LDS':>7s}")
for k, r in rows.items():
    if flt in k:
        print(f"{k[:44]:44s} {r.get('VGPRs','?'):>5s} {r.get('AGPRs','?'):>5s} {r.get('TotalSGPRs','?'):>5s} {r.get('VGPR Spill','?'):>6s} "
              f"{r.get('SGPR Spill','?'):>6s} {r.get('ScratchSize [bytes/lane]','?'):>8s} {r.get('Occupancy [waves/SIMD]','?'):>4s} {r.get('LDS Size [bytes/block]','?'):>7s}")
