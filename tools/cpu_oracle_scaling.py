"""How the CPU oracle scales with threads on this host (diagnostic for bench.py's cpu_baseline)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
os.environ["PBA_NO_TORCH"] = "1"
from pacbioassembly_amd import engine as eng
from oraclelib import Oracle
O = Oracle()
mask = eng.mask_from_pattern("111*11*11*1*1111")
g = eng.synth_genome(2, 5_000_000)
reads, offs, _ = eng.synth_reads(3, g, 64, 15000, nthreads=8)
print("affinity", len(os.sched_getaffinity(0)), "cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "-", flush=True)
try:
    print("thp", open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip(), flush=True)
except Exception as e:
    print("thp ?", e)
for nt, n in ((1, 3), (16, 64), (16, 64)):
    t = time.time()
    rows, st = O.locator(g, mask, 0.30, reads[:n * 15000], offs[:n + 1], 50, 500, nthreads=nt)
    dt = time.time() - t
    print(f"threads {nt:3d} reads {n:3d}: {dt:6.2f} s  {st['n_cells'] / dt / 1e9:.3f} GCUPS  {st['n_cells'] / dt / 1e9 / nt:.3f} /thread  located {st['n_located']}", flush=True)
