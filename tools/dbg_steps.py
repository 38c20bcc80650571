import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pacbioassembly_amd import Context, engine as eng
from pacbioassembly_amd.engine import PBA_INDEX_ALL
ctx = Context(0)
g = eng.synth_genome(2, 5_000_000)
reads, offs, _ = eng.synth_reads(3, g, 100000, 15000, 0.05, 0.05, 0.05, nthreads=16)
T = ctx.seqs_from_text(g, np.array([0, g.size], np.uint64), strict_acgt=True)
Rd = ctx.seqs_from_text(reads, offs, strict_acgt=True)
mask = eng.mask_from_pattern("111*11*11*1*1111")
for i in range(5):
    t0 = time.perf_counter(); ix = ctx.index_build(T, 0, mask, PBA_INDEX_ALL); t1 = time.perf_counter()
    rows, st = ctx.locate(ix, T, 0, Rd, 0.3, 50, 500); t2 = time.perf_counter()
    p = ctx.last_profile()
    ix.close(); t3 = time.perf_counter()
    print(i, "index %.2f locate %.2f close %.2f" % (1e3*(t1-t0), 1e3*(t2-t1), 1e3*(t3-t2)), {k: round(v, 2) for k, v in p.items() if isinstance(v, float)})
