import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pacbioassembly_amd import engine as eng, Context
ctx = Context(0)
g = eng.synth_genome(71, 9000)
reads, offs, _ = eng.synth_reads(72, g, 64, 1300)
S = ctx.seqs_from_text(reads, offs, strict_acgt=True)
mask = eng.mask_from_pattern("111*11*11*1*1111")
for kernel in (1, 2):
    print("kernel", kernel, flush=True)
    t = time.time(); got, st = ctx.overlap_all(S, mask, 0.30, 32, 64, kernel=kernel); print(len(got), st, "%.3f s" % (time.time() - t), flush=True)
