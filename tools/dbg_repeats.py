#!/usr/bin/env python3
"""Tandem repeats at scale: targets whose candidate lists outgrow one LDS sort (row-sweep form) and whose survivor lists
hold queries with hundreds of candidates (bit-vector form, both ways of sizing the slices) against the row-sweep kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pacbioassembly_amd import Context, engine as eng
from pacbioassembly_amd.engine import PBA_KERNEL_BITVEC, PBA_KERNEL_ROWSWEEP
ctx = Context(0)
rng = np.random.RandomState(5)
alpha = np.frombuffer(b"ACGT", np.uint8)
unit = alpha[rng.randint(0, 4, 23)].tobytes()
g = eng.synth_genome(91, 6000).tobytes()
genome = np.frombuffer(g[:2000] + unit * 90 + g[2000:], np.uint8)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 160
reads, offs, _ = eng.synth_reads(92, genome, n, 2600, 0.02, 0.02, 0.02)
texts = [reads[int(offs[i]):int(offs[i + 1])].tobytes() for i in range(n)]
S = ctx.seqs_from_list(texts, strict_acgt=True)
mask = eng.mask_from_pattern("111*11*11*1*1111")
os.environ.pop("PBA_OVL_ROOM", None)
want, wst = ctx.overlap_all(S, mask, 0.30, 32, 64, kernel=PBA_KERNEL_ROWSWEEP)
print("rowsweep:", len(want), "overlaps", wst["n_pairs"], "pairs", wst["n_candidates"], "candidates, big targets", wst["n_big_targets"], flush=True)
for mode in ("0", "64"):                                            # census + exact slices; equal room that overflows and is redone
    os.environ["PBA_OVL_ROOM"] = mode
    got, st = ctx.overlap_all(S, mask, 0.30, 32, 64, kernel=PBA_KERNEL_BITVEC)
    ok = got.size == want.size and (got == want).all() and st["n_pairs"] == wst["n_pairs"] and st["n_candidates"] == wst["n_candidates"]
    print("room", mode, "same" if ok else "DIFFERENT", "prefiltered", st["n_prefiltered"], "big", st["n_big_targets"], "redo", st["n_redo"], flush=True)
    if not ok:
        sys.exit(1)
