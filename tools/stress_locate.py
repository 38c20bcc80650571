#!/usr/bin/env python3
"""Randomised differential test of the locate driver (probe order, prefilter, narrow / certifying windows, first success,
counted pairs and cells) against the CPU oracle: random genome sizes, read lengths (ragged, some below the 500-base
cut), error mixes from clean to beyond R, R from 0.1 to 0.45, both kernels.  Exits 1 on the first disagreement."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from oraclelib import Oracle
from pacbioassembly_amd import Context, engine as eng
from pacbioassembly_amd.engine import PBA_INDEX_ALL, PBA_KERNEL_BITVEC, PBA_KERNEL_ROWSWEEP

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=8)
ap.add_argument("--seed", type=int, default=1)
a = ap.parse_args()
ctx, orc = Context(0), Oracle()
rng = np.random.RandomState(a.seed)
masks = [0xFF3C3FFC, 0xFFCCF3FC, 0x3FCFC3FC, 0xFFF0CCFC]
bad = 0
for rnd in range(a.rounds):
    glen = int(rng.randint(20000, 90000))
    n = int(rng.randint(120, 360))
    rl = int(rng.choice([700, 1500, 3000, 6000]))
    R = float(rng.choice([0.1, 0.15, 0.2, 0.3, 0.3, 0.45]))
    tot = rng.uniform(0.0, R * 0.9)
    mix = rng.dirichlet([1.0, 1.0, 1.0])
    e = tuple(float(x) for x in tot * mix)
    g = eng.synth_genome(1000 + a.seed * 100 + rnd, glen)
    reads, offs, _ = eng.synth_reads(2000 + a.seed * 100 + rnd, g, n, rl, *e)
    texts = [reads[int(offs[i]):int(offs[i + 1])].tobytes() for i in range(n)]
    for i in range(0, n, 9):                         # ragged: cut some reads, a few below the 500-base cut
        texts[i] = texts[i][:int(rng.randint(100, len(texts[i]) + 1))]
    reads = np.frombuffer(b"".join(texts), np.uint8)
    offs = np.cumsum([0] + [len(t) for t in texts]).astype(np.uint64)
    mask = masks[rnd % len(masks)]
    trials = int(rng.choice([10, 50]))
    want, wst = orc.locator(g, mask, R, reads, offs, trials, 500, nthreads=8)
    T = ctx.seqs_from_list([g.tobytes()], strict_acgt=True)
    Rd = ctx.seqs_from_text(reads, offs, strict_acgt=True)
    ix = ctx.index_build(T, 0, mask, PBA_INDEX_ALL)
    ok = True
    for kernel in (PBA_KERNEL_ROWSWEEP, PBA_KERNEL_BITVEC):
        rows, st = ctx.locate(ix, T, 0, Rd, R, trials, 500, kernel=kernel)
        same = all((rows[c] == want[c]).all() for c in ("nseq", "found", "j", "pos", "cost", "seglen", "matlen_a", "matlen_b", "n_pairs")) and st == wst
        ok = ok and same
        if not same:
            d = [c for c in ("nseq", "found", "j", "pos", "cost", "seglen", "matlen_a", "matlen_b", "n_pairs") if not (rows[c] == want[c]).all()]
            print("  kernel", kernel, "differs in", d, st, wst)
    print(f"round {rnd}: genome {glen} reads {n}x{rl} err {tuple(round(x, 3) for x in e)} R={R} trials={trials} "
          f"located {int(want['found'].sum())} pairs {wst['n_pairs']} redo {ctx.last_profile()['n_redo']} same={ok}", flush=True)
    if not ok:
        bad = 1
        break
sys.exit(bad)
