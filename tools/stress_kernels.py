#!/usr/bin/env python3
"""Differential stress of the two independent aligning kernels (full-band row sweep vs bit-vector array with its
prefilter, asymmetric windows and certificates) on pairs built to push the optimal path towards the window edges:
indel-biased errors (the path drifts off the diagonal), error rates up to the acceptance limit, tails on either side.
Prints one line per batch; exits 1 on the first disagreement."""
import argparse, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pacbioassembly_amd import Context, engine as eng
from pacbioassembly_amd.engine import PAIR_DTYPE, PBA_KERNEL_BITVEC, PBA_KERNEL_ROWSWEEP

ap = argparse.ArgumentParser()
ap.add_argument("--batches", type=int, default=6)
ap.add_argument("--pairs", type=int, default=300)
ap.add_argument("--max-len", type=int, default=9000)
ap.add_argument("--seed", type=int, default=1)
a = ap.parse_args()
ctx = Context(0)
rng = np.random.RandomState(a.seed)
alpha = np.frombuffer(b"ACGT", np.uint8)


def mutate(x, ins, dele, sub):
    u = rng.rand(x.size)
    keep = u >= dele
    subm = (u >= dele) & (u < dele + sub)
    y = x.copy()
    y[subm] = alpha[(np.searchsorted(alpha, y[subm]) + 1 + rng.randint(0, 3, subm.sum())) % 4]
    out = []
    insm = rng.rand(x.size) < ins
    for k in np.flatnonzero(keep):
        if insm[k]:
            out.append(alpha[rng.randint(4)])
        out.append(y[k])
    return np.array(out, np.uint8)


bad = 0
for b in range(a.batches):
    R = float(rng.choice([0.15, 0.2, 0.3, 0.3, 0.4]))
    seqs, pairs = [], []
    for q in range(a.pairs):
        m = int(rng.randint(40, a.max_len))
        x = alpha[rng.randint(0, 4, m)]
        tot = rng.uniform(0.0, R * 0.85)
        mix = rng.dirichlet([0.6, 0.6, 0.6])            # often lopsided: mostly insertions or mostly deletions
        y = mutate(x, *(tot * mix))
        tail = int(rng.choice([0, 0, 30, 400, 3000]))
        y = np.concatenate([y, alpha[rng.randint(0, 4, tail)]])
        if rng.rand() < 0.5:
            x, y = y, x
        fl = int(rng.randint(0, 4))
        xa, ya = x.tobytes(), y.tobytes()
        i = len(seqs); seqs += [xa, ya]
        pairs.append((i, len(xa) - 1 if fl & 1 else 0, len(xa), i + 1, len(ya) - 1 if fl & 2 else 0, len(ya), fl))
    S = ctx.seqs_from_list(seqs, strict_acgt=True)
    arr = np.array(pairs, PAIR_DTYPE)
    r0 = ctx.align_batch(S, S, arr, R, kernel=PBA_KERNEL_ROWSWEEP)
    r1 = ctx.align_batch(S, S, arr, R, kernel=PBA_KERNEL_BITVEC)
    prof = ctx.last_profile()
    same = all((r0[c] == r1[c]).all() for c in ("rc", "cost", "matlen_a", "matlen_b", "len_a", "len_b", "max_dst"))
    print(f"batch {b}: R={R} pairs={a.pairs} ok={int((r0['rc'] >= 0).sum())} redo={prof['n_redo']} same={same}", flush=True)
    if not same:
        d = np.flatnonzero((r0["rc"] != r1["rc"]) | (r0["cost"] != r1["cost"]) | (r0["matlen_b"] != r1["matlen_b"]))
        for q in d[:5]:
            print("  pair", q, pairs[q][2], pairs[q][5], "rowsweep", r0[q], "bitvec", r1[q])
        bad = 1
        break
sys.exit(bad)
