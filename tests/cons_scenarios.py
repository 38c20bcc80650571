"""Consensus scenarios shared by the golden generator, the oracle tests and the GPU tests: an unlocked reference
(a slice of a synthetic genome, possibly mutated), reads that overlap it and run off both ends, two voting rounds.
The tries of a round are anchored like spaced_seed.cpp:262-298 does it (an exact 16-mer of the read found in the
CURRENT reference text, forward from the read's head, backward from its tail), so every implementation derives the
same calls as long as its reference text is the same -- and the test says so when it is not."""
import hashlib

import numpy as np

from pacbioassembly_amd import engine as eng

SCENARIOS = [
    # name, genome seed, reads seed, genome length, slice start, slice length, reads, read length, weight, (ins, del, sub), mutate ref
    ("balanced", 21, 22, 9000, 3000, 3000, 60, 1200, 1, (0.05, 0.05, 0.05), False),
    ("pacbio_w3", 23, 24, 9000, 2500, 3500, 70, 1500, 3, (0.09, 0.045, 0.015), False),
    ("noisy_ref", 25, 26, 8000, 2000, 3000, 80, 1000, 1, (0.03, 0.03, 0.03), True),
]


def scenario_inputs(sc):
    name, gs, rs, glen, p0, plen, nreads, rlen, weight, err, mutate = sc
    g = eng.synth_genome(gs, glen)
    reads, offs, starts = eng.synth_reads(rs, g, nreads, rlen, *err)
    text = g[p0:p0 + plen].tobytes()
    if mutate:                      # a reference that itself carries errors: votes must repair it
        rng = np.random.RandomState(gs)
        t = bytearray(text)
        for k in rng.choice(len(t), len(t) // 25, replace=False):
            t[k] = b"ACGT"[(b"ACGT".index(t[k]) + 1 + rng.randint(3)) % 4]
        text = bytes(t)
    rd = [reads[int(offs[r]):int(offs[r + 1])].tobytes() for r in range(nreads)]
    return text, weight, rd


def round_tries(text: bytes, reads, rnd: int):
    """(pos, read id, seg bytes in memory order, fwd) for one round; round 0 takes even reads, round 1 odd ones."""
    tries = []
    for r in range(rnd, len(reads), 2):
        rd = reads[r]
        for j in range(0, 30):
            hit = text.find(rd[j:j + 16])
            if hit >= 0:
                tries.append((hit, r, rd[j:], True))
                break
        for j in range(0, 30):
            p = len(rd) - j - 16
            hit = text.find(rd[p:p + 16])
            if hit >= 0:
                tries.append((hit + 15, r, rd[:p + 16], False))
                break
    return tries


def votes_digest(sel, sup, tot) -> str:
    return hashlib.sha256(np.ascontiguousarray(sel, "<u2").tobytes() + np.ascontiguousarray(sup, "<u2").tobytes() +
                          np.ascontiguousarray(tot, "<i4").tobytes()).hexdigest()[:32]


def run_scenario(cons, reads, R=0.3):
    """Drive any consensus object with try_align / dump / text / evolve; returns the record the goldens hold."""
    rec = {"rounds": []}
    for rnd in range(2):
        e0 = cons.dump()[3]
        base = cons.text()
        tries = round_tries(base, reads, rnd)
        rows = []
        for hit, r, seg, fwd in tries:
            # positions are relative to beg; the text starts at pre
            out = cons.try_align(hit + e0[0], seg, fwd, R)
            rows.append([hit + e0[0], r, len(seg), int(fwd)] + [out[k] for k in ("ok", "matlen_b", "cost", "matlen_a", "nedit", "pre", "post")])
        sel, sup, tot, ext = cons.dump()
        before = {"extent": ext, "votes": votes_digest(sel, sup, tot), "text_sha": hashlib.sha256(cons.text()).hexdigest()[:32]}
        cons.evolve()
        sel, sup, tot, ext = cons.dump()
        after = {"extent": ext, "votes": votes_digest(sel, sup, tot), "text": cons.text().decode()}
        rec["rounds"].append({"tries": rows, "before_evolve": before, "after_evolve": after})
    return rec
