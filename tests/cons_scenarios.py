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


# ---------------------------------------------------------------------------------------------------------------
# spaced_seed's main loop for a locked reference (spaced_seed.cpp:409-452), composed from single locked rounds
MULTI = dict(genome_seed=31, genome_len=30000, reads_seed=32, n_reads=160, read_len=1200, foreign_seed=33, n_foreign=24,
             R=0.30, max_trial=3, overlap_min=64, max_round=40,
             picks=[5, 2, 7, 7, 0, 3, 6, 1, 4, 2, 2, 5, 0, 7, 3, 1, 6, 4, 5, 5, 3, 0, 2, 6, 1, 7, 4])


def multi_inputs():
    """Reference text, the binary read file (reads of the reference plus reads of a foreign genome that never align),
    record offsets."""
    m = MULTI
    g = eng.synth_genome(m["genome_seed"], m["genome_len"])
    reads, offs, _ = eng.synth_reads(m["reads_seed"], g, m["n_reads"], m["read_len"])
    f = eng.synth_genome(m["foreign_seed"], 20000)
    fr, fo, _ = eng.synth_reads(m["foreign_seed"] + 1, f, m["n_foreign"], m["read_len"])
    texts = [reads[int(offs[i]):int(offs[i + 1])].tobytes() for i in range(m["n_reads"])]
    texts += [fr[int(fo[i]):int(fo[i + 1])].tobytes() for i in range(m["n_foreign"])]
    order = np.random.RandomState(7).permutation(len(texts))
    texts = [texts[i] for i in order]
    file = b"".join(eng.text2bin(t) for t in texts)
    rec_offs = np.cumsum([0] + [4 + (len(t) + 3) // 4 for t in texts[:-1]]).astype(np.uint64)
    return g.tobytes(), file, rec_offs


def multi_rounds(round_fn, masks, n_reads):
    """The loop of spaced_seed.cpp:409-452 over round_fn(mask, pool) -> rows (one per pool entry, `found` column).
    Returns (found_round per read, log [[round, mask, tried, found], ...], rows of the finding round per read)."""
    m = MULTI
    pool = list(range(n_reads))
    found_round = [0] * n_reads
    final = [None] * n_reads
    log, nfailure, draws = [], 0, 0
    for nround in range(1, m["max_round"] + 1):
        if nfailure == 0:
            mask = masks[m["picks"][draws % len(m["picks"])] % len(masks)]; draws += 1
        else:
            mask = masks[nfailure - 1]
        rows = round_fn(mask, pool)
        nm, rest = 0, []
        for k, r in enumerate(pool):
            final[r] = [int(rows[c][k]) for c in ("found", "j", "dir", "ref_pos", "cost", "matlen_a", "matlen_b")]
            if rows["found"][k]:
                found_round[r] = nround; nm += 1
            else:
                rest.append(r)
        log.append([nround, int(mask), len(pool), nm])
        pool = rest
        if nm:
            nfailure = 0
        else:
            nfailure += 1
            if nfailure == len(masks):
                break
    return found_round, log, final


# ---------------------------------------------------------------------------------------------------------------
# the stock `locator` command line (locator.cpp): contig file + seed pattern + reads on stdin, R = 0.15
LOCATOR_CLI = dict(genome_seed=41, genome_len=100000, reads_seed=42, n_reads=400, read_len=2000, err=(0.02, 0.02, 0.02),
                   pattern="111*11*11*1*1111")


def locator_cli_inputs():
    """(contig text, list of read texts): every 7th read is cut below 500 bases (skipped without an id)."""
    m = LOCATOR_CLI
    g = eng.synth_genome(m["genome_seed"], m["genome_len"])
    reads, offs, _ = eng.synth_reads(m["reads_seed"], g, m["n_reads"], m["read_len"], *m["err"])
    texts = [reads[int(offs[i]):int(offs[i + 1])].tobytes() for i in range(m["n_reads"])]
    texts = [t[:300 + 10 * (i % 13)] if i % 7 == 3 else t for i, t in enumerate(texts)]
    return g.tobytes(), texts


# ---------------------------------------------------------------------------------------------------------------
# spaced_seed's main loop WITHOUT -l (spaced_seed.cpp:409-452): unlocked rounds -- every success votes and may grow the
# reference, which the reads after it see -- and evolve after each round.  A short slice of a genome grows over its reads.
ASSEMBLE = dict(genome_seed=61, genome_len=14000, slice=(5500, 3000), reads_seed=62, n_reads=220, read_len=1200,
                err=(0.04, 0.04, 0.04), foreign_seed=63, n_foreign=20, weight=2, R=0.30, max_trial=32, overlap_min=64,
                max_round=10, picks=[5, 2, 7, 7, 0, 3, 6, 1, 4, 2, 2, 5, 0, 7, 3, 1, 6, 4, 5, 5, 3, 0, 2, 6, 1, 7, 4])


def assemble_inputs(cfg=None):
    """Start reference, weight, the binary read file (reads of the genome in random order plus foreign ones), record
    offsets, read texts."""
    m = cfg or ASSEMBLE
    g = eng.synth_genome(m["genome_seed"], m["genome_len"])
    reads, offs, _ = eng.synth_reads(m["reads_seed"], g, m["n_reads"], m["read_len"], *m["err"])
    f = eng.synth_genome(m["foreign_seed"], 20000)
    fr, fo, _ = eng.synth_reads(m["foreign_seed"] + 1, f, m["n_foreign"], m["read_len"])
    texts = [reads[int(offs[i]):int(offs[i + 1])].tobytes() for i in range(m["n_reads"])]
    texts += [fr[int(fo[i]):int(fo[i + 1])].tobytes() for i in range(m["n_foreign"])]
    order = np.random.RandomState(m["genome_seed"]).permutation(len(texts))
    texts = [texts[i] for i in order]
    file = b"".join(eng.text2bin(t) for t in texts)
    rec_offs = np.cumsum([0] + [4 + (len(t) + 3) // 4 for t in texts[:-1]]).astype(np.uint64)
    p0, plen = m["slice"]
    return g[p0:p0 + plen].tobytes(), m["weight"], file, rec_offs, texts


def run_assembly(cons, masks, file, rec_offs, n_reads, cfg=None):
    """spaced_seed.cpp:409-452 over any object with round(mask, R, max_trial, file, rec_offs, pool) -> (rows, nmatches),
    evolve(), dump(), text().  Returns the record the goldens hold."""
    m = cfg or ASSEMBLE
    pool = list(range(n_reads))
    rec = {"rounds": []}
    nfailure, draws = 0, 0
    for nround in range(1, m["max_round"] + 1):
        if nfailure == 0:
            mask = masks[m["picks"][draws % len(m["picks"])] % len(masks)]; draws += 1
        else:
            mask = masks[nfailure - 1]
        rows, nm = cons.round(int(mask), m["R"], m["max_trial"], file, rec_offs, pool)
        found = [[int(rows[c][k]) for c in ("read", "j", "dir", "ref_pos", "cost", "matlen_a", "matlen_b")]
                 for k in range(len(pool)) if rows["found"][k]]
        assert nm == len(found)
        trials = hashlib.sha256(np.ascontiguousarray(rows["n_trials"]).tobytes() + np.ascontiguousarray(rows["n_pairs"]).tobytes()).hexdigest()[:16]
        sel, sup, tot, ext = cons.dump()
        r = {"round": nround, "mask": int(mask), "n_tried": len(pool), "found": found, "trials_pairs": trials, "extent": ext,
             "votes": votes_digest(sel, sup, tot)}
        pool = [p for k, p in enumerate(pool) if not rows["found"][k]]
        last = False
        if nm:
            nfailure = 0
        else:
            nfailure += 1
            last = nfailure == len(masks)
        if not last:
            cons.evolve()
        t = cons.text()
        r["ref_len"] = len(t); r["text_sha"] = hashlib.sha256(t).hexdigest()[:32]
        rec["rounds"].append(r)
        if last:
            break
    rec["final_text"] = cons.text().decode()
    return rec


# the reference's own spaced_seed main on the ASSEMBLE reads (tests/golden/make_golden.py: gen_spaced_seed_cli)
SPACED_SEED_CLI = dict(pattern="111*11*11*1*1111", runs={"unlocked": ["-m", "5", "-t", "32"], "locked": ["-l", "-m", "3", "-t", "32"],
                                                         "unlocked_r20": ["-m", "3", "-t", "20", "-r", "0.2"]})


def run_spaced_seed_cli(exe, workdir, extra):
    """Run a spaced_seed-compatible program on the ASSEMBLE inputs (reference text in a file WITHOUT a trailing newline:
    fgets then stops at EOF and the weight stays 1; a one-line seed file: rand() % 1) and return what the goldens hold."""
    import hashlib
    import re
    import subprocess
    import os
    text, weight, file, rec_offs, texts = assemble_inputs()
    open(os.path.join(workdir, "seqs.bin"), "wb").write(file)
    open(os.path.join(workdir, "ref.txt"), "wb").write(text)
    open(os.path.join(workdir, "seed.txt"), "w").write(SPACED_SEED_CLI["pattern"] + "\n")
    out = {}
    for name, flags in SPACED_SEED_CLI["runs"].items():
        dump = os.path.join(workdir, name + ".dump")
        r = subprocess.run([exe, "-f", "ref.txt", "-d", dump] + flags + extra + ["seqs.bin", "seed.txt"], cwd=workdir, capture_output=True,
                           check=True, timeout=600)
        lines = r.stdout.split(b"\n")
        assert lines[-1] == b""
        found = [[int(x) for x in m] for m in re.findall(r"found (\d+) at cost (\d+):\tref_ml=(\d+),\tseg_ml=(\d+)", r.stderr.decode())]
        out[name] = {"consensus_len": [len(l) for l in lines[:-1]],
                     "consensus_sha": [hashlib.sha256(l).hexdigest()[:32] for l in lines[:-1]],
                     "last_consensus": lines[-2].decode(), "found": found,
                     "dump_sha": hashlib.sha256(open(dump, "rb").read()).hexdigest()[:32], "dump_lines": open(dump, "rb").read().count(b"\n")}
    return out
