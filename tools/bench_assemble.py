"""Unlocked multi-round assembly (spaced_seed.cpp:409-452 without -l) on one GPU: a slice of a synthetic genome grows
over noisy reads through pba_cons_round / pba_cons_evolve.  Prints per-round timings and, with --check-rounds K, runs the
CPU oracle beside the first K rounds and compares rows, vote boxes and evolved text.

    python tools/bench_assemble.py --genome 300000 --reads 400 --read-len 15000 --rounds 12 --check-rounds 1
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from pacbioassembly_amd import engine as eng  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome", type=int, default=300000)
    ap.add_argument("--reads", type=int, default=400)
    ap.add_argument("--read-len", type=int, default=15000)
    ap.add_argument("--err", type=float, default=0.15)
    ap.add_argument("--start-len", type=int, default=30000)
    ap.add_argument("--rounds", type=int, default=12)
    ap.add_argument("--trials", type=int, default=32)
    ap.add_argument("--R", type=float, default=0.30)
    ap.add_argument("--check-rounds", type=int, default=0)
    ap.add_argument("--seed", type=int, default=5)
    a = ap.parse_args()

    g = eng.synth_genome(a.seed, a.genome)
    e = a.err / 3
    reads, offs, _ = eng.synth_reads(a.seed + 1, g, a.reads, a.read_len, e, e, e)
    texts = [reads[int(offs[i]):int(offs[i + 1])].tobytes() for i in range(a.reads)]
    order = np.random.RandomState(a.seed).permutation(a.reads)
    texts = [texts[i] for i in order]
    file = b"".join(eng.text2bin(t) for t in texts)
    rec_offs = np.cumsum([0] + [4 + (len(t) + 3) // 4 for t in texts[:-1]]).astype(np.uint64)
    p0 = (a.genome - a.start_len) // 2
    text = g[p0:p0 + a.start_len].tobytes()
    masks = [eng.mask_from_pattern(l.strip()) for l in open(os.path.join(ROOT, "tests", "golden", "seeds.txt")) if l.strip()]
    picks = np.random.RandomState(a.seed + 2).randint(0, 1 << 30, 64)

    ctx = eng.Context(0)
    Rd = ctx.seqs_from_records(file, 0, 1 << 30)
    cons = eng.Consensus(ctx, text, 1, max_len=800000)
    oc = None
    if a.check_rounds:
        from oraclelib import Oracle
        oc = Oracle().consensus(text, 1, max_len=800000)
    pool = list(range(a.reads))
    nfailure, draws, total = 0, 0, 0.0
    out = []
    for rnd in range(1, a.rounds + 1):
        mask = masks[picks[draws] % len(masks)] if nfailure == 0 else masks[nfailure - 1]
        draws += nfailure == 0
        t0 = time.time()
        rows, st = cons.round(Rd, pool, int(mask), a.R, a.trials, 64, buggy_seed_at=True)
        t1 = time.time()
        same = None
        if oc is not None and rnd <= a.check_rounds:
            c0 = time.time()
            want, nm = oc.round(int(mask), a.R, a.trials, file, rec_offs, pool, buggy=True)
            cpu_s = time.time() - c0
            same = all((rows[c] == want[c]).all() for c in ("found", "j", "dir", "ref_pos", "cost", "matlen_a", "matlen_b", "n_trials", "n_pairs"))
            gs, os_ = cons.dump(), oc.dump()
            same = same and all((x == y).all() for x, y in zip(gs[:3], os_[:3])) and gs[3] == os_[3]
        pool = [p for k, p in enumerate(pool) if not rows["found"][k]]
        last = False
        if st["n_found"]:
            nfailure = 0
        else:
            nfailure += 1
            last = nfailure == len(masks)
        t2 = time.time()
        if not last:
            new = cons.evolve()
            if oc is not None and rnd <= a.check_rounds:
                oc.evolve()
                same = same and oc.text() == new
        t3 = time.time()
        total += (t1 - t0) + (t3 - t2)
        rec = dict(round=rnd, mask=hex(int(mask)), tried=len(pool) + st["n_found"], found=st["n_found"], batches=st["n_batches"],
                   grown=[st["n_grown_bwd"], st["n_grown_fwd"]], deferred=st["n_deferred"], index=st["n_index"],
                   ref_len=len(cons.text()), round_ms=round((t1 - t0) * 1e3, 1), evolve_ms=round((t3 - t2) * 1e3, 1))
        if same is not None:
            rec["same_as_oracle"] = bool(same); rec["oracle_s"] = round(cpu_s, 1)
        print(json.dumps(rec), flush=True)
        out.append(rec)
        if last:
            break
    final = cons.text()
    # how good is the assembly: exact 32-mers of the genome it holds
    gb = g.tobytes()
    kmers = {gb[i:i + 32] for i in range(0, len(gb) - 32)}
    hit = sum(1 for i in range(0, len(final) - 32, 7) if final[i:i + 32] in kmers)
    print(json.dumps(dict(rounds=len(out), found=sum(r["found"] for r in out), of=a.reads, final_len=len(final), gpu_s=round(total, 3),
                          genome_32mers_in_sample=round(hit / max(1, len(range(0, len(final) - 32, 7))), 3),
                          checked_same=all(r.get("same_as_oracle", True) for r in out))))


if __name__ == "__main__":
    main()
