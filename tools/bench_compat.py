"""What the drop-in user gets: the reference's UNMODIFIED spaced_seed.cpp, compiled against include/compat/ and linked with
libpba.so (oracle/_ref/spaced_seed_compat, built by oracle/Makefile where /root/reference lies), on one locked round over
15 kb reads -- next to the same round through the batch entry point (pba_spaced_round: every read of the round in one
launch) and, on a prefix of the reads, the stock CPU build of the same source (oracle/_ref/spaced_seed).

The serial API hands the GPU one pair at a time (ref_seq::try_align -> seq_aligner::align), so what is measured is the
latency of one pair on one wavefront; the batch call keeps 8 192 wavefronts busy.  Both are reported per candidate pair.

    python tools/bench_compat.py --reads 200 --cpu-reads 12
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pacbioassembly_amd import engine as eng  # noqa: E402

PATTERN = "111*11*11*1*1111"
FOUND = re.compile(r"found (\d+) at cost (\d+):\tref_ml=(\d+),\tseg_ml=(\d+)")


def run_main(exe, workdir, bin_name, trials, R):
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "pacbioassembly_amd", "lib") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    t0 = time.time()
    r = subprocess.run([exe, "-f", "ref.txt", "-l", "-m", "1", "-t", str(trials), "-r", str(R), bin_name, "seed.txt"], cwd=workdir,
                       capture_output=True, env=env, timeout=3000)
    dt = time.time() - t0
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    err = r.stderr.decode()
    found = [[int(x) for x in m] for m in FOUND.findall(err)]
    ntrials = int(re.search(r"#trials: (\d+)", err).group(1))
    return dt, found, ntrials


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome", type=int, default=38000)          # head + tail of get_seedmap cover all of it (ref_seq.h:291-311)
    ap.add_argument("--reads", type=int, default=200)
    ap.add_argument("--read-len", type=int, default=15000)
    ap.add_argument("--err", type=float, default=0.15)
    ap.add_argument("--trials", type=int, default=32)
    ap.add_argument("--R", type=float, default=0.30)
    ap.add_argument("--cpu-reads", type=int, default=12, help="prefix of the reads the stock CPU build is timed on (0: skip)")
    ap.add_argument("--seed", type=int, default=17)
    a = ap.parse_args()

    g = eng.synth_genome(a.seed, a.genome)
    e = a.err / 3
    reads, offs, _ = eng.synth_reads(a.seed + 1, g, a.reads, a.read_len, e, e, e)
    texts = [reads[int(offs[i]):int(offs[i + 1])].tobytes() for i in range(a.reads)]
    file = b"".join(eng.text2bin(t) for t in texts)
    mask = eng.mask_from_pattern(PATTERN)
    out = {"workload": f"{a.reads} x {a.read_len} b reads @{a.err:.0%} against a {a.genome} b locked reference, one round, "
                       f"{a.trials} trials, R={a.R}", "pattern": PATTERN}

    # ---- the batch entry point: the whole round in one call (buggy seed_at like the reference's main)
    ctx = eng.Context(0)
    Rd = ctx.seqs_from_records(file, 500, 20000)
    T = ctx.seqs_from_list([g.tobytes()], strict_acgt=True)
    ix = ctx.index_build(T, 0, mask, eng.PBA_INDEX_HEAD_TAIL)
    rows = ctx.spaced_round(ix, T, 0, Rd, a.R, a.trials, 64, buggy_seed_at=True)      # warm-up (buffers, code objects)
    ctx.sync()
    t0 = time.time()
    reps = 5
    for _ in range(reps):
        rows = ctx.spaced_round(ix, T, 0, Rd, a.R, a.trials, 64, buggy_seed_at=True)
    ctx.sync()
    batch_s = (time.time() - t0) / reps
    n_pairs = int(rows["n_pairs"].sum())
    n_found = int(rows["found"].sum())
    out["batch"] = {"seconds": round(batch_s, 5), "pairs": n_pairs, "found": n_found, "us_per_pair": round(batch_s / max(n_pairs, 1) * 1e6, 2)}
    # the same round with ONE read per call: the batch entry point's own latency for a read
    one = []
    for r in np.flatnonzero(rows["found"])[:8]:
        R1 = ctx.seqs_from_list([texts[int(r)]], strict_acgt=True)
        ctx.spaced_round(ix, T, 0, R1, a.R, a.trials, 64, buggy_seed_at=True)
        t0 = time.time()
        rr = ctx.spaced_round(ix, T, 0, R1, a.R, a.trials, 64, buggy_seed_at=True)
        one.append((time.time() - t0, int(rr["n_pairs"][0])))
        R1.close()
    out["batch_one_read_per_call"] = {"ms_per_read": round(1e3 * float(np.mean([x[0] for x in one])), 3),
                                      "pairs_per_read": round(float(np.mean([x[1] for x in one])), 1)}

    # ---- one pair per call through the C ABI (what seq_aligner::align costs): a true 15 kb pair, a false one
    hit = int(np.flatnonzero((rows["found"] != 0) & (rows["dir"] > 0))[0])
    seg = texts[hit][int(rows["j"][hit]):]
    tgt = g.tobytes()[int(rows["ref_pos"][hit]):]
    lat = {}
    for name, bb in (("true_pair", tgt), ("false_pair", g.tobytes()[::-1][:len(tgt)])):
        ctx.align_text_trace(bb, seg, a.R, maxn=26000, maxm=6000)
        t0 = time.time()
        n = 20 if name == "true_pair" else 200
        for _ in range(n):
            res, ops = ctx.align_text_trace(bb, seg, a.R, maxn=26000, maxm=6000)      # a = the reference side (ref_seq.h:264)
        lat[name] = {"ms_per_call": round((time.time() - t0) / n * 1e3, 4), "rc": int(res["rc"]), "nedit": int(ops.size)}
    out["align_text_trace_latency"] = lat
    ctx.close()

    # ---- the reference's own main through compat, and its stock CPU build on a prefix
    with tempfile.TemporaryDirectory() as wd:
        open(os.path.join(wd, "seqs.bin"), "wb").write(file)
        open(os.path.join(wd, "ref.txt"), "wb").write(g.tobytes())
        open(os.path.join(wd, "seed.txt"), "w").write(PATTERN + "\n")
        exe = os.path.join(ROOT, "oracle", "_ref", "spaced_seed_compat")
        dt, found, ntr = run_main(exe, wd, "seqs.bin", a.trials, a.R)
        # what a run costs before its first read: process start, HIP initialisation, the code objects, get_seedmap -- the same
        # main on one foreign read that no seed places
        fg = eng.synth_genome(a.seed + 7, 4000)
        fr, fo, _ = eng.synth_reads(a.seed + 8, fg, 1, 2000)
        open(os.path.join(wd, "none.bin"), "wb").write(eng.text2bin(fr[int(fo[0]):int(fo[1])].tobytes()))
        dt0, f0, _ = run_main(exe, wd, "none.bin", a.trials, a.R)
        assert not f0
        want = [[int(r), int(rows["cost"][r]), int(rows["matlen_a"][r]), int(rows["matlen_b"][r])] for r in np.flatnonzero(rows["found"])]
        out["compat_main"] = {"seconds": round(dt, 3), "found": len(found), "trials": ntr, "pairs": n_pairs,
                              "startup_seconds": round(dt0, 3),
                              "us_per_pair": round(dt / max(n_pairs, 1) * 1e6, 1), "ms_per_found_read": round(dt / max(len(found), 1) * 1e3, 3),
                              "ms_per_found_read_after_startup": round((dt - dt0) / max(len(found), 1) * 1e3, 3),
                              "same_found_lines_as_batch": found == want}
        if a.cpu_reads:
            k = min(a.cpu_reads, a.reads)
            open(os.path.join(wd, "head.bin"), "wb").write(b"".join(eng.text2bin(t) for t in texts[:k]))
            cpu = os.path.join(ROOT, "oracle", "_ref", "spaced_seed")
            dt_c, found_c, ntr_c = run_main(cpu, wd, "head.bin", a.trials, a.R)
            pairs_c = int(rows["n_pairs"][:k].sum())
            out["stock_cpu_main"] = {"reads": k, "seconds": round(dt_c, 2), "found": len(found_c), "pairs": pairs_c,
                                     "us_per_pair": round(dt_c / max(pairs_c, 1) * 1e6, 1),
                                     "ms_per_found_read": round(dt_c / max(len(found_c), 1) * 1e3, 1),
                                     "same_found_lines_as_batch": found_c == [w for w in want if w[0] < k]}
    out["compat_vs_batch_per_pair"] = round(out["compat_main"]["us_per_pair"] / out["batch"]["us_per_pair"], 1)
    out["compat_vs_one_read_batch_call"] = round(out["compat_main"]["ms_per_found_read_after_startup"] / out["batch_one_read_per_call"]["ms_per_read"], 2)
    if "stock_cpu_main" in out:
        out["stock_cpu_vs_compat_per_found_read"] = round(out["stock_cpu_main"]["ms_per_found_read"] / out["compat_main"]["ms_per_found_read_after_startup"], 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
