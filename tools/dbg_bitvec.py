"""Debug harness: a few pairs through one kernel, printed next to the oracle.  usage: dbg_bitvec.py KERNEL [N]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from pacbioassembly_amd import engine as eng, Context
from oraclelib import Oracle
kernel = int(sys.argv[1]); N = int(sys.argv[2]) if len(sys.argv) > 2 else 4
O = Oracle(); ctx = Context(0)
g = eng.synth_genome(77, 60000)
for rl in (40, 100, 1000, 2500):
    reads, offs, starts = eng.synth_reads(78, g, N, rl)
    seqs = [g.tobytes()] + [reads[int(offs[r]):int(offs[r+1])].tobytes() for r in range(N)]
    S = ctx.seqs_from_list(seqs)
    pairs = []
    for r in range(N):
        tp = int(starts[r])
        pairs.append((r+1, 0, rl, 0, tp, 60000-tp, 0))
        pairs.append((r+1, 0, rl, 0, (tp+777) % 50000, 5000, 0))
    arr = np.array(pairs, eng.PAIR_DTYPE)
    print("launch rl", rl, flush=True)
    t = time.time(); out = ctx.align_batch(S, S, arr, 0.3, kernel=kernel); dt = time.time() - t
    for pr, got in zip(pairs, out):
        a = seqs[pr[0]][pr[1]:pr[1]+pr[2]]; b = seqs[pr[3]][pr[4]:pr[4]+pr[5]]
        exp = O.align(a, b, 0.3)
        ok = int(got["rc"]) == exp["rc"] and (exp["rc"] < 0 or (int(got["cost"]), int(got["matlen_a"]), int(got["matlen_b"])) == (exp["cost"], exp["matlen_a"], exp["matlen_b"]))
        print("OK " if ok else "BAD", rl, [int(got[k]) for k in got.dtype.names], [exp[k] for k in ("rc","cost","matlen_a","matlen_b")], flush=True)
    print("time %.3f s" % dt, flush=True)
