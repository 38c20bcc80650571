"""Debug aid for sets beyond 4 GiB: build N 15 kb reads shard by shard into one device buffer (as tools/rehearse_config4.py
does), check texts and planes of reads across the set, run three target ranges and print overlaps / listed / candidates per
target (20 x coverage: 5 overlaps per target at any N).  usage: NT=2000 python -u tools/dbg_bigset.py 9300000"""
import sys, os, numpy as np, torch, time
sys.path.insert(0, "/root/repo")
from pacbioassembly_amd import Context, ProbeTable, engine as eng
n = int(sys.argv[1]); rl = 15000; pk = ((rl+3)//4+15)&~15
L = n*rl//20
ctx = Context(0)
g = eng.synth_genome(2, L)
big = torch.empty(n*pk, dtype=torch.uint8, device="cuda")
sr = 250000
keep = {}
for lo in range(0, n, sr):
    hi = min(n, lo+sr)
    text, offs = eng.synth_reads_range(3, g, lo, hi, rl, nthreads=16)
    for i in (lo, hi-1):
        keep[i] = text[(i-lo)*rl:(i-lo+1)*rl].tobytes()
    sh = ctx.seqs_from_text(text, offs, strict_acgt=True)
    sh.export(big[lo*pk:].data_ptr(), (hi-lo)*pk)
    sh.close()
S = ctx.seqs_from_device_packed(big.data_ptr(), big.numel(), np.arange(n, dtype=np.uint64)*np.uint64(pk), np.full(n, rl, np.uint32))
del big
bad = [i for i, t in keep.items() if S.get_text(i) != t]
print("reads checked", len(keep), "bad", bad[:10], flush=True)
# planes: a read against itself through the bit-vector kernel (which reads the bit planes only)
from pacbioassembly_amd.engine import PAIR_DTYPE, PBA_KERNEL_BITVEC
ids = [0, 1, n // 3, n // 2, int(n * 0.9), int(n * 0.93), n - 2, n - 1]
pairs = np.array([(i, 0, 2000, i, 0, 2000, 0) for i in ids], PAIR_DTYPE)
res = ctx.align_batch(S, S, pairs, 0.3, kernel=PBA_KERNEL_BITVEC)
print("self alignments (planes on both sides -- only says the kernel runs; tests/test_gpu_parity.py: test_packed_set_beyond_4_gib checks them against planes built independently):", [(i, int(r["rc"]), int(r["cost"])) for i, r in zip(ids, res)], flush=True)
mask = eng.mask_from_pattern("111*11*11*1*1111")
slots = n*64
probes = torch.full((slots,), -1, dtype=torch.int64, device="cuda")
ctx.overlap_probes(S, 0, n, mask, 32, probes.data_ptr(), slots)
torch.cuda.synchronize()
table = ProbeTable(ctx, probes.data_ptr(), probes.numel(), mask, 32)
NT = int(os.environ.get('NT', '25000'))
for lo in (0, n//2, n-NT):
    ov, st = ctx.overlap_all_table(S, table, 0.3, 64, lo, lo+NT, cap=NT*400)
    print(lo, "overlaps/target", st["n_overlaps"]/NT, "listed/target", st["n_listed"]/NT, "cand/target", st["n_candidates"]/NT, "sort_ms", st["sort_ms"], "scan_ms", st["scan_ms"], flush=True)
    q = ov["query"]
    print("   query id quantiles", np.quantile(q, [0, .25, .5, .75, 1]).astype(int).tolist(), flush=True)
