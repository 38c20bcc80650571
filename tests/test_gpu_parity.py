"""Parity tests proper: the HIP path, called through the C ABI, against (1) the golden vectors the
reference itself produced and (2) the CPU oracle on the same seeded inputs.  Bit-exact: seed-hit
sets, hit order, integer scores and coordinates.  Needs a real MI355X (-m gpu)."""
import numpy as np
import pytest

from conftest import GOLD, MASK_PAT, gold_json, gold_npz
from pacbioassembly_amd import engine as eng
from pacbioassembly_amd.engine import (PAIR_DTYPE, PBA_INDEX_ALL, PBA_INDEX_HEAD_TAIL, PBA_KERNEL_BITVEC,
                                       PBA_KERNEL_ROWSWEEP, PbaError)
from test_oracle_golden import (check_locator_rows, index_digest, index_inputs, locator_inputs, spaced_inputs)

pytestmark = pytest.mark.gpu
KERNELS = [PBA_KERNEL_ROWSWEEP, PBA_KERNEL_BITVEC]
SOAK = int(__import__("os").environ.get("PBA_SOAK_SEED", "0"))      # other seeds for the fuzz tests (soak runs); 0 = the suite's own


def c2i_text(b: bytes) -> bytes:
    return bytes(c if c in b"ACG" else ord("T") for c in b)


# ----------------------------------------------------------------------------- sequence sets
def test_seqs_pack_roundtrip(ctx):
    seqs = [b"ACGTGTCATCGGATCAACCGGTT", b"", b"A", b"ACGTN", b"acgtNNNN", b"T" * 17, b"G" * 64, b"ACGT" * 1000 + b"AC"]
    s = ctx.seqs_from_list(seqs)
    assert s.count == len(seqs) and s.max_len == 4002
    assert s.lengths().tolist() == [len(x) for x in seqs]
    for i, x in enumerate(seqs):
        assert s.get_text(i) == c2i_text(x)          # C2I: anything but A,C,G packs as 3 (dna_seq.h:21)
    with pytest.raises(PbaError) as e:
        ctx.seqs_from_list([b"ACGT", b"ACNT"], strict_acgt=True)
    assert e.value.status == -6
    ctx.seqs_from_list([b"ACGT", b"TTTT"], strict_acgt=True)


def test_seqs_from_device_text_matches_host_path(ctx):
    import torch
    g = eng.synth_genome(3, 5000)
    reads, offs, _ = eng.synth_reads(4, g, 50, 777)
    d_text = torch.from_numpy(reads.copy()).cuda()
    d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
    torch.cuda.synchronize()
    s1 = ctx.seqs_from_device_text(d_text.data_ptr(), d_offs.data_ptr(), 50, reads.size, 777)
    s2 = ctx.seqs_from_text(reads, offs)
    for i in (0, 1, 17, 49):
        assert s1.get_text(i) == s2.get_text(i) == reads[int(offs[i]):int(offs[i + 1])].tobytes()


def test_seqs_from_records(ctx):
    lines = open(f"{GOLD}/real_align.txt").read().split()
    img = b"".join(eng.text2bin(l.encode()) for l in lines)
    s = ctx.seqs_from_records(img, 500, 20000)       # spaced_seed.cpp:334 keeps 500 < len < 20000
    keep = [l for l in lines if 500 < len(l) < 20000]
    assert s.count == len(keep)
    for i, l in enumerate(keep):
        assert s.get_text(i) == l.encode()
    assert ctx.seqs_from_records(img, 0, 1 << 30).count == 12


# ----------------------------------------------------------------------------- seed index
def test_index_golden(ctx, oracle):
    g = gold_json("index.json")
    b = g["ref_test_basic"]                          # test/ref_test.cpp:119-128
    s = ctx.seqs_from_list([b["text"].encode()])
    ix = ctx.index_build(s, 0, b["mask"], PBA_INDEX_HEAD_TAIL)
    k, p = ix.dump()
    assert k.tolist() == b["keys"] and p.tolist() == b["pos"] and ix.visited == b["rv"]
    assert len(set(k.tolist())) == b["nkeys"] == len(b["text"]) - 15 - 1
    for c in g["cases"]:
        text = index_inputs(c)
        s = ctx.seqs_from_list([b"ACGT" * 5, text])  # index sequence 1 of a set, not 0
        mode = PBA_INDEX_ALL if c["mode"] == "all" else PBA_INDEX_HEAD_TAIL
        ix = ctx.index_build(s, 1, c["mask"], mode)
        k, p = ix.dump()
        assert ix.entries == c["n"] == k.size, c["name"]
        assert index_digest(k, p) == c["sha256"], c["name"]
        if c["mode"] == "head_tail":
            assert ix.visited == c["rv"] and len(np.unique(k)) == c["nkeys"], c["name"]
        # hash_table::find for present, absent and zero keys, in list order
        ok, op, _, _ = oracle.index(text, c["mask"], c["mode"])
        rng = np.random.RandomState(7)
        probe = np.concatenate([ok[rng.randint(0, max(ok.size, 1), 200)] if ok.size else np.zeros(0, np.uint32),
                                rng.randint(0, 2 ** 32, 50, dtype=np.uint64).astype(np.uint32) & np.uint32(c["mask"]),
                                np.array([0, 0xFFFFFFFF], np.uint32)]).astype(np.uint32)
        off, pos = ix.find(probe)
        for q, key in enumerate(probe):
            want = op[ok == key].tolist() if key else []
            assert pos[int(off[q]):int(off[q + 1])].tolist() == want, (c["name"], hex(int(key)))


# ----------------------------------------------------------------------------- banded DP
def check_result(got, exp, tag):
    assert int(got["rc"]) == exp["rc"], (tag, int(got["rc"]), exp)
    for k in ("len_a", "len_b", "max_dst"):
        assert int(got[k]) == exp[k], (tag, k)
    if exp["rc"] >= 0:
        for k in ("cost", "matlen_a", "matlen_b"):
            assert int(got[k]) == exp[k], (tag, k, int(got[k]), exp)


@pytest.fixture(params=["bitvec_when_acgt", "rowsweep_always"])
def text_form(request, monkeypatch):
    """The one-pair text entry points (the compat seq_aligner::align) take an ACGT-only pair through the bit-vector array and
    anything else through the raw-byte row sweep; PBA_TEXT_ROWSWEEP=1 forces the row sweep for every pair.  Same answers."""
    if request.param == "rowsweep_always":
        monkeypatch.setenv("PBA_TEXT_ROWSWEEP", "1")
    return request.param


def test_align_text_golden(ctx, text_form):
    """Raw-byte semantics (seq_aligner.h:136), every golden case incl. the reference's own KATs."""
    for c in gold_json("align_kat.json"):
        got = ctx.align_text(c["a"].encode("latin1"), c["b"].encode("latin1"), c["R"], c["a_fwd"], c["b_fwd"])
        check_result(got, c["exp"], c["tag"])


def pairs_for(cases):
    seqs, pairs = [], []
    for c in cases:
        a, b = c["a"].encode(), c["b"].encode()
        ia, ib = len(seqs), len(seqs) + 1
        seqs += [a, b]
        fl = (0 if c["a_fwd"] else 1) | (0 if c["b_fwd"] else 2)
        pairs.append((ia, 0 if c["a_fwd"] or not a else len(a) - 1, len(a), ib,
                      0 if c["b_fwd"] or not b else len(b) - 1, len(b), fl))
    return seqs, np.array(pairs, PAIR_DTYPE)


@pytest.mark.parametrize("kernel", KERNELS)
def test_align_batch_golden(ctx, kernel):
    """Packed 2-bit path, both kernels, on every ACGT-only golden case, grouped by R."""
    cases = [c for c in gold_json("align_kat.json") if set(c["a"] + c["b"]) <= set("ACGT")]
    assert len(cases) > 280
    for R in sorted({c["R"] for c in cases}):
        sub = [c for c in cases if c["R"] == R]
        seqs, pairs = pairs_for(sub)
        S = ctx.seqs_from_list(seqs, strict_acgt=True)
        out = ctx.align_batch(S, S, pairs, R, kernel=kernel)
        for c, got in zip(sub, out):
            check_result(got, c["exp"], (c["tag"], kernel))


@pytest.mark.parametrize("kernel", KERNELS)
def test_align_size_guard(ctx, kernel):
    """seq_aligner.h:104-107: len_a >= MAXN+MAXM or max_dst >= MAXM -> -1."""
    a = b"ACGT" * 30
    S = ctx.seqs_from_list([a, a])
    pr = np.array([(0, 0, 120, 1, 0, 120, 0)], PAIR_DTYPE)
    assert int(ctx.align_batch(S, S, pr, 0.3, maxn=100, maxm=30, kernel=kernel)[0]["rc"]) == -1   # max_dst=37 >= 30
    assert int(ctx.align_batch(S, S, pr, 0.3, maxn=60, maxm=50, kernel=kernel)[0]["rc"]) == -1    # len_a=120 >= 110
    assert int(ctx.align_batch(S, S, pr, 0.3, maxn=100, maxm=40, kernel=kernel)[0]["rc"]) == 120


@pytest.mark.parametrize("kernel", KERNELS)
def test_align_random_vs_oracle(ctx, oracle, kernel):
    """Seeded synthetic pairs at sizes the oracle finishes in seconds: true overlaps, false hits,
    both directions, a longer than b and b longer than a, R in {0.15, 0.30}."""
    g = eng.synth_genome(77, 60000)
    reads, offs, starts = eng.synth_reads(78, g, 48, 2500)
    seqs = [g.tobytes()] + [reads[int(offs[r]):int(offs[r + 1])].tobytes() for r in range(48)]
    S = ctx.seqs_from_list(seqs, strict_acgt=True)
    rng = np.random.RandomState(5)
    pairs = []
    for r in range(48):
        tp, rs = int(starts[r]), r + 1
        j = int(rng.randint(0, 40))
        pairs.append((rs, j, 2500 - j, 0, tp + j, 60000 - tp - j, 0))              # true locus, locator shape
        p = int(rng.randint(0, 50000))
        pairs.append((rs, j, 2500 - j, 0, p, 60000 - p, 0))                         # false locus
        ln = int(rng.randint(300, 2000))
        pairs.append((0, tp, min(60000 - tp, ln + 900), rs, 0, ln, 0))              # a longer than b (SURVEY B4 shape)
        k = int(rng.randint(600, 2400))
        pairs.append((rs, k, k + 1, 0, tp + k, tp + k + 1, 3))                      # both backward
        pairs.append((rs, 0, 2500, rs, 0, 2500, 0))                                 # identical
    arr = np.array(pairs, PAIR_DTYPE)

    def elems(seq, pos, ln, back):
        return seqs[seq][pos - ln + 1:pos + 1] if back else seqs[seq][pos:pos + ln]

    for R in (0.30, 0.15):
        out = ctx.align_batch(S, S, arr, R, kernel=kernel)
        n_ok = 0
        for pr, got in zip(pairs, out):
            sa, pa, la, sb, pb, lb, fl = pr
            exp = oracle.align(elems(sa, pa, la, fl & 1), elems(sb, pb, lb, fl & 2), R, not (fl & 1), not (fl & 2))
            check_result(got, exp, (R, pr, kernel))
            n_ok += exp["rc"] >= 0
        assert n_ok >= 48          # the test is not vacuous: plenty of successful alignments


# ----------------------------------------------------------------------------- traceback
def test_traceback_golden_and_oracle(ctx, oracle, text_form):
    """edits[] / nedit (seq_aligner.h:214-233): every golden case through pba_align_text_trace (digest of the whole
    script produced by the reference), and the ACGT ones through the packed batch form against the oracle."""
    import hashlib
    cases = gold_json("align_kat.json")
    for c in cases:
        r, ops = ctx.align_text_trace(c["a"].encode("latin1"), c["b"].encode("latin1"), c["R"], c["a_fwd"], c["b_fwd"])
        check_result(r, c["exp"], c["tag"])
        if c["exp"]["rc"] >= 0:
            assert ops.size == c["exp"]["nedit"], c["tag"]
            assert (int(ops[0]) if ops.size else 0) == c["exp"]["first_op"], c["tag"]
            assert hashlib.sha256(bytes(ops)).hexdigest()[:24] == c["exp"]["ops_sha"], c["tag"]
        else:
            assert ops.size == 0
    sub = [c for c in cases if set(c["a"] + c["b"]) <= set("ACGT") and c["R"] == 0.3]
    seqs, pairs = pairs_for(sub)
    S = ctx.seqs_from_list(seqs, strict_acgt=True)
    for kernel in KERNELS:              # row sweep with parent bytes, bit-vector array with 2 parent bits per cell
        out, scripts = ctx.align_batch_trace(S, S, pairs, 0.3, kernel=kernel)
        for c, got, ops in zip(sub, out, scripts):
            check_result(got, c["exp"], (c["tag"], kernel))
            if c["exp"]["rc"] >= 0:
                assert ops.size == c["exp"]["nedit"] and hashlib.sha256(bytes(ops)).hexdigest()[:24] == c["exp"]["ops_sha"], \
                    (c["tag"], kernel)
            exp = oracle.align(c["a"].encode(), c["b"].encode(), 0.3, c["a_fwd"], c["b_fwd"], want_ops=True)
            assert ops.tolist() == exp["ops"].tolist(), (c["tag"], kernel)
    # a mid-size true overlap: script length and content against the oracle
    g = eng.synth_genome(5, 30000)
    reads, offs, starts = eng.synth_reads(6, g, 4, 3000)
    for r in range(4):
        a = reads[int(offs[r]):int(offs[r + 1])].tobytes(); b = g[int(starts[r]):int(starts[r]) + 4000].tobytes()
        res, ops = ctx.align_text_trace(a, b, 0.3)
        exp = oracle.align(a, b, 0.3, want_ops=True)
        check_result(res, exp, r)
        assert ops.tolist() == exp["ops"].tolist()


def test_text_entry_points_clip_before_they_size_or_copy(ctx, oracle, text_form):
    """seq_aligner.h:94-102 clips the longer accessor to the shorter + max_dst before anything else; locator.cpp:80-81 hands
    align() the whole rest of its contig (up to 800 kb) and ref_seq.h:282-286 the whole rest of the reference.  Accessors of
    100 kb, 300 kb and 790 kb (far beyond the engine's 65 000-element limit, which applies to what the DP sees) through
    pba_align_text / _trace / _matrix, forward and backward, either side the long one, ACGT and with other bytes: result,
    edit script and matrix cells equal the oracle's."""
    g = eng.synth_genome(91, 800000)
    reads, offs, starts = eng.synth_reads(92, g, 6, 2500)
    gb = g.tobytes()
    cases = []
    for r, rest in zip(range(6), (100000, 300000, 790000, 100000, 300000, 70000)):
        seg = reads[int(offs[r]):int(offs[r + 1])].tobytes()
        p = int(starts[r]) if int(starts[r]) + rest <= 800000 else 800000 - rest
        cases.append((seg, gb[p:p + rest], True, r))                                   # a read against the rest of the contig
        cases.append((seg[::-1], gb[p:p + rest][::-1], False, r))                      # the same elements through backward accessors: the
                                                                                       # ones the DP sees are the LAST bytes in memory
    cases.append((gb[1000:201000], cases[0][0], True, "long_a"))                       # the a side is the long one
    cases.append((cases[0][0].replace(b"G", b"N", 3), cases[0][1], True, "non_acgt"))  # raw-byte form whatever text_form says
    cases.append((cases[2][0][:900] + b"acgt", gb[int(starts[1]):int(starts[1]) + 120000].lower(), True, "lower"))
    n_ok = 0
    for a, b, fwd, tag in cases:
        exp = oracle.align(a, b, 0.3, fwd, fwd, want_ops=True)
        assert exp["len_a"] <= 4000 and exp["len_b"] <= 4000 and max(len(a), len(b)) >= 70000
        check_result(ctx.align_text(a, b, 0.3, fwd, fwd), exp, tag)
        res, ops = ctx.align_text_trace(a, b, 0.3, fwd, fwd)
        check_result(res, exp, tag)
        assert ops.tolist() == exp["ops"].tolist(), tag
        n_ok += exp["rc"] >= 0
    assert n_ok >= 8                     # (reads whose locus lies too close to the contig end meet a shifted window: failures)
    # the matrix of a clipped pair: every cell the reference's call wrote
    a, b, fwd, _ = cases[0]
    exp = oracle.align(a, b, 0.3, fwd, fwd)
    res, cost, par, rows = ctx.align_text_matrix(a, b, 0.3, fwd, fwd)
    check_result(res, exp, "matrix")
    md = exp["max_dst"]
    assert cost.shape == (exp["len_a"] + 1, 2 * md + 1) and rows == exp["len_a"]
    rng = np.random.RandomState(3)
    for i in rng.randint(1, exp["len_a"] + 1, 300):
        j = int(np.clip(i + rng.randint(-md, md + 1), max(0, i - md), min(exp["len_b"], i + md)))
        assert (int(cost[i, j - i + md]), int(par[i, j - i + md])) == oracle.cell(int(i), j), (i, j)
    # limits: the reference's size guard (seq_aligner.h:104-107) answers -1 for what t_aligner cannot hold; without a guard the
    # engine's own limit is an explicit error -- on the clipped lengths in both cases
    big = gb[:70000]
    r = ctx.align_text(big, gb[5:70005], 0.3, maxn=26000, maxm=6000)
    assert int(r["rc"]) == -1 and int(r["len_a"]) == 70000 and int(r["max_dst"]) == 21001
    with pytest.raises(eng.PbaError):
        ctx.align_text(big, gb[5:70005], 0.3)
    r = ctx.align_text(cases[0][0], gb, 0.3, maxn=26000, maxm=6000)           # 800 kb accessor, 2.5 kb read: fine
    assert int(r["len_b"]) == 2500 + int(r["max_dst"])


# ----------------------------------------------------------------------------- drivers
@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("name", ["cfg1_R30", "cfg1_R15", "cfg1_pacbio", "short_mix", "r15k_R30", "r15k_R15"])
def test_locate_golden(ctx, name, kernel):
    """locator.cpp:70-92 end to end: same rows (found, j, pos, cost, len-j, matlen) and the same
    number of candidate pairs per read as the reference."""
    meta = {m["name"]: m for m in gold_json("locator.json")}[name]
    want = gold_npz("locator.npz")[name]
    g, reads, offs = locator_inputs(meta)
    T = ctx.seqs_from_list([g.tobytes()], strict_acgt=True)
    Rd = ctx.seqs_from_text(reads, offs, strict_acgt=True)
    ix = ctx.index_build(T, 0, meta["mask"], PBA_INDEX_ALL)
    rows, st = ctx.locate(ix, T, 0, Rd, meta["R"], meta["trials"], meta["min_len"], kernel=kernel)
    check_locator_rows(rows, want, meta["columns"], name)
    for k, v in meta["stats"].items():
        assert st[k] == v, (name, k, st)


def test_locate_cells_match_oracle(ctx, oracle):
    meta = {m["name"]: m for m in gold_json("locator.json")}["cfg1_pacbio"]
    g, reads, offs = locator_inputs(meta)
    T = ctx.seqs_from_list([g.tobytes()])
    Rd = ctx.seqs_from_text(reads, offs)
    ix = ctx.index_build(T, 0, meta["mask"], PBA_INDEX_ALL)
    _, st = ctx.locate(ix, T, 0, Rd, meta["R"], meta["trials"], meta["min_len"], kernel=PBA_KERNEL_ROWSWEEP)
    _, so = oracle.locator(g, meta["mask"], meta["R"], reads, offs, meta["trials"], meta["min_len"], nthreads=4)
    assert st == so


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("name", ["ss_30k", "ss_50k", "ss_12k"])
def test_spaced_round_golden(ctx, name, kernel):
    """spaced_seed.cpp:420-437 (locked round) incl. seed_at's pos%4==0 behaviour."""
    meta = {m["name"]: m for m in gold_json("spaced.json")}[name]
    want = gold_npz("spaced.npz")[name]
    g, file, rec_offs = spaced_inputs(meta, eng.text2bin)
    Rf = ctx.seqs_from_list([g.tobytes()])
    Rd = ctx.seqs_from_records(file, 0, 1 << 30)
    assert Rd.count == meta["n_reads"]
    ix = ctx.index_build(Rf, 0, meta["mask"], PBA_INDEX_HEAD_TAIL)
    rows = ctx.spaced_round(ix, Rf, 0, Rd, meta["R"], meta["max_trial"], meta["overlap_min"], buggy_seed_at=True,
                            kernel=kernel)
    cols = meta["columns"]
    for ci, col in enumerate(cols):
        if col in ("dir", "ref_pos", "cost", "matlen_a", "matlen_b"):
            sel = want[:, cols.index("found")] == 1
            assert (rows[col][sel] == want[sel, ci]).all(), (name, col)
        else:
            assert (rows[col] == want[:, ci]).all(), (name, col)


# ----------------------------------------------------------------------------- exchange form of the index
def test_index_scan_gather_build_equals_direct_build(ctx):
    """pba_index_scan over k slices + concatenation (what the RCCL all-gather produces, padding
    included) + pba_index_from_entries == pba_index_build, for both visiting orders."""
    import torch
    g = gold_json("index.json")
    mask = eng.mask_from_pattern(MASK_PAT)
    for c in [x for x in g["cases"] if x["name"] in ("all_100000", "ht_50000", "ht_20017", "all_repeat", "all_33")]:
        text = index_inputs(c)
        S = ctx.seqs_from_list([text])
        mode = PBA_INDEX_ALL if c["mode"] == "all" else PBA_INDEX_HEAD_TAIL
        for nparts in (1, 2, 3, 8):
            cap = len(text) // nparts + 64
            allent = torch.full((nparts * cap,), -1, dtype=torch.int64, device="cuda")
            total = 0
            for part in range(nparts):
                sl = allent[part * cap:(part + 1) * cap]
                total += ctx.index_scan(S, 0, c["mask"], mode, part, nparts, sl.data_ptr(), cap)
            torch.cuda.synchronize()
            assert total == c["n"], (c["name"], nparts)
            ix = ctx.index_from_entries(allent.data_ptr(), nparts * cap, c["mask"], mode, len(text))
            k, p = ix.dump()
            assert index_digest(k, p) == c["sha256"], (c["name"], nparts)
            if c["mode"] == "head_tail":
                assert ix.visited == c["rv"]


# ----------------------------------------------------------------------------- band certificate / re-run
def test_bitvec_uncertified_pairs_rerun_at_reference_band(ctx, oracle):
    """Reads at ~21 % error, long enough that the first-pass window (as wide as a one-block ring lets it be: 1 384) is
    below their cost: the narrow pass cannot certify the goal, the pair is re-run at the reference band, and the answer
    is still the reference's.  (Shorter pairs get the whole band in the first pass and never need the re-run.)"""
    g = eng.synth_genome(55, 60000)
    L, nr = 7300, 64
    reads, offs, starts = eng.synth_reads(56, g, nr, L, 0.07, 0.07, 0.07)
    seqs = [g.tobytes()] + [reads[int(offs[r]):int(offs[r + 1])].tobytes() for r in range(nr)]
    S = ctx.seqs_from_list(seqs, strict_acgt=True)
    pairs = [(r + 1, 0, L, 0, int(starts[r]), min(60000 - int(starts[r]), L + 2200), 0) for r in range(nr)]
    out = ctx.align_batch(S, S, np.array(pairs, PAIR_DTYPE), 0.30, kernel=PBA_KERNEL_BITVEC)
    prof = ctx.last_profile()
    assert prof["nb_first"] == 1
    n_ok = 0
    for pr, got in zip(pairs, out):
        exp = oracle.align(seqs[pr[0]][:L], seqs[0][pr[4]:pr[4] + pr[5]], 0.30)
        check_result(got, exp, pr)
        n_ok += exp["rc"] >= 0 and exp["cost"] > 1384
    assert n_ok >= 4 and prof["n_redo"] >= n_ok          # the re-run path was really taken
    # same through the locate driver
    T = ctx.seqs_from_list([g.tobytes()])
    Rd = ctx.seqs_from_text(reads, offs)
    mask = eng.mask_from_pattern(MASK_PAT)
    ix = ctx.index_build(T, 0, mask, PBA_INDEX_ALL)
    rows, st = ctx.locate(ix, T, 0, Rd, 0.30, 50, 500, kernel=PBA_KERNEL_BITVEC)
    assert ctx.last_profile()["n_redo"] > 0
    want, wst = oracle.locator(g, mask, 0.30, reads, offs, 50, 500, nthreads=8)
    for c in ("found", "j", "pos", "cost", "seglen", "matlen_a", "matlen_b", "n_pairs"):
        assert (rows[c] == want[c]).all(), c
    assert st == wst


# ----------------------------------------------------------------------------- full-size properties
def test_fullsize_kernels_agree_and_recover_planted_loci(ctx):
    """BASELINE config-2 shaped pairs (15 kb reads, R = 0.30, band 9003): too slow for the CPU oracle
    beyond the golden handful, so check size-independent properties: the two independent kernels
    (full-band row sweep vs bit-vector array) agree bit for bit, every read is located at the locus it
    was sampled from, and cost is symmetric under swapping the two sequences."""
    g = eng.synth_genome(2, 600000)
    n = 40
    reads, offs, starts = eng.synth_reads(3, g, n, 15000)
    T = ctx.seqs_from_text(g, np.array([0, g.size], np.uint64), strict_acgt=True)
    Rd = ctx.seqs_from_text(reads, offs, strict_acgt=True)
    mask = eng.mask_from_pattern(MASK_PAT)
    ix = ctx.index_build(T, 0, mask, PBA_INDEX_ALL)
    r_bv, s_bv = ctx.locate(ix, T, 0, Rd, 0.30, 50, 500, kernel=PBA_KERNEL_BITVEC)
    r_rs, s_rs = ctx.locate(ix, T, 0, Rd, 0.30, 50, 500, kernel=PBA_KERNEL_ROWSWEEP)
    assert s_bv == s_rs
    for c in r_bv.dtype.names:
        assert (r_bv[c] == r_rs[c]).all(), c
    f = r_bv["found"] == 1
    assert f.sum() >= n * 0.7
    # located position == sampled start + probe offset, up to the indels inside the first j bases
    assert (np.abs(r_bv["pos"][f] - (starts[f].astype(np.int64) + r_bv["j"][f])) <= 12).all()
    assert (r_bv["cost"][f] < 0.3 * 15000).all() and (r_bv["matlen_b"][f] >= r_bv["seglen"][f] * 0.7).all()
    # transposition symmetry on explicit pairs: align(a,b).cost == align(b,a).cost, matlens swap
    pairs = [(0, int(p), 15000 - int(j) + 4501, r, int(j), 15000 - int(j), 0)
             for r, (p, j) in enumerate(zip(r_bv["pos"], r_bv["j"])) if f[r]][:12]
    fwd = ctx.align_batch(T, Rd, np.array(pairs, PAIR_DTYPE), 0.30, kernel=PBA_KERNEL_BITVEC)
    rev = ctx.align_batch(Rd, T, np.array([(p[3], p[4], p[5], p[0], p[1], p[2], 0) for p in pairs], PAIR_DTYPE), 0.30,
                          kernel=PBA_KERNEL_BITVEC)
    for x, y in zip(fwd, rev):
        assert int(x["cost"]) == int(y["cost"]) and int(x["matlen_a"]) == int(y["matlen_b"]) \
            and int(x["matlen_b"]) == int(y["matlen_a"]) and int(x["rc"]) == int(x["matlen_b"])


# ----------------------------------------------------------------------------- the reference's API surface
def test_compat_headers_cpp(lib):
    """include/compat/{dna_seq,seq_aligner,ref_seq}.h: the reference's own test expectations (dna_test,
    aligner_test incl. edit scripts, all twelve ref_test cases: votes, growth, evolve) compiled with g++ against the
    compat headers and run on the GPU."""
    import os
    import subprocess
    from conftest import ROOT
    out = os.path.join(ROOT, "tests", "cpp", "_build")
    os.makedirs(out, exist_ok=True)
    exe = os.path.join(out, "compat_test")
    libdir = os.path.join(ROOT, "pacbioassembly_amd", "lib")
    subprocess.run(["g++", "-O1", "-std=c++11", "-Wall", "-I", os.path.join(ROOT, "include"), "-I",
                    os.path.join(ROOT, "include", "compat"), "-o", exe, os.path.join(ROOT, "tests", "cpp", "compat_test.cpp"),
                    "-L", libdir, "-lpba", f"-Wl,-rpath,{libdir}"], check=True)
    r = subprocess.run([exe, os.path.join(GOLD, "real_align.txt")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 failed" in r.stdout


# ----------------------------------------------------------------------------- error behaviour of the boundary
def test_error_codes_not_exceptions_across_the_abi(ctx):
    S = ctx.seqs_from_list([b"ACGT" * 50, b"ACGTTGCA" * 20])
    # pair outside its sequence / negative length / bad R / unknown kernel -> PBA_E_INVALID, never a crash
    for pr in [(0, 0, 500, 1, 0, 100, 0), (0, 190, 50, 1, 0, 100, 0), (5, 0, 10, 1, 0, 10, 0), (0, 10, 50, 1, 0, 20, 1)]:
        with pytest.raises(PbaError) as e:
            ctx.align_batch(S, S, np.array([pr], PAIR_DTYPE), 0.3)
        assert e.value.status == -1, pr
    for R in (0.0, 1.0, -0.1):
        with pytest.raises(PbaError) as e:
            ctx.align_batch(S, S, np.array([(0, 0, 100, 1, 0, 100, 0)], PAIR_DTYPE), R)
        assert e.value.status == -1
    with pytest.raises(PbaError) as e:
        ctx.align_batch(S, S, np.array([(0, 0, 100, 1, 0, 100, 0)], PAIR_DTYPE), 0.3, kernel=7)
    assert e.value.status == -1
    # a read longer than the engine limit -> PBA_E_TOOLONG from the drivers
    big = ctx.seqs_from_list([b"A" * 70000])
    T = ctx.seqs_from_list([b"ACGT" * 5000])
    ix = ctx.index_build(T, 0, eng.mask_from_pattern(MASK_PAT), PBA_INDEX_ALL)
    with pytest.raises(PbaError) as e:
        ctx.locate(ix, T, 0, big, 0.3)
    assert e.value.status == -4
    # locate wants a PBA_INDEX_ALL index of that very sequence
    ix2 = ctx.index_build(T, 0, eng.mask_from_pattern(MASK_PAT), PBA_INDEX_HEAD_TAIL)
    with pytest.raises(PbaError) as e:
        ctx.locate(ix2, T, 0, S, 0.3)
    assert e.value.status == -1
    # bytes outside ACGT: the index may be built (C2I codes, like the reference's), aligning is refused loudly
    Sn = ctx.seqs_from_list([b"ACGTNACGT" * 20, b"ACGT" * 50])
    ctx.index_build(Sn, 0, eng.mask_from_pattern(MASK_PAT), PBA_INDEX_ALL)
    with pytest.raises(PbaError) as e:
        ctx.align_batch(Sn, Sn, np.array([(0, 0, 100, 1, 0, 100, 0)], PAIR_DTYPE), 0.3)
    assert e.value.status == -6
    # empty batches are fine
    assert ctx.align_batch(S, S, np.zeros(0, PAIR_DTYPE), 0.3).size == 0
    empty = ctx.seqs_from_list([])
    rows, st = ctx.locate(ix, T, 0, empty, 0.3)
    assert rows.size == 0 and st["n_pairs"] == 0


def test_runs_on_a_caller_owned_stream(ctx, oracle):
    """pba_ctx_set_stream: the engine enqueues on torch's stream (how bench.py and a torch host would use it)."""
    import torch
    g = eng.synth_genome(91, 20000)
    reads, offs, _ = eng.synth_reads(92, g, 32, 900)
    s = torch.cuda.Stream()
    ctx.set_stream(s.cuda_stream)
    try:
        T = ctx.seqs_from_text(g, np.array([0, g.size], np.uint64))
        Rd = ctx.seqs_from_text(reads, offs)
        mask = eng.mask_from_pattern(MASK_PAT)
        ix = ctx.index_build(T, 0, mask, PBA_INDEX_ALL)
        rows, st = ctx.locate(ix, T, 0, Rd, 0.3)
    finally:
        ctx.set_stream(None)
    want, wst = oracle.locator(g, mask, 0.3, reads, offs, 50, 500, nthreads=2)
    for c in ("found", "j", "pos", "cost", "n_pairs"):
        assert (rows[c] == want[c]).all(), c
    assert st == wst


# ----------------------------------------------------------------------------- fuzz
def fuzz_pairs(seed, count, max_len=4000):
    """Random pair shapes: lengths 1..max_len, related / unrelated, tails, swapped roles, all four direction
    combinations, R from 0.05 to 0.49."""
    rng = np.random.RandomState(seed)
    alpha = np.frombuffer(b"ACGT", np.uint8)
    seqs, pairs, Rs = [], [], []
    for t in range(count):
        la = int(np.exp(rng.uniform(0, np.log(max_len))))
        a = alpha[rng.randint(0, 4, la)]
        kind = rng.randint(0, 10)
        if kind == 0:
            b = alpha[rng.randint(0, 4, int(np.exp(rng.uniform(0, np.log(max_len)))))]       # unrelated
        else:
            e = [0.0, 0.02, 0.08, 0.15, 0.15, 0.22, 0.30, 0.40, 0.15, 0.05][kind]
            u = rng.rand(la)
            out = []
            for ch, x in zip(a, u):
                if x < e / 3:
                    out += [alpha[rng.randint(4)], ch]
                elif x < 2 * e / 3:
                    pass
                elif x < e:
                    out.append(alpha[rng.randint(4)])
                else:
                    out.append(ch)
            tail = int(rng.choice([0, 0, 1, 17, 300, 2000]))
            b = np.concatenate([np.array(out, np.uint8), alpha[rng.randint(0, 4, tail)]]).astype(np.uint8)
        if rng.rand() < 0.35:
            a, b = b, a
        fa, fb = (int(rng.rand() < 0.25), int(rng.rand() < 0.25))
        a, b = a.tobytes(), b.tobytes()
        ia = len(seqs); seqs += [a, b]
        pairs.append((ia, len(a) - 1 if fa and a else 0, len(a), ia + 1, len(b) - 1 if fb and b else 0, len(b), fa | (fb << 1)))
        Rs.append(float(rng.choice([0.05, 0.1, 0.15, 0.2, 0.3, 0.3, 0.4, 0.49])))
    return seqs, pairs, Rs


@pytest.mark.parametrize("kernel", KERNELS)
def test_align_fuzz_vs_oracle(ctx, oracle, kernel):
    """700 random pair shapes through both kernels against the oracle, bit for bit."""
    seqs, pairs, Rs = fuzz_pairs(424242 + SOAK, 700)
    S = ctx.seqs_from_list(seqs, strict_acgt=True)
    for R in sorted(set(Rs)):
        sel = [q for q in range(len(pairs)) if Rs[q] == R]
        out = ctx.align_batch(S, S, np.array([pairs[q] for q in sel], PAIR_DTYPE), R, kernel=kernel)
        for q, got in zip(sel, out):
            sa, pa, la_, sb, pb, lb_, fl = pairs[q]
            a = seqs[sa][::-1] if fl & 1 else seqs[sa]          # accessor elements in order
            b = seqs[sb][::-1] if fl & 2 else seqs[sb]
            exp = oracle.align(a, b, R)                         # forward over the reversed copy == backward accessor
            check_result(got, exp, (q, R, la_, lb_, fl, kernel))


@pytest.mark.parametrize("kernel", KERNELS)
def test_traceback_fuzz_vs_oracle(ctx, oracle, kernel):
    """Edit scripts (seq_aligner.h:214-233) of 400 random pair shapes, both trace kernels, against the oracle's
    find_path: same ops in the same order, including the tie-breaks (MATCH, then INSERT, then DELETE)."""
    seqs, pairs, Rs = fuzz_pairs(777 + SOAK, 400, 3000)
    S = ctx.seqs_from_list(seqs, strict_acgt=True)
    n_scripts = 0
    for R in sorted(set(Rs)):
        sel = [q for q in range(len(pairs)) if Rs[q] == R]
        out, scripts = ctx.align_batch_trace(S, S, np.array([pairs[q] for q in sel], PAIR_DTYPE), R, kernel=kernel)
        for q, got, ops in zip(sel, out, scripts):
            sa, pa, la_, sb, pb, lb_, fl = pairs[q]
            a = seqs[sa][::-1] if fl & 1 else seqs[sa]
            b = seqs[sb][::-1] if fl & 2 else seqs[sb]
            exp = oracle.align(a, b, R, want_ops=True)
            check_result(got, exp, (q, R, la_, lb_, fl, kernel))
            assert ops.tolist() == exp["ops"].tolist(), (q, R, la_, lb_, fl, kernel)
            n_scripts += exp["rc"] >= 0
    assert n_scripts > 150


def test_traceback_uncertified_and_fullsize(ctx, oracle):
    """The bit-vector trace kernel where its first pass cannot certify the goal (7.3 kb reads at ~21 % error: the pair is
    re-swept at the reference band by the second launch and that sweep is the one walked back), and at BASELINE size
    (15 kb reads, band 9003: 27 MB of parent bits per pair) against the oracle's script."""
    g = eng.synth_genome(55, 60000)
    L, nr = 7300, 40
    reads, offs, starts = eng.synth_reads(56, g, nr, L, 0.07, 0.07, 0.07)
    seqs = [g.tobytes()] + [reads[int(offs[r]):int(offs[r + 1])].tobytes() for r in range(nr)]
    S = ctx.seqs_from_list(seqs, strict_acgt=True)
    pairs = [(r + 1, 0, L, 0, int(starts[r]), min(60000 - int(starts[r]), L + 2200), 0) for r in range(nr)]
    pairs += [(0, int(starts[r]), min(L + 2200, 60000 - int(starts[r])), r + 1, 0, L, 0) for r in range(nr)]   # roles swapped: a is the longer side
    out, scripts = ctx.align_batch_trace(S, S, np.array(pairs, PAIR_DTYPE), 0.30, kernel=PBA_KERNEL_BITVEC)
    assert ctx.last_profile()["n_redo"] >= 3
    n_wide = 0
    for pr, got, ops in zip(pairs, out, scripts):
        exp = oracle.align(seqs[pr[0]][pr[1]:pr[1] + pr[2]], seqs[pr[3]][pr[4]:pr[4] + pr[5]], 0.30, want_ops=True)
        check_result(got, exp, pr)
        assert ops.tolist() == exp["ops"].tolist(), pr
        n_wide += exp["rc"] >= 0 and exp["cost"] > 1384
    assert n_wide >= 3
    # full size
    g = eng.synth_genome(2, 200000)
    reads, offs, starts = eng.synth_reads(3, g, 5, 15000)
    seqs = [g.tobytes()] + [reads[int(offs[r]):int(offs[r + 1])].tobytes() for r in range(5)]
    S = ctx.seqs_from_list(seqs, strict_acgt=True)
    pairs = [(r + 1, 0, 15000, 0, int(starts[r]), min(25000, 200000 - int(starts[r])), 0) for r in range(5)]
    out, scripts = ctx.align_batch_trace(S, S, np.array(pairs, PAIR_DTYPE), 0.30)
    n_ok = 0
    for pr, got, ops in zip(pairs, out, scripts):
        exp = oracle.align(seqs[pr[0]][pr[1]:pr[1] + pr[2]], seqs[0][pr[4]:pr[4] + pr[5]], 0.30, want_ops=True)
        check_result(got, exp, pr)
        assert ops.tolist() == exp["ops"].tolist(), pr
        n_ok += exp["rc"] > 0 and ops.size > 15000
    assert n_ok >= 2                # (the others fail the reference's row-11 check: an indel among the first bases)


def test_locate_ecoli_scale_genome_vs_oracle(ctx, oracle):
    """BASELINE config 3 shape (locator.cpp path against a 4.6 Mb target; the real E. coli data cannot be
    fetched, so a synthetic genome of that size): identical rows (nseq, pos, cost, len-j) and pair counts."""
    g = eng.synth_genome(33, 4_600_000)
    reads, offs, _ = eng.synth_reads(34, g, 400, 2000, 0.09, 0.045, 0.015)      # PacBio-like error mix
    mask = eng.mask_from_pattern(MASK_PAT)
    T = ctx.seqs_from_text(g, np.array([0, g.size], np.uint64), strict_acgt=True)
    Rd = ctx.seqs_from_text(reads, offs, strict_acgt=True)
    ix = ctx.index_build(T, 0, mask, PBA_INDEX_ALL)
    assert ix.entries > 4_500_000
    want, wst = oracle.locator(g, mask, 0.30, reads, offs, 50, 500, nthreads=8)
    for kernel in KERNELS:
        rows, st = ctx.locate(ix, T, 0, Rd, 0.30, 50, 500, kernel=kernel)
        for c in ("nseq", "found", "j", "pos", "cost", "seglen", "matlen_a", "matlen_b", "n_pairs"):
            assert (rows[c] == want[c]).all(), (kernel, c)
        assert st == wst
    assert wst["n_located"] > 250 and wst["n_pairs"] > wst["n_located"]


@pytest.fixture(params=["census_then_exact_slices", "equal_room", "equal_room_overflows"])
def prekeep(request, monkeypatch):
    """The all-vs-all tests run three times, once per way the scan of the bit-vector kernels (overlap.h: k_ovl_scan -- the first
    32 rows of every candidate run where it is found, only survivors are written) sizes the survivors' slices: a census
    launch and exact slices (what the first range of a table gets), equal room for every target (what later ranges get from
    the census of an earlier one; forced here), and equal room that some target outgrows, so that the range is scanned
    again with exact slices.  Same overlaps, same pair counts, all against the oracle."""
    if request.param == "equal_room":
        monkeypatch.setenv("PBA_OVL_ROOM", "16384")
    if request.param == "equal_room_overflows":
        monkeypatch.setenv("PBA_OVL_ROOM", "4")
    return request.param


# ----------------------------------------------------------------------------- all-vs-all overlap
@pytest.mark.parametrize("kernel", KERNELS)
def test_overlap_all_vs_oracle_composition(ctx, oracle, kernel, prekeep):
    """pba_overlap_all == running the oracle's locked spaced_seed round once per target read (every read in
    the reference role, every other read a query, intended seed_at): same successful (target, query) set, same
    j / dir / hit position / cost / match lengths, for whole ranges and for target shards."""
    g = eng.synth_genome(71, 9000)
    n, rl = 64, 1300
    reads, offs, _ = eng.synth_reads(72, g, n, rl)
    texts = [reads[int(offs[i]):int(offs[i + 1])].tobytes() for i in range(n)]
    texts[5] = texts[5][:700]            # ragged: a short read, and one shorter than OVERLAP_MIN + 16
    texts[9] = texts[9][:70]
    file = b"".join(eng.text2bin(t) for t in texts)
    rec_offs = np.cumsum([0] + [4 + (len(t) + 3) // 4 for t in texts[:-1]]).astype(np.uint64)
    mask = eng.mask_from_pattern(MASK_PAT)
    want, pairs = [], 0
    for t in range(n):
        rows = oracle.spaced_round(texts[t], mask, 0.30, file, rec_offs, 32, 64, buggy=False, nthreads=8)
        pairs += int(rows["n_pairs"].sum()) - int(rows["n_pairs"][t])
        for q in range(n):
            if q != t and rows["found"][q]:
                want.append((t, q, int(rows["j"][q]), int(rows["dir"][q]), int(rows["ref_pos"][q]), int(rows["cost"][q]),
                             int(rows["matlen_a"][q]), int(rows["matlen_b"][q])))
    assert len(want) > 100
    S = ctx.seqs_from_list(texts, strict_acgt=True)
    got, st = ctx.overlap_all(S, mask, 0.30, 32, 64, kernel=kernel)
    assert [tuple(int(x) for x in r) for r in got] == want
    assert st["n_overlaps"] == len(want) and st["n_pairs"] == pairs and st["n_candidates"] >= st["n_pairs"]
    assert (st["n_prefiltered"] > 0) == (kernel != PBA_KERNEL_ROWSWEEP)       # failed their first 32 rows in the scan: counted, never written
    assert kernel == PBA_KERNEL_ROWSWEEP or st["cap_overflow"] == (prekeep == "equal_room_overflows")
    # target shards (what ranks of a multi-GPU run do) concatenate to the same answer
    parts = [ctx.overlap_all(S, mask, 0.30, 32, 64, t_lo=a, t_hi=b, kernel=kernel)[0] for a, b in ((0, 20), (20, 21), (21, 64))]
    assert [tuple(int(x) for x in r) for p in parts for r in p] == want
    # exchange form: every "rank" emits the probes of its query shard, the padded buffers are concatenated (what
    # the RCCL all-gather produces) and the gathered table drives the same target scan
    import torch
    shards, cap = ((0, 22), (22, 43), (43, 64)), 22 * 64
    gathered = torch.full((len(shards) * cap,), -1, dtype=torch.int64, device="cuda")
    for k, (a, b) in enumerate(shards):
        ctx.overlap_probes(S, a, b, mask, 32, gathered[k * cap:(k + 1) * cap].data_ptr(), cap)
    torch.cuda.synchronize()
    got2, st2 = ctx.overlap_all_probes(S, gathered.data_ptr(), gathered.numel(), mask, 0.30, 32, 64, kernel=kernel)
    assert [tuple(int(x) for x in r) for r in got2] == want and st2["n_probe_entries"] == st["n_probe_entries"]
    # target ranges against one probe table built once (how configs 4-5 stay below 2^32 candidates per call)
    got3, st3 = ctx.overlap_all_sharded(S, mask, 0.30, 32, 64, targets_per_call=17, kernel=kernel)
    assert [tuple(int(x) for x in r) for r in got3] == want and st3["n_pairs"] == st["n_pairs"]


def test_matrix_cells_and_diagonal_end_vs_oracle(ctx, oracle):
    """seq_aligner::get_cost / get_parent (seq_aligner.h:131-134; locator.cpp:86 prints get_cost(len - j, len - j)):
    pba_align_text_matrix returns every cell the reference's call writes -- borders, the band of every row swept, up to
    an early failure -- with the oracle's cost and parent (which tests/test_oracle_vs_ref.py pins to the reference's own
    array), everything else unwritten; and pba_result::diag_cost is the cell at the end of the diagonal in both kernels."""
    rng = np.random.RandomState(77)
    alpha = np.frombuffer(b"ACGT", np.uint8)
    seqs_a, seqs_b = [], []
    for t in range(36):
        la = int(rng.randint(12, 420))
        a = alpha[rng.randint(0, 4, la)]
        e = float(rng.choice([0.05, 0.2, 0.5]))
        b = a[rng.rand(la) > e / 2].copy()
        flip = rng.rand(b.size) < e / 2
        b[flip] = alpha[rng.randint(0, 4, int(flip.sum()))]
        b = np.concatenate([b, alpha[rng.randint(0, 4, int(rng.choice([0, 7, 90])))]])
        if t % 3 == 0:
            a, b = b, a
        seqs_a.append(a.tobytes()); seqs_b.append(b.tobytes())
    n_failed = 0
    for t, (a, b) in enumerate(zip(seqs_a, seqs_b)):
        fwd = bool(t % 2)
        x = oracle.align(a, b, 0.3, fwd, fwd)
        res, cost, par, rows = ctx.align_text_matrix(a, b, 0.3, fwd, fwd)
        assert int(res["rc"]) == x["rc"] and rows == (x["fail_row"] or x["len_a"])
        n_failed += bool(x["fail_row"])
        md = x["max_dst"]
        written = np.zeros(cost.shape, bool)
        for i in range(0, rows + 1):
            for j in range(max(0, i - md), min(x["len_b"], i + md) + 1):
                if i == 0 and j > md:
                    continue
                assert (int(cost[i, j - i + md]), int(par[i, j - i + md])) == oracle.cell(i, j), (t, i, j)
                written[i, j - i + md] = True
        assert (cost[~written] == 0xFFFF).all() and (par[~written] == 0).all()
        m = min(x["len_a"], x["len_b"])
        assert int(res["diag_cost"]) == (oracle.cell(m, m)[0] if not x["fail_row"] else -1)
    assert 3 < n_failed < 30
    # the batch kernels report the same cell
    A = ctx.seqs_from_list(seqs_a, strict_acgt=True)
    B = ctx.seqs_from_list(seqs_b, strict_acgt=True)
    pairs = np.zeros(len(seqs_a), PAIR_DTYPE)
    pairs["a_seq"] = pairs["b_seq"] = np.arange(len(seqs_a))
    pairs["a_len"] = [len(s) for s in seqs_a]; pairs["b_len"] = [len(s) for s in seqs_b]
    for kernel in KERNELS:
        out = ctx.align_batch(A, B, pairs, 0.3, kernel=kernel)
        for t, (a, b) in enumerate(zip(seqs_a, seqs_b)):
            x = oracle.align(a, b, 0.3)
            m = min(x["len_a"], x["len_b"])
            if not x["fail_row"] and m > 10:
                assert int(out["diag_cost"][t]) == oracle.cell(m, m)[0], (kernel, t)
            elif x["fail_row"]:
                assert int(out["diag_cost"][t]) == -1, (kernel, t)


def test_read_shards_exported_and_regathered_give_the_same_overlaps(ctx):
    """The multi-GPU read exchange in one process: three "ranks" pack their shards (one of them from a binary read file,
    whose payloads are not 16-byte aligned), export the packed arenas into one padded buffer -- what the RCCL all-gather of
    pacbioassembly_amd.distributed.all_gather_packed produces -- and the set rebuilt from that buffer gives the same reads,
    the same probe table and the same overlaps as the set packed in one piece."""
    import torch
    g = eng.synth_genome(171, 7000)
    n, rl = 48, 1200
    reads, offs, _ = eng.synth_reads(172, g, n, rl)
    texts = [reads[int(offs[i]):int(offs[i + 1])].tobytes() for i in range(n)]
    texts[7] = texts[7][:801]
    texts[30] = texts[30][:650]
    mask = eng.mask_from_pattern(MASK_PAT)
    whole = ctx.seqs_from_list(texts, strict_acgt=True)
    want, wst = ctx.overlap_all(whole, mask, 0.30, 32, 64)
    assert len(want) > 60
    shards = [(0, 17), (17, 33), (33, 48)]
    sets = [ctx.seqs_from_list(texts[a:b], strict_acgt=True) if k != 1 else
            ctx.seqs_from_records(b"".join(eng.text2bin(t) for t in texts[a:b]), 0, 1 << 30) for k, (a, b) in enumerate(shards)]
    stride = (max(s.packed_bytes for s in sets) + 15) // 16 * 16
    buf = torch.zeros(len(sets) * stride, dtype=torch.uint8, device="cuda")
    all_offs, all_lens = [], []
    for k, s in enumerate(sets):
        o = s.export(buf[k * stride:].data_ptr(), stride)
        all_offs.append(o + np.uint64(k * stride))
        all_lens.append(s.lengths())
    torch.cuda.synchronize()
    S = ctx.seqs_from_device_packed(buf.data_ptr(), buf.numel(), np.concatenate(all_offs), np.concatenate(all_lens))
    del buf
    assert S.count == n and [S.get_text(i) for i in (0, 7, 17, 30, 47)] == [texts[i] for i in (0, 7, 17, 30, 47)]
    assert not S.non_acgt
    got, st = ctx.overlap_all(S, mask, 0.30, 32, 64)
    assert [tuple(int(x) for x in r) for r in got] == [tuple(int(x) for x in r) for r in want]
    assert st["n_pairs"] == wst["n_pairs"] and st["n_candidates"] == wst["n_candidates"] and st["n_probe_entries"] == wst["n_probe_entries"]
    # one probe table, many target ranges (pba_probe_table_*): the form ranks and the 10 M-read configuration use
    cap = n * 64
    probes = torch.full((cap,), -1, dtype=torch.int64, device="cuda")
    assert ctx.overlap_probes(S, 0, n, mask, 32, probes.data_ptr(), cap) == st["n_probe_entries"]
    from pacbioassembly_amd import ProbeTable
    table = ProbeTable(ctx, probes.data_ptr(), cap, mask, 32)
    assert table.entries == st["n_probe_entries"]
    parts = [ctx.overlap_all_table(S, table, 0.30, 64, a, b)[0] for a, b in ((0, 5), (5, 6), (6, 48))]
    assert [tuple(int(x) for x in r) for p in parts for r in p] == [tuple(int(x) for x in r) for r in want]
    with pytest.raises(PbaError) as e:                               # a sequence that does not fit the buffer it is said to lie in
        ctx.seqs_from_device_packed(probes.data_ptr(), 64, np.array([60], np.uint64), np.array([100], np.uint32))
    assert e.value.status == -1


@pytest.mark.parametrize("n", [1_250_000, 9_300_000])
def test_packed_set_beyond_4_gib(ctx, n):
    """A sequence set over more than 4 GiB of packed bytes handed over on the device (what pba_seqs_from_device_packed gets
    after the packed-read all-gather of BASELINE configs[4]: 37.6 GB), at two sizes: 4.7 GB, and 35 GB = 4.36 G plane words,
    more than a kernel launch's 32-bit global size.  Every read is the bytes it was given, first to last -- its packed bases
    (get_text), and its BIT PLANES, which is what the aligning kernels read: a read of the big set against the same text
    uploaded on its own aligns at cost 0 on the bit-vector array.  (tools/rehearse_config4.py found both: a device-to-device
    copy of that size goes in pieces of 1 GiB now, and the plane builder walks its words with a grid-stride loop -- a thread
    per word silently built the planes of the first 141 000 reads only, and nine overlaps in ten went missing.)"""
    import torch
    rl = 15000
    pk = ((rl + 3) // 4 + 15) & ~15
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    buf = torch.empty(n * pk, dtype=torch.uint8, device="cuda")
    for lo in range(0, n * pk, 1 << 30):                          # (random bytes are random bases)
        buf[lo:lo + (1 << 30)] = torch.randint(0, 256, (min(1 << 30, n * pk - lo),), dtype=torch.uint8, device="cuda", generator=g)
    torch.cuda.synchronize()                                      # (the engine runs on its own stream)
    offs = np.arange(n, dtype=np.uint64) * np.uint64(pk)
    S = ctx.seqs_from_device_packed(buf.data_ptr(), buf.numel(), offs, np.full(n, rl, np.uint32))
    assert S.count == n and S.packed_bytes == n * pk

    def text_of(i):                                               # dna_seq.h:147-159: first base in bits 7:6
        b = buf[i * pk:i * pk + (rl + 3) // 4].cpu().numpy()
        codes = np.stack([(b >> 6) & 3, (b >> 4) & 3, (b >> 2) & 3, b & 3], 1).reshape(-1)[:rl]
        return np.frombuffer(b"ACGT", np.uint8)[codes].tobytes()
    ids = [0, 1, n // 4, 141_000, 1_142_322, 1_142_323, n - 2, n - 1]      # (read 1 142 322 straddles byte 2^32)
    texts = [text_of(i) for i in ids]
    for i, t in zip(ids, texts):
        assert S.get_text(i) == t, i
    small = ctx.seqs_from_list(texts, strict_acgt=True)           # the same reads with planes of their own
    pairs = np.array([(i, 0, rl, k, 0, rl, 0) for k, i in enumerate(ids)], PAIR_DTYPE)
    res = ctx.align_batch(S, small, pairs, 0.3, kernel=PBA_KERNEL_BITVEC)
    assert [(int(r["rc"]), int(r["cost"])) for r in res] == [(rl, 0)] * len(ids), res
    small.close()
    if n < 2_000_000:                                             # ... and back out through pba_seqs_export
        back = torch.zeros(n * pk, dtype=torch.uint8, device="cuda")
        o = S.export(back.data_ptr(), back.numel())
        assert (o == offs).all()
        for lo in (0, (1 << 32) - 4096, n * pk - 8192):
            assert torch.equal(back[lo:lo + 8192], buf[lo:lo + 8192]), lo
    S.close()


def test_overlap_many_target_ranges_and_the_limits_of_a_call(ctx, oracle, monkeypatch):
    """BASELINE config 5's way through the engine at a size a test can afford: 200 000 short reads, ONE probe table, the
    targets in 25 ranges -- same overlaps and the same number of pairs as one call over everything, the oracle's
    composition on sampled targets of different ranges -- and the limits of a call answer PBA_E_TOOLONG instead of a
    wrapped count: candidates per call (here lowered through the test hook), probe ids beyond 32 bits."""
    import torch
    from pacbioassembly_amd import ProbeTable
    n, rl = 200_000, 500
    g = eng.synth_genome(401, n * rl // 20)
    reads, offs, _ = eng.synth_reads(402, g, n, rl, nthreads=16)
    mask = eng.mask_from_pattern(MASK_PAT)
    S = ctx.seqs_from_text(reads, offs, strict_acgt=True)
    one, st1 = ctx.overlap_all(S, mask, 0.30, 32, 64, cap=n * 80)
    assert st1["n_overlaps"] == len(one) > 500_000 and st1["n_candidates"] > 50_000_000
    many, stm = ctx.overlap_all_sharded(S, mask, 0.30, 32, 64, targets_per_call=8000, cap_per_target=400)
    assert (many == one).all() and stm["n_pairs"] == st1["n_pairs"] and stm["n_candidates"] == st1["n_candidates"]
    texts = lambda i: reads[int(offs[i]):int(offs[i + 1])].tobytes()
    file = b"".join(eng.text2bin(texts(i)) for i in range(n))
    rec_offs = (np.arange(n, dtype=np.uint64) * np.uint64(4 + (rl + 3) // 4))
    for t in (0, 7999, 8000, 123_456, n - 1):
        rows = oracle.spaced_round(texts(t), mask, 0.30, file, rec_offs, 32, 64, buggy=False, nthreads=16)
        exp = [(t, q, int(rows["j"][q]), int(rows["dir"][q]), int(rows["ref_pos"][q]), int(rows["cost"][q]), int(rows["matlen_a"][q]),
                int(rows["matlen_b"][q])) for q in np.nonzero(rows["found"])[0] if q != t]
        lo, hi = np.searchsorted(many["target"], [t, t + 1])
        assert [tuple(int(x) for x in r) for r in many[lo:hi]] == exp and len(exp) >= 1, t
    # a call that would hold more candidates than its offsets can address refuses; ranges below the limit go through
    # (what is held are the LISTED candidates: the survivors of the scan's 32 rows -- a tenth of the candidates of these short reads)
    assert 0 < st1["n_listed"] < st1["n_candidates"] // 5 and st1["n_listed"] + st1["n_prefiltered"] >= st1["n_pairs"]
    monkeypatch.setenv("PBA_OVL_MAX_CANDIDATES", str(st1["n_listed"] // 10))
    with pytest.raises(PbaError) as e:
        ctx.overlap_all(S, mask, 0.30, 32, 64, cap=16)
    assert e.value.status == -4                                      # PBA_E_TOOLONG
    again, sta = ctx.overlap_all_sharded(S, mask, 0.30, 32, 64, targets_per_call=8000, cap_per_target=400)
    assert sta["n_overlaps"] == st1["n_overlaps"]
    monkeypatch.delenv("PBA_OVL_MAX_CANDIDATES")
    # probe ids are 32 bits: reads x 2 x max_trial must stay below 2^32 (refused before anything is read)
    buf = torch.zeros(16, dtype=torch.int64, device="cuda")
    with pytest.raises(PbaError) as e:
        ProbeTable(ctx, buf.data_ptr(), 1 << 32, mask, 32)
    assert e.value.status == -4
    with pytest.raises(PbaError) as e:                               # max_trial beyond what a candidate's 7 probe bits hold
        ProbeTable(ctx, buf.data_ptr(), 16, mask, 64)
    assert e.value.status == -1


def test_locate_baseline_config1_shape_256_reads_vs_oracle(ctx, oracle):
    """BASELINE configs[1] at its own shape (15 kb reads @15 % against the 5 Mb genome, R = 0.30, 50 probe offsets), 256
    reads of it: every row, the pair count and the band-cell count of every read against the oracle on the host cores."""
    g = eng.synth_genome(2, 5_000_000)
    reads, offs, _ = eng.synth_reads(3, g, 256, 15_000, 0.05, 0.05, 0.05, nthreads=16)
    mask = eng.mask_from_pattern(MASK_PAT)
    T = ctx.seqs_from_text(g, np.array([0, g.size], np.uint64), strict_acgt=True)
    Rd = ctx.seqs_from_text(reads, offs, strict_acgt=True)
    ix = ctx.index_build(T, 0, mask, PBA_INDEX_ALL)
    oracle.prefault(16, 15_000, 0.30)
    want, wst = oracle.locator(g, mask, 0.30, reads, offs, 50, 500, nthreads=16)
    oracle.release()
    rows, st = ctx.locate(ix, T, 0, Rd, 0.30, 50, 500)
    for c in ("nseq", "found", "j", "pos", "cost", "seglen", "matlen_a", "matlen_b", "n_pairs"):
        assert (rows[c] == want[c]).all(), c
    assert st == wst and wst["n_located"] > 180 and wst["n_pairs"] > 1500 and wst["n_cells"] > 2 * 10 ** 10
    assert (rows["diag_cost"][rows["found"] == 1] >= rows["cost"][rows["found"] == 1]).all()


def test_overlap_all_parks_and_resumes_uncertified_runs(ctx, oracle, prekeep):
    """6.5 kb reads at ~12 % error each overlap at ~24 % between them: the first-pass window (1 384 in a one-block ring)
    cannot certify the longest true overlaps, the (target, query) run is parked and resumed at the reference band by
    the second launch -- and the answer is still the oracle's composition, with the number of pairs aligned equal to
    the oracle's count."""
    g = eng.synth_genome(81, 8000)
    n, rl = 24, 6500
    reads, offs, _ = eng.synth_reads(82, g, n, rl, 0.04, 0.04, 0.04)
    texts = [reads[int(offs[i]):int(offs[i + 1])].tobytes() for i in range(n)]
    file = b"".join(eng.text2bin(t) for t in texts)
    rec_offs = np.cumsum([0] + [4 + (len(t) + 3) // 4 for t in texts[:-1]]).astype(np.uint64)
    mask = eng.mask_from_pattern(MASK_PAT)
    want, pairs = [], 0
    for t in range(n):
        rows = oracle.spaced_round(texts[t], mask, 0.30, file, rec_offs, 32, 64, buggy=False, nthreads=8)
        pairs += int(rows["n_pairs"].sum()) - int(rows["n_pairs"][t])
        for q in range(n):
            if q != t and rows["found"][q]:
                want.append((t, q, int(rows["j"][q]), int(rows["dir"][q]), int(rows["ref_pos"][q]), int(rows["cost"][q]),
                             int(rows["matlen_a"][q]), int(rows["matlen_b"][q])))
    S = ctx.seqs_from_list(texts, strict_acgt=True)
    got, st = ctx.overlap_all(S, mask, 0.30, 32, 64, kernel=PBA_KERNEL_BITVEC)
    assert [tuple(int(x) for x in r) for r in got] == want
    assert len(want) > 60 and st["n_redo"] > 5
    assert st["n_pairs"] == pairs


def test_overlap_all_cascade_through_the_middle_ring(ctx, oracle, monkeypatch, prekeep):
    """14 kb reads at 15 % error each (~27 % between two of them): max_dst = 4 201 puts the narrow window in a two-block
    ring (window 2 729) and the reference band in a four-block one, so the runs the first stage parks are resumed in
    the three-block ring (window 4 072: all the room it has) and only what that cannot certify goes on to the reference
    band -- same overlaps and pair counts as the row-sweep kernel (which has no windows at all), with and without the
    sampled decision, and as the oracle's composition for the first targets."""
    g = eng.synth_genome(301, 60000)
    n, rl = 36, 14000
    reads, offs, _ = eng.synth_reads(302, g, n, rl, 0.05, 0.05, 0.05)
    texts = [reads[int(offs[i]):int(offs[i + 1])].tobytes() for i in range(n)]
    mask = eng.mask_from_pattern(MASK_PAT)
    S = ctx.seqs_from_list(texts, strict_acgt=True)
    want, st_rs = ctx.overlap_all(S, mask, 0.30, 32, 64, kernel=PBA_KERNEL_ROWSWEEP)
    assert len(want) > 30
    for sample in (None, "2"):
        if sample:
            monkeypatch.setenv("PBA_OVL_SAMPLE_MIN", sample)         # the rest of the items goes through the sampled decision
        got, st = ctx.overlap_all(S, mask, 0.30, 32, 64, kernel=PBA_KERNEL_BITVEC)
        assert [tuple(int(x) for x in r) for r in got] == [tuple(int(x) for x in r) for r in want], sample
        assert st["n_pairs"] == st_rs["n_pairs"] and st["n_redo"] > 3
    file = b"".join(eng.text2bin(t) for t in texts)
    rec_offs = np.cumsum([0] + [4 + (len(t) + 3) // 4 for t in texts[:-1]]).astype(np.uint64)
    for t in range(3):
        rows = oracle.spaced_round(texts[t], mask, 0.30, file, rec_offs, 32, 64, buggy=False, nthreads=16)
        exp = [(t, q, int(rows["j"][q]), int(rows["dir"][q]), int(rows["ref_pos"][q]), int(rows["cost"][q]), int(rows["matlen_a"][q]),
                int(rows["matlen_b"][q])) for q in range(n) if q != t and rows["found"][q]]
        assert [tuple(int(x) for x in r) for r in got if int(r["target"]) == t] == exp, t


# ----------------------------------------------------------------------------- consensus (ref_seq, unlocked)
class GpuCons:
    """ref_seq::try_align (ref_seq.h:259-276) composed from the C ABI: align + edit script on the GPU, votes and
    growth in the device-resident vote boxes (pba_cons_*).  Same face as the oracle's / reference's objects."""

    def __init__(self, ctx, text, weight):
        self.ctx = ctx
        self.c = eng.Consensus(ctx, text, weight)

    def try_align(self, pos, seg, fwd, R=0.3):
        e = self.c.extent()
        text = self.c.text()
        at = pos - e[0]                                         # index of reference position `pos` inside [pre, post)
        a = text[at:] if fwd else text[:at + 1]                 # get_accessor, ref_seq.h:282-286
        res, ops = self.ctx.align_text_trace(a, seg, R, fwd, fwd, maxn=26000, maxm=6000)   # t_aligner; a = the reference
        ok = int(res["rc"]) >= 0 and int(res["matlen_a"]) >= 64
        if ok:
            self.c.elect([pos], [fwd], [ops], [eng.script_vals(ops, seg, fwd)])
            if int(res["matlen_a"]) == len(a):
                add = len(seg) - int(res["matlen_b"])
                if fwd:
                    self.c.append(seg[len(seg) - add:] if add else b"")
                else:
                    self.c.prepend(seg[:add])
        e = self.c.extent()
        return {"ok": int(ok), "matlen_b": int(res["matlen_b"]) if ok else 0, "cost": int(res["cost"]) if ok else 0,
                "matlen_a": int(res["matlen_a"]) if ok else 0, "nedit": int(ops.size) if ok else 0, "pre": e[0], "post": e[1]}

    def dump(self):
        return self.c.dump()

    def text(self):
        return self.c.text()

    def evolve(self):
        self.c.evolve()


def test_consensus_golden_and_oracle(ctx, oracle):
    """Votes (k_cons_elect), growth and evolve (k_cons_evolve) against what the reference itself produced
    (tests/golden/consensus.json) and, on a fresh scenario, against the oracle; then a batch elect of all scripts of a
    round at once against the one-by-one result (votes commute)."""
    from cons_scenarios import SCENARIOS, round_tries, run_scenario, scenario_inputs
    gold = {g["name"]: g for g in gold_json("consensus.json")}
    for sc in SCENARIOS:
        text, weight, reads = scenario_inputs(sc)
        got = run_scenario(GpuCons(ctx, text, weight), reads)
        for k, (a, b) in enumerate(zip(got["rounds"], gold[sc[0]]["rounds"])):
            assert a["tries"] == b["tries"], (sc[0], k)
            assert a["before_evolve"] == b["before_evolve"], (sc[0], k)
            assert a["after_evolve"] == b["after_evolve"], (sc[0], k)
    sc = ("fresh_gpu", 141, 142, 7000, 1500, 3200, 70, 1100, 2, (0.06, 0.04, 0.04), True)
    text, weight, reads = scenario_inputs(sc)
    assert run_scenario(GpuCons(ctx, text, weight), reads) == run_scenario(oracle.consensus(text, weight), reads)
    # batch form: every forward script of round 0 in one pba_cons_elect call (no growth: interior tries only)
    one, many = eng.Consensus(ctx, text, weight), eng.Consensus(ctx, text, weight)
    pos, fw, scripts, vals = [], [], [], []
    for hit, r, seg, fwd in round_tries(text, reads, 0):
        a = text[hit:] if fwd else text[:hit + 1]
        res, ops = ctx.align_text_trace(a, seg, 0.3, fwd, fwd)
        if int(res["rc"]) < 0:
            continue
        one.elect([hit], [fwd], [ops], [eng.script_vals(ops, seg, fwd)])
        pos.append(hit); fw.append(fwd); scripts.append(ops); vals.append(eng.script_vals(ops, seg, fwd))
    assert len(pos) >= 8
    many.elect(pos, fw, scripts, vals)
    for x, y in zip(one.dump()[:3], many.dump()[:3]):
        assert (x == y).all()
    assert one.evolve() == many.evolve()


def test_kernels_agree_on_drifting_paths():
    """tools/stress_kernels.py (indel-biased pairs whose optimal path drifts towards the window edges, error rates up to
    the acceptance limit, tails, all direction combinations): the full-band row sweep and the bit-vector array with its
    prefilter, asymmetric windows and certificates agree bit for bit."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_kernels.py"), "--batches", "3", "--pairs", "200",
                        "--max-len", "6000", "--seed", "3"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("same=True") == 3


def test_spaced_multi_golden(ctx):
    """pba_spaced_multi (spaced_seed.cpp:409-452 for a locked reference) against the chain of the reference's own
    locked rounds (tests/golden/spaced_multi.json): same seed per round, same reads found per round, same rows."""
    from cons_scenarios import MULTI, multi_inputs
    gold = gold_json("spaced_multi.json")
    text, file, rec_offs = multi_inputs()
    Rf = ctx.seqs_from_list([text])
    Rd = ctx.seqs_from_records(file, 0, 1 << 30)
    assert Rd.count == rec_offs.size
    rows, fr, log = ctx.spaced_multi(Rf, 0, Rd, MULTI["R"], gold["masks"], MULTI["picks"], MULTI["max_round"], MULTI["max_trial"],
                                     MULTI["overlap_min"], buggy_seed_at=True)
    assert [[l["round"], l["mask"], l["n_tried"], l["n_found"]] for l in log] == gold["log"]
    assert fr.tolist() == gold["found_round"]
    for r, want in enumerate(gold["final"]):
        assert int(rows["found"][r]) == want[0]
        if want[0]:
            assert [int(rows[c][r]) for c in ("found", "j", "dir", "ref_pos", "cost", "matlen_a", "matlen_b")] == want, r


class GpuAsm:
    """The face run_assembly drives (round / evolve / dump / text) over pba_cons_round and the device vote boxes."""

    def __init__(self, ctx, text, weight, reads, kernel=eng.PBA_KERNEL_AUTO):
        self.c = eng.Consensus(ctx, text, weight, max_len=100000)
        self.reads, self.kernel = reads, kernel
        self.stats = []

    def round(self, mask, R, max_trial, file, rec_offs, pool, buggy=True):
        rows, st = self.c.round(self.reads, pool, mask, R, max_trial, 64, buggy_seed_at=buggy, kernel=self.kernel)
        self.stats.append(st)
        return rows, st["n_found"]

    def evolve(self):
        self.c.evolve()

    def dump(self):
        return self.c.dump()

    def text(self):
        return self.c.text()


def test_assemble_unlocked_rounds_golden(ctx):
    """pba_cons_round (an unlocked round of spaced_seed.cpp:420-446 as a few device batches: every pending read walked
    at once, rows accepted in pool order until a growth they depend on, votes straight from the traceback walk) against
    the reference's own serial loop on its ref_seq (tests/golden/assemble.json): per round the same reads found with the
    same rows, the same probe and pair counts, the same extent and vote boxes before evolve, the same evolved text --
    and the same 13 kb assembly grown from a 3 kb slice."""
    from cons_scenarios import assemble_inputs, run_assembly
    gold = gold_json("assemble.json")
    text, weight, file, rec_offs, texts = assemble_inputs()
    Rd = ctx.seqs_from_records(file, 0, 1 << 30)
    assert Rd.count == len(texts)
    for kernel in KERNELS:
        asm = GpuAsm(ctx, text, weight, Rd, kernel)
        got = run_assembly(asm, gold["masks"], file, rec_offs, len(texts))
        for a, b in zip(got["rounds"], gold["rounds"]):
            assert a == b, (kernel, a["round"], {k: (a[k], b[k]) for k in a if a[k] != b[k] and k != "found"})
        assert got["final_text"] == gold["final_text"]
        # the batching really happened: growth on both ends, reads put back behind it, far fewer launches than reads
        assert sum(s["n_grown_fwd"] for s in asm.stats) >= 5 and sum(s["n_grown_bwd"] for s in asm.stats) >= 5
        assert sum(s["n_deferred"] for s in asm.stats) > 0
        assert sum(s["n_batches"] for s in asm.stats) < sum(len(r["found"]) for r in gold["rounds"])


def test_assemble_whole_loop_golden(ctx):
    """pba_cons_assemble (the whole of spaced_seed.cpp:409-452 without -l behind one call) == the golden chain: seeds,
    pool sizes, matches and reference length per round, the round each read was found in, the final text."""
    from cons_scenarios import ASSEMBLE, assemble_inputs
    gold = gold_json("assemble.json")
    text, weight, file, rec_offs, texts = assemble_inputs()
    Rd = ctx.seqs_from_records(file, 0, 1 << 30)
    c = eng.Consensus(ctx, text, weight, max_len=100000)
    rows, fr, log = c.assemble(Rd, ASSEMBLE["R"], gold["masks"], ASSEMBLE["picks"], ASSEMBLE["max_round"], ASSEMBLE["max_trial"],
                               ASSEMBLE["overlap_min"], buggy_seed_at=True)
    assert [[l["round"], l["mask"], l["n_tried"], l["n_found"], l["ref_len"]] for l in log] == \
        [[r["round"], r["mask"], r["n_tried"], len(r["found"]), r["ref_len"]] for r in gold["rounds"]]
    want_round = {f[0]: r["round"] for r in gold["rounds"] for f in r["found"]}
    assert fr.tolist() == [want_round.get(i, 0) for i in range(len(texts))]
    for r in gold["rounds"]:
        for f in r["found"]:
            assert [int(rows[c_][f[0]]) for c_ in ("read", "j", "dir", "ref_pos", "cost", "matlen_a", "matlen_b")] == f
    assert c.text().decode() == gold["final_text"]


def test_assemble_fresh_inputs_vs_oracle(ctx, oracle):
    """Unlocked rounds on inputs no golden holds (other seeds, an insertion-heavy error mix, the intended seed_at as well
    as the reference's): GPU == oracle, round by round."""
    from cons_scenarios import ASSEMBLE, assemble_inputs, run_assembly
    masks = [oracle.mask_from_pattern(p) for p in ("111*11*11*1*1111", "1111*1*11**11*111", "11*1111**1*11*111")]
    cfg = dict(ASSEMBLE, genome_seed=81, reads_seed=82, foreign_seed=83, err=(0.02, 0.08, 0.03), slice=(3000, 2500), max_round=6)
    text, weight, file, rec_offs, texts = assemble_inputs(cfg)
    Rd = ctx.seqs_from_records(file, 0, 1 << 30)
    for buggy in (True, False):
        class O:                                   # the oracle with this seed_at
            def __init__(s): s.c = oracle.consensus(text, weight)
            def round(s, mask, R, mt, f, ro, pool): return s.c.round(mask, R, mt, f, ro, pool, buggy=buggy)
            def evolve(s): s.c.evolve()
            def dump(s): return s.c.dump()
            def text(s): return s.c.text()
        want = run_assembly(O(), masks, file, rec_offs, len(texts), cfg)
        asm = GpuAsm(ctx, text, weight, Rd)
        asm_round = asm.round
        asm.round = lambda mask, R, mt, f, ro, pool: asm_round(mask, R, mt, f, ro, pool, buggy=buggy)
        got = run_assembly(asm, masks, file, rec_offs, len(texts), cfg)
        for a, b in zip(got["rounds"], want["rounds"]):
            assert a == b, (buggy, a["round"], {k: (a[k], b[k]) for k in a if a[k] != b[k] and k != "found"})
        assert got["final_text"] == want["final_text"] and sum(len(r["found"]) for r in want["rounds"]) > 40


def test_edge_cases_of_the_widened_entry_points(ctx, oracle):
    """Empty / degenerate inputs of the traceback, consensus, multi-round and overlap entry points: defined answers or
    status codes, never a crash."""
    S = ctx.seqs_from_list([b"ACGT" * 50, b"ACGTTGCA" * 20, b"", b"A"])
    # empty batch, empty sequences, single bases through the trace kernels
    out, scripts = ctx.align_batch_trace(S, S, np.zeros(0, PAIR_DTYPE), 0.3)
    assert out.size == 0 and scripts == []
    pairs = np.array([(2, 0, 0, 2, 0, 0, 0), (3, 0, 1, 3, 0, 1, 0), (2, 0, 0, 3, 0, 1, 0), (0, 0, 200, 0, 0, 200, 0)], PAIR_DTYPE)
    for kernel in KERNELS:
        out, scripts = ctx.align_batch_trace(S, S, pairs, 0.3, kernel=kernel)
        for pr, got, ops in zip(pairs, out, scripts):
            a = S.get_text(int(pr["a_seq"]))[:int(pr["a_len"])]; b = S.get_text(int(pr["b_seq"]))[:int(pr["b_len"])]
            exp = oracle.align(a, b, 0.3, want_ops=True)
            check_result(got, exp, (tuple(pr), kernel))
            assert ops.tolist() == exp["ops"].tolist(), (tuple(pr), kernel)
    with pytest.raises(PbaError) as e:                      # ops_off must leave a_len + b_len slots: checked, not trusted
        off = np.array([0, 10], np.uint64); ne = np.zeros(1, np.int32); o = np.zeros(1, eng.RESULT_DTYPE); ops = np.zeros(16, np.uint8)
        pr = np.array([(0, 0, 200, 0, 0, 200, 0)], PAIR_DTYPE)
        ctx.check(ctx.lib.pba_align_batch_trace(ctx.h, S.h, S.h, pr.ctypes.data, 1, 0.3, 0, 0, 0, o.ctypes.data, ops.ctypes.data,
                                                off.ctypes.data, ne.ctypes.data), "trace")
    assert e.value.status == -1
    # consensus: an empty reference, votes outside it, evolve of nothing
    c = eng.Consensus(ctx, b"", 1, max_len=1000)
    assert c.extent() == [0, 0, 0] and c.text() == b"" and c.evolve() == b""
    c.append(b"ACGT")
    assert c.text() == b"ACGT" and c.extent() == [0, 4, 0]
    with pytest.raises(PbaError) as e:
        c.elect([7], [True], [np.array([1], np.uint8)], [b"A"])      # "pos should be contained" (ref_seq.h:351)
    assert e.value.status == -1
    with pytest.raises(PbaError) as e:
        eng.Consensus(ctx, b"ACGT" * 10, 1, max_len=8)               # longer than max_len
    assert e.value.status == -1
    c2 = eng.Consensus(ctx, b"ACGT", 1, max_len=4)
    with pytest.raises(PbaError) as e:
        c2.prepend(b"ACGTA")                                         # would grow past the buffer before the origin
    assert e.value.status == -4
    # votes that delete everything: evolve leaves an empty reference (every box invalid, nothing before the first to absorb)
    c3 = eng.Consensus(ctx, b"ACGT", 1, max_len=100)
    for _ in range(3):
        c3.elect([0], [True], [np.array([3, 3, 3, 3], np.uint8)], [b"\\0\\0\\0\\0"])
    assert c3.evolve() == b"" and c3.extent() == [0, 0, 0]
    # multi-round driver with nothing to do, overlap of a single read
    T = ctx.seqs_from_list([b"ACGT" * 200])
    empty = ctx.seqs_from_list([])
    rows, fr, log = ctx.spaced_multi(T, 0, empty, 0.3, [0xFFCCF3FC], [0], 5)
    assert rows.size == 0 and fr.size == 0 and len(log) == 1 and log[0]["n_tried"] == 0     # one round, no match, one seed: stop
    one = ctx.seqs_from_list([b"ACGT" * 300])
    ov, st = ctx.overlap_all(one, 0xFFCCF3FC, 0.3, 8, 64)
    assert ov.size == 0 and st["n_overlaps"] == 0
    # unlocked rounds: an empty pool, a reference shorter than a seed, an empty reference, a read that IS the reference
    rd = eng.synth_genome(5, 900).tobytes()
    Rd = ctx.seqs_from_records(eng.text2bin(rd) + eng.text2bin(rd[100:800]), 0, 1 << 30)
    c4 = eng.Consensus(ctx, rd, 1, max_len=5000)
    rows, st = c4.round(Rd, [], 0xFFCCF3FC, 0.3)
    assert rows.size == 0 and st["n_found"] == 0 and st["n_batches"] == 0
    for tiny in (b"ACGTACGTAC", b""):
        rows, st = eng.Consensus(ctx, tiny, 1, max_len=5000).round(Rd, [0, 1], 0xFFCCF3FC, 0.3)
        assert not rows["found"].any() and st["n_index"] == 0
    rows, st = c4.round(Rd, [1, 0], 0xFFCCF3FC, 0.3)                   # exact copies: found at j = 0, cost 0, nothing grows
    assert rows["found"].tolist() == [1, 1] and rows["cost"].tolist() == [0, 0] and rows["ref_pos"].tolist() == [100, 0]
    assert st["n_grown_fwd"] + st["n_grown_bwd"] <= 1 and c4.extent() == [0, 900, 900]
    sel, sup, tot, _ = c4.dump()
    assert tot[:100].tolist() == [2] * 100 and tot[100:800].tolist() == [3] * 700 and c4.evolve() == rd
    rows, fr, log = eng.Consensus(ctx, rd, 1, max_len=5000).assemble(empty, 0.3, [0xFFCCF3FC], [0], 5)
    assert rows.size == 0 and len(log) == 1


def test_cons_vote_pairs_equals_scripts_then_elect(ctx):
    """pba_cons_vote_pairs (sweep, walk and vote on the device, no script in memory) == edit scripts from
    pba_align_batch_trace applied with pba_cons_elect behind the OVERLAP_MIN gate: same boxes, same evolved text; forward
    and backward tries, reads running off both ends of the reference (their votes past it are dropped alike)."""
    from cons_scenarios import round_tries, scenario_inputs
    sc = ("vote", 151, 152, 8000, 2000, 3500, 120, 1300, 2, (0.05, 0.05, 0.05), True)
    text, weight, reads = scenario_inputs(sc)
    A = ctx.seqs_from_list([b"ACGT" * 10, text], strict_acgt=True)       # the reference is sequence 1 of its set
    B = ctx.seqs_from_list(reads, strict_acgt=True)
    pairs = []
    for rnd in (0, 1):
        for hit, r, seg, fwd in round_tries(text, reads, rnd):
            if fwd:
                pairs.append((1, hit, len(text) - hit, r, len(reads[r]) - len(seg), len(seg), 0))
            else:
                pairs.append((1, hit, hit + 1, r, len(seg) - 1, len(seg), 3))
    pairs = np.array(pairs, PAIR_DTYPE)
    assert pairs.size >= 30 and (pairs["flags"] == 3).sum() >= 8
    one, many = eng.Consensus(ctx, text, weight), eng.Consensus(ctx, text, weight)
    out, scripts = ctx.align_batch_trace(A, B, pairs, 0.3)
    n_voted = 0
    for pr, res, ops in zip(pairs, out, scripts):
        if int(res["rc"]) < 0 or int(res["matlen_a"]) < 64:
            continue
        fwd = int(pr["flags"]) == 0
        rd = reads[int(pr["b_seq"])]
        seg = rd[int(pr["b_pos"]):] if fwd else rd[:int(pr["b_pos"]) + 1]
        one.elect([int(pr["a_pos"])], [fwd], [ops], [eng.script_vals(ops, seg, fwd)])
        n_voted += 1
    out2 = many.vote_pairs(A, 1, B, pairs, 0.3, 64)
    assert n_voted >= 20
    for c in ("rc", "cost", "matlen_a", "matlen_b"):
        assert (out[c] == out2[c]).all(), c
    for x, y in zip(one.dump()[:3], many.dump()[:3]):
        assert (x == y).all()
    assert one.evolve() == many.evolve()
    with pytest.raises(PbaError) as e:                               # a must be the reference of these boxes
        many.vote_pairs(A, 0, B, pairs[:1], 0.3, 64)
    assert e.value.status == -1


def test_overlap_all_with_tandem_repeats(ctx, oracle, prekeep):
    """Reads that share a tandem repeat: one probe hits hundreds of positions of every other read, so a (target, query)
    run is thousands of candidates long -- it spans many 64-candidate work items (the owner rule), a target's list
    outgrows the LDS sort, and the first success sits deep inside a run.  Same answer as the oracle's composition."""
    rng = np.random.RandomState(5)
    alpha = np.frombuffer(b"ACGT", np.uint8)
    unit = alpha[rng.randint(0, 4, 23)].tobytes()
    g = eng.synth_genome(91, 6000).tobytes()
    genome = g[:2000] + unit * 70 + g[2000:]                      # 1 610 bases of period 23 inside a 7.6 kb genome
    garr = np.frombuffer(genome.encode() if isinstance(genome, str) else genome, np.uint8)
    reads, offs, _ = eng.synth_reads(92, garr, 24, 2600, 0.02, 0.02, 0.02)
    texts = [reads[int(offs[i]):int(offs[i + 1])].tobytes() for i in range(24)]
    file = b"".join(eng.text2bin(t) for t in texts)
    rec_offs = np.cumsum([0] + [4 + (len(t) + 3) // 4 for t in texts[:-1]]).astype(np.uint64)
    mask = eng.mask_from_pattern(MASK_PAT)
    want, pairs = [], 0
    for t in range(24):
        rows = oracle.spaced_round(texts[t], mask, 0.30, file, rec_offs, 32, 64, buggy=False, nthreads=8)
        pairs += int(rows["n_pairs"].sum()) - int(rows["n_pairs"][t])
        for q in range(24):
            if q != t and rows["found"][q]:
                want.append((t, q, int(rows["j"][q]), int(rows["dir"][q]), int(rows["ref_pos"][q]), int(rows["cost"][q]),
                             int(rows["matlen_a"][q]), int(rows["matlen_b"][q])))
    S = ctx.seqs_from_list(texts, strict_acgt=True)
    for kernel in KERNELS:
        got, st = ctx.overlap_all(S, mask, 0.30, 32, 64, kernel=kernel)
        assert [tuple(int(x) for x in r) for r in got] == want, kernel
        assert st["n_pairs"] == pairs and st["n_candidates"] > 20 * st["n_pairs"] / 10
    assert len(want) > 100 and pairs > 3000


def test_locator_gpu_example_prints_what_the_reference_locator_prints(lib, tmp_path):
    """examples/locator_gpu.cpp (the reference's `locator` command line over the C ABI, plain g++) against the stdout of
    the reference's own main, compiled unmodified and run on the same files (tests/golden/locator_cli.json): same TSV,
    columns 1-4, including the ids that skip the reads shorter than 500 bases."""
    import os
    import subprocess
    from conftest import ROOT
    from cons_scenarios import LOCATOR_CLI, locator_cli_inputs
    gold = gold_json("locator_cli.json")
    out = os.path.join(ROOT, "tests", "cpp", "_build")
    os.makedirs(out, exist_ok=True)
    exe = os.path.join(out, "locator_gpu")
    libdir = os.path.join(ROOT, "pacbioassembly_amd", "lib")
    subprocess.run(["g++", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"), "-o", exe,
                    os.path.join(ROOT, "examples", "locator_gpu.cpp"), "-L", libdir, "-lpba", f"-Wl,-rpath,{libdir}"], check=True)
    contig, texts = locator_cli_inputs()
    cf = tmp_path / "contig.txt"
    cf.write_bytes(contig + b"\n")
    r = subprocess.run([exe, str(cf), LOCATOR_CLI["pattern"]], input=b"\n".join(texts) + b"\n", capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode()
    rows = [[int(x) for x in line.split()] for line in r.stdout.decode().splitlines()]
    assert [x[:4] for x in rows] == gold["rows"] and len(rows) > 250
    assert [x[4] for x in rows] == gold["col5"] and min(gold["col5"]) >= 0      # get_cost(len - j, len - j), locator.cpp:86


def test_reference_mains_linked_against_compat_print_what_they_print_on_the_cpu(lib, tmp_path):
    """The drop-in claim itself: the reference's own locator.cpp and spaced_seed.cpp, UNMODIFIED, compiled against
    include/compat/ instead of its src/ headers and linked with libpba.so (oracle/Makefile: locator_compat,
    spaced_seed_compat -- built in the container where /root/reference lies, the binaries travel like the other checkers)
    print on the GPU what the same sources print with their own headers on the CPU: all five TSV columns of locator, the
    consensus after every round / the found lines / the dump file of spaced_seed."""
    import os
    import subprocess
    from conftest import ROOT
    from cons_scenarios import LOCATOR_CLI, locator_cli_inputs, run_spaced_seed_cli
    exe = os.path.join(ROOT, "oracle", "_ref", "locator_compat")
    exe2 = os.path.join(ROOT, "oracle", "_ref", "spaced_seed_compat")
    if not (os.path.exists(exe) and os.path.exists(exe2)):
        pytest.skip("oracle/_ref/*_compat are built only where /root/reference exists")
    # they are the engine's clients, not the stock CPU programs (round 2's were: quote includes resolve beside the source
    # first -- oracle/Makefile feeds the sources on stdin now): libpba.so is needed, and without a usable device they stop
    for e in (exe, exe2):
        assert "libpba.so" in subprocess.run(["readelf", "-d", e], capture_output=True, text=True, check=True).stdout
        assert "pba_align_text_trace" in subprocess.run(["nm", "-D", "--undefined-only", e], capture_output=True, text=True, check=True).stdout
    gold = gold_json("locator_cli.json")
    contig, texts = locator_cli_inputs()
    cf = tmp_path / "contig.txt"
    cf.write_bytes(contig + b"\n")
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "pacbioassembly_amd", "lib") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    bad = subprocess.run([exe, str(cf), LOCATOR_CLI["pattern"]], input=b"\n".join(texts[:30]) + b"\n", capture_output=True, timeout=600,
                         env=dict(env, PBA_DEVICE="4096"))
    assert bad.returncode != 0 and bad.stdout == b"" and b"cannot create a device context" in bad.stderr
    r = subprocess.run([exe, str(cf), LOCATOR_CLI["pattern"]], input=b"\n".join(texts) + b"\n", capture_output=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    rows = [[int(x) for x in line.split()] for line in r.stdout.decode().splitlines()]
    assert [x[:4] for x in rows] == gold["rows"] and [x[4] for x in rows] == gold["col5"]
    gold2 = gold_json("spaced_seed_cli.json")["runs"]
    os.environ["LD_LIBRARY_PATH"] = env["LD_LIBRARY_PATH"]
    got = run_spaced_seed_cli(exe2, str(tmp_path), [])
    for name in gold2:
        assert got[name] == gold2[name], (name, [k for k in gold2[name] if got[name][k] != gold2[name][k]])


def test_spaced_seed_gpu_example_prints_what_the_reference_spaced_seed_prints(lib, tmp_path):
    """examples/spaced_seed_gpu.cpp (the reference's `spaced_seed` command line over the C ABI, plain g++) against the
    reference's own main, compiled unmodified and run on the same files (tests/golden/spaced_seed_cli.json): the same
    consensus on stdout after every round, the same `found` lines in the log, the same -d dump file -- unlocked (the
    assembly: votes, growth, evolve), unlocked at another ratio / trial count, and locked."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    from cons_scenarios import run_spaced_seed_cli
    gold = gold_json("spaced_seed_cli.json")["runs"]
    out = os.path.join(ROOT, "tests", "cpp", "_build")
    os.makedirs(out, exist_ok=True)
    exe = os.path.join(out, "spaced_seed_gpu")
    libdir = os.path.join(ROOT, "pacbioassembly_amd", "lib")
    subprocess.run(["g++", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"), "-o", exe,
                    os.path.join(ROOT, "examples", "spaced_seed_gpu.cpp"), "-L", libdir, "-lpba", f"-Wl,-rpath,{libdir}"], check=True)
    got = run_spaced_seed_cli(exe, str(tmp_path), [])
    for name in gold:
        assert got[name] == gold[name], (name, {k: (got[name][k], gold[name][k]) for k in gold[name] if got[name][k] != gold[name][k] and k != "last_consensus"})


def test_assemble_tool_other_configs_vs_oracle():
    """tools/bench_assemble.py with the oracle beside every round (rows, probe / pair counts, vote boxes, evolved text):
    other genome sizes, read lengths, error rates and trial counts than the goldens hold; a random read order."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    for args in (["--genome", "40000", "--reads", "250", "--read-len", "2500", "--err", "0.12", "--start-len", "6000", "--rounds", "6",
                  "--check-rounds", "6", "--seed", "11"],
                 ["--genome", "30000", "--reads", "200", "--read-len", "1800", "--err", "0.18", "--start-len", "4000", "--rounds", "5",
                  "--check-rounds", "5", "--trials", "20", "--seed", "12"]):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_assemble.py")] + args, capture_output=True, text=True,
                           timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        recs = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
        assert recs[-1]["checked_same"] and all(x["same_as_oracle"] for x in recs[:-1]), r.stdout
        assert recs[-1]["found"] >= 10 and sum(x["batches"] for x in recs[:-1]) > len(recs) - 1


def test_locate_random_configs_vs_oracle():
    """tools/stress_locate.py: random genome sizes, ragged read lengths, error mixes up to the acceptance limit, R from 0.1
    to 0.45, 10 or 50 probe offsets, both kernels -- rows and counted pairs / cells equal the oracle's."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_locate.py"), "--rounds", "4", "--seed", "5"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("same=True") == 4


def test_overlap_later_ranges_of_a_table_skip_the_count_pass(ctx, monkeypatch):
    """After the census of one range of >= 1 024 targets the next ranges against the same probe table get equal room per target
    (1.25 x the largest need seen) instead of a census launch; a target that outgrows its room sends the range through the
    scan again with exact slices.  2 600 short reads in ranges of 1 300 targets: same rows and counts as the row-sweep kernel
    in one call -- with the room as sized, with room for half the largest need (every later range overflows and is redone),
    and with the mode switched off."""
    g = eng.synth_genome(401, 260000)
    n, rl = 2600, 1500
    reads, offs, _ = eng.synth_reads(402, g, n, rl, 0.02, 0.02, 0.02)
    S = ctx.seqs_from_text(reads, offs, strict_acgt=True)
    mask = eng.mask_from_pattern(MASK_PAT)
    want, wst = ctx.overlap_all(S, mask, 0.30, 32, 64, kernel=PBA_KERNEL_ROWSWEEP)
    assert len(want) > 10000
    for pct, n_cap, n_over in (("125", 1, 0), ("50", 1, 1), ("0", 0, 0)):
        monkeypatch.setenv("PBA_OVL_CAPFILL_PCT", pct)
        got, st = ctx.overlap_all_sharded(S, mask, 0.30, 32, 64, targets_per_call=1300, kernel=PBA_KERNEL_BITVEC)
        assert (got == want).all() and st["n_pairs"] == wst["n_pairs"] and st["n_candidates"] == wst["n_candidates"], pct
        assert (st["cap_fill"] + st["cap_overflow"] > 0) == bool(n_cap) and (st["cap_overflow"] > 0) == bool(n_over), (pct, st)


@pytest.mark.parametrize("pattern", ["1111111111111111", "111111111111111*", "1111111*11111111"])
def test_overlap_all_heavy_masks_hashed_table(ctx, oracle, pattern, prekeep):
    """Masks with more than 26 care bits: the probe table is a 2^26-bucket hash with the probes' keys stored beside them
    (overlap.h: HASHED) -- a bucket can hold probes of other keys, which the scan must skip.  300 reads at 3 % error (whole
    16-mers have to match): the bit-vector form against the row-sweep form, and both against the oracle's locked rounds."""
    g = eng.synth_genome(611, 60000)
    n, rl = 300, 2500
    reads, offs, _ = eng.synth_reads(612, g, n, rl, 0.01, 0.01, 0.01)
    texts = [reads[int(offs[i]):int(offs[i + 1])].tobytes() for i in range(n)]
    file = b"".join(eng.text2bin(t) for t in texts)
    rec_offs = np.cumsum([0] + [4 + (len(t) + 3) // 4 for t in texts[:-1]]).astype(np.uint64)
    mask = eng.mask_from_pattern(pattern)
    assert bin(mask).count("1") > 26
    S = ctx.seqs_from_list(texts, strict_acgt=True)
    want, wst = ctx.overlap_all(S, mask, 0.30, 32, 64, kernel=PBA_KERNEL_ROWSWEEP)
    got, st = ctx.overlap_all(S, mask, 0.30, 32, 64, kernel=PBA_KERNEL_BITVEC)
    assert len(want) > 1500 and (got == want).all()
    assert st["n_pairs"] == wst["n_pairs"] and st["n_candidates"] == wst["n_candidates"] and st["n_probe_entries"] == wst["n_probe_entries"]
    pairs = 0
    for t in (0, 7, 150, n - 1):
        rows = oracle.spaced_round(texts[t], mask, 0.30, file, rec_offs, 32, 64, buggy=False, nthreads=8)
        exp = [(t, q, int(rows["j"][q]), int(rows["dir"][q]), int(rows["ref_pos"][q]), int(rows["cost"][q]), int(rows["matlen_a"][q]),
                int(rows["matlen_b"][q])) for q in np.nonzero(rows["found"])[0] if q != t]
        lo, hi = np.searchsorted(got["target"], [t, t + 1])
        assert [tuple(int(x) for x in r) for r in got[lo:hi]] == exp and len(exp) >= 3, t
        one, st1 = ctx.overlap_all(S, mask, 0.30, 32, 64, t_lo=t, t_hi=t + 1)
        assert st1["n_pairs"] == int(rows["n_pairs"].sum()) - int(rows["n_pairs"][t]), t


def test_overlap_15kb_reads_sampled_targets_vs_oracle(ctx, oracle, monkeypatch):
    """BASELINE configs[3]'s read shape at a size the oracle can spot-check: 2 400 x 15 kb reads @15 % at 20 x coverage through
    ONE probe table in three target ranges -- sampled census, equal-room slices from the second range on, the scan's 32 rows,
    the sampled narrow / wide decision, parked runs through the rings, pairs = candidates - what lies behind a success -- and
    the oracle's locked round (every read a query) on five targets spread over the ranges: same rows.  The whole set again
    with room that overflows (every range scanned twice): same rows, same counts."""
    n, rl = 2400, 15000
    g = eng.synth_genome(77, n * rl // 20)
    reads, offs, _ = eng.synth_reads(78, g, n, rl, nthreads=16)
    mask = eng.mask_from_pattern(MASK_PAT)
    S = ctx.seqs_from_text(reads, offs, strict_acgt=True)
    monkeypatch.setenv("PBA_OVL_SAMPLE_MIN", "64")                  # the sampled window decision at this size
    got, st = ctx.overlap_all_sharded(S, mask, 0.30, 32, 64, targets_per_call=800, cap_per_target=400)
    assert st["n_overlaps"] == len(got) > 8000 and st["n_prefiltered"] > 5 * st["n_listed"] and st["cap_fill"] >= 2
    texts = lambda i: reads[int(offs[i]):int(offs[i + 1])].tobytes()
    file = b"".join(eng.text2bin(texts(i)) for i in range(n))
    rec_offs = (np.arange(n, dtype=np.uint64) * np.uint64(4 + (rl + 3) // 4))
    pairs = 0
    for t in (0, 799, 800, 1733, n - 1):
        rows = oracle.spaced_round(texts(t), mask, 0.30, file, rec_offs, 32, 64, buggy=False, nthreads=16)
        exp = [(t, q, int(rows["j"][q]), int(rows["dir"][q]), int(rows["ref_pos"][q]), int(rows["cost"][q]), int(rows["matlen_a"][q]),
                int(rows["matlen_b"][q])) for q in np.nonzero(rows["found"])[0] if q != t]
        lo, hi = np.searchsorted(got["target"], [t, t + 1])
        assert [tuple(int(x) for x in r) for r in got[lo:hi]] == exp and len(exp) >= 1, t
        # ... and the pairs of that target alone, through a one-target call
        one, st1 = ctx.overlap_all(S, mask, 0.30, 32, 64, t_lo=t, t_hi=t + 1)
        assert st1["n_pairs"] == int(rows["n_pairs"].sum()) - int(rows["n_pairs"][t]), t
    monkeypatch.setenv("PBA_OVL_ROOM", "12")                        # every range overflows and is scanned again with exact slices
    again, st2 = ctx.overlap_all_sharded(S, mask, 0.30, 32, 64, targets_per_call=800, cap_per_target=400)
    assert (again == got).all() and st2["n_pairs"] == st["n_pairs"] and st2["n_candidates"] == st["n_candidates"] and st2["cap_overflow"] >= 2


def test_overlap_random_read_sets_bitvec_forms_vs_rowsweep():
    """tools/stress_overlap.py: all-vs-all on random read sets (60 ... 16 000 bases, 1-17 % error in indel- and substitution-
    heavy mixes, R 0.15-0.35, up to 4 000 reads) -- the bit-vector walk, with the pre-sort prefilter stage forced on and
    without it, gives the row-sweep kernel's overlaps row by row and its pair and candidate counts."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_overlap.py"), "--rounds", "5", "--seed", str(7 + SOAK)],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("same") == 10 and "DIFFERENT" not in r.stdout and "all rounds agree" in r.stdout


def test_overlap_repeats_outgrow_one_sort_with_and_without_the_pre_sort_stage():
    """tools/dbg_repeats.py: 160 reads over a genome with 2 kb of period-23 tandem repeat -- 7.4 M candidates, most of them
    real matches, 142 targets whose lists outgrow one LDS sort (cut into pieces without the stage; packed lists beyond one
    sort go through the global pass with it): the bit-vector walk gives the row-sweep kernel's rows and counts both ways."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dbg_repeats.py"), "160"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count(" same ") == 2 and "big targets 142" in r.stdout


def test_c_level_exchange_over_rccl_world_of_one(ctx, lib, tmp_path):
    """include/pba_dist.h (libpba_dist.so): the multi-GPU exchange as C entry points over RCCL, for hosts that are not Python.
    A box with one GPU can run a world of one -- RCCL itself wants a GPU per rank -- which still goes through every call:
    communicator bring-up from a unique id, the collectives, and the three composite steps, each against its single-process
    counterpart: pba_dist_index_build == pba_index_build, pba_dist_gather_reads gives the set back read for read,
    pba_dist_probe_table + pba_overlap_all_table == pba_overlap_all.  Then examples/overlap_dist.cpp (plain g++ over the two
    headers) prints the same totals.  (More ranks: tests/dist_overlap_worker.py runs the same protocol over gloo.)"""
    import ctypes as C
    import os
    import subprocess
    import torch
    from conftest import ROOT
    from pacbioassembly_amd import _lib, ProbeTable
    from pacbioassembly_amd.engine import SeqSet, SeedIndex
    dl = _lib.load_dist()
    ident = (C.c_uint8 * 128)()
    assert dl.pba_dist_unique_id(ident) == 0
    comm = C.c_void_p()
    assert dl.pba_dist_comm_create(ctx.h, 0, 1, ident, C.byref(comm)) == 0, ctx.lib.pba_ctx_error(ctx.h)
    assert dl.pba_dist_rank(comm) == 0 and dl.pba_dist_world(comm) == 1
    vals = (C.c_uint64 * 3)(5, 7, 9)
    assert dl.pba_dist_all_reduce_u64(comm, vals, 3, 0) == 0 and list(vals) == [5, 7, 9]
    a = torch.arange(4096, dtype=torch.uint8, device="cuda")
    b = torch.zeros(4096, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    assert dl.pba_dist_all_gather(comm, C.c_void_p(a.data_ptr()), 4096, C.c_void_p(b.data_ptr())) == 0 and torch.equal(a, b)
    # the seed index built "by all ranks" == the single-process build
    g = eng.synth_genome(901, 300000)
    mask = eng.mask_from_pattern(MASK_PAT)
    T = ctx.seqs_from_list([g.tobytes()], strict_acgt=True)
    for mode in (PBA_INDEX_ALL, PBA_INDEX_HEAD_TAIL):
        h = C.c_void_p()
        assert dl.pba_dist_index_build(comm, T.h, 0, mask, mode, C.byref(h)) == 0, ctx.lib.pba_ctx_error(ctx.h)
        ix2, ix1 = SeedIndex(ctx, h), ctx.index_build(T, 0, mask, mode)
        (k1, p1), (k2, p2) = ix1.dump(), ix2.dump()
        assert ix1.entries == ix2.entries > 30000 and (k1 == k2).all() and (p1 == p2).all()
    # the read set from "the ranks' shards", the probe table from "the ranks' probes": the all-vs-all is the single-process one
    n, rl = 600, 1500
    genome = eng.synth_genome(2, n * rl // 20)
    reads, offs = eng.synth_reads_range(3, genome, 0, n, rl)
    mine = ctx.seqs_from_text(reads, offs, strict_acgt=True)
    h = C.c_void_p()
    assert dl.pba_dist_gather_reads(comm, mine.h, C.byref(h)) == 0, ctx.lib.pba_ctx_error(ctx.h)
    allr = SeqSet(ctx, h)
    assert allr.count == n and all(allr.get_text(i) == mine.get_text(i) for i in (0, 1, 299, n - 1))
    th = C.c_void_p()
    assert dl.pba_dist_probe_table(comm, allr.h, 0, n, mask, 32, C.byref(th)) == 0, ctx.lib.pba_ctx_error(ctx.h)
    table = ProbeTable.__new__(ProbeTable)
    table.ctx, table.h = ctx, th
    got, st = ctx.overlap_all_table(allr, table, 0.30, 64)
    want, wst = ctx.overlap_all(mine, mask, 0.30, 32, 64)
    assert len(want) > 1000 and (got == want).all() and st["n_pairs"] == wst["n_pairs"] and st["n_candidates"] == wst["n_candidates"]
    table.close()
    dl.pba_dist_comm_destroy(comm)
    # the same from C++
    libdir = os.path.join(ROOT, "pacbioassembly_amd", "lib")
    exe = str(tmp_path / "overlap_dist")
    subprocess.run(["g++", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-o", exe, os.path.join(ROOT, "examples", "overlap_dist.cpp"),
                    "-L", libdir, "-lpba_dist", "-lpba", f"-Wl,-rpath,{libdir}", "-Wl,-rpath-link,/opt/rocm/lib"], check=True)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([exe, str(n), str(rl)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert f"all ranks: {wst['n_pairs']} candidate pairs, {wst['n_overlaps']} overlaps" in r.stdout, r.stdout


def test_bench_two_ranks_share_the_gpu_and_agree_with_one(lib):
    """The N > 1 paths of bench.py on the GPU box there is: two ranks (started by bench.py's own launcher) share the one GPU
    and exchange over gloo instead of RCCL (which needs a GPU per rank) -- seed-index exchange and locate in weak scaling,
    then the all-vs-all leg with its packed-read all-gather, probe exchange and target shards.  The strong-scaling counts
    (candidates, pairs, overlaps) equal the one-rank run's."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["PBA_BENCH_SHARE_GPU"] = "1"
    common = ["--reads", "3000", "--overlap-reads", "12000", "--steps", "1", "--warmup", "1", "--cpu-sample", "0"]

    def run(*args, rc=0):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args, *common], env=env, capture_output=True, timeout=600)
        assert p.returncode == rc, (p.returncode, p.stderr.decode()[-3000:])
        lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
        assert len(lines) == 1
        return json.loads(lines[0])

    two = run("--gpus", "2", "--backend", "gloo")
    one = run("--gpus", "1")
    assert two["n_gpus"] == 2 and two["scaling"] == "weak" and two["pairs_per_step"] > one["pairs_per_step"] > 10000
    a, b = two["overlap_strong"], one["overlap_strong"]
    assert a["world_size"] == 2 and a["scaling"] == "strong" and len(a["per_rank"]["walk_s"]) == 2
    for k in ("pairs_per_step", "overlaps_per_step", "candidates_per_step"):
        assert a[k] == b[k] > 1000, k
    ovl = run("--gpus", "2", "--backend", "gloo", "--mode", "overlap")
    assert ovl["scaling"] == "strong" and ovl["overlap"]["pairs_per_step"] == b["pairs_per_step"]
    # a rank that dies in the extra all-vs-all leg costs the run that leg, not its headline line: the failing rank leaves,
    # the other one gives the leg up at its timeout, rank 0 prints the line with the error noted FIRST, and the run then ends
    # with exit code 3 (bench.py: EXIT_LEG_LOST), never 0: the lost leg is visible to whoever started it
    for bad in ("1", "0"):
        env["PBA_BENCH_TEST_FAIL_RANK"] = bad
        hurt = run("--gpus", "2", "--backend", "gloo", "--overlap-timeout", "25", rc=3)
        assert hurt["n_gpus"] == 2 and hurt["pairs_per_step"] == two["pairs_per_step"]
        assert "error" in hurt["overlap_strong"], hurt["overlap_strong"]
    # --mode overlap is bounded too: a rank lost there ends the run non-zero (the failing rank's exception; its peer would
    # leave at --timeout with code 4 if the launcher had not taken it down already)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--mode", "overlap",
                        "--timeout", "40", *common], env=env, capture_output=True, timeout=600)
    assert p.returncode != 0 and not [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    del env["PBA_BENCH_TEST_FAIL_RANK"]

