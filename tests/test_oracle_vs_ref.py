"""Pin oracle/pba_oracle.c directly against the reference compiled from /root/reference (oracle/_ref/libpba_ref.so)
on FRESH random inputs, beyond the committed goldens.  Only where that library exists (the build container)."""
import numpy as np
import pytest

from oraclelib import Ref, have_ref
from pacbioassembly_amd import engine as eng

pytestmark = pytest.mark.skipif(not have_ref(), reason="oracle/_ref/libpba_ref.so is built only where /root/reference exists")


@pytest.fixture(scope="module")
def ref():
    return Ref()


def test_codec_random(oracle, ref):
    rng = np.random.RandomState(1)
    for _ in range(200):
        n = int(rng.randint(0, 200))
        s = bytes(rng.choice(list(b"ACGTNacgt\n"), n).astype(np.uint8))
        assert oracle.text2bin(s) == ref.text2bin(s) == eng.text2bin(s)
        if n >= 16:
            assert oracle.encode(s[:16]) == ref.encode(s[:16]) == eng.encode(s[:16])
    t = bytes(rng.choice(list(b"ACGT"), 300).astype(np.uint8))
    rec = oracle.text2bin(t)
    for pos in range(0, 64):
        assert oracle.seed_at(rec, pos) == ref.seed_at(rec, pos) == eng.seed_at(rec, pos)


def test_align_random_full_scripts(oracle, ref):
    """rc, cost, match lengths, nedit and the whole edit script, for shapes the goldens do not contain."""
    rng = np.random.RandomState(2)
    alpha = np.frombuffer(b"ACGT", np.uint8)
    for t in range(150):
        la = int(rng.randint(1, 700))
        a = alpha[rng.randint(0, 4, la)]
        e = float(rng.choice([0.0, 0.1, 0.2, 0.35]))
        keep = rng.rand(la) > e / 2
        b = a[keep].copy()
        flip = rng.rand(b.size) < e / 2
        b[flip] = alpha[rng.randint(0, 4, int(flip.sum()))]
        b = np.concatenate([b, alpha[rng.randint(0, 4, int(rng.choice([0, 5, 150])))]])
        if rng.rand() < 0.4:
            a, b = b, a
        R = float(rng.choice([0.1, 0.3, 0.45]))
        fwd = bool(rng.rand() < 0.6)
        x = oracle.align(a.tobytes(), b.tobytes(), R, fwd, fwd, want_ops=True)
        y = ref.align(a.tobytes(), b.tobytes(), R, fwd, fwd, want_ops=True)
        assert x["rc"] == y["rc"] and (x["len_a"], x["len_b"], x["max_dst"]) == (y["len_a"], y["len_b"], y["max_dst"])
        if y["rc"] >= 0:
            assert (x["cost"], x["matlen_a"], x["matlen_b"], x["nedit"]) == (y["cost"], y["matlen_a"], y["matlen_b"], y["nedit"])
            assert x["ops"].tolist() == y["ops"].tolist()


def test_matrix_cells_get_cost_get_parent(oracle, ref):
    """seq_aligner::get_cost / get_parent (seq_aligner.h:131-134, locator.cpp:86): every cell a call writes -- the
    borders of init_cell, the band of every row swept, up to the row of an early failure -- holds the same cost and parent
    in the oracle's matrix as in the reference's."""
    rng = np.random.RandomState(12)
    alpha = np.frombuffer(b"ACGT", np.uint8)
    for t in range(40):
        la = int(rng.randint(12, 260))
        a = alpha[rng.randint(0, 4, la)]
        e = float(rng.choice([0.05, 0.2, 0.5]))
        b = a[rng.rand(la) > e / 2].copy()
        flip = rng.rand(b.size) < e / 2
        b[flip] = alpha[rng.randint(0, 4, int(flip.sum()))]
        b = np.concatenate([b, alpha[rng.randint(0, 4, int(rng.choice([0, 7, 90])))]])
        if t % 3 == 0:
            a, b = b, a
        fwd = bool(t % 2)
        x = oracle.align(a.tobytes(), b.tobytes(), 0.3, fwd, fwd)
        y = ref.align(a.tobytes(), b.tobytes(), 0.3, fwd, fwd)
        assert x["rc"] == y["rc"]
        rows = x["fail_row"] if x["fail_row"] else x["len_a"]
        md, n = x["max_dst"], 0
        for i in range(0, rows + 1):
            for j in range(max(0, i - md), min(x["len_b"], i + md) + 1):
                if i == 0 and j > md:
                    continue
                assert oracle.cell(i, j) == ref.cell(i, j), (t, i, j)
                n += 1
        assert n > 100 and oracle.cell(rows + 1, rows + 1) is None
        if x["rc"] >= 0 and x["len_b"] >= x["len_a"]:         # locator.cpp:86: the diagonal cell at the end of a
            assert oracle.cell(x["len_a"], x["len_a"])[0] >= 0


def test_stock_aligner_agrees_where_well_defined(oracle, ref):
    """The stock seq_aligner<26000,6000> typedef (no wide MAXM) agrees with the canonical one when 2*max_dst+1 <= MAXM."""
    g = eng.synth_genome(8, 40000)
    reads, offs, starts = eng.synth_reads(9, g, 6, 4000)
    for r in range(6):
        a = reads[int(offs[r]):int(offs[r + 1])].tobytes(); b = g[int(starts[r]):int(starts[r]) + 6000].tobytes()
        x, y = ref.align(a, b, 0.3), ref.align(a, b, 0.3, stock=True)
        assert x == y == {k: oracle.align(a, b, 0.3)[k] for k in x}


def test_index_and_locator_random(oracle, ref):
    mask = eng.mask_from_pattern("11*11*1*1*11*111")
    g = eng.synth_genome(10, 60000)
    k1, p1, rv1, nk1 = oracle.index(g.tobytes(), mask, "head_tail")
    k2, p2, rv2, nk2 = ref.get_seedmap(g.tobytes(), mask)
    assert (k1 == k2).all() and (p1 == p2).all() and (rv1, nk1) == (rv2, nk2)
    k1, p1, _, _ = oracle.index(g.tobytes()[:7000], mask, "all")
    k2, p2 = ref.locator_index(g.tobytes()[:7000], mask)
    assert (k1 == k2).all() and (p1 == p2).all()
    reads, offs, _ = eng.synth_reads(11, g, 120, 800, 0.09, 0.045, 0.015)
    r1, s1 = oracle.locator(g, mask, 0.25, reads, offs, 40, 500, nthreads=4)
    r2, s2 = ref.locator(g, mask, 0.25, reads, offs, 40, 500)
    for c in r1.dtype.names:
        assert (r1[c] == r2[c]).all(), c
    assert all(s1[k] == s2[k] for k in s2)


def test_consensus_fresh_inputs(oracle, ref):
    """Consensus voting / growth / evolve on seeds the goldens do not hold: oracle == reference, step by step."""
    from cons_scenarios import run_scenario, scenario_inputs
    for sc in [("fresh_a", 131, 132, 7000, 1500, 3200, 70, 1100, 2, (0.06, 0.04, 0.04), True),
               ("fresh_b", 133, 134, 6000, 2000, 2500, 50, 900, 1, (0.02, 0.08, 0.03), False)]:
        text, weight, reads = scenario_inputs(sc)
        a = run_scenario(oracle.consensus(text, weight), reads)
        b = run_scenario(ref.consensus(text, weight), reads)
        assert a == b, sc[0]
        assert sum(t[4] for r in b["rounds"] for t in r["tries"]) >= 8


def test_assembly_fresh_inputs(oracle, ref):
    """Unlocked multi-round assembly (spaced_seed.cpp:409-452 without -l) on seeds and error mixes the golden does not
    hold: oracle == the reference's ref_seq, round by round."""
    from cons_scenarios import ASSEMBLE, assemble_inputs, run_assembly
    masks = [oracle.mask_from_pattern(p) for p in ("111*11*11*1*1111", "1111*1*11**11*111", "11*1111**1*11*111")]
    for k, over in enumerate([dict(genome_seed=71, reads_seed=72, foreign_seed=73, err=(0.02, 0.08, 0.03), slice=(3000, 2500), max_round=6),
                              dict(genome_seed=74, reads_seed=75, foreign_seed=76, err=(0.05, 0.05, 0.05), slice=(9000, 4000), weight=1,
                                   n_reads=150, read_len=1500, max_round=5)]):
        cfg = dict(ASSEMBLE, **over)
        text, weight, file, rec_offs, texts = assemble_inputs(cfg)
        a = run_assembly(oracle.consensus(text, weight), masks, file, rec_offs, len(texts), cfg)
        b = run_assembly(ref.consensus(text, weight), masks, file, rec_offs, len(texts), cfg)
        assert a == b, k
        assert sum(len(r["found"]) for r in b["rounds"]) > 40 and len(b["final_text"]) > len(text) + 1000
