"""Host half of the boundary (pure functions of libpba.so) against the reference's known answers."""
import hashlib

import numpy as np

from conftest import GOLD, gold_json
from pacbioassembly_amd import engine as eng


def test_reference_dna_test_kats(lib):
    # test/dna_test.cpp:18-29
    s = b"ACGTGTCATCGGATCAACCGGTT"
    rec = eng.text2bin(s)
    assert len(rec) == 10 and eng.bin2text(rec) == s and len(eng.bin2text(rec)) == 23
    for pos, want in [(0, 0x34DAB41B), (1, 0xD068D36E), (2, 0x41A34DBB), (7, 0xAF058D36)]:
        assert eng.seed_at(rec, pos) == want


def test_codec_golden(lib):
    g = gold_json("codec.json")
    for w, code in g["encode"]:
        assert eng.encode(w.encode()) == code
        if set(w) <= set("ACGT"):
            assert eng.decode(code) == w.encode()
    rec = eng.text2bin(g["seed_at_text"].encode())
    for pos, want in g["seed_at"]:            # includes the pos%4==0 behaviour
        assert eng.seed_at(rec, pos) == want, pos
        if pos % 4 or pos == 0:
            assert eng.seed_at(rec, pos, fixed=True) == want
        assert eng.seed_at(rec, pos, fixed=True) == eng.encode(g["seed_at_text"].encode()[pos:pos + 16])
    for pat, m in g["masks"]:
        assert eng.mask_from_pattern(pat) == m, pat
    for s, hexrec, back in g["text2bin"]:
        assert eng.text2bin(s.encode()).hex() == hexrec
        assert eng.bin2text(bytes.fromhex(hexrec)).decode() == back


def test_binary_file_image_and_record_walk(lib):
    g = gold_json("codec.json")
    lines = open(f"{GOLD}/real_align.txt").read().split()
    img = b"".join(eng.text2bin(l.encode()) for l in lines)
    assert len(img) == g["real_align_bin_len"] == 2533
    assert hashlib.sha256(img).hexdigest() == g["real_align_bin_sha"]
    offs, total = eng.open_binary(img, 500, 20000)      # spaced_seed.cpp:330-342
    assert total == 12
    want = [i for i, l in enumerate(lines) if 500 < len(l) < 20000]
    starts = np.concatenate([[0], np.cumsum([4 + (len(l) + 3) // 4 for l in lines])])[:-1]
    assert offs.tolist() == [int(starts[i]) for i in want]
    offs0, _ = eng.open_binary(img, 0, 1 << 30)
    assert len(offs0) == 12


def test_synth_is_deterministic(lib):
    g1 = eng.synth_genome(5, 1000)
    g2 = eng.synth_genome(5, 1000)
    assert (g1 == g2).all() and set(bytes(g1)) <= set(b"ACGT")
    # prefix property: the genome is a counter stream
    assert (eng.synth_genome(5, 333) == g1[:333]).all()
    r1, o1, s1 = eng.synth_reads(9, g1, 20, 100, nthreads=1)
    r2, o2, s2 = eng.synth_reads(9, g1, 20, 100, nthreads=4)
    assert (r1 == r2).all() and (s1 == s2).all()
    # pinned digests: the golden locator fixtures were generated from exactly these bytes
    assert hashlib.sha256(bytes(eng.synth_genome(1, 4096))).hexdigest()[:16] == "4c7ddc8983ca6f57"
    g = eng.synth_genome(1, 100000)
    r, _, _ = eng.synth_reads(11, g, 1000, 1000)
    assert hashlib.sha256(bytes(r)).hexdigest()[:16] == "17cce4a5731b6d0f"
