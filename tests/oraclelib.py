"""ctypes access to the checkers: oracle/_build/libpba_oracle.so (our C restatement) and, when it
was built in the container, oracle/_ref/libpba_ref.so (the reference itself).  Test infrastructure:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "_build", "libpba_oracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libpba_ref.so")

_P = C.c_void_p


class OrcResult(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("rc", "cost", "matlen_a", "matlen_b", "len_a", "len_b", "max_dst", "nedit", "fail_row")] + \
               [("cells", C.c_int64)]


ORC_LOC_ROW = np.dtype([(n, "<i4") for n in
                        ("read", "nseq", "found", "j", "pos", "cost", "seglen", "matlen_a", "matlen_b", "n_pairs")])
ORC_SS_ROW = np.dtype([(n, "<i4") for n in
                       ("read", "found", "j", "dir", "ref_pos", "cost", "matlen_a", "matlen_b", "n_trials",
                        "n_pairs")])


class OrcLocStats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("n_reads_kept", "n_probe_hits", "n_pairs", "n_located", "n_cells")]


def build_oracle() -> str:
    """(Re)build the C restatement with gcc; works anywhere."""
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"], check=True)
    return ORACLE_SO


def _ptr(a):
    return C.c_void_p(a.ctypes.data)


class Oracle:
    def __init__(self):
        if not os.path.exists(ORACLE_SO):
            build_oracle()
        L = self.lib = C.CDLL(ORACLE_SO)
        L.orc_encode.restype = C.c_uint32; L.orc_encode.argtypes = [C.c_char_p]
        L.orc_encode_padded.restype = C.c_uint32; L.orc_encode_padded.argtypes = [C.c_char_p, C.c_long]
        L.orc_decode.argtypes = [C.c_uint32, C.c_char_p]
        L.orc_c2i.restype = C.c_int; L.orc_c2i.argtypes = [C.c_int]
        L.orc_text2bin.restype = C.c_size_t; L.orc_text2bin.argtypes = [C.c_char_p, C.c_size_t, _P, C.c_size_t]
        L.orc_bin2text.restype = C.c_size_t; L.orc_bin2text.argtypes = [_P, C.c_char_p, C.c_size_t]
        L.orc_seed_at.restype = C.c_uint32; L.orc_seed_at.argtypes = [_P, C.c_int]
        L.orc_seed_at_fixed.restype = C.c_uint32; L.orc_seed_at_fixed.argtypes = [_P, C.c_int]
        L.orc_mask_from_pattern.restype = C.c_uint32; L.orc_mask_from_pattern.argtypes = [C.c_char_p]
        L.orc_aligner_new.restype = _P; L.orc_aligner_new.argtypes = [C.c_int, C.c_int]
        L.orc_aligner_free.argtypes = [_P]
        L.orc_align.restype = C.c_int
        L.orc_align.argtypes = [_P, _P, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_double, C.POINTER(OrcResult), _P]
        L.orc_aligner_cell.restype = C.c_int
        L.orc_aligner_cell.argtypes = [_P, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_seedmap_new.restype = _P; L.orc_seedmap_new.argtypes = [C.c_size_t]
        L.orc_seedmap_free.argtypes = [_P]
        L.orc_seedmap_size.restype = C.c_size_t; L.orc_seedmap_size.argtypes = [_P]
        L.orc_seedmap_entries.restype = C.c_size_t; L.orc_seedmap_entries.argtypes = [_P]
        L.orc_index_all.restype = C.c_size_t; L.orc_index_all.argtypes = [_P, _P, C.c_int, C.c_uint32]
        L.orc_index_head_tail.restype = C.c_uint; L.orc_index_head_tail.argtypes = [_P, _P, C.c_int, C.c_uint32]
        L.orc_seedmap_find.restype = C.c_int; L.orc_seedmap_find.argtypes = [_P, C.c_uint32, _P, C.c_int]
        L.orc_seedmap_dump.restype = C.c_size_t; L.orc_seedmap_dump.argtypes = [_P, _P, _P, C.c_size_t]
        L.orc_locator_run.restype = C.c_int
        L.orc_locator_run.argtypes = [_P, C.c_int, C.c_uint32, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int,
                                      _P, _P, C.c_int, C.c_int, _P, C.POINTER(OrcLocStats)]
        L.orc_spaced_round.restype = C.c_int
        L.orc_spaced_round.argtypes = [_P, C.c_int, C.c_uint32, C.c_double, C.c_int, C.c_int, C.c_int, _P, _P,
                                       C.c_int, C.c_int, _P]
        L.orc_open_binary.restype = C.c_size_t
        L.orc_open_binary.argtypes = [_P, C.c_size_t, C.c_uint32, C.c_uint32, _P, C.c_size_t, _P]
        L.orc_prefault.restype = C.c_int; L.orc_prefault.argtypes = [C.c_int, C.c_int, C.c_double]
        L.orc_pool_release.restype = None; L.orc_pool_release.argtypes = []
        L.orc_cons_new.restype = _P; L.orc_cons_new.argtypes = [_P, C.c_int, C.c_int, C.c_int]
        L.orc_cons_free.argtypes = [_P]
        L.orc_cons_try.restype = C.c_int
        L.orc_cons_try.argtypes = [_P, _P, C.c_int, _P, C.c_int, C.c_int, C.c_double, C.c_int, _P]
        L.orc_cons_elect.argtypes = [_P, C.c_int, C.c_int, _P, _P, C.c_int]
        L.orc_cons_append.argtypes = [_P, _P, C.c_int]; L.orc_cons_prepend.argtypes = [_P, _P, C.c_int]
        L.orc_cons_evolve.argtypes = [_P]
        L.orc_cons_round.restype = C.c_int
        L.orc_cons_round.argtypes = [_P, _P, C.c_uint32, C.c_double, C.c_int, C.c_int, C.c_int, _P, _P, _P, C.c_int, _P]
        L.orc_cons_dump.restype = C.c_int; L.orc_cons_dump.argtypes = [_P, _P, _P, _P, C.c_int, _P]
        L.orc_cons_text.restype = C.c_int; L.orc_cons_text.argtypes = [_P, _P, C.c_int]
        self._aligner = None

    def prefault(self, nthreads: int, len_a: int, R: float):
        """Map + touch the per-thread DP matrices before a timed run (the reference's static aligner is warm
        after its first alignment; first-touch faults cost seconds per GB in sandboxed containers)."""
        assert self.lib.orc_prefault(nthreads, len_a, R) == 0

    def release(self):
        self.lib.orc_pool_release()

    # codec
    def encode(self, t: bytes) -> int:
        return self.lib.orc_encode(t)

    def decode(self, code: int) -> bytes:
        b = C.create_string_buffer(17)
        self.lib.orc_decode(code, b)
        return b.raw[:16]

    def text2bin(self, text: bytes) -> bytes:
        cap = 4 + (len(text) + 3) // 4
        out = np.zeros(cap + 8, np.uint8)
        n = self.lib.orc_text2bin(text, len(text), _ptr(out), cap)
        return out[:n].tobytes()

    def bin2text(self, rec: bytes) -> bytes:
        buf = np.frombuffer(rec, np.uint8)
        ln = int(np.frombuffer(rec[:4], "<u4")[0])
        b = C.create_string_buffer(ln + 1)
        n = self.lib.orc_bin2text(_ptr(buf), b, ln + 1)
        return b.raw[:n]

    def seed_at(self, rec: bytes, pos: int, fixed=False) -> int:
        buf = np.frombuffer(rec + b"\0" * 64, np.uint8)
        return (self.lib.orc_seed_at_fixed if fixed else self.lib.orc_seed_at)(_ptr(buf), pos)

    def mask_from_pattern(self, pat: str) -> int:
        return self.lib.orc_mask_from_pattern(pat.encode())

    def consensus(self, text: bytes, weight: int = 1, max_len: int = 0, overlap_min: int = 64):
        """The unlocked ref_seq (votes, growth, evolve) as restated by the oracle."""
        return OracleCons(self, text, weight, max_len or max(4 * len(text), 100000), overlap_min)

    # DP
    def align(self, a: bytes, b: bytes, R: float, a_fwd=True, b_fwd=True, maxn=0, maxm=0, want_ops=False):
        """a/b hold the accessor's elements in memory order (a backward accessor starts at the last byte)."""
        if self._aligner is None:
            self._aligner = self.lib.orc_aligner_new(0, 0)
        al = self._aligner if maxn <= 0 else self.lib.orc_aligner_new(maxn, maxm)
        abuf = np.frombuffer(a + b"\0", np.uint8); bbuf = np.frombuffer(b + b"\0", np.uint8)
        pa = abuf.ctypes.data + (0 if a_fwd or not a else len(a) - 1)
        pb = bbuf.ctypes.data + (0 if b_fwd or not b else len(b) - 1)
        res = OrcResult()
        ops = np.zeros(len(a) + len(b) + 1, np.uint8) if want_ops else None
        self.lib.orc_align(al, C.c_void_p(pa), int(a_fwd), len(a), C.c_void_p(pb), int(b_fwd), len(b), R,
                           C.byref(res), _ptr(ops) if want_ops else None)
        if maxn > 0:
            self.lib.orc_aligner_free(al)
        d = {n: getattr(res, n) for n, _ in OrcResult._fields_}
        if want_ops:
            d["ops"] = ops[:res.nedit].copy() if res.rc >= 0 else ops[:0]
        return d

    def cell(self, i: int, j: int):
        """(cost, parent) of cell (i, j) as the most recent align() left it (seq_aligner.h:131,133), None if not written."""
        c, p = C.c_int(), C.c_int()
        if self.lib.orc_aligner_cell(self._aligner, i, j, C.byref(c), C.byref(p)) != 0:
            return None
        return c.value, p.value

    # index
    def index(self, text: bytes, mask: int, mode: str = "all"):
        """Returns (keys, pos) sorted by key with hit order inside a key, and the builder's return value."""
        sm = self.lib.orc_seedmap_new(1 << 21)
        buf = np.frombuffer(text + b"\0" * 32, np.uint8)
        if mode == "all":
            rv = self.lib.orc_index_all(sm, _ptr(buf), len(text), mask)
        else:
            rv = self.lib.orc_index_head_tail(sm, _ptr(buf), len(text), mask)
        n = self.lib.orc_seedmap_entries(sm)
        keys = np.zeros(max(n, 1), np.uint32); pos = np.zeros(max(n, 1), np.int32)
        self.lib.orc_seedmap_dump(sm, _ptr(keys), _ptr(pos), n)
        nkeys = self.lib.orc_seedmap_size(sm)
        self.lib.orc_seedmap_free(sm)
        return keys[:n], pos[:n], int(rv), int(nkeys)

    # drivers
    def locator(self, contig: np.ndarray, mask: int, R: float, reads: np.ndarray, offs: np.ndarray, trials=50,
                min_len=500, maxn=0, maxm=0, nthreads=1):
        contig = np.ascontiguousarray(contig, np.uint8)
        reads = np.ascontiguousarray(np.concatenate([reads, np.zeros(32, np.uint8)]), np.uint8)
        offs = np.ascontiguousarray(offs, np.uint64)
        n = offs.size - 1
        rows = np.zeros(max(n, 1), ORC_LOC_ROW)
        st = OrcLocStats()
        rc = self.lib.orc_locator_run(_ptr(contig), contig.size, mask, R, trials, min_len, maxn, maxm, _ptr(reads),
                                      _ptr(offs), n, nthreads, _ptr(rows), C.byref(st))
        assert rc == 0
        return rows[:n], {k: getattr(st, k) for k, _ in OrcLocStats._fields_}

    def spaced_round(self, ref: bytes, mask: int, R: float, file: bytes, rec_offs: np.ndarray, max_trial=32,
                     overlap_min=64, buggy=False, nthreads=1):
        refb = np.frombuffer(ref + b"\0" * 32, np.uint8)
        fileb = np.frombuffer(file + b"\0" * 65536, np.uint8)
        rec_offs = np.ascontiguousarray(rec_offs, np.uint64)
        n = rec_offs.size
        rows = np.zeros(max(n, 1), ORC_SS_ROW)
        rc = self.lib.orc_spaced_round(_ptr(refb), len(ref), mask, R, max_trial, overlap_min, int(buggy),
                                       _ptr(fileb), _ptr(rec_offs), n, nthreads, _ptr(rows))
        assert rc == 0
        return rows[:n]

    def open_binary(self, file: bytes, min_excl=500, max_excl=20000):
        buf = np.frombuffer(file, np.uint8)
        total = C.c_size_t()
        kept = self.lib.orc_open_binary(_ptr(buf), len(file), min_excl, max_excl, None, 0, C.byref(total))
        offs = np.zeros(max(kept, 1), np.uint64)
        self.lib.orc_open_binary(_ptr(buf), len(file), min_excl, max_excl, _ptr(offs), kept, None)
        return offs[:kept], int(total.value)


def have_ref() -> bool:
    return os.path.exists(REF_SO)


class _ConsBase:
    """Common face of the oracle's and the reference's consensus objects: try_align / evolve / dump / text."""

    def try_align(self, pos: int, seg: bytes, fwd: bool, R: float = 0.3):
        """seg holds the accessor's elements in memory order (backward: the accessor starts at the last byte).
        Returns dict(ok, matlen_b, cost, matlen_a, nedit, pre, post)."""
        buf = np.frombuffer(b"\0" * 8 + seg + b"\0" * 8, np.uint8)
        origin = buf.ctypes.data + 8 + (0 if fwd or not seg else len(seg) - 1)
        out = np.zeros(7, np.int32)
        self._try(pos, origin, len(seg), int(fwd), R, out)
        return dict(zip(("ok", "matlen_b", "cost", "matlen_a", "nedit", "pre", "post"), out.tolist()))

    def dump(self, cap: int = 1 << 22):
        sel = np.zeros((cap, 4), np.uint16); sup = np.zeros((cap, 4), np.uint16); tot = np.zeros(cap, np.int32)
        ext = np.zeros(3, np.int32)
        n = self._dump(sel, sup, tot, cap, ext)
        return sel[:n].copy(), sup[:n].copy(), tot[:n].copy(), ext.tolist()

    def text(self, cap: int = 1 << 22) -> bytes:
        b = C.create_string_buffer(cap)
        n = self._text(b, cap)
        return b.raw[:n]


class OracleCons(_ConsBase):
    def __init__(self, orc, text, weight, max_len, overlap_min):
        self.orc, self.overlap_min = orc, overlap_min
        self.h = orc.lib.orc_cons_new(text, len(text), weight, max_len)
        self.al = orc.lib.orc_aligner_new(26000, 6000)          # t_aligner, seq_aligner.h:260

    def _try(self, pos, origin, n, fwd, R, out):
        self.orc.lib.orc_cons_try(self.h, self.al, pos, C.c_void_p(origin), n, fwd, R, self.overlap_min, _ptr(out))

    def elect(self, pos: int, fwd: bool, ops: np.ndarray, vals: bytes):
        ops = np.ascontiguousarray(ops, np.uint8)
        self.orc.lib.orc_cons_elect(self.h, pos, int(fwd), _ptr(ops), vals, ops.size)

    def append(self, seg: bytes):
        self.orc.lib.orc_cons_append(self.h, seg, len(seg))

    def prepend(self, seg: bytes):
        self.orc.lib.orc_cons_prepend(self.h, seg, len(seg))

    def evolve(self):
        self.orc.lib.orc_cons_evolve(self.h)

    def round(self, mask, R, max_trial, file: bytes, rec_offs, pool, buggy=True):
        """one unlocked round of spaced_seed.cpp:420-446 over the records pool (in order); rows in pool order"""
        buf = np.frombuffer(file + b"\0" * 65536, np.uint8)
        offs = np.ascontiguousarray(rec_offs, np.uint64); pool = np.ascontiguousarray(pool, np.int32)
        rows = np.zeros(max(pool.size, 1), ORC_SS_ROW)
        nm = self.orc.lib.orc_cons_round(self.h, self.al, mask, R, max_trial, self.overlap_min, int(buggy), _ptr(buf), _ptr(offs),
                                         _ptr(pool), pool.size, _ptr(rows))
        return rows[:pool.size], nm

    def _dump(self, sel, sup, tot, cap, ext):
        return self.orc.lib.orc_cons_dump(self.h, _ptr(sel), _ptr(sup), _ptr(tot), cap, _ptr(ext))

    def _text(self, b, cap):
        return self.orc.lib.orc_cons_text(self.h, b, cap)

    def __del__(self):
        try:
            self.orc.lib.orc_cons_free(self.h); self.orc.lib.orc_aligner_free(self.al)
        except Exception:
            pass


class RefCons(_ConsBase):
    """ref_seq itself (one live object per process), OVERLAP_MIN = 64 as compiled."""

    def __init__(self, ref, text, weight):
        self.lib = ref.lib
        self.lib.ref_cons_new(text, len(text), weight)

    def _try(self, pos, origin, n, fwd, R, out):
        self.lib.ref_cons_try(pos, C.c_void_p(origin), n, fwd, C.c_double(R), _ptr(out))

    def evolve(self):
        self.lib.ref_cons_evolve()

    def round(self, mask, R, max_trial, file: bytes, rec_offs, pool, buggy=True):
        assert buggy, "the reference only has its own seed_at"
        buf = np.frombuffer(file + b"\0" * 65536, np.uint8).copy()
        offs = np.ascontiguousarray(rec_offs, np.uint64); pool = np.ascontiguousarray(pool, np.int32)
        rows = np.zeros((max(pool.size, 1), 10), np.int32)
        self.lib.ref_cons_round.restype = C.c_int
        self.lib.ref_cons_round.argtypes = [C.c_uint32, C.c_double, C.c_int, _P, _P, _P, C.c_int, _P]
        nm = self.lib.ref_cons_round(mask, R, max_trial, _ptr(buf), _ptr(offs), _ptr(pool), pool.size, _ptr(rows))
        out = np.zeros(max(pool.size, 1), ORC_SS_ROW)
        for k, name in enumerate(ORC_SS_ROW.names):
            out[name] = rows[:, k]
        return out[:pool.size], nm

    def _dump(self, sel, sup, tot, cap, ext):
        return self.lib.ref_cons_dump(_ptr(sel), _ptr(sup), _ptr(tot), cap, _ptr(ext))

    def _text(self, b, cap):
        return self.lib.ref_cons_text(b, cap)


class Ref:
    """The reference itself (oracle/_ref/libpba_ref.so); build container only."""

    def __init__(self):
        L = self.lib = C.CDLL(REF_SO)
        L.ref_encode.restype = C.c_uint32; L.ref_encode.argtypes = [C.c_char_p]
        L.ref_decode.argtypes = [C.c_uint32, C.c_char_p]
        L.ref_seed_at.restype = C.c_uint32; L.ref_seed_at.argtypes = [_P, C.c_int]
        L.ref_text2bin.restype = C.c_uint32; L.ref_text2bin.argtypes = [C.c_char_p, _P, C.c_uint32]
        L.ref_bin2text.restype = C.c_uint32; L.ref_bin2text.argtypes = [_P, C.c_char_p, C.c_uint32]
        L.ref_c2i.restype = C.c_int; L.ref_c2i.argtypes = [C.c_int]
        L.ref_mask_from_pattern.restype = C.c_uint32; L.ref_mask_from_pattern.argtypes = [C.c_char_p]
        for f in (L.ref_align, L.ref_align_stock):
            f.restype = C.c_int
            f.argtypes = [_P, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_double, _P, _P]
        L.ref_get_seedmap.restype = C.c_long
        L.ref_get_seedmap.argtypes = [_P, C.c_int, C.c_uint32, _P, _P, C.c_long, C.POINTER(C.c_uint32),
                                      C.POINTER(C.c_uint32)]
        L.ref_locator_index.restype = C.c_long
        L.ref_locator_index.argtypes = [_P, C.c_int, C.c_uint32, _P, _P, C.c_long]
        L.ref_locator.restype = C.c_int
        L.ref_locator.argtypes = [_P, C.c_int, C.c_uint32, C.c_double, C.c_int, C.c_int, _P, _P, C.c_int, _P, _P]
        L.ref_spaced_round.restype = C.c_int
        L.ref_spaced_round.argtypes = [_P, C.c_int, C.c_uint32, C.c_double, C.c_int, C.c_int, _P, _P, C.c_int, _P]
        L.ref_cons_new.argtypes = [C.c_char_p, C.c_int, C.c_int]
        L.ref_cons_try.restype = C.c_int; L.ref_cons_try.argtypes = [C.c_int, _P, C.c_int, C.c_int, C.c_double, _P]
        L.ref_cons_dump.restype = C.c_int; L.ref_cons_dump.argtypes = [_P, _P, _P, C.c_int, _P]
        L.ref_cons_text.restype = C.c_int; L.ref_cons_text.argtypes = [_P, C.c_int]

    def consensus(self, text: bytes, weight: int = 1):
        return RefCons(self, text, weight)

    def encode(self, t: bytes) -> int:
        return self.lib.ref_encode(t)

    def text2bin(self, text: bytes) -> bytes:
        cap = 4 + (len(text) + 3) // 4
        out = np.zeros(cap + 8, np.uint8)
        n = self.lib.ref_text2bin(text + b"\0", _ptr(out), cap)
        return out[:n].tobytes()

    def bin2text(self, rec: bytes) -> bytes:
        buf = np.frombuffer(rec, np.uint8)
        ln = int(np.frombuffer(rec[:4], "<u4")[0])
        b = C.create_string_buffer(ln + 1)
        n = self.lib.ref_bin2text(_ptr(buf), b, ln + 1)
        return b.raw[:n]

    def seed_at(self, rec: bytes, pos: int) -> int:
        buf = np.frombuffer(rec + b"\0" * 65536, np.uint8).copy()
        return self.lib.ref_seed_at(_ptr(buf), pos)

    def mask_from_pattern(self, pat: str) -> int:
        return self.lib.ref_mask_from_pattern(pat.encode())

    def align(self, a: bytes, b: bytes, R: float, a_fwd=True, b_fwd=True, stock=False, want_ops=False):
        abuf = np.frombuffer(a + b"\0", np.uint8).copy(); bbuf = np.frombuffer(b + b"\0", np.uint8).copy()
        pa = abuf.ctypes.data + (0 if a_fwd or not a else len(a) - 1)
        pb = bbuf.ctypes.data + (0 if b_fwd or not b else len(b) - 1)
        out = np.zeros(8, np.int32)
        ops = np.zeros(len(a) + len(b) + 1, np.uint8)
        f = self.lib.ref_align_stock if stock else self.lib.ref_align
        f(C.c_void_p(pa), int(a_fwd), len(a), C.c_void_p(pb), int(b_fwd), len(b), R, _ptr(out), _ptr(ops))
        d = dict(zip(("rc", "cost", "matlen_a", "matlen_b", "len_a", "len_b", "max_dst", "nedit"), map(int, out)))
        if want_ops:
            d["ops"] = ops[:d["nedit"]].copy()
        return d

    def cell(self, i: int, j: int):
        """seq_aligner::get_cost / get_parent of the reference's aligner after the last align()."""
        out = np.zeros(2, np.int32)
        self.lib.ref_cell(i, j, _ptr(out))
        return int(out[0]), int(out[1])

    def get_seedmap(self, text: bytes, mask: int):
        cap = len(text) + 16
        keys = np.zeros(cap, np.uint32); pos = np.zeros(cap, np.int32)
        rv, nk = C.c_uint32(), C.c_uint32()
        n = self.lib.ref_get_seedmap(text + b"\0" * 32, len(text), mask, _ptr(keys), _ptr(pos), cap, C.byref(rv),
                                     C.byref(nk))
        return keys[:n], pos[:n], int(rv.value), int(nk.value)

    def locator_index(self, text: bytes, mask: int):
        cap = len(text) + 16
        keys = np.zeros(cap, np.uint32); pos = np.zeros(cap, np.int32)
        n = self.lib.ref_locator_index(text, len(text), mask, _ptr(keys), _ptr(pos), cap)
        return keys[:n], pos[:n]

    def locator(self, contig: np.ndarray, mask: int, R: float, reads: np.ndarray, offs: np.ndarray, trials=50,
                min_len=500):
        contig = np.ascontiguousarray(contig, np.uint8)
        reads = np.ascontiguousarray(np.concatenate([reads, np.zeros(32, np.uint8)]), np.uint8)
        offs = np.ascontiguousarray(offs, np.uint64)
        n = offs.size - 1
        rows = np.zeros((max(n, 1), 10), np.int32)
        stats = np.zeros(4, np.int64)
        self.lib.ref_locator(_ptr(contig), contig.size, mask, R, trials, min_len, _ptr(reads), _ptr(offs), n,
                             _ptr(rows), _ptr(stats))
        out = np.zeros(n, ORC_LOC_ROW)
        for k, name in enumerate(ORC_LOC_ROW.names):
            out[name] = rows[:n, k]
        return out, dict(zip(("n_reads_kept", "n_probe_hits", "n_pairs", "n_located"), map(int, stats)))

    def spaced_round(self, ref: bytes, mask: int, R: float, file: bytes, rec_offs: np.ndarray, max_trial=32,
                     overlap_min=64):
        fileb = np.frombuffer(file + b"\0" * 65536, np.uint8).copy()
        rec_offs = np.ascontiguousarray(rec_offs, np.uint64)
        n = rec_offs.size
        rows = np.zeros((max(n, 1), 10), np.int32)
        self.lib.ref_spaced_round(ref, len(ref), mask, R, max_trial, overlap_min, _ptr(fileb), _ptr(rec_offs), n,
                                  _ptr(rows))
        out = np.zeros(n, ORC_SS_ROW)
        for k, name in enumerate(ORC_SS_ROW.names):
            out[name] = rows[:n, k]
        return out
