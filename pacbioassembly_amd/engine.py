"""Thin Python host layer over the C ABI (include/pba.h) -- used by tests/ and bench.py.

The reference is C++; its drop-in host API is the compat headers in include/compat/.  This module
only wraps the same C entry points for pytest and the benchmark: numpy arrays in, numpy arrays
out, every non-zero status raised as PbaError.  Nothing here computes on the CPU.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import _lib
from ._lib import (PBA_INDEX_ALL, PBA_INDEX_HEAD_TAIL, PBA_KERNEL_AUTO, PBA_KERNEL_BITVEC, PBA_KERNEL_ROWSWEEP,
                   PbaLocRow, PbaLocStats, PbaPair, PbaResult, PbaSsRow)

PAIR_DTYPE = np.dtype([("a_seq", "<u4"), ("a_pos", "<i4"), ("a_len", "<i4"), ("b_seq", "<u4"), ("b_pos", "<i4"),
                       ("b_len", "<i4"), ("flags", "<u4")])
RESULT_DTYPE = np.dtype([(n, "<i4") for n in ("rc", "cost", "matlen_a", "matlen_b", "len_a", "len_b", "max_dst", "diag_cost")])
LOC_ROW_DTYPE = np.dtype([(n, "<i4") for n in
                          ("read", "nseq", "found", "j", "pos", "cost", "seglen", "matlen_a", "matlen_b", "n_pairs", "diag_cost")])
SS_ROW_DTYPE = np.dtype([(n, "<i4") for n in
                         ("read", "found", "j", "dir", "ref_pos", "cost", "matlen_a", "matlen_b", "n_trials",
                          "n_pairs")])
OVERLAP_DTYPE = np.dtype([(n, "<i4") for n in ("target", "query", "j", "dir", "ref_pos", "cost", "matlen_a", "matlen_b")])
assert PAIR_DTYPE.itemsize == C.sizeof(PbaPair) and RESULT_DTYPE.itemsize == C.sizeof(PbaResult)
assert LOC_ROW_DTYPE.itemsize == C.sizeof(PbaLocRow) and SS_ROW_DTYPE.itemsize == C.sizeof(PbaSsRow)


class PbaError(RuntimeError):
    def __init__(self, status: int, detail: str = ""):
        self.status = status
        msg = _lib.load().pba_strerror(status).decode()
        super().__init__(f"pba status {status} ({msg})" + (f": {detail}" if detail else ""))


def _ptr(a: np.ndarray) -> C.c_void_p:
    return C.c_void_p(a.ctypes.data)


# ----------------------------------------------------------------------------- host codec
def encode(text16: bytes) -> int:
    assert len(text16) >= 16
    return _lib.load().pba_encode16(text16)


def decode(code: int) -> bytes:
    buf = C.create_string_buffer(17)
    _lib.load().pba_decode16(code, buf)
    return buf.raw[:16]


def text2bin(text: bytes) -> bytes:
    cap = 4 + (len(text) + 3) // 4
    out = np.zeros(cap, np.uint8)
    n = _lib.load().pba_text2bin(text, len(text), _ptr(out), cap)
    assert n == cap
    return out.tobytes()


def bin2text(record: bytes) -> bytes:
    rec = np.frombuffer(record, np.uint8)
    ln = int(np.frombuffer(record[:4], "<u4")[0])
    buf = C.create_string_buffer(ln + 1)
    n = _lib.load().pba_bin2text(_ptr(rec), buf, ln + 1)
    return buf.raw[:n]


def seed_at(record: bytes, pos: int, fixed: bool = False) -> int:
    rec = np.frombuffer(record + b"\0" * 64, np.uint8)   # seed_at reads past short records (B1)
    lib = _lib.load()
    return (lib.pba_seed_at_fixed if fixed else lib.pba_seed_at)(_ptr(rec), pos)


def mask_from_pattern(pattern: str) -> int:
    return _lib.load().pba_mask_from_pattern(pattern.encode())


def open_binary(file: bytes, min_excl: int = 500, max_excl: int = 20000):
    buf = np.frombuffer(file, np.uint8)
    total = C.c_size_t(0)
    lib = _lib.load()
    kept = lib.pba_open_binary(_ptr(buf), len(file), min_excl, max_excl, None, 0, C.byref(total))
    offs = np.zeros(max(kept, 1), np.uint64)
    lib.pba_open_binary(_ptr(buf), len(file), min_excl, max_excl, _ptr(offs), kept, None)
    return offs[:kept], int(total.value)


# ----------------------------------------------------------------------------- synthetic data
def synth_genome(seed: int, n: int) -> np.ndarray:
    out = np.empty(n, np.uint8)
    _lib.load().pba_synth_genome(seed, _ptr(out), n)
    return out


def synth_reads(seed: int, genome: np.ndarray, n_reads: int, read_len: int, p_ins: float = 0.05,
                p_del: float = 0.05, p_sub: float = 0.05, nthreads: int = 8):
    """Returns (text[n_reads*read_len] uint8, offsets[n_reads+1] uint64, starts[n_reads] uint32)."""
    genome = np.ascontiguousarray(genome, np.uint8)
    out = np.empty(n_reads * read_len, np.uint8)
    starts = np.empty(max(n_reads, 1), np.uint32)
    st = _lib.load().pba_synth_reads(seed, _ptr(genome), genome.size, n_reads, read_len, p_ins, p_del, p_sub,
                                     _ptr(out), _ptr(starts), nthreads)
    if st != 0:
        raise PbaError(st, "pba_synth_reads")
    offs = (np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(read_len))
    return out, offs, starts[:n_reads]


def synth_reads_range(seed: int, genome: np.ndarray, r_lo: int, r_hi: int, read_len: int, p_ins: float = 0.05,
                      p_del: float = 0.05, p_sub: float = 0.05, nthreads: int = 8):
    """Reads [r_lo, r_hi) of the set synth_reads(seed, ...) makes: (text, offsets) of just those reads."""
    genome = np.ascontiguousarray(genome, np.uint8)
    n = r_hi - r_lo
    out = np.empty(max(n, 0) * read_len, np.uint8)
    st = _lib.load().pba_synth_reads_range(seed, _ptr(genome), genome.size, r_lo, r_hi, read_len, p_ins, p_del, p_sub,
                                           _ptr(out), None, nthreads)
    if st != 0:
        raise PbaError(st, "pba_synth_reads_range")
    return out, (np.arange(n + 1, dtype=np.uint64) * np.uint64(read_len))


def concat(seqs: Sequence[bytes]):
    """Concatenate byte strings into (text uint8[], offsets uint64[n+1])."""
    offs = np.zeros(len(seqs) + 1, np.uint64)
    if seqs:
        offs[1:] = np.cumsum([len(s) for s in seqs], dtype=np.uint64)
    text = np.frombuffer(b"".join(seqs), np.uint8).copy() if seqs else np.zeros(0, np.uint8)
    return text, offs


# ----------------------------------------------------------------------------- device objects
class Context:
    """One GPU, one stream (pba_ctx).  Raises PbaError(PBA_E_NODEVICE) without a gfx950 device."""

    def __init__(self, device: int = 0):
        self.lib = _lib.load()
        h = C.c_void_p()
        st = self.lib.pba_ctx_create(device, C.byref(h))
        if st != 0:
            raise PbaError(st, "pba_ctx_create")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.pba_ctx_destroy(self.h)
            self.h = None

    __del__ = close

    def check(self, st: int, what: str = ""):
        if st != 0:
            raise PbaError(st, f"{what}: {self.lib.pba_ctx_error(self.h).decode()}")

    def set_stream(self, stream_handle: Optional[int]):
        self.check(self.lib.pba_ctx_set_stream(self.h, C.c_void_p(stream_handle or 0)), "set_stream")

    def sync(self):
        self.check(self.lib.pba_ctx_sync(self.h), "sync")

    def trim(self):
        """Give the work buffers kept between calls back to the device (pba_ctx_trim)."""
        self.check(self.lib.pba_ctx_trim(self.h), "trim")

    def last_profile(self) -> dict:
        pr = _lib.PbaProfile()
        self.check(self.lib.pba_ctx_last_profile(self.h, C.byref(pr)), "last_profile")
        return {n: getattr(pr, n) for n, _ in _lib.PbaProfile._fields_}

    def device_info(self):
        name = C.create_string_buffer(256)
        ncu, mhz, hbm = C.c_int(), C.c_int(), C.c_uint64()
        self.check(self.lib.pba_ctx_device_info(self.h, name, 256, C.byref(ncu), C.byref(mhz), C.byref(hbm)))
        return {"name": name.value.decode(), "n_cu": ncu.value, "clock_mhz": mhz.value, "hbm_bytes": hbm.value}

    # -- sequence sets
    def seqs_from_text(self, text: np.ndarray, offsets: np.ndarray, strict_acgt: bool = False) -> "SeqSet":
        text = np.ascontiguousarray(text, np.uint8)
        offsets = np.ascontiguousarray(offsets, np.uint64)
        h = C.c_void_p()
        self.check(self.lib.pba_seqs_from_text(self.h, _ptr(text), _ptr(offsets), offsets.size - 1,
                                               int(strict_acgt), C.byref(h)), "seqs_from_text")
        return SeqSet(self, h)

    def seqs_from_list(self, seqs: Sequence[bytes], strict_acgt: bool = False) -> "SeqSet":
        return self.seqs_from_text(*concat(seqs), strict_acgt=strict_acgt)

    def seqs_from_device_text(self, d_text_ptr: int, d_offsets_ptr: int, n: int, total_bytes: int,
                              max_len: int) -> "SeqSet":
        h = C.c_void_p()
        self.check(self.lib.pba_seqs_from_device_text(self.h, C.c_void_p(d_text_ptr), C.c_void_p(d_offsets_ptr), n,
                                                      total_bytes, max_len, C.byref(h)), "seqs_from_device_text")
        return SeqSet(self, h)

    def seqs_from_records(self, file: bytes, min_excl: int = 500, max_excl: int = 20000) -> "SeqSet":
        buf = np.frombuffer(file, np.uint8)
        h = C.c_void_p()
        self.check(self.lib.pba_seqs_from_records(self.h, _ptr(buf), len(file), min_excl, max_excl, C.byref(h)),
                   "seqs_from_records")
        return SeqSet(self, h)

    def seqs_from_device_packed(self, d_packed_ptr: int, n_bytes: int, offsets: np.ndarray, lengths: np.ndarray,
                                non_acgt: bool = False) -> "SeqSet":
        """A set over packed bytes already on the device (the all-gathered read shards): no re-packing."""
        offsets = np.ascontiguousarray(offsets, np.uint64)
        lengths = np.ascontiguousarray(lengths, np.uint32)
        h = C.c_void_p()
        self.check(self.lib.pba_seqs_from_device_packed(self.h, C.c_void_p(d_packed_ptr), n_bytes, _ptr(offsets), _ptr(lengths),
                                                        lengths.size, int(non_acgt), C.byref(h)), "seqs_from_device_packed")
        return SeqSet(self, h)

    # -- index
    def index_build(self, target: "SeqSet", seq: int, mask: int, mode: int = PBA_INDEX_ALL) -> "SeedIndex":
        h = C.c_void_p()
        self.check(self.lib.pba_index_build(self.h, target.h, seq, mask, mode, C.byref(h)), "index_build")
        return SeedIndex(self, h)

    def index_scan(self, target: "SeqSet", seq: int, mask: int, mode: int, part: int, nparts: int,
                   d_entries_ptr: int, cap: int) -> int:
        """Rank `part`'s slice of the index entries into a device buffer; returns how many were written."""
        n = C.c_uint64()
        self.check(self.lib.pba_index_scan(self.h, target.h, seq, mask, mode, part, nparts, C.c_void_p(d_entries_ptr),
                                           cap, C.byref(n)), "index_scan")
        return int(n.value)

    def index_from_entries(self, d_entries_ptr: int, n: int, mask: int, mode: int, seq_len: int) -> "SeedIndex":
        h = C.c_void_p()
        self.check(self.lib.pba_index_from_entries(self.h, C.c_void_p(d_entries_ptr), n, mask, mode, seq_len,
                                                   C.byref(h)), "index_from_entries")
        return SeedIndex(self, h)

    # -- alignment
    def align_batch(self, A: "SeqSet", B: "SeqSet", pairs: np.ndarray, R: float, maxn: int = 0, maxm: int = 0,
                    kernel: int = PBA_KERNEL_AUTO) -> np.ndarray:
        pairs = np.ascontiguousarray(pairs, PAIR_DTYPE)
        out = np.zeros(pairs.size, RESULT_DTYPE)
        self.check(self.lib.pba_align_batch(self.h, A.h, B.h, _ptr(pairs), pairs.size, R, maxn, maxm, kernel,
                                            _ptr(out)), "align_batch")
        return out

    def align_text(self, a: bytes, b: bytes, R: float, a_fwd: bool = True, b_fwd: bool = True, maxn: int = 0,
                   maxm: int = 0) -> np.ndarray:
        """a/b hold the accessor's elements in memory order; a backward accessor starts at the last byte."""
        abuf = np.frombuffer(a + b"\0", np.uint8)
        bbuf = np.frombuffer(b + b"\0", np.uint8)
        pa = abuf.ctypes.data + (0 if a_fwd or not a else len(a) - 1)
        pb = bbuf.ctypes.data + (0 if b_fwd or not b else len(b) - 1)
        out = np.zeros(1, RESULT_DTYPE)
        self.check(self.lib.pba_align_text(self.h, C.c_void_p(pa), int(a_fwd), len(a), C.c_void_p(pb), int(b_fwd),
                                           len(b), R, maxn, maxm, _ptr(out)), "align_text")
        return out[0]

    def align_text_trace(self, a: bytes, b: bytes, R: float, a_fwd: bool = True, b_fwd: bool = True, maxn: int = 0,
                         maxm: int = 0):
        """align_text plus the edit script: returns (result, ops uint8[nedit]) with 1 MATCH, 2 INSERT, 3 DELETE."""
        abuf = np.frombuffer(a + b"\0", np.uint8)
        bbuf = np.frombuffer(b + b"\0", np.uint8)
        pa = abuf.ctypes.data + (0 if a_fwd or not a else len(a) - 1)
        pb = bbuf.ctypes.data + (0 if b_fwd or not b else len(b) - 1)
        out = np.zeros(1, RESULT_DTYPE)
        cap = len(a) + len(b) + 1
        ops = np.zeros(cap, np.uint8)
        ne = C.c_int32()
        self.check(self.lib.pba_align_text_trace(self.h, C.c_void_p(pa), int(a_fwd), len(a), C.c_void_p(pb), int(b_fwd), len(b),
                                                 R, maxn, maxm, _ptr(out), _ptr(ops), cap, C.byref(ne)), "align_text_trace")
        return out[0], ops[:ne.value].copy()

    def align_text_matrix(self, a: bytes, b: bytes, R: float, a_fwd: bool = True, b_fwd: bool = True, maxn: int = 0, maxm: int = 0):
        """The DP matrix of one pair (pba_align_text_matrix): returns (result, cost uint16[len_a+1, 2*max_dst+1], parent
        uint8[same], rows swept); cell (i, j) sits at [i, j - i + max_dst]."""
        abuf = np.frombuffer(a + b"\0", np.uint8)
        bbuf = np.frombuffer(b + b"\0", np.uint8)
        pa = abuf.ctypes.data + (0 if a_fwd or not a else len(a) - 1)
        pb = bbuf.ctypes.data + (0 if b_fwd or not b else len(b) - 1)
        la, lb = len(a), len(b)
        md = 1 + int((la if lb >= la else lb) * R)
        len_a = la if lb >= la else min(la, lb + md)
        W = 2 * md + 1
        cost = np.zeros((len_a + 1, W), np.uint16)
        par = np.zeros((len_a + 1, W), np.uint8)
        out = np.zeros(1, RESULT_DTYPE)
        rows = C.c_int32()
        self.check(self.lib.pba_align_text_matrix(self.h, C.c_void_p(pa), int(a_fwd), la, C.c_void_p(pb), int(b_fwd), lb, R, maxn, maxm,
                                                  _ptr(out), _ptr(cost), _ptr(par), cost.size, C.byref(rows)), "align_text_matrix")
        return out[0], cost, par, rows.value

    def align_batch_trace(self, A: "SeqSet", B: "SeqSet", pairs: np.ndarray, R: float, maxn: int = 0, maxm: int = 0,
                          kernel: int = PBA_KERNEL_AUTO):
        """Returns (results, list of ops arrays)."""
        pairs = np.ascontiguousarray(pairs, PAIR_DTYPE)
        out = np.zeros(pairs.size, RESULT_DTYPE)
        off = np.zeros(pairs.size + 1, np.uint64)
        off[1:] = np.cumsum(pairs["a_len"].astype(np.int64) + pairs["b_len"].astype(np.int64)).astype(np.uint64)
        ops = np.zeros(int(off[-1]) + 1, np.uint8)
        ne = np.zeros(max(pairs.size, 1), np.int32)
        self.check(self.lib.pba_align_batch_trace(self.h, A.h, B.h, _ptr(pairs), pairs.size, R, maxn, maxm, kernel, _ptr(out), _ptr(ops),
                                                  _ptr(off), _ptr(ne)), "align_batch_trace")
        return out, [ops[int(off[q]):int(off[q]) + int(ne[q])].copy() for q in range(pairs.size)]

    # -- drivers
    def locate(self, ix: "SeedIndex", target: "SeqSet", target_seq: int, reads: "SeqSet", R: float,
               trials: int = 50, min_len: int = 500, maxn: int = 0, maxm: int = 0, kernel: int = PBA_KERNEL_AUTO):
        rows = np.zeros(max(reads.count, 1), LOC_ROW_DTYPE)
        stats = PbaLocStats()
        self.check(self.lib.pba_locate(self.h, ix.h, target.h, target_seq, reads.h, R, trials, min_len, maxn, maxm,
                                       kernel, _ptr(rows), C.byref(stats)), "locate")
        return rows[:reads.count], {n: getattr(stats, n) for n, _ in PbaLocStats._fields_}

    def spaced_round(self, ix: "SeedIndex", ref: "SeqSet", ref_seq: int, reads: "SeqSet", R: float,
                     max_trial: int = 32, overlap_min: int = 64, buggy_seed_at: bool = False,
                     kernel: int = PBA_KERNEL_AUTO) -> np.ndarray:
        rows = np.zeros(max(reads.count, 1), SS_ROW_DTYPE)
        self.check(self.lib.pba_spaced_round(self.h, ix.h, ref.h, ref_seq, reads.h, R, max_trial, overlap_min,
                                             int(buggy_seed_at), kernel, _ptr(rows)), "spaced_round")
        return rows[:reads.count]


def _spaced_multi(self, ref, ref_seq, reads, R, masks, picks, max_round=100, max_trial=32, overlap_min=64,
                  buggy_seed_at=False, kernel=PBA_KERNEL_AUTO):
    """spaced_seed's main loop for a locked reference; returns (rows, found_round, log list of dicts)."""
    n = max(reads.count, 1)
    rows = np.zeros(n, SS_ROW_DTYPE)
    fr = np.zeros(n, np.int32)
    log = np.zeros(max(max_round, 1), np.dtype([("round", "<i4"), ("mask", "<u4"), ("n_tried", "<i4"), ("n_found", "<i4")]))
    masks = np.ascontiguousarray(masks, np.uint32); picks = np.ascontiguousarray(picks, np.uint32)
    nr = C.c_int()
    self.check(self.lib.pba_spaced_multi(self.h, ref.h, ref_seq, reads.h, R, max_trial, overlap_min, int(buggy_seed_at), kernel,
                                         _ptr(masks), masks.size, _ptr(picks), picks.size, max_round, _ptr(rows), _ptr(fr),
                                         _ptr(log), log.size, C.byref(nr)), "spaced_multi")
    return rows[:reads.count], fr[:reads.count], [dict(zip(log.dtype.names, (int(x) for x in l))) for l in log[:nr.value]]


Context.spaced_multi = _spaced_multi


def _overlap_all(self, reads, mask, R, max_trial=32, overlap_min=64, t_lo=0, t_hi=None, kernel=PBA_KERNEL_AUTO,
                 cap=None):
    """All-vs-all overlap of a read set (targets t_lo..t_hi); returns (overlaps sorted by (target, query), stats)."""
    t_hi = reads.count if t_hi is None else t_hi
    cap = cap if cap is not None else max(1, (t_hi - t_lo) * max(reads.count - 1, 1))
    out = np.empty(cap, OVERLAP_DTYPE)                 # (not zeroed: 320 MB per target range at ten million reads; rows beyond n are never handed out)
    n = C.c_uint64()
    st = _lib.PbaOverlapStats()
    self.check(self.lib.pba_overlap_all(self.h, reads.h, t_lo, t_hi, mask, R, max_trial, overlap_min, kernel, _ptr(out), cap,
                                        C.byref(n), C.byref(st)), "overlap_all")
    return out[:min(int(n.value), cap)], {k: getattr(st, k) for k, _ in _lib.PbaOverlapStats._fields_}


def _overlap_probes(self, reads, q_lo, q_hi, mask, max_trial, d_entries_ptr, cap):
    """Probe entries of queries [q_lo, q_hi) into a device buffer (multi-GPU exchange form); returns the count."""
    n = C.c_uint64()
    self.check(self.lib.pba_overlap_probes(self.h, reads.h, q_lo, q_hi, mask, max_trial, C.c_void_p(d_entries_ptr), cap,
                                           C.byref(n)), "overlap_probes")
    return int(n.value)


def _overlap_all_probes(self, reads, d_entries_ptr, n_slots, mask, R, max_trial=32, overlap_min=64, t_lo=0, t_hi=None,
                        kernel=PBA_KERNEL_AUTO, cap=None):
    t_hi = reads.count if t_hi is None else t_hi
    cap = cap if cap is not None else max(1, (t_hi - t_lo) * max(reads.count - 1, 1))
    out = np.empty(cap, OVERLAP_DTYPE)                 # (not zeroed: 320 MB per target range at ten million reads; rows beyond n are never handed out)
    n = C.c_uint64()
    st = _lib.PbaOverlapStats()
    self.check(self.lib.pba_overlap_all_probes(self.h, reads.h, t_lo, t_hi, C.c_void_p(d_entries_ptr), n_slots, mask, R, max_trial,
                                               overlap_min, kernel, _ptr(out), cap, C.byref(n), C.byref(st)), "overlap_all_probes")
    return out[:min(int(n.value), cap)], {k: getattr(st, k) for k, _ in _lib.PbaOverlapStats._fields_}


class ProbeTable:
    """The probe table of a read set (pba_probe_table): built once from a device entry list, scanned by every target range."""

    def __init__(self, ctx, d_entries_ptr, n_slots, mask, max_trial):
        self.ctx = ctx
        self.h = C.c_void_p()
        ctx.check(ctx.lib.pba_probe_table_create(ctx.h, C.c_void_p(d_entries_ptr), n_slots, mask, max_trial, C.byref(self.h)),
                  "probe_table_create")

    def close(self):
        if getattr(self, "h", None) and getattr(self.ctx, "h", None):
            self.ctx.lib.pba_probe_table_destroy(self.h)
        self.h = None

    __del__ = close

    @property
    def entries(self) -> int:
        return self.ctx.lib.pba_probe_table_entries(self.h)


def _overlap_all_table(self, reads, table, R, overlap_min=64, t_lo=0, t_hi=None, kernel=PBA_KERNEL_AUTO, cap=None):
    t_hi = reads.count if t_hi is None else t_hi
    cap = cap if cap is not None else max(1, (t_hi - t_lo) * max(reads.count - 1, 1))
    out = np.empty(cap, OVERLAP_DTYPE)                 # (not zeroed: 320 MB per target range at ten million reads; rows beyond n are never handed out)
    n = C.c_uint64()
    st = _lib.PbaOverlapStats()
    self.check(self.lib.pba_overlap_all_table(self.h, reads.h, t_lo, t_hi, table.h, R, overlap_min, kernel, _ptr(out), cap,
                                              C.byref(n), C.byref(st)), "overlap_all_table")
    return out[:min(int(n.value), cap)], {k: getattr(st, k) for k, _ in _lib.PbaOverlapStats._fields_}


def _overlap_all_sharded(self, reads, mask, R, max_trial=32, overlap_min=64, targets_per_call=10000, kernel=PBA_KERNEL_AUTO,
                         cap_per_target=None, t_lo=0, t_hi=None, table=None):
    """pba_overlap_all for read sets whose candidate lists do not fit one call (false candidates grow with the square of
    the read count; a call takes at most 2^32): the probe table is built once on the device (or handed in: the multi-GPU
    form builds it from the all-gathered entries), the targets [t_lo, t_hi) go through in ranges.  Same result as one call
    (the ranges are independent: this is also what ranks of a multi-GPU run do)."""
    import torch
    n = reads.count
    t_hi = n if t_hi is None else t_hi
    own = table is None
    if own:
        slots = n * 2 * max_trial
        probes = torch.full((max(slots, 1),), -1, dtype=torch.int64, device="cuda")
        self.overlap_probes(reads, 0, n, mask, max_trial, probes.data_ptr(), slots)
        torch.cuda.synchronize()
        table = ProbeTable(self, probes.data_ptr(), probes.numel(), mask, max_trial)
        del probes
    parts, total = [], None
    for lo in range(t_lo, t_hi, targets_per_call):
        hi = min(t_hi, lo + targets_per_call)
        cap = (hi - lo) * (cap_per_target or max(n - 1, 1))
        ov, st = self.overlap_all_table(reads, table, R, overlap_min, lo, hi, kernel, cap)
        parts.append(ov)
        if total is None:
            total = dict(st)
        else:
            for k in ("n_candidates", "n_pairs", "n_overlaps", "n_redo", "scan_ms", "sort_ms", "walk_ms", "n_big_targets", "n_prefiltered", "cap_fill", "cap_overflow", "n_listed"):
                total[k] += st[k]
            total["wide_first"] = max(total["wide_first"], st["wide_first"])
    if own:
        table.close()
    out = np.concatenate(parts) if parts else np.zeros(0, OVERLAP_DTYPE)
    return out, (total or {})


Context.overlap_all = _overlap_all
Context.overlap_probes = _overlap_probes
Context.overlap_all_probes = _overlap_all_probes
Context.overlap_all_sharded = _overlap_all_sharded
Context.overlap_all_table = _overlap_all_table


class SeqSet:
    def __init__(self, ctx: Context, h):
        self.ctx, self.h = ctx, h

    def close(self):
        if getattr(self, "h", None) and getattr(self.ctx, "h", None):
            self.ctx.lib.pba_seqs_destroy(self.h)
        self.h = None

    __del__ = close

    @property
    def count(self) -> int:
        return self.ctx.lib.pba_seqs_count(self.h)

    @property
    def max_len(self) -> int:
        return self.ctx.lib.pba_seqs_max_len(self.h)

    @property
    def packed_bytes(self) -> int:
        return self.ctx.lib.pba_seqs_packed_bytes(self.h)

    def lengths(self) -> np.ndarray:
        out = np.zeros(max(self.count, 1), np.uint32)
        self.ctx.check(self.ctx.lib.pba_seqs_lengths(self.h, _ptr(out), self.count))
        return out[:self.count]

    def export(self, d_dst_ptr: int, cap: int) -> np.ndarray:
        """Copy the packed arena into a device buffer; returns the byte offset of every sequence in it."""
        offs = np.zeros(max(self.count, 1), np.uint64)
        self.ctx.check(self.ctx.lib.pba_seqs_export(self.ctx.h, self.h, C.c_void_p(d_dst_ptr), cap, _ptr(offs)), "seqs_export")
        return offs[:self.count]

    @property
    def non_acgt(self) -> bool:
        return bool(self.ctx.lib.pba_seqs_non_acgt(self.h))

    def get_text(self, i: int) -> bytes:
        ln = int(self.lengths()[i])
        buf = C.create_string_buffer(ln + 1)
        self.ctx.check(self.ctx.lib.pba_seqs_get_text(self.ctx.h, self.h, i, buf, ln + 1), "get_text")
        return buf.raw[:ln]


class Consensus:
    """Vote boxes of an unlocked reference, resident in HBM (ref_seq.h: base_vote, vote_box, elect, evolve)."""

    def __init__(self, ctx: "Context", text: bytes, weight: int = 1, max_len: int = 0):
        self.ctx = ctx
        self.max_len = max_len or max(4 * len(text), 100000)
        self.h = C.c_void_p()
        ctx.check(ctx.lib.pba_cons_create(ctx.h, text, len(text), weight, self.max_len, C.byref(self.h)), "cons_create")

    def __del__(self):
        try:
            if self.h:
                self.ctx.lib.pba_cons_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def extent(self):
        e = np.zeros(3, np.int32)
        self.ctx.check(self.ctx.lib.pba_cons_extent(self.h, _ptr(e)))
        return e.tolist()

    def append(self, seg: bytes):
        self.ctx.check(self.ctx.lib.pba_cons_append(self.ctx.h, self.h, seg, len(seg)), "cons_append")

    def prepend(self, seg: bytes):
        self.ctx.check(self.ctx.lib.pba_cons_prepend(self.ctx.h, self.h, seg, len(seg)), "cons_prepend")

    def elect(self, pos, fwd, scripts, vals):
        """scripts / vals: one uint8 op array and one bytes object per script (vals[k] = the b element of op k)."""
        n = len(scripts)
        off = np.zeros(n + 1, np.uint64)
        off[1:] = np.cumsum([len(x) for x in scripts])
        ops = np.concatenate([np.asarray(x, np.uint8) for x in scripts] + [np.zeros(1, np.uint8)])
        vb = np.frombuffer(b"".join(vals) + b"\0", np.uint8)
        ne = np.array([len(x) for x in scripts], np.int32)
        pos = np.ascontiguousarray(pos, np.int32); fw = np.ascontiguousarray(fwd, np.uint8)
        self.ctx.check(self.ctx.lib.pba_cons_elect(self.ctx.h, self.h, n, _ptr(pos), _ptr(fw), _ptr(ops), _ptr(vb), _ptr(off),
                                                   _ptr(ne)), "cons_elect")

    def vote_pairs(self, A: "SeqSet", ref_seq: int, B: "SeqSet", pairs: np.ndarray, R: float, overlap_min: int = 64,
                   maxn: int = 0, maxm: int = 0) -> np.ndarray:
        """align + gate + elect for a batch, on the device (pba_cons_vote_pairs); returns the alignment results."""
        pairs = np.ascontiguousarray(pairs, PAIR_DTYPE)
        out = np.zeros(max(pairs.size, 1), RESULT_DTYPE)
        self.ctx.check(self.ctx.lib.pba_cons_vote_pairs(self.ctx.h, self.h, A.h, ref_seq, B.h, _ptr(pairs), pairs.size, R, maxn, maxm,
                                                        overlap_min, _ptr(out)), "cons_vote_pairs")
        return out[:pairs.size]

    def round(self, reads: "SeqSet", pool, mask: int, R: float, max_trial: int = 32, overlap_min: int = 64,
              buggy_seed_at: bool = False, kernel: int = PBA_KERNEL_AUTO, maxn: int = 26000, maxm: int = 6000):
        """One unlocked round of spaced_seed.cpp:420-446 over the reads `pool` (ids, in order): pba_cons_round.
        Returns (rows of the pool's reads in pool order, stats dict); the caller evolves."""
        pool = np.ascontiguousarray(pool, np.uint32)
        rows = np.zeros(max(reads.count, 1), SS_ROW_DTYPE)
        st = np.zeros(6, np.int32)
        self.ctx.check(self.ctx.lib.pba_cons_round(self.ctx.h, self.h, reads.h, _ptr(pool), pool.size, mask, R, max_trial,
                                                   overlap_min, int(buggy_seed_at), kernel, maxn, maxm, _ptr(rows), _ptr(st)),
                       "cons_round")
        names = ("n_found", "n_batches", "n_grown_fwd", "n_grown_bwd", "n_deferred", "n_index")
        return rows[pool], dict(zip(names, (int(x) for x in st)))

    def assemble(self, reads: "SeqSet", R: float, masks, picks, max_round: int = 100, max_trial: int = 32,
                 overlap_min: int = 64, buggy_seed_at: bool = False, kernel: int = PBA_KERNEL_AUTO, maxn: int = 26000,
                 maxm: int = 6000):
        """spaced_seed's main loop without -l (pba_cons_assemble); returns (rows, found_round, log list of dicts)."""
        n = max(reads.count, 1)
        rows = np.zeros(n, SS_ROW_DTYPE)
        fr = np.zeros(n, np.int32)
        log = np.zeros(max(max_round, 1), np.dtype([("round", "<i4"), ("mask", "<u4"), ("n_tried", "<i4"), ("n_found", "<i4")]))
        rl = np.zeros(max(max_round, 1), np.int32)
        masks = np.ascontiguousarray(masks, np.uint32); picks = np.ascontiguousarray(picks, np.uint32)
        nr = C.c_int()
        self.ctx.check(self.ctx.lib.pba_cons_assemble(self.ctx.h, self.h, reads.h, R, max_trial, overlap_min, int(buggy_seed_at),
                                                      kernel, maxn, maxm, _ptr(masks), masks.size, _ptr(picks), picks.size,
                                                      max_round, _ptr(rows), _ptr(fr), _ptr(log), _ptr(rl), log.size,
                                                      C.byref(nr)), "cons_assemble")
        out = [dict(zip(log.dtype.names, (int(x) for x in l))) for l in log[:nr.value]]
        for k, d in enumerate(out):
            d["ref_len"] = int(rl[k])
        return rows[:reads.count], fr[:reads.count], out

    def evolve(self) -> bytes:
        cap = 3 * self.max_len
        buf = C.create_string_buffer(cap)
        n = C.c_int32()
        self.ctx.check(self.ctx.lib.pba_cons_evolve(self.ctx.h, self.h, buf, cap, C.byref(n)), "cons_evolve")
        return buf.raw[:n.value]

    def dump(self):
        e = self.extent()
        cap = e[1] - e[0]
        sel = np.zeros((max(cap, 1), 4), np.uint16); sup = np.zeros((max(cap, 1), 4), np.uint16); tot = np.zeros(max(cap, 1), np.int32)
        n = C.c_int32()
        self.ctx.check(self.ctx.lib.pba_cons_dump(self.ctx.h, self.h, _ptr(sel), _ptr(sup), _ptr(tot), cap, C.byref(n)), "cons_dump")
        return sel[:n.value], sup[:n.value], tot[:n.value], e

    def text(self) -> bytes:
        e = self.extent()
        cap = e[1] - e[0]
        buf = C.create_string_buffer(cap + 1)
        n = C.c_int32()
        self.ctx.check(self.ctx.lib.pba_cons_text(self.ctx.h, self.h, buf, cap, C.byref(n)), "cons_text")
        return buf.raw[:n.value]


def script_vals(ops: np.ndarray, seg: bytes, fwd: bool = True) -> bytes:
    """edits[k].val of an edit script (seq_aligner.h:218,224): the b element a MATCH / INSERT consumes, 0 for a DELETE.
    seg: the accessor's elements in memory order (a backward accessor starts at its last byte)."""
    elems = np.frombuffer(seg if fwd else seg[::-1], np.uint8)
    ops = np.asarray(ops, np.uint8)
    takes = ops != 3
    idx = np.cumsum(takes) - 1
    out = np.where(takes, elems[np.minimum(idx, max(elems.size - 1, 0))] if elems.size else 0, 0).astype(np.uint8)
    return out.tobytes()


class SeedIndex:
    def __init__(self, ctx: Context, h):
        self.ctx, self.h = ctx, h

    def close(self):
        if getattr(self, "h", None) and getattr(self.ctx, "h", None):
            self.ctx.lib.pba_index_destroy(self.h)
        self.h = None

    __del__ = close

    @property
    def entries(self) -> int:
        return self.ctx.lib.pba_index_entries(self.h)

    @property
    def visited(self) -> int:
        return self.ctx.lib.pba_index_visited(self.h)

    def dump(self):
        n = self.entries
        keys = np.zeros(max(n, 1), np.uint32)
        pos = np.zeros(max(n, 1), np.int32)
        got = C.c_uint64()
        self.ctx.check(self.ctx.lib.pba_index_dump(self.ctx.h, self.h, _ptr(keys), _ptr(pos), n, C.byref(got)),
                       "index_dump")
        return keys[:n], pos[:n]

    def find(self, keys: np.ndarray):
        keys = np.ascontiguousarray(keys, np.uint32)
        off = np.zeros(keys.size + 1, np.uint64)
        lib, c = self.ctx.lib, self.ctx
        c.check(lib.pba_index_find(c.h, self.h, _ptr(keys), keys.size, _ptr(off), None, 0), "index_find")
        total = int(off[-1])
        pos = np.zeros(max(total, 1), np.int32)
        c.check(lib.pba_index_find(c.h, self.h, _ptr(keys), keys.size, _ptr(off), _ptr(pos), total), "index_find")
        return off, pos[:total]
