"""ctypes binding of libpba.so (include/pba.h).  No fallback: a missing library is an error."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PBA_LIB_PATH") or os.path.join(HERE, "lib", "libpba.so")   # (PBA_LIB_PATH: a tuning build, tools/build_variant.py)

PBA_OK = 0
PBA_E_INVALID, PBA_E_NOMEM, PBA_E_HIP, PBA_E_TOOLONG, PBA_E_NODEVICE, PBA_E_ALPHABET = -1, -2, -3, -4, -5, -6
PBA_INDEX_ALL, PBA_INDEX_HEAD_TAIL = 0, 1
PBA_KERNEL_AUTO, PBA_KERNEL_ROWSWEEP, PBA_KERNEL_BITVEC = 0, 1, 2
PBA_A_BACKWARD, PBA_B_BACKWARD = 1, 2


class PbaPair(C.Structure):
    _fields_ = [("a_seq", C.c_uint32), ("a_pos", C.c_int32), ("a_len", C.c_int32),
                ("b_seq", C.c_uint32), ("b_pos", C.c_int32), ("b_len", C.c_int32), ("flags", C.c_uint32)]


class PbaResult(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("rc", "cost", "matlen_a", "matlen_b", "len_a", "len_b", "max_dst", "diag_cost")]


class PbaLocRow(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("read", "nseq", "found", "j", "pos", "cost", "seglen", "matlen_a", "matlen_b", "n_pairs", "diag_cost")]


class PbaLocStats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("n_reads_kept", "n_probe_hits", "n_pairs", "n_located", "n_cells")]


class PbaOverlapStats(C.Structure):
    _fields_ = [("n_probe_entries", C.c_uint64), ("n_candidates", C.c_uint64), ("n_pairs", C.c_uint64),
                ("n_overlaps", C.c_uint64), ("n_redo", C.c_uint64), ("scan_ms", C.c_float), ("sort_ms", C.c_float),
                ("walk_ms", C.c_float), ("wide_first", C.c_uint32), ("table_ms", C.c_float), ("n_big_targets", C.c_uint32),
                ("n_prefiltered", C.c_uint64), ("cap_fill", C.c_uint32), ("cap_overflow", C.c_uint32), ("n_listed", C.c_uint64)]


class PbaProfile(C.Structure):
    _fields_ = [("index_ms", C.c_float), ("align_ms", C.c_float), ("align_redo_ms", C.c_float),
                ("nb_first", C.c_uint32), ("nb_redo", C.c_uint32), ("n_first", C.c_uint32), ("n_redo", C.c_uint32)]


class PbaSsRow(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("read", "found", "j", "dir", "ref_pos", "cost", "matlen_a", "matlen_b", "n_trials", "n_pairs")]


# every symbol include/pba.h declares: (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "pba_encode16": (C.c_uint32, [C.c_char_p]),
    "pba_decode16": (None, [C.c_uint32, C.c_char_p]),
    "pba_text2bin": (C.c_size_t, [C.c_char_p, C.c_size_t, _P, C.c_size_t]),
    "pba_bin2text": (C.c_size_t, [_P, C.c_char_p, C.c_size_t]),
    "pba_seed_at": (C.c_uint32, [_P, C.c_int]),
    "pba_seed_at_fixed": (C.c_uint32, [_P, C.c_int]),
    "pba_mask_from_pattern": (C.c_uint32, [C.c_char_p]),
    "pba_value_at": (C.c_char, [C.c_uint8, C.c_int]),
    "pba_open_binary": (C.c_size_t, [_P, C.c_size_t, C.c_uint32, C.c_uint32, _P, C.c_size_t, _P]),
    "pba_synth_genome": (None, [C.c_uint64, _P, C.c_size_t]),
    "pba_synth_reads": (C.c_int, [C.c_uint64, _P, C.c_size_t, C.c_uint32, C.c_uint32, C.c_double, C.c_double,
                                  C.c_double, _P, _P, C.c_int]),
    "pba_synth_reads_range": (C.c_int, [C.c_uint64, _P, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, C.c_double,
                                        C.c_double, _P, _P, C.c_int]),
    "pba_ctx_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "pba_ctx_destroy": (None, [_P]),
    "pba_ctx_error": (C.c_char_p, [_P]),
    "pba_ctx_set_stream": (C.c_int, [_P, _P]),
    "pba_ctx_sync": (C.c_int, [_P]),
    "pba_ctx_trim": (C.c_int, [_P]),
    "pba_ctx_last_profile": (C.c_int, [_P, _P]),
    "pba_ctx_device_info": (C.c_int, [_P, C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                      C.POINTER(C.c_uint64)]),
    "pba_seqs_from_text": (C.c_int, [_P, _P, _P, C.c_uint32, C.c_int, C.POINTER(_P)]),
    "pba_seqs_from_device_text": (C.c_int, [_P, _P, _P, C.c_uint32, C.c_uint64, C.c_uint32, C.POINTER(_P)]),
    "pba_seqs_from_records": (C.c_int, [_P, _P, C.c_size_t, C.c_uint32, C.c_uint32, C.POINTER(_P)]),
    "pba_seqs_export": (C.c_int, [_P, _P, _P, C.c_uint64, _P]),
    "pba_seqs_from_device_packed": (C.c_int, [_P, _P, C.c_uint64, _P, _P, C.c_uint32, C.c_int, C.POINTER(_P)]),
    "pba_seqs_non_acgt": (C.c_int, [_P]),
    "pba_seqs_destroy": (None, [_P]),
    "pba_seqs_count": (C.c_uint32, [_P]),
    "pba_seqs_max_len": (C.c_uint32, [_P]),
    "pba_seqs_packed_bytes": (C.c_uint64, [_P]),
    "pba_seqs_lengths": (C.c_int, [_P, _P, C.c_uint32]),
    "pba_seqs_get_text": (C.c_int, [_P, _P, C.c_uint32, C.c_char_p, C.c_size_t]),
    "pba_index_build": (C.c_int, [_P, _P, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(_P)]),
    "pba_index_scan": (C.c_int, [_P, _P, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.c_uint32, _P, C.c_uint64,
                                 C.POINTER(C.c_uint64)]),
    "pba_index_from_entries": (C.c_int, [_P, _P, C.c_uint64, C.c_uint32, C.c_int, C.c_uint32, C.POINTER(_P)]),
    "pba_index_destroy": (None, [_P]),
    "pba_index_entries": (C.c_uint64, [_P]),
    "pba_index_visited": (C.c_uint32, [_P]),
    "pba_index_dump": (C.c_int, [_P, _P, _P, _P, C.c_uint64, C.POINTER(C.c_uint64)]),
    "pba_index_find": (C.c_int, [_P, _P, _P, C.c_uint32, _P, _P, C.c_uint64]),
    "pba_align_batch": (C.c_int, [_P, _P, _P, _P, C.c_size_t, C.c_double, C.c_int, C.c_int, C.c_int, _P]),
    "pba_align_text": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, _P]),
    "pba_align_text_trace": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, _P, _P,
                                       C.c_int32, C.POINTER(C.c_int32)]),
    "pba_align_text_matrix": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, _P, _P, _P,
                                        C.c_uint64, C.POINTER(C.c_int32)]),
    "pba_align_batch_trace": (C.c_int, [_P, _P, _P, _P, C.c_size_t, C.c_double, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P]),
    "pba_locate": (C.c_int, [_P, _P, _P, C.c_uint32, _P, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                             _P, _P]),
    "pba_spaced_round": (C.c_int, [_P, _P, _P, C.c_uint32, _P, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "pba_spaced_multi": (C.c_int, [_P, _P, C.c_uint32, _P, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, _P, C.c_int,
                                   C.c_int, _P, _P, _P, C.c_int, C.POINTER(C.c_int)]),
    "pba_overlap_all": (C.c_int, [_P, _P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, C.c_int, C.c_int, C.c_int, _P,
                                  C.c_uint64, C.POINTER(C.c_uint64), _P]),
    "pba_overlap_probes": (C.c_int, [_P, _P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, _P, C.c_uint64, C.POINTER(C.c_uint64)]),
    "pba_overlap_all_probes": (C.c_int, [_P, _P, C.c_uint32, C.c_uint32, _P, C.c_uint64, C.c_uint32, C.c_double, C.c_int, C.c_int,
                                         C.c_int, _P, C.c_uint64, C.POINTER(C.c_uint64), _P]),
    "pba_probe_table_create": (C.c_int, [_P, _P, C.c_uint64, C.c_uint32, C.c_int, C.POINTER(_P)]),
    "pba_probe_table_destroy": (None, [_P]),
    "pba_probe_table_entries": (C.c_uint64, [_P]),
    "pba_overlap_all_table": (C.c_int, [_P, _P, C.c_uint32, C.c_uint32, _P, C.c_double, C.c_int, C.c_int, _P, C.c_uint64,
                                        C.POINTER(C.c_uint64), _P]),
    "pba_cons_create": (C.c_int, [_P, C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    "pba_cons_destroy": (None, [_P]),
    "pba_cons_extent": (C.c_int, [_P, _P]),
    "pba_cons_append": (C.c_int, [_P, _P, C.c_char_p, C.c_int]),
    "pba_cons_prepend": (C.c_int, [_P, _P, C.c_char_p, C.c_int]),
    "pba_cons_elect": (C.c_int, [_P, _P, C.c_uint32, _P, _P, _P, _P, _P, _P]),
    "pba_cons_vote_pairs": (C.c_int, [_P, _P, _P, C.c_uint32, _P, _P, C.c_size_t, C.c_double, C.c_int, C.c_int, C.c_int, _P]),
    "pba_cons_round": (C.c_int, [_P, _P, _P, _P, C.c_uint32, C.c_uint32, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_int, _P, _P]),
    "pba_cons_assemble": (C.c_int, [_P, _P, _P, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int,
                                    _P, C.c_int, C.c_int, _P, _P, _P, _P, C.c_int, C.POINTER(C.c_int)]),
    "pba_cons_evolve": (C.c_int, [_P, _P, _P, C.c_int, C.POINTER(C.c_int32)]),
    "pba_cons_dump": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.POINTER(C.c_int32)]),
    "pba_cons_text": (C.c_int, [_P, _P, _P, C.c_int, C.POINTER(C.c_int32)]),
    "pba_strerror": (C.c_char_p, [C.c_int]),
}

# include/pba_dist.h (libpba_dist.so: the exchange over RCCL for C / C++ hosts)
DIST_LIB_PATH = os.path.join(os.path.dirname(LIB_PATH), "libpba_dist.so")
DIST_SYMBOLS = {
    "pba_dist_unique_id": (C.c_int, [_P]),
    "pba_dist_comm_create": (C.c_int, [_P, C.c_int, C.c_int, _P, C.POINTER(_P)]),
    "pba_dist_comm_destroy": (None, [_P]),
    "pba_dist_rank": (C.c_int, [_P]),
    "pba_dist_world": (C.c_int, [_P]),
    "pba_dist_shard": (None, [C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "pba_dist_all_gather": (C.c_int, [_P, _P, C.c_uint64, _P]),
    "pba_dist_all_reduce_u64": (C.c_int, [_P, _P, C.c_uint32, C.c_int]),
    "pba_dist_index_build": (C.c_int, [_P, _P, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(_P)]),
    "pba_dist_gather_reads": (C.c_int, [_P, _P, C.POINTER(_P)]),
    "pba_dist_probe_table": (C.c_int, [_P, _P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(_P)]),
}

_lib = None
_dist = None


def load_dist() -> C.CDLL:
    """Load libpba_dist.so (after libpba.so, so that both bind the HIP runtime and the RCCL torch has mapped)."""
    global _dist
    if _dist is not None:
        return _dist
    load()
    if not os.path.exists(DIST_LIB_PATH):
        raise RuntimeError(f"{DIST_LIB_PATH} is missing: build it with `python -m pacbioassembly_amd.build`")
    lib = C.CDLL(DIST_LIB_PATH)
    for name, (res, args) in DIST_SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _dist = lib
    return lib


def load() -> C.CDLL:
    """Load libpba.so and bind every declared symbol.  Raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m pacbioassembly_amd.build` "
            "(there is no CPU fallback for the HIP path)")
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64.so.7 / libhsa-runtime64 and
    # libpba.so needs the same SONAME.  Importing torch first makes the loader bind libpba.so to the
    # copy torch already mapped, so torch tensors, streams and RCCL share one runtime with our kernels
    # (two HSA runtimes in one process leave the second without a GPU).
    if os.environ.get("PBA_NO_TORCH", "0") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)       # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
