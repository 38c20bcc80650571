/*
 * pba.h -- C ABI of the MI355X seed-and-extend overlap engine (libpba.so).
 *
 * This is the drop-in boundary.  The reference (vmingchen/PacBioAssembly) has no
 * FFI layer: its mains and tests compile against the header-level C++ API in
 * src/dna_seq.h, src/seq_aligner.h and src/ref_seq.h.  include/compat/ re-creates
 * that API (same class, method and field names) on top of the entry points below;
 * each entry point names the reference interface it replaces (file:line into
 * /root/reference/).  Plain pointers and sizes only; no exceptions cross the
 * boundary; every function that can fail returns a pba_status.
 *
 * Threading: a pba_ctx is bound to one GPU and one HIP stream and must be used by
 * one host thread at a time (the reference is single-threaded with global
 * singletons, spaced_seed.cpp:71-96).  One process per GPU, one ctx per process.
 *
 * There is no CPU fallback: with no usable GPU pba_ctx_create fails with
 * PBA_E_NODEVICE and nothing else in the device API can be called.
 */
#ifndef PBA_H
#define PBA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PBA_VERSION 1

typedef enum {
    PBA_OK = 0,
    PBA_E_INVALID = -1,     /* bad argument */
    PBA_E_NOMEM = -2,       /* host or device allocation failed */
    PBA_E_HIP = -3,         /* a HIP call or kernel failed; see pba_ctx_error */
    PBA_E_TOOLONG = -4,     /* sequence longer than the engine supports */
    PBA_E_NODEVICE = -5,    /* no usable gfx950 device */
    PBA_E_ALPHABET = -6     /* byte outside ACGT where the packed path needs ACGT */
} pba_status;

/* ------------------------------------------------------------------------ */
/* Host-side codec.  Pure functions, no ctx, bit-compatible with dna_seq.   */
/* ------------------------------------------------------------------------ */

/* dna_seq::encode (dna_seq.h:86): 16 chars -> u32, byte k = bases 4k..4k+3 */
uint32_t pba_encode16(const char *text16);
/* dna_seq::decode (dna_seq.h:101) */
void pba_decode16(uint32_t code, char *text16);
/* dna_seq::text2bin (dna_seq.h:113) with an explicit length: writes the record
 * [u32 len][ceil(len/4) packed bytes]; returns bytes written, 0 if cap is too small */
size_t pba_text2bin(const char *text, size_t tlen, uint8_t *record, size_t cap);
/* dna_seq::bin2text (dna_seq.h:133): returns the length, writes the NUL; 0 if cap <= len */
size_t pba_bin2text(const uint8_t *record, char *text, size_t cap);
/* dna_seq::seed_at (dna_seq.h:62), BUG-COMPATIBLE: for pos%4==0 it returns the word at
 * byte offset pos, exactly like the reference */
uint32_t pba_seed_at(const uint8_t *record, int pos);
/* the window seed_at was meant to return: == pba_encode16(text + pos) */
uint32_t pba_seed_at_fixed(const uint8_t *record, int pos);
/* parse_pattern (spaced_seed.cpp:167) / locator.cpp:51-54: '1' -> care, else don't care */
uint32_t pba_mask_from_pattern(const char *pattern);
/* dna_seq::value_at (dna_seq.h:78) */
char pba_value_at(uint8_t packed_byte, int idx);
/* record walk of open_binary (spaced_seed.cpp:330-342): byte offsets of the records with
 * min_excl < len < max_excl; returns how many were kept (writes at most cap offsets) */
size_t pba_open_binary(const uint8_t *file, size_t file_len, uint32_t min_excl, uint32_t max_excl,
                       uint64_t *offsets, size_t cap, size_t *n_records_total);

/* ------------------------------------------------------------------------ */
/* Synthetic workload generator (bench/test infrastructure, host code).     */
/* Integer-only counter RNG: the same bytes on every machine.               */
/* ------------------------------------------------------------------------ */
void pba_synth_genome(uint64_t seed, char *out, size_t n);
/* n_reads reads of read_len bases, forward strand, start uniform in [0, L - 1.5*read_len),
 * per-step error p_ins / p_del / p_sub.  out holds n_reads*read_len chars
 * (no separators); starts (nullable) receives each read's genome start. */
int pba_synth_reads(uint64_t seed, const char *genome, size_t L, uint32_t n_reads, uint32_t read_len,
                    double p_ins, double p_del, double p_sub, char *out, uint32_t *starts, int nthreads);
/* reads [r_lo, r_hi) of the same set (read r is a function of (seed, r) only): what one rank of a multi-GPU run generates.
 * out holds (r_hi - r_lo) * read_len chars, starts (nullable) r_hi - r_lo slots. */
int pba_synth_reads_range(uint64_t seed, const char *genome, size_t L, uint32_t r_lo, uint32_t r_hi, uint32_t read_len,
                          double p_ins, double p_del, double p_sub, char *out, uint32_t *starts, int nthreads);

/* ------------------------------------------------------------------------ */
/* Context                                                                  */
/* ------------------------------------------------------------------------ */
typedef struct pba_ctx pba_ctx;
int pba_ctx_create(int device_id, pba_ctx **ctx);
void pba_ctx_destroy(pba_ctx *ctx);
/* text of the last failure on this ctx (never NULL) */
const char *pba_ctx_error(const pba_ctx *ctx);
/* run on a caller-owned hipStream_t (e.g. torch's current stream); NULL = the ctx's own */
int pba_ctx_set_stream(pba_ctx *ctx, void *hip_stream);
int pba_ctx_sync(pba_ctx *ctx);
/* The ctx keeps the work buffers of its drivers between calls (candidate arrays, per-read rows, traceback scratch: mapping
 * gigabytes anew costs more than the kernels that use them).  pba_ctx_trim gives them back to the device. */
int pba_ctx_trim(pba_ctx *ctx);
/* HIP-event timings (on the ctx's stream) of the most recent pba_index_build / pba_locate /
 * pba_align_batch / pba_spaced_round on this ctx: measurement support for bench.py */
typedef struct {
    float index_ms;        /* pba_index_build: first kernel to last kernel */
    float align_ms;        /* first launch of the aligning kernel (all pairs / reads) */
    float align_redo_ms;   /* second launch (pairs / reads the narrow band could not certify); 0 if none */
    uint32_t nb_first;     /* blocks per lane of the bit-vector array in the first launch; 0 = row sweep */
    uint32_t nb_redo;
    uint32_t n_first;      /* pairs or reads in the first launch */
    uint32_t n_redo;       /* pairs or reads re-run at the reference band */
} pba_profile;
int pba_ctx_last_profile(const pba_ctx *ctx, pba_profile *out);
/* device facts for the bench report */
int pba_ctx_device_info(const pba_ctx *ctx, char *name, size_t name_cap, int *n_cu, int *clock_mhz,
                        uint64_t *hbm_bytes);

/* ------------------------------------------------------------------------ */
/* Sequence sets resident in HBM, 2-bit packed in the reference byte layout */
/* (dna_seq.h:147-159: first base in bits 7:6), each sequence 16-B aligned. */
/* Replaces: the mmap'd read buffer (spaced_seed.cpp:310-345), contig[] and */
/* sequence[] (locator.cpp:26-27), ref_seq::txt_buf (ref_seq.h:370).        */
/* ------------------------------------------------------------------------ */
typedef struct pba_seqs pba_seqs;
/* text: concatenated ASCII; sequence i = text[offsets[i] .. offsets[i+1]).  Bytes are packed
 * with C2I (dna_seq.h:21).  If strict_acgt != 0 a byte outside "ACGT" fails with
 * PBA_E_ALPHABET (the packed DP compares codes, the reference compares bytes: they agree
 * exactly on ACGT input).  Packing runs on the GPU.  With strict_acgt == 0 such bytes are packed
 * as code 3 (that is what the seed index of locator.cpp:62-66 sees), the set remembers it, and the
 * aligning entry points (pba_align_batch, pba_locate, pba_spaced_round) refuse it with
 * PBA_E_ALPHABET rather than return scores the reference would not: use pba_align_text there. */
int pba_seqs_from_text(pba_ctx *ctx, const char *text, const uint64_t *offsets, uint32_t n,
                       int strict_acgt, pba_seqs **out);
/* same, but text/offsets are DEVICE pointers (inputs already resident in HBM) */
int pba_seqs_from_device_text(pba_ctx *ctx, const void *d_text, const void *d_offsets_u64, uint32_t n,
                              uint64_t total_bytes, uint32_t max_len, pba_seqs **out);
/* reference binary read file ([u32 len][packed])*, kept records min_excl < len < max_excl
 * (spaced_seed.cpp:330-342); payloads are uploaded as they are, no re-packing */
int pba_seqs_from_records(pba_ctx *ctx, const uint8_t *file, size_t file_len, uint32_t min_excl,
                          uint32_t max_excl, pba_seqs **out);
/* Multi-GPU exchange of packed reads (every rank packs its own shard, the shards are all-gathered over RCCL):
 * pba_seqs_export copies the set's packed arena (pba_seqs_packed_bytes bytes) into the DEVICE buffer d_dst and returns the
 * byte offset of every sequence's first packed byte in it (offsets: n host slots); pba_seqs_from_device_packed builds a
 * set from such bytes resident on the device -- the gathered buffer, offsets[i] = where sequence i starts in it (rank
 * base + exported offset), lengths in bases -- without re-packing (one device copy; the bit planes are rebuilt).
 * non_acgt: whether any contributing set was built with strict_acgt == 0 and met a byte outside ACGT. */
int pba_seqs_export(pba_ctx *ctx, const pba_seqs *s, void *d_dst, uint64_t cap, uint64_t *offsets);
int pba_seqs_from_device_packed(pba_ctx *ctx, const void *d_packed, uint64_t n_bytes, const uint64_t *offsets,
                                const uint32_t *lengths, uint32_t n, int non_acgt, pba_seqs **out);
int pba_seqs_non_acgt(const pba_seqs *s);
void pba_seqs_destroy(pba_seqs *s);
uint32_t pba_seqs_count(const pba_seqs *s);
uint32_t pba_seqs_max_len(const pba_seqs *s);
uint64_t pba_seqs_packed_bytes(const pba_seqs *s);
int pba_seqs_lengths(const pba_seqs *s, uint32_t *lengths, uint32_t cap);
/* unpack sequence i back to text (bin2text), for round-trip checks */
int pba_seqs_get_text(pba_ctx *ctx, const pba_seqs *s, uint32_t i, char *text, size_t cap);

/* ------------------------------------------------------------------------ */
/* Seed-hit index.  Replaces hash_table = hash_map<unsigned, list<int>>     */
/* (common.h:54) and its two builders.                                      */
/* ------------------------------------------------------------------------ */
typedef struct pba_index pba_index;
typedef enum {
    PBA_INDEX_ALL = 0,        /* locator.cpp:62-66: every position [0,len), tail windows padded with code 3 */
    PBA_INDEX_HEAD_TAIL = 1   /* ref_seq::get_seedmap, ref_seq.h:291-311: head ascending, then tail descending */
} pba_index_mode;
/* index sequence `seq` of `target` under `mask`; entries whose masked key is 0 are dropped
 * (ref_seq.h:300,307; locator.cpp:64); per-key hit order = the reference's insertion order */
int pba_index_build(pba_ctx *ctx, const pba_seqs *target, uint32_t seq, uint32_t mask, int mode,
                    pba_index **out);
/* Multi-GPU form of the build (one process per GPU; the exchange itself is the caller's RCCL
 * all-gather on device buffers): rank `part` of `nparts` scans its contiguous slice of the
 * reference's visiting order and writes the raw entries (key << 32 | ordinal, any order) to the
 * DEVICE buffer d_entries (cap u64 slots) ... */
int pba_index_scan(pba_ctx *ctx, const pba_seqs *target, uint32_t seq, uint32_t mask, int mode, uint32_t part,
                   uint32_t nparts, void *d_entries, uint64_t cap, uint64_t *n_out);
/* ... and every rank builds the same index from the gathered DEVICE list (n slots; slots holding
 * all-ones are padding).  seq_len / mode / mask must be those of the scan. */
int pba_index_from_entries(pba_ctx *ctx, const void *d_entries, uint64_t n, uint32_t mask, int mode,
                           uint32_t seq_len, pba_index **out);
void pba_index_destroy(pba_index *ix);
uint64_t pba_index_entries(const pba_index *ix);
/* what get_seedmap returns (ref_seq.h:310): positions visited, not entries kept */
uint32_t pba_index_visited(const pba_index *ix);
/* all entries sorted by key, reference hit order within a key; returns PBA_OK and *n */
int pba_index_dump(pba_ctx *ctx, const pba_index *ix, uint32_t *keys, int32_t *pos, uint64_t cap, uint64_t *n);
/* hash_table::find (locator.cpp:76, spaced_seed.cpp:265) for a batch of keys: for key q the hits are
 * hit_pos[hit_off[q] .. hit_off[q+1]) in reference list order.  hit_off has n_keys+1 slots. */
int pba_index_find(pba_ctx *ctx, const pba_index *ix, const uint32_t *keys, uint32_t n_keys,
                   uint64_t *hit_off, int32_t *hit_pos, uint64_t hit_cap);

/* ------------------------------------------------------------------------ */
/* Banded edit-distance extension.  Replaces seq_aligner<MAXN,MAXM>::align  */
/* (seq_aligner.h:92-125) and its result fields (seq_aligner.h:73-81).      */
/* ------------------------------------------------------------------------ */
typedef struct {
    uint32_t a_seq;      /* sequence id in set A */
    int32_t  a_pos;      /* accessor origin: base index inside the sequence (dna_seq.h:191) */
    int32_t  a_len;      /* accessor length */
    uint32_t b_seq;
    int32_t  b_pos;
    int32_t  b_len;
    uint32_t flags;      /* PBA_A_BACKWARD / PBA_B_BACKWARD: element k is seq[pos-k] (dna_seq.h:211,221) */
} pba_pair;
#define PBA_A_BACKWARD 1u
#define PBA_B_BACKWARD 2u

typedef struct {
    int32_t rc;          /* align()'s return: -1 or matlen_b (seq_aligner.h:106,111,114,124) */
    int32_t cost;        /* final_cost() (seq_aligner.h:130); 0 when rc < 0 */
    int32_t matlen_a;    /* 0 when rc < 0 unless only the acceptance test failed */
    int32_t matlen_b;
    int32_t len_a;       /* parameter block, seq_aligner.h:94-102 */
    int32_t len_b;
    int32_t max_dst;
    int32_t diag_cost;   /* get_cost(m, m), m = min(len_a, len_b): the end of the diagonal -- what locator.cpp:86 prints as
                          * get_cost(len - j, len - j) when len_b >= len_a; -1 when the sweep stopped before that row */
} pba_result;

typedef enum {
    PBA_KERNEL_AUTO = 0,
    PBA_KERNEL_ROWSWEEP = 1,   /* full-band row sweep, band row in LDS */
    PBA_KERNEL_BITVEC = 2      /* bit-parallel delta encoding, exact */
} pba_kernel;

/* maxn/maxm: the template limits of the seq_aligner instantiation being replaced (size guard,
 * seq_aligner.h:104: len_a >= maxn+maxm || max_dst >= maxm -> -1); maxn <= 0 disables the guard */
int pba_align_batch(pba_ctx *ctx, const pba_seqs *A, const pba_seqs *B, const pba_pair *pairs, size_t n,
                    double R, int maxn, int maxm, int kernel, pba_result *out);
/* one pair given as host text, RAW BYTE comparison exactly like seq_aligner.h:136 (any bytes,
 * case-sensitive).  a/b are accessor origins: element k is p[k] when fwd, p[-k] otherwise. */
int pba_align_text(pba_ctx *ctx, const char *a, int a_fwd, int a_len, const char *b, int b_fwd, int b_len,
                   double R, int maxn, int maxm, pba_result *out);

/* Traceback (seq_aligner.h:115-116, 214-233): the same alignment plus its edit script, ops[k] = 1 MATCH,
 * 2 INSERT, 3 DELETE in the order seq_aligner::edits[] holds them (the `val` of a MATCH / INSERT is the b
 * element it consumes, which the caller can read off b while replaying the ops).
 * Text form: raw bytes, full-band row sweep with one parent code per band cell in HBM
 * ((len_a+1) * (2*max_dst+1) bytes): for single pairs (display, the compat seq_aligner). */
int pba_align_text_trace(pba_ctx *ctx, const char *a, int a_fwd, int a_len, const char *b, int b_fwd, int b_len,
                         double R, int maxn, int maxm, pba_result *out, uint8_t *ops, int32_t ops_cap, int32_t *nedit);
/* The DP matrix itself, for callers that read it (seq_aligner<>::mat, get_cost / get_parent, seq_aligner.h:81,131-134;
 * locator.cpp:86): cost[i * W + c] and parent[i * W + c] for cell (i, j) with W = 2*max_dst + 1 and c = j - i + max_dst, the
 * reference's own diagonal-stripe layout; (len_a + 1) * W cells (cap_cells must hold them: len_a and max_dst follow from the
 * lengths and R as in seq_aligner.h:94-102).  Cells the call writes -- init_cell's borders and the band of rows
 * 1 .. *rows_swept (all rows, or up to the row of the early failure) -- hold their values; every other cell holds cost
 * 0xFFFF and parent 0 (the reference would hand back whatever an earlier call left there).  One pair, raw bytes. */
int pba_align_text_matrix(pba_ctx *ctx, const char *a, int a_fwd, int a_len, const char *b, int b_fwd, int b_len, double R,
                          int maxn, int maxm, pba_result *out, uint16_t *cost, uint8_t *parent, uint64_t cap_cells,
                          int32_t *rows_swept);
/* Batch form on packed sets: pair q's script goes to ops[ops_off[q] .. ops_off[q+1]) (needs a_len + b_len
 * slots), its length to nedit[q] (0 when rc < 0).  kernel: PBA_KERNEL_AUTO / _BITVEC run the bit-vector array
 * and stream 2 parent bits per processed cell into a per-wavefront scratch area that the same wavefront walks
 * back (HBM-bound: ~16 MB written per 15 kb pair, any batch size); PBA_KERNEL_ROWSWEEP keeps one parent byte
 * per band cell for every pair of the batch at once (135 MB per 15 kb pair; the cross-check). */
int pba_align_batch_trace(pba_ctx *ctx, const pba_seqs *A, const pba_seqs *B, const pba_pair *pairs, size_t n,
                          double R, int maxn, int maxm, int kernel, pba_result *out, uint8_t *ops,
                          const uint64_t *ops_off, int32_t *nedit);

/* ------------------------------------------------------------------------ */
/* Consensus voting and reference growth: the unlocked half of ref_seq      */
/* (ref_seq.h:25-188 base_vote / vote_box, :207-242 ctor / append / prepend, */
/* :317-362 evolve / elect).  One vote box per reference position, resident */
/* in HBM; max_len plays MAX_SEQ_LEN (common.h:31): the object spans        */
/* 3*max_len positions with its origin (`beg`) at max_len.                  */
/* ------------------------------------------------------------------------ */
typedef struct pba_cons pba_cons;
/* ref_seq(const char*, int len, bool, int w) (ref_seq.h:218-225): every box starts as vote_box(text[i], weight) */
int pba_cons_create(pba_ctx *ctx, const char *text, int len, int weight, int max_len, pba_cons **out);
void pba_cons_destroy(pba_cons *c);
/* extent[0..2] = pre - beg, post - beg, end - beg (ref_seq.h:364-367) */
int pba_cons_extent(const pba_cons *c, int32_t *extent);
/* ref_seq::append / prepend (ref_seq.h:227-242): seg holds the new characters in text order */
int pba_cons_append(pba_ctx *ctx, pba_cons *c, const char *seg, int len);
int pba_cons_prepend(pba_ctx *ctx, pba_cons *c, const char *seg, int len);
/* ref_seq::elect + apply_edits (ref_seq.h:352-362, 25-41) for n edit scripts at once (votes commute): script q
 * starts at reference position pos[q] (relative to beg, must be contained), runs forward (fwd[q] != 0) or
 * backward, and is ops/vals[ops_off[q] .. +nedit[q]) with ops as pba_align_*_trace returns them and vals[k] =
 * edits[k].val (the b element of a MATCH / INSERT, seq_aligner.h:218,224). */
int pba_cons_elect(pba_ctx *ctx, pba_cons *c, uint32_t n, const int32_t *pos, const uint8_t *fwd, const uint8_t *ops,
                   const char *vals, const uint64_t *ops_off, const int32_t *nedit);
/* The batch form of try_align's align + OVERLAP_MIN gate + elect (ref_seq.h:264-267), everything on the device: pair q
 * aligns a = A[ref_seq] from pairs[q].a_pos (the reference text of these boxes, as the caller uploaded it) against its
 * b, and if it succeeds with matlen_a >= overlap_min the path is voted straight from the traceback walk -- no edit
 * script leaves the GPU.  Both accessors of a pair run in the same direction.  No growth: append / prepend stay the
 * caller's (a round of interior reads).  out[q] as pba_align_batch returns it. */
int pba_cons_vote_pairs(pba_ctx *ctx, pba_cons *c, const pba_seqs *A, uint32_t ref_seq, const pba_seqs *B,
                        const pba_pair *pairs, size_t n, double R, int maxn, int maxm, int overlap_min, pba_result *out);
/* ref_seq::evolve (ref_seq.h:317-349): votes -> next reference; the boxes keep their counts, the new text
 * (new_len characters, up to cap copied) starts at beg and pre = beg, post = end = beg + new_len. */
int pba_cons_evolve(pba_ctx *ctx, pba_cons *c, char *text_out, int cap, int32_t *new_len);
/* the boxes of [pre, post) in order: sel/sup 4 u16 each (A,C,G,T), tot; *n = their number (up to cap copied) */
int pba_cons_dump(pba_ctx *ctx, const pba_cons *c, uint16_t *sel, uint16_t *sup, int32_t *tot, int cap, int32_t *n);
/* the text of [pre, post) as ref_seq::get_accessor sees it */
int pba_cons_text(pba_ctx *ctx, const pba_cons *c, char *out, int cap, int32_t *n);

/* ------------------------------------------------------------------------ */
/* Drivers: the reference's ordered first-success loops, run on the GPU.    */
/* ------------------------------------------------------------------------ */
typedef struct {
    int32_t read;        /* index into the read set */
    int32_t nseq;        /* locator's id: index among reads with len >= min_len, else -1 (locator.cpp:72,91) */
    int32_t found;
    int32_t j;           /* probe offset of the successful candidate, -1 if none */
    int32_t pos;         /* TSV column 2 (locator.cpp:84) */
    int32_t cost;        /* TSV column 3 */
    int32_t seglen;      /* TSV column 4: len - j */
    int32_t matlen_a, matlen_b;
    int32_t n_pairs;     /* candidate pairs the reference loop hands to align for this read */
    int32_t diag_cost;   /* TSV column 5 (locator.cpp:86): get_cost(len - j, len - j), a written cell when the contig remainder is
                          * at least as long as the read remainder; -1 if none */
} pba_loc_row;

typedef struct {
    int64_t n_reads_kept, n_probe_hits, n_pairs, n_located;
    int64_t n_cells;     /* band cells the reference loop would evaluate for those pairs */
} pba_loc_stats;

/* locator.cpp:70-92 with R / trials / min_len as parameters (stock: 0.15 / 50 / 500).
 * ix must be a PBA_INDEX_ALL index of sequence target_seq of `target`. */
int pba_locate(pba_ctx *ctx, const pba_index *ix, const pba_seqs *target, uint32_t target_seq,
               const pba_seqs *reads, double R, int trials, int min_len, int maxn, int maxm, int kernel,
               pba_loc_row *rows, pba_loc_stats *stats);

typedef struct {
    int32_t read, found, j, dir;   /* dir +1 forward / -1 backward (spaced_seed.cpp:426) */
    int32_t ref_pos;               /* hit position (list value) */
    int32_t cost, matlen_a, matlen_b;
    int32_t n_trials;              /* probes that hit the map (DBG _ntrials, spaced_seed.cpp:268-270) */
    int32_t n_pairs;
} pba_ss_row;

/* one locked round of spaced_seed.cpp:420-437 against a PBA_INDEX_HEAD_TAIL index.
 * buggy_seed_at != 0 reproduces dna_seq::seed_at's pos%4==0 behaviour (needs reads built with
 * pba_seqs_from_records so the bytes after each record are the file's). */
int pba_spaced_round(pba_ctx *ctx, const pba_index *ix, const pba_seqs *ref, uint32_t ref_seq,
                     const pba_seqs *reads, double R, int max_trial, int overlap_min, int buggy_seed_at,
                     int kernel, pba_ss_row *rows);

/* spaced_seed's main loop (spaced_seed.cpp:409-452) for a LOCKED reference (-l; the reference never changes, so a
 * round is pba_spaced_round over the reads not found yet): the seed of a round is masks[picks[k] % n_masks] for the
 * k-th draw (picks[] stands in for rand(), spaced_seed.cpp:412) after a round that found something, else the seeds in
 * order; found reads leave the pool; the loop ends after max_round rounds or when every seed failed in a row.
 * rows[r] = the row of the round that found read r (found = 0: its last failed round), found_round[r] = that round
 * (1-based) or 0; log[k] describes round k+1 (up to log_cap), *n_rounds = rounds run. */
typedef struct {
    int32_t round;
    uint32_t mask;
    int32_t n_tried, n_found;
} pba_ss_round_log;
int pba_spaced_multi(pba_ctx *ctx, const pba_seqs *ref, uint32_t ref_seq, const pba_seqs *reads, double ratio, int max_trial,
                     int overlap_min, int buggy_seed_at, int kernel, const uint32_t *masks, int n_masks,
                     const uint32_t *picks, int n_picks, int max_round, pba_ss_row *rows, int32_t *found_round,
                     pba_ss_round_log *log, int log_cap, int *n_rounds);

/* One round of spaced_seed.cpp:420-446 against the UNLOCKED reference c: the reads pool[0..n_pool) in that order, each
 * walked like spaced_seed.cpp:424-437 (first success over j, forward then backward) with ref_seq::try_align voting and
 * growing the reference as it goes (ref_seq.h:259-276) -- so a read sees the text as the reads before it left it.  The
 * seed index is get_seedmap's (ref_seq.h:291-311) over [beg, end) as the round finds it.  maxn / maxm: the size guard of
 * the caller's aligner (t_aligner: 26000, 6000; 0, 0 = none).  rows[read id] is filled for the reads of the pool.  The
 * caller calls pba_cons_evolve afterwards (spaced_seed.cpp:451).  Replaces: the loop body of spaced_seed.cpp:410-446
 * for a reference that is not locked.  Runs as batches on the device (see pba_device.hip); results are those of the
 * serial loop. */
typedef struct {
    int32_t n_found;               /* nmatches, spaced_seed.cpp:434 */
    int32_t n_batches;             /* launches of the round kernel (1 + one per growth that mattered to a later read) */
    int32_t n_grown_fwd, n_grown_bwd;   /* append / prepend calls, ref_seq.h:270-273 */
    uint32_t n_deferred;           /* reads put back behind a growth, summed over the batches */
    uint32_t n_index;              /* entries of the round's seed index */
} pba_cons_round_stats;
int pba_cons_round(pba_ctx *ctx, pba_cons *c, const pba_seqs *reads, const uint32_t *pool, uint32_t n_pool, uint32_t mask,
                   double R, int max_trial, int overlap_min, int buggy_seed_at, int kernel, int maxn, int maxm,
                   pba_ss_row *rows, pba_cons_round_stats *stats);
/* spaced_seed's main loop (spaced_seed.cpp:409-452) WITHOUT -l: pba_cons_round over the reads not found yet, the seed of
 * a round chosen as in pba_spaced_multi, pba_cons_evolve after every round except one that ends the loop (every seed
 * failed in a row, spaced_seed.cpp:450).  rows / found_round / log as in pba_spaced_multi; ref_len_log[k] = length of the
 * reference after round k+1. */
int pba_cons_assemble(pba_ctx *ctx, pba_cons *c, const pba_seqs *reads, double R, int max_trial, int overlap_min,
                      int buggy_seed_at, int kernel, int maxn, int maxm, const uint32_t *masks, int n_masks,
                      const uint32_t *picks, int n_picks, int max_round, pba_ss_row *rows, int32_t *found_round,
                      pba_ss_round_log *log, int32_t *ref_len_log, int log_cap, int *n_rounds);

/* ------------------------------------------------------------------------ */
/* All-vs-all overlap.  Not a loop the reference  */
/* has, but built only from its pieces: every read t in [t_lo, t_hi) takes the */
/* reference role (ref_seq::get_seedmap index of t, ref_seq.h:291-311) and     */
/* every other read q is walked like one read of a locked spaced_seed round    */
/* (spaced_seed.cpp:420-437 with the intended seed_at): first       */
/* success per (t, q); every successful pair is reported.                      */
/* ------------------------------------------------------------------------ */
typedef struct {
    int32_t target, query;         /* read ids; target = the `a` side (ref_seq.h:264) */
    int32_t j, dir, ref_pos;       /* probe offset, +1 forward / -1 backward, hit position in the target */
    int32_t cost, matlen_a, matlen_b;
} pba_overlap;

typedef struct {
    uint64_t n_probe_entries;      /* probe keys indexed (2*max_trial per read, zero keys dropped) */
    uint64_t n_candidates;         /* (target position, probe) matches */
    uint64_t n_pairs;              /* candidate pairs handed to the banded DP (stops at the first success per pair of reads) */
    uint64_t n_overlaps;           /* successful (target, query) pairs */
    uint64_t n_redo;               /* (target, query) runs resumed at the reference band (narrow window not certified) */
    float scan_ms, sort_ms, walk_ms;
    uint32_t wide_first;           /* 1: a sample showed the narrow window rarely certifies, the rest went straight to the reference band */
    float table_ms;                /* build of the probe table this call scanned against (once per table, not per call) */
    uint32_t n_big_targets;        /* targets whose candidates outgrew one LDS sort and were cut into pieces of consecutive queries */
    uint64_t n_prefiltered;        /* bit-vector kernels: candidates that failed the reference's diagonal check within their first 32
                                      rows where the scan found them -- pairs the reference aligned and dropped there: counted in
                                      n_pairs, never written (row-sweep kernel: 0, every candidate is written, sorted and walked) */
    uint32_t cap_fill;             /* 1: the survivors' slices were not sized by a census launch first but given equal room, sized by an earlier range of the table */
    uint32_t cap_overflow;         /* 1: a slice outgrew that room and the range was scanned again with exact slices */
    uint64_t n_listed;             /* candidates written, sorted and walked: the survivors of the scan's 32 rows (row-sweep kernel: all) */
} pba_overlap_stats;

/* Limits of the all-vs-all entry points (explicit PBA_E_TOOLONG beyond them, never a wrapped count):
 *   reads                      < PBA_OVL_MAX_READS       (a candidate packs the query id next to 23 bits of probe and ordinal)
 *   reads * 2 * max_trial      < PBA_OVL_MAX_PROBES      (a probe id is 32 bits)
 *   LISTED candidates of ONE call < PBA_OVL_MAX_CANDIDATES (n_listed; offsets into the candidate array are 32 bits: go through the targets
 *                                                         in ranges [t_lo, t_hi) against one probe table -- BASELINE config 5,
 *                                                         10 M reads, takes ~4 000 targets per call)
 *   max_trial                  in [1, 63], read length <= 65 000 */
#define PBA_OVL_MAX_READS (1u << 24)
#define PBA_OVL_MAX_PROBES (1ull << 32)
#define PBA_OVL_MAX_CANDIDATES 0xFFFFFFF0ull

/* out: caller-allocated, cap entries; *n_out = overlaps found (may exceed cap: then only cap are written).
 * Results are sorted by (target, query).  Targets shard across GPUs through [t_lo, t_hi). */
int pba_overlap_all(pba_ctx *ctx, const pba_seqs *reads, uint32_t t_lo, uint32_t t_hi, uint32_t mask, double R,
                    int max_trial, int overlap_min, int kernel, pba_overlap *out, uint64_t cap, uint64_t *n_out,
                    pba_overlap_stats *stats);
/* Multi-GPU form (the one exchange of SURVEY 8e): a rank emits the probe entries of ITS queries [q_lo, q_hi)
 * into a DEVICE buffer (2*max_trial slots per query are enough), the ranks all-gather those buffers over RCCL
 * (slots holding all-ones are padding), and every rank hands the gathered list to pba_overlap_all_probes,
 * which is pba_overlap_all with the probe table given instead of built. */
int pba_overlap_probes(pba_ctx *ctx, const pba_seqs *reads, uint32_t q_lo, uint32_t q_hi, uint32_t mask, int max_trial,
                       void *d_entries, uint64_t cap, uint64_t *n_out);
int pba_overlap_all_probes(pba_ctx *ctx, const pba_seqs *reads, uint32_t t_lo, uint32_t t_hi, const void *d_probe_entries,
                           uint64_t n_probe_slots, uint32_t mask, double R, int max_trial, int overlap_min, int kernel,
                           pba_overlap *out, uint64_t cap, uint64_t *n_out, pba_overlap_stats *stats);
/* The same in two steps, for read sets that go through the targets in many ranges (or many ranks' worth of them): the
 * probe table -- what hash_table is to one reference (common.h:54), for the probes of every read: bucket offsets by key,
 * the probe ids bucket by bucket, one presence bit per key -- is built ONCE from the (gathered) DEVICE entry list and
 * every pba_overlap_all_table call scans its target range against it. */
typedef struct pba_probe_table pba_probe_table;
int pba_probe_table_create(pba_ctx *ctx, const void *d_probe_entries, uint64_t n_probe_slots, uint32_t mask, int max_trial,
                           pba_probe_table **out);
void pba_probe_table_destroy(pba_probe_table *t);
uint64_t pba_probe_table_entries(const pba_probe_table *t);
int pba_overlap_all_table(pba_ctx *ctx, const pba_seqs *reads, uint32_t t_lo, uint32_t t_hi, const pba_probe_table *tab, double R,
                          int overlap_min, int kernel, pba_overlap *out, uint64_t cap, uint64_t *n_out, pba_overlap_stats *stats);

const char *pba_strerror(int status);

#ifdef __cplusplus
}
#endif
#endif
