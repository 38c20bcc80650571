/*
 * pba_oracle.h -- CPU restatement of the reference seed-and-extend path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under pacbioassembly_amd/ (the product)
 * may include, link or call this.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and there only as the checker / the CPU
 * baseline being timed, never as the thing shipped.
 *
 * Parity status: PINNED.  Every function here is checked (tests/test_oracle_*.py)
 * against (1) the reference's own known-answer vectors (test/dna_test.cpp:23-29,
 * test/aligner_test.cpp:44-117, test/ref_test.cpp:119-128, test/real_align.txt,
 * seeds.txt) and (2) golden vectors produced by the reference itself, compiled
 * from /root/reference/src by oracle/Makefile into oracle/_ref/ (see
 * tests/golden/make_golden.py).
 *
 * All file:line citations are into /root/reference/.
 */
#ifndef PBA_ORACLE_H
#define PBA_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- codec (src/dna_seq.h:21-176) ------------------------------------- */
int      orc_c2i(int ch);                                   /* dna_seq.h:21  */
uint32_t orc_encode(const char *text16);                    /* dna_seq.h:86  */
void     orc_decode(uint32_t code, char *text16);           /* dna_seq.h:101 */
/* encode with an explicit end: bytes at/after `avail` count as NUL (code 3),
 * which is what locator.cpp:62-63 reads past the contig in its zeroed global */
uint32_t orc_encode_padded(const char *text, long avail);
size_t   orc_text2bin(const char *text, size_t tlen, uint8_t *rec, size_t cap); /* dna_seq.h:113 */
size_t   orc_bin2text(const uint8_t *rec, char *text, size_t cap);              /* dna_seq.h:133 */
uint32_t orc_seed_at(const uint8_t *rec, int pos);          /* dna_seq.h:62, bug-compatible */
uint32_t orc_seed_at_fixed(const uint8_t *rec, int pos);    /* what seed_at meant: == encode(text+pos) */
uint32_t orc_mask_from_pattern(const char *pattern);        /* spaced_seed.cpp:167-180 */

/* ---- banded DP (src/seq_aligner.h:92-233), canonical semantics SURVEY A.4 */
typedef struct {
    int32_t rc;        /* -1 or matlen_b (seq_aligner.h:106,111,114,124) */
    int32_t cost;      /* final_cost() (seq_aligner.h:130); valid when rc >= 0 */
    int32_t matlen_a, matlen_b;
    int32_t len_a, len_b, max_dst;
    int32_t nedit;     /* traceback length (seq_aligner.h:115-116); valid when rc >= 0 */
    int32_t fail_row;  /* row of early failure (seq_aligner.h:185), 0 if none */
    int64_t cells;     /* DP cells evaluated (work accounting only) */
} orc_result;

typedef struct orc_aligner orc_aligner;
/* maxn/maxm: the template limits of seq_aligner<MAXN,MAXM> (size guard at
 * seq_aligner.h:104); maxn <= 0 means no guard.  The matrix is allocated so
 * that every band row has its own 2*max_dst+1 cells (no aliasing)
 * at the reference's 8 bytes per cell. */
orc_aligner *orc_aligner_new(int maxn, int maxm);
void         orc_aligner_free(orc_aligner *al);
/* a/b are accessor origins (dna_seq.h:191): element k is p[k] when fwd, p[-k] otherwise.
 * ops (nullable, cap >= la+lb) receives the edit ops 1=MATCH 2=INSERT 3=DELETE. */
int orc_align(orc_aligner *al, const char *a, int a_fwd, int la,
              const char *b, int b_fwd, int lb, double R,
              orc_result *res, uint8_t *ops);

/* get_cost / get_parent (seq_aligner.h:131,133) of the most recent orc_align on `al`: 0 and the values when (i, j) is a
 * cell that call wrote, -1 otherwise (the reference returns stale memory there) */
int orc_aligner_cell(const orc_aligner *al, int i, int j, int *cost, int *parent);

/* Map and touch the DP matrices of n pooled aligners (the drivers below draw their per-thread aligners
 * from that pool) for up to len_a rows at ratio R; orc_pool_release frees them. */
int  orc_prefault(int n, int len_a, double R);
void orc_pool_release(void);

/* ---- seed index (common.h:54; locator.cpp:62-66; ref_seq.h:291-311) ---- */
typedef struct orc_seedmap orc_seedmap;
orc_seedmap *orc_seedmap_new(size_t nbuckets);
void         orc_seedmap_clear(orc_seedmap *sm);
void         orc_seedmap_free(orc_seedmap *sm);
size_t       orc_seedmap_size(const orc_seedmap *sm);       /* number of keys */
size_t       orc_seedmap_entries(const orc_seedmap *sm);    /* number of positions */
/* locator.cpp:62-66: every position [0,len), tail windows padded with code 3 */
size_t orc_index_all(orc_seedmap *sm, const char *text, int len, uint32_t mask);
/* ref_seq.h:291-311: head ascending then tail descending; returns its return value */
unsigned orc_index_head_tail(orc_seedmap *sm, const char *text, int len, uint32_t mask);
/* hits of `key` in insertion order; returns count, copies up to cap */
int orc_seedmap_find(const orc_seedmap *sm, uint32_t key, int32_t *pos, int cap);
/* all entries sorted by key, insertion order within a key; returns count */
size_t orc_seedmap_dump(const orc_seedmap *sm, uint32_t *keys, int32_t *pos, size_t cap);

/* ---- locator driver (locator.cpp:70-92), R / trials / min_len as parameters */
typedef struct {
    int32_t read;      /* index into the input reads */
    int32_t nseq;      /* locator's running id: index among reads with len >= min_len */
    int32_t found;     /* 1 when some candidate aligned */
    int32_t j;         /* probe offset of the successful candidate */
    int32_t pos;       /* contig position (TSV column 2) */
    int32_t cost;      /* TSV column 3 */
    int32_t seglen;    /* len - j, TSV column 4 */
    int32_t matlen_a, matlen_b;
    int32_t n_pairs;   /* candidate pairs handed to align for this read */
} orc_loc_row;

typedef struct {
    int64_t n_reads_kept, n_probe_hits, n_pairs, n_located, n_cells;
} orc_loc_stats;

int orc_locator_run(const char *contig, int clen, uint32_t mask, double R,
                    int trials, int min_len, int maxn, int maxm,
                    const char *reads, const uint64_t *offs, int nreads,
                    int nthreads, orc_loc_row *rows, orc_loc_stats *stats);

/* ---- spaced_seed locked round (spaced_seed.cpp:420-439, ref_seq.h:259-265) */
typedef struct {
    int32_t read, found, j, dir;   /* dir +1 forward, -1 backward */
    int32_t ref_pos;               /* hit position in the reference (list value) */
    int32_t cost, matlen_a, matlen_b;
    int32_t n_trials;              /* seed probes that hit the map (the DBG _ntrials counter) */
    int32_t n_pairs;
} orc_ss_row;

int orc_spaced_round(const char *ref, int ref_len, uint32_t mask, double R,
                     int max_trial, int overlap_min, int buggy_seed_at,
                     const uint8_t *records, const uint64_t *rec_offs, int nreads,
                     int nthreads, orc_ss_row *rows);

/* binary read file walk (spaced_seed.cpp:330-342): offsets of kept records */
size_t orc_open_binary(const uint8_t *buf, size_t len, uint32_t min_excl, uint32_t max_excl,
                       uint64_t *offs, size_t cap, size_t *n_total);

/* ---- consensus voting and reference growth (ref_seq.h:25-188, 207-242, 259-276, 317-362) ------------------
 * The unlocked half of ref_seq: one vote box per reference position (4 selection + 4 suppliment u16 counters and
 * `total`), elect() applies an edit script to the boxes, evolve() turns the votes into the next reference.
 * max_len plays MAX_SEQ_LEN (common.h:31): the text buffer holds 3*max_len chars, the origin sits at max_len. */
typedef struct orc_cons orc_cons;
orc_cons *orc_cons_new(const char *text, int len, int weight, int max_len);          /* ref_seq.h:218-225 */
void      orc_cons_free(orc_cons *c);
void      orc_cons_append(orc_cons *c, const char *seg, int len);                      /* ref_seq.h:227-233 */
void      orc_cons_prepend(orc_cons *c, const char *seg, int len);                     /* ref_seq.h:235-242 */
/* ref_seq.h:352-362 + apply_edits :25-41.  vals[k] = edits[k].val (the b element of a MATCH / INSERT) */
void      orc_cons_elect(orc_cons *c, int pos, int fwd, const uint8_t *ops, const char *vals, int nedit);
/* ref_seq::try_align, unlocked (ref_seq.h:259-276).  out: ok, matlen_b, cost, matlen_a, nedit, pre-beg, post-beg */
int       orc_cons_try(orc_cons *c, orc_aligner *al, int pos, const char *seg_origin, int seg_len, int fwd,
                       double R, int overlap_min, int32_t *out);
void      orc_cons_evolve(orc_cons *c);                                                /* ref_seq.h:317-349 */
/* one unlocked round of spaced_seed.cpp:420-446 (serial, pool order; try_align votes and grows); rows[k] <-> pool[k];
 * returns nmatches.  The caller evolves (spaced_seed.cpp:451). */
int       orc_cons_round(orc_cons *c, orc_aligner *al, uint32_t mask, double R, int max_trial, int overlap_min,
                         int buggy_seed_at, const uint8_t *records, const uint64_t *rec_offs, const int32_t *pool,
                         int npool, orc_ss_row *rows);
/* vote list in list order; extent = {pre-beg, post-beg, end-beg}; returns the number of boxes */
int       orc_cons_dump(const orc_cons *c, uint16_t *sel, uint16_t *sup, int32_t *tot, int cap, int32_t *extent);
int       orc_cons_text(const orc_cons *c, char *out, int cap);                        /* text [pre, post) */

#ifdef __cplusplus
}
#endif
#endif
