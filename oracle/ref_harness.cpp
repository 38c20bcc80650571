/*
 * ref_harness.cpp -- C entry points around the UNMODIFIED reference headers.
 *
 * TEST INFRASTRUCTURE ONLY.  Built by oracle/Makefile into oracle/_ref/libpba_ref.so
 * with -I/root/reference/src: the reference sources are compiled where they lie and
 * are never copied into this repository.  The library exists only in the build
 * container (it cannot be rebuilt on the GPU box, /root/reference is absent there);
 * it is used by tests/golden/make_golden.py to produce the committed golden vectors
 * and by tests/test_oracle_vs_ref.py to pin oracle/pba_oracle.c.
 *
 * What is the reference's and what is this file's:
 *   reference (called, not restated): dna_seq::{encode,decode,seed_at,text2bin,bin2text},
 *     seq_aligner<>::align and its public result fields, ref_seq::{get_seedmap,get_accessor},
 *     hash_table.
 *   reference, consensus half (ref_cons_*): ref_seq's constructor, try_align (unlocked: elect, append, prepend),
 *     evolve, get_accessor; the vote list is read through the private members (`#define private public`
 *     around the include -- the header itself is untouched).
 *   this file: the driver loops of locator.cpp:62-92 and spaced_seed.cpp:262-298,420-437
 *     (those live in main() files and cannot be linked), with R / trials / buffer sizes
 *     as parameters, plus the canonicalisation of SURVEY A.4 / B3 / B4:
 *       - the aligner is instantiated as seq_aligner<40000,12288> so that
 *         2*max_dst+1 <= MAXM for every read <= 20 kb at R <= 0.30 (no row aliasing);
 *       - before each call the diagonal cells mat[i][max_dst], i in (len_b, len_a], are
 *         zeroed, which is what a freshly allocated aligner holds there.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

#include "common.h"
#include "dna_seq.h"
#include "seq_aligner.h"
#define private public          /* the consensus entry points below read ref_seq's vote list (ref_seq.h:359-369) */
#include "ref_seq.h"
#undef private

typedef seq_aligner<40000, 12288> big_aligner;
static big_aligner *g_big = NULL;
static t_aligner *g_stock = NULL;

static big_aligner *big() {
    if (!g_big) g_big = new big_aligner();      /* 3.9 GB virtual, touched lazily */
    return g_big;
}

template <class AL>
static void canonicalise(AL *al, int la, int lb) {
    /* same parameter block as seq_aligner.h:94-102, only to know which cells to clear */
    int len_a, len_b, max_dst;
    if (lb >= la) { len_a = la; max_dst = 1 + (int)(len_a * al->R); len_b = std::min(lb, len_a + max_dst); }
    else { len_b = lb; max_dst = 1 + (int)(len_b * al->R); len_a = std::min(la, len_b + max_dst); }
    for (int i = len_b + 1; i <= len_a; ++i) al->mat[i][max_dst].cost = 0;
}

template <class AL>
static int run_align(AL *al, char *a, int a_fwd, int la, char *b, int b_fwd, int lb,
                     double R, int32_t *out, uint8_t *ops) {
    al->R = R;
    canonicalise(al, la, lb);
    seq_accessor sa(a, a_fwd != 0, la), sb(b, b_fwd != 0, lb);
    int rc = al->align(&sa, &sb);
    out[0] = rc;
    out[4] = al->len_a; out[5] = al->len_b; out[6] = al->max_dst;
    if (rc >= 0) {
        out[1] = al->final_cost(); out[2] = al->matlen_a; out[3] = al->matlen_b; out[7] = al->nedit;
        if (ops) for (int k = 0; k < al->nedit; ++k) ops[k] = (uint8_t)al->edits[k].op;
    } else {
        out[1] = out[2] = out[3] = out[7] = 0;
    }
    return rc;
}

extern "C" {

uint32_t ref_encode(const char *t) { return dna_seq::encode(t); }
void ref_decode(uint32_t code, char *t) { dna_seq::decode(code, t); }
uint32_t ref_seed_at(uint8_t *rec, int pos) { return dna_seq::seed_at(rec, pos); }
uint32_t ref_text2bin(const char *text, uint8_t *rec, uint32_t cap) { return dna_seq::text2bin(text, rec, cap); }
uint32_t ref_bin2text(const uint8_t *rec, char *text, uint32_t cap) { return dna_seq::bin2text(rec, text, cap); }
int ref_c2i(int ch) { char x = (char)ch; return C2I(x); }

/* pattern -> mask the way spaced_seed.cpp:167-180 does it, through dna_seq::encode */
uint32_t ref_mask_from_pattern(const char *pat) {
    char w[17] = "AAAAAAAAAAAAAAAA";
    size_t n = std::min(strlen(pat), (size_t)16);
    for (size_t i = 0; i < n; ++i) w[i] = pat[i] == '1' ? 'T' : 'A';
    return dna_seq::encode(w);
}

/* out: rc, cost, matlen_a, matlen_b, len_a, len_b, max_dst, nedit */
int ref_align(char *a, int a_fwd, int la, char *b, int b_fwd, int lb, double R, int32_t *out, uint8_t *ops) {
    return run_align(big(), a, a_fwd, la, b, b_fwd, lb, R, out, ops);
}

/* seq_aligner::get_cost / get_parent (seq_aligner.h:131,133) on the canonical aligner, after the last ref_align */
void ref_cell(int i, int j, int32_t *out) {
    out[0] = big()->get_cost(i, j);
    out[1] = big()->get_parent(i, j);
}

/* the stock typedef (seq_aligner<26000,6000>), canonicalised the same way */
int ref_align_stock(char *a, int a_fwd, int la, char *b, int b_fwd, int lb, double R, int32_t *out, uint8_t *ops) {
    if (!g_stock) g_stock = new t_aligner();
    return run_align(g_stock, a, a_fwd, la, b, b_fwd, lb, R, out, ops);
}

/* ref_seq::get_seedmap (ref_seq.h:291); entries sorted by key, list order within a key */
long ref_get_seedmap(const char *text, int len, uint32_t mask, uint32_t *keys, int32_t *pos, long cap,
                     uint32_t *retval, uint32_t *nkeys) {
    ref_seq *r = new ref_seq(text, len, true);
    hash_table sm;
    unsigned rv = r->get_seedmap(sm, mask);
    if (retval) *retval = rv;
    if (nkeys) *nkeys = (uint32_t)sm.size();
    std::vector<unsigned> ks;
    for (sm_it it = sm.begin(); it != sm.end(); ++it) ks.push_back(it->first);
    std::sort(ks.begin(), ks.end());
    long n = 0;
    for (size_t i = 0; i < ks.size(); ++i) {
        std::list<int> &l = sm[ks[i]];
        for (std::list<int>::iterator p = l.begin(); p != l.end(); ++p, ++n)
            if (n < cap) { keys[n] = ks[i]; pos[n] = *p; }
    }
    delete r;
    return n;
}

/* index of locator.cpp:62-66 over an own, NUL-padded buffer */
static void locator_index(hash_table &sm, const char *contig, int clen, uint32_t mask) {
    for (int i = 0; i < clen; ++i) {
        int sd = dna_seq::encode(contig + i);
        if (sd & mask) sm[sd & mask].push_back(i);
    }
}

long ref_locator_index(const char *text, int len, uint32_t mask, uint32_t *keys, int32_t *pos, long cap) {
    std::vector<char> c(len + 32, '\0');
    memcpy(&c[0], text, len);
    hash_table sm(1 << 20);
    locator_index(sm, &c[0], len, mask);
    std::vector<unsigned> ks;
    for (sm_it it = sm.begin(); it != sm.end(); ++it) ks.push_back(it->first);
    std::sort(ks.begin(), ks.end());
    long n = 0;
    for (size_t i = 0; i < ks.size(); ++i) {
        std::list<int> &l = sm[ks[i]];
        for (std::list<int>::iterator p = l.begin(); p != l.end(); ++p, ++n)
            if (n < cap) { keys[n] = ks[i]; pos[n] = *p; }
    }
    return n;
}

/* rows: read, nseq, found, j, pos, cost, seglen, matlen_a, matlen_b, n_pairs (10 ints per read)
 * stats: reads kept, probe hits, candidate pairs, located */
int ref_locator(const char *contig_in, int clen, uint32_t mask, double R, int trials, int min_len,
                const char *reads, const uint64_t *offs, int nreads, int32_t *rows, int64_t *stats) {
    std::vector<char> contig(clen + 32, '\0');
    memcpy(&contig[0], contig_in, clen);
    hash_table sm(1 << 23);
    locator_index(sm, &contig[0], clen, mask);
    big_aligner *al = big();
    al->R = R;
    std::vector<char> seq;
    int nseq = 0;
    int64_t kept = 0, hits = 0, pairs = 0, located = 0;
    for (int r = 0; r < nreads; ++r) {
        int len = (int)(offs[r + 1] - offs[r]);
        int32_t *row = rows + 10 * r;
        row[0] = r; row[1] = -1; row[2] = 0; row[3] = -1; row[4] = -1; row[5] = -1;
        row[6] = 0; row[7] = 0; row[8] = 0; row[9] = 0;
        if (len < min_len) continue;                               /* locator.cpp:72 */
        ++kept;
        seq.assign(len + 32, '\0');
        memcpy(&seq[0], reads + offs[r], len);
        bool found = false;
        for (int j = 0; j < trials && !found; ++j) {               /* locator.cpp:74 */
            int seed = dna_seq::encode(&seq[0] + j) & mask;
            sm_it sit = sm.find(seed);
            if (sit == sm.end()) continue;
            ++hits;
            for (std::list<int>::iterator it = sit->second.begin(); it != sit->second.end(); ++it) {
                ++pairs; ++row[9];
                int32_t out[8];
                int rc = run_align(al, &seq[0] + j, 1, len - j, &contig[0] + *it, 1, clen - *it, R, out, NULL);
                if (rc > 0) {                                      /* locator.cpp:82 */
                    found = true; ++located;
                    row[2] = 1; row[3] = j; row[4] = *it; row[5] = out[1]; row[6] = len - j;
                    row[7] = out[2]; row[8] = out[3];
                    break;
                }
            }
        }
        row[1] = nseq++;                                           /* locator.cpp:91 */
    }
    if (stats) { stats[0] = kept; stats[1] = hits; stats[2] = pairs; stats[3] = located; }
    return 0;
}

/* one locked round of spaced_seed.cpp:420-437 over binary records.
 * rows: read, found, j, dir, ref_pos, cost, matlen_a, matlen_b, n_trials, n_pairs (10 ints)
 * `records` must be followed by >= 32 KB of readable bytes (seed_at's pos%4==0 path reads far
 * past the record). */
static bool ss_try(ref_seq *pref, hash_table &sm, big_aligner *al, uint8_t *rec, char *txt, int seg_len,
                   long pos, int dir, uint32_t mask, int overlap_min, int32_t *row) {
    sm_it sit = sm.find(dna_seq::seed_at(rec, (int)pos) & mask);   /* spaced_seed.cpp:265 */
    if (sit == sm.end()) return false;
    ++row[8];
    bool forward = dir == 1;
    int s_offset = forward ? (int)pos : (int)pos + 16 - 1;
    int s_len = forward ? seg_len - s_offset : s_offset + 1;
    if (s_len < overlap_min) return false;
    for (std::list<int>::iterator it = sit->second.begin(); it != sit->second.end(); ++it) {
        int r_offset = forward ? (*it) : (*it) + 16 - 1;
        seq_accessor ac_ref = pref->get_accessor(r_offset, forward);   /* ref_seq.h:261 */
        ++row[9];
        int32_t out[8];
        /* ref_seq.h:264: the reference is `a`, the read is `b` */
        int rc = run_align(al, ac_ref.pt(0), forward, ac_ref.length(), txt + s_offset, forward, s_len,
                           al->R, out, NULL);
        if (rc < 0) continue;
        if (out[2] < overlap_min) continue;                        /* ref_seq.h:265 */
        row[1] = 1; row[3] = dir; row[4] = *it; row[5] = out[1]; row[6] = out[2]; row[7] = out[3];
        return true;
    }
    return false;
}

int ref_spaced_round(const char *ref, int ref_len, uint32_t mask, double R, int max_trial, int overlap_min,
                     uint8_t *records, const uint64_t *rec_offs, int nreads, int32_t *rows) {
    ref_seq *pref = new ref_seq(ref, ref_len, true);
    hash_table sm(1 << 20);
    pref->get_seedmap(sm, mask);
    big_aligner *al = big();
    al->R = R;
    std::vector<char> txt(1 << 20);
    for (int r = 0; r < nreads; ++r) {
        uint8_t *rec = records + rec_offs[r];
        int32_t *row = rows + 10 * r;
        memset(row, 0, 10 * sizeof(int32_t));
        row[0] = r; row[2] = -1;
        int slen = (int)dna_seq::bin2text(rec, &txt[0], 1 << 20);
        for (int j = 0; j < max_trial; ++j) {
            if (ss_try(pref, sm, al, rec, &txt[0], slen, j, 1, mask, overlap_min, row) ||
                ss_try(pref, sm, al, rec, &txt[0], slen, (long)slen - j - 16, -1, mask, overlap_min, row)) {
                row[2] = j;
                break;
            }
        }
    }
    delete pref;
    return 0;
}


/* ---- consensus voting and reference growth (ref_seq.h:25-188, 259-276, 317-362) -------------------------------
 * One ref_seq at a time, driven from Python: new -> try* -> dump / evolve -> ... -> free.
 * try_align needs the stock t_aligner (seq_aligner<26000,6000>); OVERLAP_MIN is the reference's 64. */
static ref_seq *g_cons = NULL;

int ref_cons_new(const char *text, int len, int weight) {
    delete g_cons;
    g_cons = new ref_seq(text, len, false, weight);
    return 0;
}
void ref_cons_free(void) { delete g_cons; g_cons = NULL; }

/* out: ok, matlen_b, cost, matlen_a, nedit (all 0 unless ok), pre-beg, post-beg after the call */
int ref_cons_try(int pos, char *seg_origin, int seg_len, int fwd, double R, int32_t *out) {
    if (!g_stock) g_stock = new t_aligner();
    g_stock->R = R;
    seq_accessor ac(seg_origin, fwd != 0, seg_len);
    seq_accessor ar = g_cons->get_accessor(pos, fwd != 0);
    canonicalise(g_stock, ar.length(), seg_len);
    bool ok = g_cons->try_align(g_stock, pos, &ac);
    out[0] = ok; out[1] = ok ? g_stock->matlen_b : 0; out[2] = ok ? g_stock->final_cost() : 0; out[3] = ok ? g_stock->matlen_a : 0;
    out[4] = ok ? g_stock->nedit : 0; out[5] = g_cons->pre - g_cons->beg; out[6] = g_cons->post - g_cons->beg;
    return ok;
}

void ref_cons_evolve(void) { g_cons->evolve(); }

/* one UNLOCKED round of spaced_seed.cpp:420-446 on g_cons: get_seedmap, then every read of the pool in order through
 * try_align (spaced_seed.cpp:262-298) with the reference's own ref_seq::try_align (votes, growth) on the stock
 * t_aligner, OVERLAP_MIN = 64 and the reference's seed_at.  rows[k] <-> pool[k], 10 ints as ref_spaced_round.
 * Returns nmatches.  `records` must be followed by >= 32 KB of readable bytes. */
static bool cr_try(hash_table &sm, uint8_t *rec, char *txt, int seg_len, long pos, int dir, uint32_t mask, int32_t *row) {
    sm_it sit = sm.find(dna_seq::seed_at(rec, (int)pos) & mask);   /* spaced_seed.cpp:265 */
    if (sit == sm.end()) return false;
    ++row[8];
    bool forward = dir == 1;
    int s_offset = forward ? (int)pos : (int)pos + 16 - 1;
    int s_len = forward ? seg_len - s_offset : s_offset + 1;
    seq_accessor ac_seg(txt + s_offset, forward, s_len);
    if (s_len < OVERLAP_MIN) return false;
    for (std::list<int>::iterator it = sit->second.begin(); it != sit->second.end(); ++it) {
        int r_offset = forward ? (*it) : (*it) + 16 - 1;
        seq_accessor ac_ref = g_cons->get_accessor(r_offset, forward);
        canonicalise(g_stock, ac_ref.length(), s_len);
        ++row[9];
        ac_seg.reset(0);
        if (g_cons->try_align(g_stock, r_offset, &ac_seg)) {       /* spaced_seed.cpp:286 */
            row[1] = 1; row[3] = dir; row[4] = *it; row[5] = g_stock->final_cost(); row[6] = g_stock->matlen_a; row[7] = g_stock->matlen_b;
            return true;
        }
    }
    return false;
}

int ref_cons_round(uint32_t mask, double R, int max_trial, uint8_t *records, const uint64_t *rec_offs, const int32_t *pool,
                   int npool, int32_t *rows) {
    if (!g_stock) g_stock = new t_aligner();
    g_stock->R = R;
    hash_table sm(1 << 20);
    g_cons->get_seedmap(sm, mask);                                  /* spaced_seed.cpp:415 */
    std::vector<char> txt(1 << 20);
    int nmatches = 0;
    for (int k = 0; k < npool; ++k) {
        uint8_t *rec = records + rec_offs[pool[k]];
        int32_t *row = rows + 10 * k;
        memset(row, 0, 10 * sizeof(int32_t));
        row[0] = pool[k]; row[2] = -1;
        int slen = (int)dna_seq::bin2text(rec, &txt[0], 1 << 20);
        for (int j = 0; j < max_trial; ++j) {
            if (cr_try(sm, rec, &txt[0], slen, j, 1, mask, row) ||
                cr_try(sm, rec, &txt[0], slen, (long)slen - j - 16, -1, mask, row)) {
                row[2] = j; ++nmatches;
                break;
            }
        }
    }
    return nmatches;
}


/* the vote list in list order: sel[4k..], sup[4k..], tot[k]; returns the number of boxes */
int ref_cons_dump(uint16_t *sel, uint16_t *sup, int32_t *tot, int cap, int32_t *extent) {
    int n = 0;
    for (std::list<vote_box>::iterator it = g_cons->consensus.begin(); it != g_cons->consensus.end(); ++it, ++n) {
        if (n >= cap) continue;
        for (int c = 0; c < 4; ++c) { sel[4 * n + c] = it->selection.acgt[c]; sup[4 * n + c] = it->suppliment.acgt[c]; }
        tot[n] = it->total;
    }
    if (extent) {
        extent[0] = g_cons->pre - g_cons->beg; extent[1] = g_cons->post - g_cons->beg; extent[2] = g_cons->end - g_cons->beg;
    }
    return n;
}

/* text [pre, post) as get_accessor sees it; returns its length */
int ref_cons_text(char *out, int cap) {
    int n = g_cons->post - g_cons->pre;
    memcpy(out, g_cons->txt_buf + g_cons->pre, n < cap ? n : cap);
    return n;
}

}  /* extern "C" */
