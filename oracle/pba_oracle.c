/*
 * pba_oracle.c -- CPU restatement of the reference seed-and-extend path.
 *
 * TEST INFRASTRUCTURE ONLY (see pba_oracle.h).  Plain C, scalar, and on purpose
 * "faithful": node-based hash map with per-key position lists, full-band row
 * sweep storing 8 bytes per DP cell, first-success driver loops -- so that the
 * time it takes is an honest stand-in for the reference CPU path when bench.py
 * times it next to the GPU.  Parity status: PINNED (header comment).
 *
 * All file:line citations are into /root/reference/.
 */
#define _GNU_SOURCE
#include "pba_oracle.h"

#include <pthread.h>
#include <stdio.h>
#include <sys/mman.h>
#include <stdlib.h>
#include <string.h>

/* ======================================================================== */
/* codec                                                                    */
/* ======================================================================== */

/* dna_seq.h:21 -- A,C,G map to 0,1,2; every other byte (T, N, lowercase, NUL, '\n') to 3 */
int orc_c2i(int ch)
{
    switch (ch) {
    case 'A': return 0;
    case 'C': return 1;
    case 'G': return 2;
    default:  return 3;
    }
}

static const char ORC_I2C[4] = { 'A', 'C', 'G', 'T' };   /* dna_seq.h:30 */

/* dna_seq.h:147-159 -- first base in bits 7:6; a short tail is zero-padded low */
static uint8_t pack4(const char *p, long n)
{
    uint8_t b = 0;
    if (n > 4) n = 4;
    for (long k = 0; k < n; ++k)
        b |= (uint8_t)(orc_c2i((unsigned char)p[k]) << (6 - 2 * k));
    return b;
}

/* dna_seq.h:86-96 -- byte k of the little-endian word holds bases 4k..4k+3 */
uint32_t orc_encode(const char *t)
{
    return (uint32_t)pack4(t, 4) | (uint32_t)pack4(t + 4, 4) << 8 |
           (uint32_t)pack4(t + 8, 4) << 16 | (uint32_t)pack4(t + 12, 4) << 24;
}

uint32_t orc_encode_padded(const char *t, long avail)
{
    char w[16];
    for (int k = 0; k < 16; ++k)
        w[k] = (k < avail) ? t[k] : '\0';     /* NUL -> code 3 (locator.cpp:26,62-63) */
    return orc_encode(w);
}

/* dna_seq.h:101-107 */
void orc_decode(uint32_t code, char *t)
{
    for (int k = 0; k < 4; ++k) {
        uint8_t b = (uint8_t)(code >> (8 * k));
        for (int q = 0; q < 4; ++q)
            t[4 * k + q] = ORC_I2C[(b >> (6 - 2 * q)) & 3];
    }
}

/* dna_seq.h:113-127 -- record = u32 length in bases + ceil(len/4) packed bytes */
size_t orc_text2bin(const char *text, size_t tlen, uint8_t *rec, size_t cap)
{
    size_t blen = 4 + (tlen + 3) / 4;
    if (cap < blen) return 0;
    uint32_t l32 = (uint32_t)tlen;
    memcpy(rec, &l32, 4);
    uint8_t *pb = rec + 4;
    for (size_t i = 0; i < tlen; i += 4)
        *pb++ = pack4(text + i, (long)(tlen - i));
    return blen;
}

/* dna_seq.h:133-145 -- writes the NUL as well */
size_t orc_bin2text(const uint8_t *rec, char *text, size_t cap)
{
    uint32_t tlen;
    memcpy(&tlen, rec, 4);
    if (cap <= tlen) return 0;
    for (uint32_t i = 0; i < tlen; ++i)
        text[i] = ORC_I2C[(rec[4 + (i >> 2)] >> (6 - 2 * (i & 3))) & 3];
    text[tlen] = '\0';
    return tlen;
}

/* dna_seq.h:62-76.  For pos%4==0 the reference adds `pos` as a BYTE offset, so it
 * returns the window of base 4*pos (correct only at pos 0): kept as is. */
uint32_t orc_seed_at(const uint8_t *rec, int pos)
{
    const uint8_t *p = rec + 4;
    uint32_t w;
    if ((pos & 3) == 0) {
        memcpy(&w, p + pos, 4);
        return w;
    }
    p += pos >> 2;
    unsigned ls = (unsigned)(pos & 3) << 1, rs = 8 - ls;
    uint8_t s[4];
    for (int k = 0; k < 4; ++k)
        s[k] = (uint8_t)((p[k] << ls) | (p[k + 1] >> rs));
    memcpy(&w, s, 4);
    return w;
}

uint32_t orc_seed_at_fixed(const uint8_t *rec, int pos)
{
    const uint8_t *p = rec + 4 + (pos >> 2);
    unsigned ls = (unsigned)(pos & 3) << 1;
    uint8_t s[4];
    for (int k = 0; k < 4; ++k)
        s[k] = ls ? (uint8_t)((p[k] << ls) | (p[k + 1] >> (8 - ls))) : p[k];
    uint32_t w;
    memcpy(&w, s, 4);
    return w;
}

/* spaced_seed.cpp:167-180 / locator.cpp:51-54: '1' -> T (both bits set), else A; pad with A */
uint32_t orc_mask_from_pattern(const char *pat)
{
    char w[16];
    size_t n = strlen(pat);
    if (n > 16) n = 16;
    for (size_t k = 0; k < 16; ++k)
        w[k] = (k < n && pat[k] == '1') ? 'T' : 'A';
    return orc_encode(w);
}

/* ======================================================================== */
/* banded DP                                                                */
/* ======================================================================== */

typedef struct { int cost; int parent; } orc_cell;     /* seq_aligner.h:60-63 */
enum { OP_MATCH = 1, OP_INSERT = 2, OP_DELETE = 3 };   /* seq_aligner.h:32-36 */

struct orc_aligner {
    int maxn, maxm;
    orc_cell *mat;
    size_t cap;       /* cells allocated */
    int last_len_a, last_len_b, last_max_dst, last_rows;   /* the band of the most recent orc_align (orc_aligner_cell) */
};

/* Aligners are pooled across driver calls: like the reference's one static aligner per process
 * (spaced_seed.cpp:215, locator.cpp:68) a worker's DP matrix is mapped and faulted in once and then
 * reused, so a timed run does not pay first-touch page faults (which cost ~25 us each in sandboxed
 * containers, i.e. seconds per GB) for every call. */
#define ORC_POOL_MAX 256
static orc_aligner *g_pool[ORC_POOL_MAX];
static int g_pool_n = 0;
static pthread_mutex_t g_pool_mu = PTHREAD_MUTEX_INITIALIZER;

static orc_aligner *pool_get(int maxn, int maxm)
{
    orc_aligner *al = NULL;
    pthread_mutex_lock(&g_pool_mu);
    if (g_pool_n > 0) al = g_pool[--g_pool_n];
    pthread_mutex_unlock(&g_pool_mu);
    if (!al) al = orc_aligner_new(0, 0);
    if (al) { al->maxn = maxn; al->maxm = maxm; }
    return al;
}

static void pool_put(orc_aligner *al)
{
    if (!al) return;
    pthread_mutex_lock(&g_pool_mu);
    if (g_pool_n < ORC_POOL_MAX) { g_pool[g_pool_n++] = al; al = NULL; }
    pthread_mutex_unlock(&g_pool_mu);
    if (al) orc_aligner_free(al);
}

static int aligner_reserve(orc_aligner *al, size_t need);

/* Map and touch the DP matrices of `n` pooled aligners for alignments of up to len_a rows at ratio R. */
int orc_prefault(int n, int len_a, double R)
{
    const size_t W = 2 * (size_t)(1 + (int)(len_a * R)) + 1;
    orc_aligner *tmp[ORC_POOL_MAX];
    if (n > ORC_POOL_MAX) n = ORC_POOL_MAX;
    for (int i = 0; i < n; ++i) {
        tmp[i] = pool_get(0, 0);
        if (!tmp[i] || aligner_reserve(tmp[i], ((size_t)len_a + 1) * W) != 0) return -1;
        memset(tmp[i]->mat, 0, tmp[i]->cap * sizeof(orc_cell));
    }
    for (int i = 0; i < n; ++i) pool_put(tmp[i]);
    return 0;
}

void orc_pool_release(void)
{
    pthread_mutex_lock(&g_pool_mu);
    while (g_pool_n > 0) orc_aligner_free(g_pool[--g_pool_n]);
    pthread_mutex_unlock(&g_pool_mu);
}

orc_aligner *orc_aligner_new(int maxn, int maxm)
{
    orc_aligner *al = (orc_aligner *)calloc(1, sizeof *al);
    if (al) { al->maxn = maxn; al->maxm = maxm; }
    return al;
}

void orc_aligner_free(orc_aligner *al)
{
    if (!al) return;
    free(al->mat);
    free(al);
}

/* Grow-only with headroom, like the reference's static mat[MAXN][MAXM] (seq_aligner.h:81) that is mapped
 * once per process and reused.  2 MB alignment + MADV_HUGEPAGE so a host with THP in "madvise" mode
 * behaves like one with THP "always". */
static int aligner_reserve(orc_aligner *al, size_t need)
{
    if (need <= al->cap) return 0;
    const size_t want = (need + need / 8) * sizeof(orc_cell);
    const size_t bytes = (want + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
    void *p = NULL;
    free(al->mat);
    al->mat = NULL; al->cap = 0;
    if (posix_memalign(&p, (size_t)2 << 20, bytes) != 0) return -1;
#ifdef MADV_HUGEPAGE
    madvise(p, bytes, MADV_HUGEPAGE);
#endif
    al->mat = (orc_cell *)p;
    al->cap = bytes / sizeof(orc_cell);
    return 0;
}

static inline char acc_at(const char *p, int fwd, int k) { return fwd ? p[k] : p[-k]; }

/* seq_aligner::get_cost / get_parent (seq_aligner.h:131,133) of the most recent orc_align on this aligner: 0 and the
 * cell's cost / parent when (i, j) is a cell that call wrote (rows 0 .. the last row swept, inside the band and the
 * matrix: init_cell's borders included), else -1 -- the reference would hand back whatever an earlier call left there. */
int orc_aligner_cell(const orc_aligner *al, int i, int j, int *cost, int *parent)
{
    const int md = al->last_max_dst;
    if (al->last_rows < 0 || i < 0 || j < 0 || i > al->last_rows || j > al->last_len_b || j - i > md || i - j > md) return -1;
    if (i == 0 && j > md) return -1;
    const orc_cell *c = &al->mat[(size_t)i * (2 * (size_t)md + 1) + (size_t)(j - i + md)];
    if (cost) *cost = c->cost;
    if (parent) *parent = c->parent;
    return 0;
}

int orc_align(orc_aligner *al, const char *a, int a_fwd, int la,
              const char *b, int b_fwd, int lb, double R,
              orc_result *res, uint8_t *ops)
{
    int len_a, len_b, max_dst;
    memset(res, 0, sizeof *res);
    res->rc = -1;

    /* seq_aligner.h:94-102 */
    if (lb >= la) {
        len_a = la;
        max_dst = 1 + (int)(len_a * R);
        len_b = lb < len_a + max_dst ? lb : len_a + max_dst;
    } else {
        len_b = lb;
        max_dst = 1 + (int)(len_b * R);
        len_a = la < len_b + max_dst ? la : len_b + max_dst;
    }
    res->len_a = len_a; res->len_b = len_b; res->max_dst = max_dst;
    al->last_rows = -1;

    /* seq_aligner.h:104-107 */
    if (al->maxn > 0 && (len_a >= al->maxn + al->maxm || max_dst >= al->maxm))
        return -1;

    const size_t W = 2 * (size_t)max_dst + 1;           /* own cells for every band row */
    const size_t need = ((size_t)len_a + 1) * W;
    if (aligner_reserve(al, need) != 0) return -1;
    orc_cell *mat = al->mat;
    al->last_len_a = len_a; al->last_len_b = len_b; al->last_max_dst = max_dst; al->last_rows = 0;
#define CELL(i, j) mat[(size_t)(i) * W + (size_t)((j) - (i) + max_dst)]

    /* init_cell, seq_aligner.h:139-150 */
    for (int i = 1; i <= max_dst && i <= len_a; ++i) { CELL(i, 0).cost = i; CELL(i, 0).parent = OP_DELETE; }
    for (int j = 1; j <= max_dst; ++j) { CELL(0, j).cost = j; CELL(0, j).parent = OP_INSERT; }
    CELL(0, 0).cost = 0; CELL(0, 0).parent = 0;

    /* search, seq_aligner.h:151-190 */
    int64_t cells = 0;
    for (int i = 1; i <= len_a; ++i) {
        const char c = acc_at(a, a_fwd, i - 1);
        const int beg = i - max_dst > 1 ? i - max_dst : 1;
        const int end = i + max_dst < len_b ? i + max_dst : len_b;
        for (int j = beg; j <= end; ++j) {
            const char d = acc_at(b, b_fwd, j - 1);
            int t;
            int cost = CELL(i - 1, j - 1).cost + (c != d);
            int src = OP_MATCH;
            if (i - j < max_dst && (t = CELL(i, j - 1).cost + 1) < cost) { cost = t; src = OP_INSERT; }
            if (j - i < max_dst && (t = CELL(i - 1, j).cost + 1) < cost) { cost = t; src = OP_DELETE; }
            CELL(i, j).cost = cost;
            CELL(i, j).parent = src;
        }
        cells += end >= beg ? end - beg + 1 : 0;
        /* early failure, seq_aligner.h:185; the i<=len_b guard is the canonical
         * reading of an unwritten diagonal cell */
        al->last_rows = i;
        if (i > 10 && i <= len_b && (double)CELL(i, i).cost > i * R) {
            res->fail_row = i;
            res->cells = cells;
            return -1;
        }
    }
    res->cells = cells;

    /* goal_cell, seq_aligner.h:191-213 */
    int matlen_a, matlen_b;
    if (len_a > len_b) {
        matlen_a = matlen_b = len_b;
        int best = CELL(len_b, len_b).cost;
        for (int i = len_b + 1; i <= len_a; ++i)
            if (CELL(i, len_b).cost < best) { best = CELL(i, len_b).cost; matlen_a = i; }
    } else {
        matlen_a = matlen_b = len_a;
        int best = CELL(len_a, len_a).cost;
        for (int j = len_a + 1; j <= len_b; ++j)
            if (CELL(len_a, j).cost < best) { best = CELL(len_a, j).cost; matlen_b = j; }
    }
    res->matlen_a = matlen_a; res->matlen_b = matlen_b;
    res->cost = CELL(matlen_a, matlen_b).cost;

    /* acceptance, seq_aligner.h:114 */
    if (matlen_b < len_b * (1 - R)) return -1;

    /* find_path, seq_aligner.h:214-233, walked iteratively from the goal */
    int n = 0, i = matlen_a, j = matlen_b;
    for (;;) {
        int p = CELL(i, j).parent;
        if (p == OP_MATCH) { --i; --j; }
        else if (p == OP_INSERT) { --j; }
        else if (p == OP_DELETE) { --i; }
        else break;
        if (ops) ops[n] = (uint8_t)p;
        ++n;
    }
    if (ops)
        for (int x = 0, y = n - 1; x < y; ++x, --y) { uint8_t t = ops[x]; ops[x] = ops[y]; ops[y] = t; }
    res->nedit = n;
#undef CELL
    res->rc = matlen_b;
    return matlen_b;
}

/* ======================================================================== */
/* seed index: chained hash map, key -> list of positions in insertion order */
/* (common.h:54: hash_map<unsigned, list<int>>, identity hash)               */
/* ======================================================================== */

typedef struct orc_pnode { int32_t pos; struct orc_pnode *next; } orc_pnode;
typedef struct orc_knode {
    uint32_t key; int32_t n;
    orc_pnode *head, *tail;
    struct orc_knode *next;
} orc_knode;

struct orc_seedmap {
    size_t nb, nkeys, nentries;
    orc_knode **b;
};

/* __gnu_cxx::hash_map rounds the bucket hint up to a prime (its identity hash is then reduced modulo that
 * prime, which spreads masked keys whose low bits are mostly zero); same here. */
static size_t next_prime(size_t n)
{
    if (n < 17) n = 17;
    for (n |= 1;; n += 2) {
        int ok = 1;
        for (size_t d = 3; d * d <= n; d += 2)
            if (n % d == 0) { ok = 0; break; }
        if (ok) return n;
    }
}

orc_seedmap *orc_seedmap_new(size_t nb)
{
    orc_seedmap *sm = (orc_seedmap *)calloc(1, sizeof *sm);
    if (!sm) return NULL;
    nb = next_prime(nb);
    sm->nb = nb;
    sm->b = (orc_knode **)calloc(nb, sizeof *sm->b);
    if (!sm->b) { free(sm); return NULL; }
    return sm;
}

void orc_seedmap_clear(orc_seedmap *sm)
{
    for (size_t i = 0; i < sm->nb; ++i) {
        orc_knode *k = sm->b[i];
        while (k) {
            orc_pnode *p = k->head;
            while (p) { orc_pnode *q = p->next; free(p); p = q; }
            orc_knode *kn = k->next; free(k); k = kn;
        }
        sm->b[i] = NULL;
    }
    sm->nkeys = sm->nentries = 0;
}

void orc_seedmap_free(orc_seedmap *sm)
{
    if (!sm) return;
    orc_seedmap_clear(sm);
    free(sm->b);
    free(sm);
}

size_t orc_seedmap_size(const orc_seedmap *sm) { return sm->nkeys; }
size_t orc_seedmap_entries(const orc_seedmap *sm) { return sm->nentries; }

static const orc_knode *sm_find(const orc_seedmap *sm, uint32_t key)
{
    for (const orc_knode *k = sm->b[key % sm->nb]; k; k = k->next)
        if (k->key == key) return k;
    return NULL;
}

static void sm_push(orc_seedmap *sm, uint32_t key, int32_t pos)
{
    orc_knode **slot = &sm->b[key % sm->nb], *k;
    for (k = *slot; k; k = k->next)
        if (k->key == key) break;
    if (!k) {
        k = (orc_knode *)calloc(1, sizeof *k);
        k->key = key; k->next = *slot; *slot = k;
        ++sm->nkeys;
    }
    orc_pnode *p = (orc_pnode *)malloc(sizeof *p);
    p->pos = pos; p->next = NULL;
    if (k->tail) k->tail->next = p; else k->head = p;
    k->tail = p;
    ++k->n; ++sm->nentries;
}

/* locator.cpp:62-66 */
size_t orc_index_all(orc_seedmap *sm, const char *text, int len, uint32_t mask)
{
    for (int i = 0; i < len; ++i) {
        uint32_t sd = orc_encode_padded(text + i, (long)len - i);
        if (sd & mask) sm_push(sm, sd & mask, i);
    }
    return sm->nentries;
}

/* ref_seq.h:291-311; MAX_READ_LEN=20000 (common.h:33), N_SEQ_WORD=16 (dna_seq.h:26) */
unsigned orc_index_head_tail(orc_seedmap *sm, const char *text, int len, uint32_t mask)
{
    const int MAXRD = 20000, NW = 16;
    int nmax = len - NW;
    int nhead = nmax < MAXRD ? nmax : MAXRD;
    orc_seedmap_clear(sm);
    for (int i = 0; i < nhead; ++i) {
        uint32_t sd = orc_encode(text + i);
        if (sd & mask) sm_push(sm, sd & mask, i);
    }
    int ntail = len - MAXRD - NW < MAXRD ? len - MAXRD - NW : MAXRD;
    for (int i = 0; i < ntail; ++i) {
        uint32_t sd = orc_encode(text + len - NW - i);
        if (sd & mask) sm_push(sm, sd & mask, len - i - NW);
    }
    return (unsigned)(nhead + (ntail < 0 ? 0 : ntail));
}

int orc_seedmap_find(const orc_seedmap *sm, uint32_t key, int32_t *pos, int cap)
{
    const orc_knode *k = sm_find(sm, key);
    if (!k) return 0;
    int n = 0;
    for (const orc_pnode *p = k->head; p; p = p->next, ++n)
        if (pos && n < cap) pos[n] = p->pos;
    return n;
}

static int cmp_knode(const void *x, const void *y)
{
    uint32_t a = (*(const orc_knode *const *)x)->key, b = (*(const orc_knode *const *)y)->key;
    return a < b ? -1 : a > b;
}

size_t orc_seedmap_dump(const orc_seedmap *sm, uint32_t *keys, int32_t *pos, size_t cap)
{
    const orc_knode **ks = (const orc_knode **)malloc((sm->nkeys + 1) * sizeof *ks);
    size_t nk = 0, n = 0;
    for (size_t i = 0; i < sm->nb; ++i)
        for (const orc_knode *k = sm->b[i]; k; k = k->next) ks[nk++] = k;
    qsort(ks, nk, sizeof *ks, cmp_knode);
    for (size_t i = 0; i < nk; ++i)
        for (const orc_pnode *p = ks[i]->head; p; p = p->next, ++n)
            if (n < cap) { keys[n] = ks[i]->key; pos[n] = p->pos; }
    free(ks);
    return n;
}

/* ======================================================================== */
/* locator driver                                                           */
/* ======================================================================== */

typedef struct {
    const char *contig; int clen; uint32_t mask; double R; int trials, min_len, maxn, maxm;
    const char *reads; const uint64_t *offs; int nreads;
    const orc_seedmap *sm;
    orc_loc_row *rows;
    volatile int next;
    pthread_mutex_t mu;
    orc_loc_stats st;
} loc_job;

/* one read of the loop at locator.cpp:70-92 */
static void locate_one(loc_job *J, orc_aligner *al, int r, orc_loc_stats *st)
{
    const char *seq = J->reads + J->offs[r];
    const int len = (int)(J->offs[r + 1] - J->offs[r]);
    orc_loc_row *row = &J->rows[r];
    row->found = 0; row->j = -1; row->pos = -1; row->cost = -1; row->seglen = 0;
    row->matlen_a = row->matlen_b = 0; row->n_pairs = 0;
    if (len < J->min_len) return;                                   /* locator.cpp:72 */
    ++st->n_reads_kept;
    for (int j = 0; j < J->trials && !row->found; ++j) {            /* locator.cpp:74 */
        uint32_t seed = orc_encode_padded(seq + j, (long)len - j) & J->mask;
        const orc_knode *k = sm_find(J->sm, seed);                  /* locator.cpp:76 */
        if (!k) continue;
        ++st->n_probe_hits;
        for (const orc_pnode *p = k->head; p; p = p->next) {        /* locator.cpp:79 */
            orc_result res;
            ++row->n_pairs; ++st->n_pairs;
            int rc = orc_align(al, seq + j, 1, len - j,              /* a = read   (locator.cpp:78) */
                               J->contig + p->pos, 1, J->clen - p->pos, /* b = contig (locator.cpp:80) */
                               J->R, &res, NULL);
            st->n_cells += res.cells;
            if (rc > 0) {                                           /* locator.cpp:82 */
                row->found = 1; row->j = j; row->pos = p->pos; row->cost = res.cost;
                row->seglen = len - j; row->matlen_a = res.matlen_a; row->matlen_b = res.matlen_b;
                ++st->n_located;
                break;
            }
        }
    }
}

static void *loc_worker(void *arg)
{
    loc_job *J = (loc_job *)arg;
    orc_aligner *al = pool_get(J->maxn, J->maxm);
    orc_loc_stats st; memset(&st, 0, sizeof st);
    for (;;) {
        int r = __sync_fetch_and_add(&J->next, 1);
        if (r >= J->nreads) break;
        locate_one(J, al, r, &st);
    }
    pool_put(al);
    pthread_mutex_lock(&J->mu);
    J->st.n_reads_kept += st.n_reads_kept; J->st.n_probe_hits += st.n_probe_hits;
    J->st.n_pairs += st.n_pairs; J->st.n_located += st.n_located; J->st.n_cells += st.n_cells;
    pthread_mutex_unlock(&J->mu);
    return NULL;
}

int orc_locator_run(const char *contig, int clen, uint32_t mask, double R,
                    int trials, int min_len, int maxn, int maxm,
                    const char *reads, const uint64_t *offs, int nreads,
                    int nthreads, orc_loc_row *rows, orc_loc_stats *stats)
{
    orc_seedmap *sm = orc_seedmap_new((size_t)1 << 23);             /* locator.cpp:28 */
    if (!sm) return -1;
    orc_index_all(sm, contig, clen, mask);

    int nseq = 0;                                                   /* locator.cpp:69,72,91 */
    for (int r = 0; r < nreads; ++r) {
        int len = (int)(offs[r + 1] - offs[r]);
        rows[r].read = r;
        rows[r].nseq = len < min_len ? -1 : nseq++;
    }

    loc_job J; memset(&J, 0, sizeof J);
    J.contig = contig; J.clen = clen; J.mask = mask; J.R = R; J.trials = trials;
    J.min_len = min_len; J.maxn = maxn; J.maxm = maxm;
    J.reads = reads; J.offs = offs; J.nreads = nreads; J.sm = sm; J.rows = rows;
    pthread_mutex_init(&J.mu, NULL);
    if (nthreads < 1) nthreads = 1;
    if (nthreads == 1) {
        loc_worker(&J);
    } else {
        pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
        for (int t = 0; t < nthreads; ++t) pthread_create(&th[t], NULL, loc_worker, &J);
        for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
        free(th);
    }
    pthread_mutex_destroy(&J.mu);
    if (stats) *stats = J.st;
    orc_seedmap_free(sm);
    return 0;
}

/* ======================================================================== */
/* spaced_seed locked round                                                 */
/* ======================================================================== */

typedef struct {
    const char *ref; int ref_len; uint32_t mask; double R; int max_trial, overlap_min, buggy;
    const uint8_t *records; const uint64_t *rec_offs; int nreads;
    const orc_seedmap *sm; orc_ss_row *rows; volatile int next;
} ss_job;

/* spaced_seed.cpp:262-298 with ref_seq::try_align (ref_seq.h:259-265) in locked mode */
static int ss_try(ss_job *J, orc_aligner *al, const uint8_t *rec, const char *txt, int seg_len,
                  long pos, int dir, orc_ss_row *row)
{
    uint32_t sd = J->buggy ? orc_seed_at(rec, (int)pos) : orc_seed_at_fixed(rec, (int)pos);
    const orc_knode *k = sm_find(J->sm, sd & J->mask);              /* spaced_seed.cpp:265 */
    if (!k) return 0;
    ++row->n_trials;
    const int fwd = dir == 1;
    const int s_off = fwd ? (int)pos : (int)pos + 15;               /* spaced_seed.cpp:274 */
    const int s_len = fwd ? seg_len - s_off : s_off + 1;            /* spaced_seed.cpp:275 */
    if (s_len < J->overlap_min) return 0;                           /* spaced_seed.cpp:280 */
    for (const orc_pnode *p = k->head; p; p = p->next) {
        const int r_off = fwd ? p->pos : p->pos + 15;               /* spaced_seed.cpp:285 */
        const int r_len = fwd ? J->ref_len - r_off : r_off + 1;     /* ref_seq.h:284-285 */
        orc_result res;
        ++row->n_pairs;
        int rc = orc_align(al, J->ref + r_off, fwd, r_len,          /* a = reference (ref_seq.h:264) */
                           txt + s_off, fwd, s_len, J->R, &res, NULL);
        if (rc < 0) continue;                                       /* ref_seq.h:264 */
        if (res.matlen_a < J->overlap_min) continue;                /* ref_seq.h:265 */
        row->found = 1; row->dir = dir; row->ref_pos = p->pos; row->cost = res.cost;
        row->matlen_a = res.matlen_a; row->matlen_b = res.matlen_b;
        return 1;
    }
    return 0;
}

static void *ss_worker(void *arg)
{
    ss_job *J = (ss_job *)arg;
    orc_aligner *al = pool_get(0, 0);                               /* canonical t_aligner: no aliasing, no size guard */
    char *txt = (char *)malloc(1 << 20);
    for (;;) {
        int r = __sync_fetch_and_add(&J->next, 1);
        if (r >= J->nreads) break;
        const uint8_t *rec = J->records + J->rec_offs[r];
        orc_ss_row *row = &J->rows[r];
        memset(row, 0, sizeof *row);
        row->read = r; row->j = -1;
        int slen = (int)orc_bin2text(rec, txt, 1 << 20);            /* spaced_seed.cpp:116 */
        for (int j = 0; j < J->max_trial; ++j) {                    /* spaced_seed.cpp:424-426 */
            if (ss_try(J, al, rec, txt, slen, j, 1, row) ||
                ss_try(J, al, rec, txt, slen, (long)slen - j - 16, -1, row)) {
                row->j = j;
                break;
            }
        }
    }
    free(txt);
    pool_put(al);
    return NULL;
}

int orc_spaced_round(const char *ref, int ref_len, uint32_t mask, double R,
                     int max_trial, int overlap_min, int buggy_seed_at,
                     const uint8_t *records, const uint64_t *rec_offs, int nreads,
                     int nthreads, orc_ss_row *rows)
{
    orc_seedmap *sm = orc_seedmap_new((size_t)1 << 20);             /* spaced_seed.cpp:88 */
    if (!sm) return -1;
    orc_index_head_tail(sm, ref, ref_len, mask);                    /* spaced_seed.cpp:415 */
    ss_job J; memset(&J, 0, sizeof J);
    J.ref = ref; J.ref_len = ref_len; J.mask = mask; J.R = R; J.max_trial = max_trial;
    J.overlap_min = overlap_min; J.buggy = buggy_seed_at;
    J.records = records; J.rec_offs = rec_offs; J.nreads = nreads; J.sm = sm; J.rows = rows;
    if (nthreads <= 1) {
        ss_worker(&J);
    } else {
        pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
        for (int t = 0; t < nthreads; ++t) pthread_create(&th[t], NULL, ss_worker, &J);
        for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
        free(th);
    }
    orc_seedmap_free(sm);
    return 0;
}

/* spaced_seed.cpp:330-342: keep min_excl < len < max_excl (500 / 20000 in the reference) */
size_t orc_open_binary(const uint8_t *buf, size_t len, uint32_t min_excl, uint32_t max_excl,
                       uint64_t *offs, size_t cap, size_t *n_total)
{
    size_t kept = 0, total = 0;
    for (size_t off = 0; off + 4 <= len; ) {
        uint32_t sl;
        memcpy(&sl, buf + off, 4);
        if (sl > min_excl && sl < max_excl) {
            if (kept < cap) offs[kept] = off;
            ++kept;
        }
        ++total;
        off += 4 + ((size_t)sl + 3) / 4;
    }
    if (n_total) *n_total = total;
    return kept;
}


/* =====================================================================================================
 * consensus voting and reference growth (ref_seq.h), restated on flat arrays: box k of the list is text
 * position pre + k.
 * ===================================================================================================== */
typedef struct { uint16_t sel[4], sup[4]; int32_t tot; } cbox;
struct orc_cons {
    int max_len;            /* MAX_SEQ_LEN */
    char *txt;              /* 3 * max_len */
    int beg, end, pre, post;
    cbox *box;              /* list begin = box[0]; capacity 3 * max_len */
    int nbox;
};

static cbox cbox_of(char ch, int n) {            /* vote_box(char c, int n): selection(c, n), total(1) -- ref_seq.h:118 */
    cbox b;
    memset(&b, 0, sizeof b);
    b.sel[orc_c2i(ch)] = (uint16_t)(b.sel[orc_c2i(ch)] + n);
    b.tot = 1;
    return b;
}
static int cmax4(const uint16_t *v) {            /* base_vote::max_vote, ref_seq.h:88-91 */
    int m = v[0];
    if (v[1] > m) m = v[1];
    if (v[2] > m) m = v[2];
    if (v[3] > m) m = v[3];
    return m;
}
static char cwinner(const uint16_t *v) {         /* base_vote::winner, ref_seq.h:96-100 */
    const int mv = cmax4(v);
    return mv == v[0] ? 'A' : (mv == v[1] ? 'C' : (mv == v[2] ? 'G' : 'T'));
}

orc_cons *orc_cons_new(const char *text, int len, int weight, int max_len) {
    if (len > max_len) return NULL;
    orc_cons *c = (orc_cons *)calloc(1, sizeof *c);
    c->max_len = max_len;
    c->txt = (char *)calloc((size_t)3 * max_len + 64, 1);
    c->box = (cbox *)calloc((size_t)3 * max_len + 64, sizeof(cbox));
    c->beg = c->pre = max_len;
    c->end = c->post = c->beg + len;
    memcpy(c->txt + c->beg, text, (size_t)len);
    for (int i = 0; i < len; ++i) c->box[i] = cbox_of(text[i], weight);
    c->nbox = len;
    return c;
}
void orc_cons_free(orc_cons *c) {
    if (!c) return;
    free(c->txt); free(c->box); free(c);
}
void orc_cons_append(orc_cons *c, const char *seg, int len) {
    memmove(c->txt + c->post, seg, (size_t)len);
    c->post += len;
    for (int i = 0; i < len; ++i) c->box[c->nbox++] = cbox_of(seg[i], 1);
}
void orc_cons_prepend(orc_cons *c, const char *seg, int len) {
    c->pre -= len;
    memmove(c->txt + c->pre, seg, (size_t)len);
    memmove(c->box + len, c->box, (size_t)c->nbox * sizeof(cbox));
    for (int i = 0; i < len; ++i) c->box[i] = cbox_of(seg[i], 1);   /* push_front from the last char backwards */
    c->nbox += len;
}

void orc_cons_elect(orc_cons *c, int pos, int fwd, const uint8_t *ops, const char *vals, int nedit) {
    int it = pos + c->beg - c->pre;              /* both iterators start at the box of `pos` (ref_seq.h:355,359) */
    const int step = fwd ? 1 : -1;
    for (int k = 0; k < nedit; ++k) {
        if (ops[k] == 3) {                       /* DELETE: it->ignore(); ++it */
            if (it >= 0 && it < c->nbox) c->box[it].tot++;
            it += step;
        } else if (ops[k] == 1) {                /* MATCH: it->select(val); ++it */
            if (it >= 0 && it < c->nbox) { c->box[it].sel[orc_c2i(vals[k])]++; c->box[it].tot++; }
            it += step;
        } else if (ops[k] == 2) {                /* INSERT: forward supplies the box before `it`, backward `it` itself */
            const int at = fwd ? it - 1 : it;
            if (at >= 0 && at < c->nbox) c->box[at].sup[orc_c2i(vals[k])]++;
        }
    }
}

int orc_cons_try(orc_cons *c, orc_aligner *al, int pos, const char *seg_origin, int seg_len, int fwd, double R,
                 int overlap_min, int32_t *out) {
    const char *a = c->txt + c->beg + pos;                                   /* get_accessor, ref_seq.h:282-286 */
    const int la = fwd ? c->post - c->beg - pos : pos + c->beg - c->pre + 1;
    orc_result res;
    uint8_t *ops = (uint8_t *)malloc((size_t)la + seg_len + 8);
    char *vals = (char *)malloc((size_t)la + seg_len + 8);
    int ok = 0;
    orc_align(al, a, fwd, la, seg_origin, fwd, seg_len, R, &res, ops);       /* ref_seq.h:264: a = the reference */
    if (res.rc >= 0 && res.matlen_a >= overlap_min) {                        /* ref_seq.h:264-265 */
        ok = 1;
        int j = 0;
        for (int k = 0; k < res.nedit; ++k)                                  /* edits[k].val, seq_aligner.h:218,224 */
            if (ops[k] == 1 || ops[k] == 2) { vals[k] = fwd ? seg_origin[j] : seg_origin[-j]; ++j; } else vals[k] = 0;
        orc_cons_elect(c, pos, fwd, ops, vals, res.nedit);                   /* ref_seq.h:267 */
        if (res.matlen_a == la) {                                            /* ref_seq.h:268-275 */
            const int add = seg_len - res.matlen_b;
            if (fwd) orc_cons_append(c, seg_origin + res.matlen_b, add);
            else orc_cons_prepend(c, seg_origin - (seg_len - 1), add);
        }
    }
    if (out) {
        out[0] = ok; out[1] = ok ? res.matlen_b : 0; out[2] = ok ? res.cost : 0; out[3] = ok ? res.matlen_a : 0;
        out[4] = ok ? res.nedit : 0; out[5] = c->pre - c->beg; out[6] = c->post - c->beg;
    }
    free(ops); free(vals);
    return ok;
}

/* One UNLOCKED round of spaced_seed.cpp:420-446: the loop of orc_spaced_round, but serial and in pool order, because
 * pref->try_align (ref_seq.h:259-276) votes and grows the reference as it goes.  The seed map is get_seedmap's over
 * [beg, end) as the round finds it (spaced_seed.cpp:415).  rows[k] belongs to pool[k].  Returns nmatches. */
static int cr_try(orc_cons *c, orc_aligner *al, const orc_seedmap *sm, uint32_t mask, double R, int overlap_min, int buggy,
                  const uint8_t *rec, const char *txt, int seg_len, long pos, int dir, orc_ss_row *row)
{
    uint32_t sd = buggy ? orc_seed_at(rec, (int)pos) : orc_seed_at_fixed(rec, (int)pos);
    const orc_knode *k = sm_find(sm, sd & mask);                    /* spaced_seed.cpp:265 */
    if (!k) return 0;
    ++row->n_trials;
    const int fwd = dir == 1;
    const int s_off = fwd ? (int)pos : (int)pos + 15;               /* spaced_seed.cpp:274 */
    const int s_len = fwd ? seg_len - s_off : s_off + 1;            /* spaced_seed.cpp:275 */
    if (s_len < overlap_min) return 0;                              /* spaced_seed.cpp:280 */
    for (const orc_pnode *p = k->head; p; p = p->next) {
        const int r_off = fwd ? p->pos : p->pos + 15;               /* spaced_seed.cpp:285 */
        int32_t out[8];
        ++row->n_pairs;
        if (!orc_cons_try(c, al, r_off, txt + s_off, s_len, fwd, R, overlap_min, out)) continue;   /* spaced_seed.cpp:286 */
        row->found = 1; row->dir = dir; row->ref_pos = p->pos; row->cost = out[2];
        row->matlen_a = out[3]; row->matlen_b = out[1];
        return 1;
    }
    return 0;
}

int orc_cons_round(orc_cons *c, orc_aligner *al, uint32_t mask, double R, int max_trial, int overlap_min, int buggy_seed_at,
                   const uint8_t *records, const uint64_t *rec_offs, const int32_t *pool, int npool, orc_ss_row *rows)
{
    orc_seedmap *sm = orc_seedmap_new((size_t)1 << 20);
    if (!sm) return -1;
    orc_index_head_tail(sm, c->txt + c->beg, c->end - c->beg, mask);     /* ref_seq.h:291-311 */
    char *txt = (char *)malloc(1 << 20);
    int nmatches = 0;
    for (int k = 0; k < npool; ++k) {
        const uint8_t *rec = records + rec_offs[pool[k]];
        orc_ss_row *row = &rows[k];
        memset(row, 0, sizeof *row);
        row->read = pool[k]; row->j = -1;
        int slen = (int)orc_bin2text(rec, txt, 1 << 20);
        for (int j = 0; j < max_trial; ++j) {                       /* spaced_seed.cpp:424-426 */
            if (cr_try(c, al, sm, mask, R, overlap_min, buggy_seed_at, rec, txt, slen, j, 1, row) ||
                cr_try(c, al, sm, mask, R, overlap_min, buggy_seed_at, rec, txt, slen, (long)slen - j - 16, -1, row)) {
                row->j = j; ++nmatches;
                break;
            }
        }
    }
    free(txt);
    orc_seedmap_free(sm);
    return nmatches;
}

void orc_cons_evolve(orc_cons *c) {
    cbox *nb = (cbox *)calloc((size_t)3 * c->max_len + 64, sizeof(cbox));
    int nn = 0;
    c->end = c->pre = c->beg = c->max_len;
    char *p = c->txt + c->beg;
    for (int i = 0; i < c->nbox; ++i) {
        cbox cur = c->box[i], vb;
        int has_vb = 0;
        if ((double)cmax4(cur.sup) > 0.5 * cur.tot) {       /* has_supply(0.5): split, insert after cur */
            memset(&vb, 0, sizeof vb);
            memcpy(vb.sel, cur.sup, sizeof vb.sel);
            vb.tot = cur.tot;
            memset(cur.sup, 0, sizeof cur.sup);
            has_vb = 1;
        }
        for (int rep = 0; rep <= has_vb; ++rep) {           /* cur, then the box just inserted behind it */
            const cbox *b = rep ? &vb : &cur;
            if ((double)cmax4(b->sel) > 0.5 * b->tot) {     /* is_valid(0.5): keep */
                *p++ = cwinner(b->sel);
                ++c->end;
                nb[nn++] = *b;
            } else if (nn > 0) {                            /* delete: the previous box absorbs its selection */
                for (int k = 0; k < 4; ++k) nb[nn - 1].sup[k] = (uint16_t)(nb[nn - 1].sup[k] + b->sel[k]);
            }
        }
    }
    c->post = c->end;
    free(c->box);
    c->box = nb;
    c->nbox = nn;
}

int orc_cons_dump(const orc_cons *c, uint16_t *sel, uint16_t *sup, int32_t *tot, int cap, int32_t *extent) {
    for (int k = 0; k < c->nbox && k < cap; ++k) {
        memcpy(sel + 4 * k, c->box[k].sel, 8);
        memcpy(sup + 4 * k, c->box[k].sup, 8);
        tot[k] = c->box[k].tot;
    }
    if (extent) { extent[0] = c->pre - c->beg; extent[1] = c->post - c->beg; extent[2] = c->end - c->beg; }
    return c->nbox;
}
int orc_cons_text(const orc_cons *c, char *out, int cap) {
    const int n = c->post - c->pre;
    memcpy(out, c->txt + c->pre, (size_t)(n < cap ? n : cap));
    return n;
}
