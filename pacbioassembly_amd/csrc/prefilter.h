// prefilter.h -- the first 32 rows of seq_aligner<>::align for 64 candidate pairs at once, ONE PAIR PER LANE.
//
// Most candidates a seed lookup returns are false hits, and the reference's sweep throws them out at one of its
// first diagonal checks (cost(i,i) > i*R for i > 10, /root/reference/src/seq_aligner.h:185): a random pair fails by
// row ~11-20.  The cell (i,i) depends only on the square [1..i] x [1..i], and on that square the banded matrix and
// the plain edit-distance matrix U decide the check identically (point 1: a path that passes the check
// cannot afford to leave the band).  So one 32-row Myers block per lane, swept over the first 32 columns, gives the
// reference's verdict for rows 11..32 exactly: the first failing row if there is one.  Pairs that survive go to the
// wavefront-wide aligner as before; pairs that fail here never get a wavefront.
// ~16 lane-ops per column instead of a 64-lane array stepping through its ramp: a false candidate costs ~10
// wave-instructions instead of several thousand.
#ifndef PBA_PREFILTER_H
#define PBA_PREFILTER_H

#include "align_bitvec.h"
#include "dev_common.h"

#define PBA_PRE_ROWS 32

// floor(i * R) for i = 0..32, FP64 exactly as the reference forms i*R; an integer d satisfies
// (double)d > (double)i*R  <=>  d > floor((double)i*R).  R < 1, so a threshold fits a byte: four to a register (33 separate
// scalars were more than the scalar file had left next to the aligner's state -- they ended up in scratch memory, one
// scratch load per column of the sweep).
struct PreThresholds {
    uint32_t p[(PBA_PRE_ROWS + 4) / 4];
    PreThresholds() = default;
    // formed on the host for a kernel that would otherwise form them once per thread (same FP64 products: -ffp-contract=off)
    static PreThresholds on_host(double R) {
        PreThresholds t;
        for (int w = 0; w < (PBA_PRE_ROWS + 4) / 4; ++w) {
            t.p[w] = 0;
            for (int b = 0; b < 4; ++b) {
                const int i = 4 * w + b;
                if (i <= PBA_PRE_ROWS) t.p[w] |= (uint32_t)(int)__builtin_floor((double)i * R) << (8 * b);
            }
        }
        return t;
    }
    __device__ __forceinline__ explicit PreThresholds(double R) {
#pragma unroll
        for (int w = 0; w < (PBA_PRE_ROWS + 4) / 4; ++w) {
            uint32_t x = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int i = 4 * w + b;
                if (i <= PBA_PRE_ROWS) x |= (uint32_t)(int)floor((double)i * R) << (8 * b);
            }
            p[w] = (uint32_t)__builtin_amdgcn_readfirstlane((int)x);
        }
    }
    __device__ __forceinline__ int at(int i) const { return (int)((p[i >> 2] >> (8 * (i & 3))) & 0xFFu); }
};

// the sweep itself: a's first 32 elements as rows (alo / ahi), b's first 32 as columns (blo / bhi: bit k = column k+1)
__device__ __forceinline__ int prefilter32_planes(uint32_t alo, uint32_t ahi, uint32_t blo, uint32_t bhi, const PreThresholds &T) {
    uint32_t Pv = ~0u, Mv = 0u;                // column 0: D(i,0) = i
    int score = 0, fr = 0;                     // score = D(j,j)
#pragma unroll
    for (int k = 0; k < PBA_PRE_ROWS; ++k) {
        const uint32_t clo = bit_mask(blo, k), chi = bit_mask(bhi, k);
        const uint32_t Eq = ~(alo ^ clo) & ~(ahi ^ chi);
        const uint32_t Xv = Eq | Mv;
        const uint32_t Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
        uint32_t Ph = Mv | ~(Xh | Pv);
        uint32_t Mh = Pv & Xh;
        const uint32_t D0 = Xh | Mv;           // bit r: D(r+1, k+1) == D(r, k)
        score += 1 - (int)((D0 >> k) & 1u);    // D(k+1, k+1) = D(k, k) + (the diagonal cell's D0 ? 0 : 1)
        Ph = (Ph << 1) | 1u;                   // row 0 grows by one per column: D(0,j) = j
        Mh <<= 1;
        Pv = Mh | ~(Xv | Ph);
        Mv = Ph & Xv;
        if (k + 1 > 10 && fr == 0 && score > T.at(k + 1)) fr = k + 1;
    }
    return fr;
}

// The same 32 rows when only the VERDICT is wanted (the all-vs-all scan: a candidate either survives or is counted) -- and it is
// wanted 57 G times per million reads, so every instruction of the column shows up in the scan's time:
//  * the diagonal is not scored column by column: the D0 bit of cell (k+1, k+1) is put aside (one op), and afterwards
//    D(i,i) = i - popcount of the first i of those bits;
//  * D(i,i) never decreases with i, and floor(i*R) is constant over stretches of rows, so a stretch fails iff its LAST row
//    fails: only those rows are checked (7 of the 22 at R = 0.30; PreChecks::rows, formed on the host);
//  * the column's logic in v_bitop3_b32 forms (three inputs, any truth table, full rate): 16 vector ops per column, 4 of them
//    at the half rate of shifts and bit-field extracts, where the scoring form has ~23.
// Same cells, same FP64-derived thresholds, same verdict (every all-vs-all test holds it against the row-sweep kernel, which
// has no prefilter).
// rows the scan sweeps (<= 32; tuning hook).  Fewer rows cost less per candidate and let more candidates through to the walk,
// whose own stage (rows 1..64 of every listed candidate) is exact whatever got here: measured at a million reads, §5.1
#ifndef PBA_SCAN_ROWS
#define PBA_SCAN_ROWS 32
#endif
struct PreChecks {
    PreThresholds t;
    uint32_t rows;                  // bit i-1: row i (11..PBA_SCAN_ROWS) is the last of a stretch of equal thresholds -- check it
    static PreChecks on_host(double R) {
        PreChecks c;
        c.t = PreThresholds::on_host(R);
        c.rows = 0;
        auto T = [&](int i) { return (int)((c.t.p[i >> 2] >> (8 * (i & 3))) & 0xFFu); };
        for (int i = 11; i <= PBA_SCAN_ROWS; ++i)
            if (i == PBA_SCAN_ROWS || T(i + 1) != T(i)) c.rows |= 1u << (i - 1);
        return c;
    }
};
// (a ^ b) | c,  a | ~(b | c)
__device__ __forceinline__ uint32_t bitop_xor_or(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t d;
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xbe" : "=v"(d) : "v"(a), "v"(b), "v"(c));   // (0xF0 ^ 0xCC) | 0xAA
    return d;
}
__device__ __forceinline__ uint32_t bitop_or_nor(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t d;
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xf1" : "=v"(d) : "v"(a), "v"(b), "v"(c));   // 0xF0 | ~(0xCC | 0xAA)
    return d;
}
// true: some row 11..32 fails the reference's check (seq_aligner.h:185)
__device__ __forceinline__ bool prefilter32_fails(uint32_t alo, uint32_t ahi, uint32_t blo, uint32_t bhi, const PreChecks &C) {
    uint32_t Pv = ~0u, Mv = 0u, diag = 0u;     // column 0: D(i,0) = i;  diag bit k: D(k+1, k+1) == D(k, k)
#pragma unroll
    for (int k = 0; k < PBA_SCAN_ROWS; ++k) {                             // (cell (i,i), i <= PBA_SCAN_ROWS, needs columns 1..i only)
        const uint32_t clo = bit_mask(blo, k), chi = bit_mask(bhi, k);
        const uint32_t Eq = eq_mask(alo ^ clo, ahi, chi);                 // ~(alo ^ clo) & ~(ahi ^ chi)
        const uint32_t Xh = bitop_xor_or((Eq & Pv) + Pv, Pv, Eq);         // (((Eq & Pv) + Pv) ^ Pv) | Eq
        const uint32_t Xv = Eq | Mv;
        uint32_t Ph = bitop_or_nor(Mv, Xh, Pv);                           // Mv | ~(Xh | Pv)
        uint32_t Mh = Pv & Xh;
        diag |= (Xh | Mv) & (1u << k);                                    // D0 of the diagonal cell
        Ph = (Ph << 1) | 1u;                                              // row 0 grows by one per column: D(0,j) = j
        Mh <<= 1;
        Pv = bitop_or_nor(Mh, Xv, Ph);                                    // Mh | ~(Xv | Ph)
        Mv = Ph & Xv;
    }
    bool fail = false;
#pragma unroll
    for (int i = 11; i <= PBA_SCAN_ROWS; ++i)
        if ((C.rows >> (i - 1)) & 1u)                                     // (wave-uniform: a kernel argument)
            fail = fail || i - (int)__builtin_popcount(i == 32 ? diag : diag & ((1u << i) - 1u)) > C.t.at(i);
    return fail;
}

// whether the prefilter applies to a pair of these accessor lengths (else the full aligner decides)
__device__ __forceinline__ bool prefilter32_applies(bool active, int la, int lb, double R, int maxn, int maxm, AlnOut &o) {
    aln_params(la, lb, R, o);
    if (!active) return false;
    if (maxn > 0 && (o.len_a >= maxn + maxm || o.max_dst >= maxm)) return false;   // seq_aligner.h:104-107: the caller's path
    return o.len_a >= PBA_PRE_ROWS && o.len_b >= PBA_PRE_ROWS;
}

// Returns the first failing row (11..32) of the pair this LANE holds, or 0 when rows 11..32 all pass or the
// prefilter does not apply (a sequence shorter than 32 after the reference's length clipping, the size guard) --
// then the full aligner decides.  `active`: lanes without a pair must pass false (they return 0).
__device__ __forceinline__ int prefilter32(bool active, const PackedFetch &fa, int la, const PackedFetch &fb, int lb,
                                           double R, int maxn, int maxm, const PreThresholds &T, AlnOut &o) {
    if (!prefilter32_applies(active, la, lb, R, maxn, maxm, o)) return 0;
    uint32_t alo, ahi, blo, bhi;
    load_planes32(fa, 0, alo, ahi);            // rows: a[0..31]
    load_planes32(fb, 0, blo, bhi);            // columns: b[0..31], bit k = column k+1
    return prefilter32_planes(alo, ahi, blo, bhi, T);
}

// ---- second stage: rows 33..64, still one pair per lane
// About one random pair in seventy survives its first 32 rows; nearly all of those fail before row 64.  Handing each
// of them to the wavefront-wide array costs ~2 500 instructions a piece (one pair in flight, the array mostly in its
// ramp); the same 64 rows as a two-block Myers column sweep in ONE lane cost about as much for up to 64 survivors at
// once.  Callers collect the survivors of many groups and run this on them together.  Same exactness argument as
// above: cell (i,i), i <= 64, depends on the square [1..i] x [1..i] only.
#define PBA_PRE2_ROWS 64
// true: some row 11..64 fails the reference's check (seq_aligner.h:185); false: all pass, or the stage does not apply
// (a sequence shorter than 64 after clipping) and the full aligner decides
__device__ __forceinline__ bool prefilter64(bool active, const PackedFetch &fa, int la, const PackedFetch &fb, int lb, double R) {
    AlnOut o;
    aln_params(la, lb, R, o);
    if (!active || o.len_a < PBA_PRE2_ROWS || o.len_b < PBA_PRE2_ROWS) return false;
    uint32_t alo[2], ahi[2], blo[2], bhi[2];
    load_planes32(fa, 0, alo[0], ahi[0]);
    load_planes32(fa, 32, alo[1], ahi[1]);
    load_planes32(fb, 0, blo[0], bhi[0]);
    load_planes32(fb, 32, blo[1], bhi[1]);
    uint32_t Pv0 = ~0u, Mv0 = 0u, Pv1 = ~0u, Mv1 = 0u;
    int score = 0;                              // D(k, k)
    bool failed = false;
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
#pragma unroll 4
        for (int kk = 0; kk < 32; ++kk) {
            const int k = 32 * half + kk;
            const uint32_t clo = bit_mask(blo[half], kk), chi = bit_mask(bhi[half], kk);
            // rows 1..32
            const uint32_t Eq0 = ~(alo[0] ^ clo) & ~(ahi[0] ^ chi);
            const uint32_t Xv0 = Eq0 | Mv0;
            const uint32_t Xh0 = (((Eq0 & Pv0) + Pv0) ^ Pv0) | Eq0;
            uint32_t Ph0 = Mv0 | ~(Xh0 | Pv0), Mh0 = Pv0 & Xh0;
            const uint32_t D00 = Xh0 | Mv0;     // bit r: D(r+1, k+1) == D(r, k)
            const uint32_t hp = Ph0 >> 31, hm = Mh0 >> 31;                   // the delta leaving row 32
            Ph0 = (Ph0 << 1) | 1u;              // row 0 grows by one per column: D(0,j) = j
            Mh0 <<= 1;
            Pv0 = Mh0 | ~(Xv0 | Ph0);
            Mv0 = Ph0 & Xv0;
            // rows 33..64, the delta of row 32 coming in at the top
            const uint32_t Eq1 = ~(alo[1] ^ clo) & ~(ahi[1] ^ chi);
            const uint32_t Xv1 = Eq1 | Mv1;
            const uint32_t Eq1c = Eq1 | hm;
            const uint32_t Xh1 = (((Eq1c & Pv1) + Pv1) ^ Pv1) | Eq1c;
            uint32_t Ph1 = Mv1 | ~(Xh1 | Pv1), Mh1 = Pv1 & Xh1;
            const uint32_t D01 = Xh1 | Mv1;
            Ph1 = (Ph1 << 1) | hp;
            Mh1 = (Mh1 << 1) | hm;
            Pv1 = Mh1 | ~(Xv1 | Ph1);
            Mv1 = Ph1 & Xv1;
            score += 1 - (int)(((half ? D01 : D00) >> kk) & 1u);             // D(k+1, k+1)
            if (k + 1 > 10 && (double)score > (double)(k + 1) * R) failed = true;   // seq_aligner.h:185, FP64 like the reference
        }
    }
    return failed;
}

#endif
