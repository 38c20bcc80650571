// align_bitvec.h -- bit-parallel kernel body for 2-bit packed sequences: one candidate pair per
// wavefront, the DP band swept as a 64-lane systolic array.
//
// What is computed.  seq_aligner<>::align (/root/reference/src/seq_aligner.h:92-213, canonical
// reading SURVEY.md A.4) exposes only: whether some diagonal cell fails cost(i,i) > i*R for
// i > 10 (:185), and the first strict minimum of the last row (or column) past the diagonal
// (:191-213).  Both are functions of the plain unit-cost edit-distance matrix U with
// U(i,0)=i, U(0,j)=j, because (DESIGN.md "exactness of the bit-vector kernel"):
//   * banded D >= U, and a path that leaves the band |i-j| <= max_dst and comes back to
//     diagonal offset d costs at least 2*(max_dst+1) - |d|;
//   * a passing diagonal cell has cost <= floor(i*R) <= max_dst-1, so banded and unbanded
//     values agree wherever the comparison could go either way, and every goal-row cell that
//     could be a strict minimum costs less than D(m,m) <= max_dst-1.
// U is evaluated in Myers/Hyyro vertical-delta encoding: a 32-row block holds +1/-1 bit masks
// (Pv, Mv); one text column updates the block with ~20 integer ops, i.e. 32 DP cells per
// ~20 lane-ops instead of ~8 ops per cell.
//
// Mapping to the wavefront.  Rows (the shorter sequence, m of them) are cut into superblocks of
// NB*32 rows; superblock s lives in lane s mod 64 and processes column j at step t = j + s, so
// a column flows down the lanes one lane per step: the horizontal delta leaving the bottom row
// of a superblock (hout) and the running diagonal score travel to the next lane with one DPP
// wave-rotate per step.  Only columns within w of the superblock's rows are processed
// (lo..hi); cells left of the window are taken as "+1 per row" and the row above the window
// as "+1 per column", which are real (if expensive) paths, so everything computed is an upper
// bound W >= U that equals U whenever U's optimal path stays within |i-j| <= w.
// With w >= max_dst/2 every diagonal verdict is exact (an out-of-window path costs >= 2w+2 >
// floor(i*R)); the goal row is exact when its minimum is <= w (any unseen path costs >= w+1);
// otherwise the pair is re-run with w = max_dst, the reference's own band, where the two
// bullets above apply directly.  Results are bit-identical to the reference in all cases.
#ifndef PBA_ALIGN_BITVEC_H
#define PBA_ALIGN_BITVEC_H

#include <limits.h>

#include "align_rowsweep.h"
#include "dev_common.h"

#define PBA_BV_MAX_NB 8
#define PBA_BV_BAND_NUM 9      // first-pass half width = max(max_dst/2, 9/16 * max_dst) + 1
#define PBA_BV_BAND_DEN 16

// largest half-width a superblock of NB blocks supports: lane s must be done with superblock s
// before superblock s+64 starts (2w <= 63*32*NB + 64)
__device__ __host__ inline int bv_max_w(int nb) { return 1008 * nb + 32; }
__device__ __host__ inline int bv_nb_for(int w) {
    const int need = (w - 32 + 1007) / 1008;
    const int nb = need < 1 ? 1 : need;
    return nb <= 4 ? nb : (nb <= 6 ? 6 : (nb <= 8 ? 8 : 0));   // instantiated: 1,2,3,4,6,8; 0 = too wide
}

// sequential reader of an accessor's 2-bit elements through a 64-bit register window
struct BaseStream {
    uint64_t X;
    // position on element e of accessor f (e may be slightly out of range: sequence sets carry
    // 1 KB of readable slack on both sides and out-of-range elements are never used)
    __device__ __forceinline__ void seek(const PackedFetch &f, int e) {
        const int idx = f.org + f.dir * e;
        if (f.dir > 0) X = __builtin_bswap64(ld_u64(f.seq + (idx >> 2))) << (2 * (idx & 3));
        else X = __builtin_bswap64(ld_u64(f.seq + (idx >> 2) - 7)) >> (2 * (3 - (idx & 3)));
    }
    __device__ __forceinline__ int next(int dir) {     // valid for >= 16 calls after a seek
        int c;
        if (dir > 0) { c = (int)(X >> 62); X <<= 2; }
        else { c = (int)(X & 3); X >>= 2; }
        return c;
    }
};

__device__ __forceinline__ uint32_t wave_ror1(uint32_t v) {
    // lane i receives lane (i-1) mod 64
#ifdef PBA_BV_USE_BPERMUTE
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)((((threadIdx.x & 63) + 63) & 63) << 2), (int)v);
#else
    // DPP wave_ror:1 (0x13C), available on gfx9-family ISAs
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x13C, 0xF, 0xF, false);
#endif
}

// One sweep with half-width w.  Returns 0 when all diagonal checks pass (then best / bestj hold the
// goal-row minimum and its column), else the first failing row.
template <int NB>
__device__ __forceinline__ int bitvec_pass(const PackedFetch &rowsF, int m, const PackedFetch &colsF, int n, int w, double R,
                           int &best_out, int &bestj_out) {
    constexpr int RB = 32 * NB;                 // rows per superblock
    const int lane = threadIdx.x & (PBA_WAVE - 1);
    const int S = (m + RB - 1) / RB;            // superblocks
    const int s_m = S - 1;                      // superblock, block and bit of row m
    const int nb_m = ((m - 1) - s_m * RB) >> 5, r_m = (m - 1) & 31;
    const int hi_last = min(n, s_m * RB + RB + w);
    const int t1 = m + s_m;                     // step at which cell (m,m) is produced
    const int t_end = hi_last + s_m;

    uint32_t Pv[NB], Mv[NB], Plo[NB], Phi[NB];
    int s_cur = lane, lo, hi, base_row;

    auto open_superblock = [&]() {
        base_row = s_cur * RB;
        if (s_cur < S) {
            lo = max(1, base_row + 1 - w);
            hi = min(n, base_row + RB + w);
            // bit planes of the pattern rows: bit r of Plo/Phi = low/high bit of row base_row+32*nb+r+1
            BaseStream rs;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                uint32_t pl = 0, ph = 0;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    rs.seek(rowsF, base_row + 32 * nb + 16 * half);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const uint32_t c = (uint32_t)rs.next(rowsF.dir);
                        pl |= (c & 1u) << (16 * half + r);
                        ph |= (c >> 1) << (16 * half + r);
                    }
                }
                Plo[nb] = pl; Phi[nb] = ph;
            }
        } else {
            lo = INT_MAX; hi = INT_MAX;          // nothing left for this lane
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) { Plo[nb] = 0; Phi[nb] = 0; }
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) { Pv[nb] = ~0u; Mv[nb] = 0u; }
    };
    open_superblock();

    BaseStream tx;
    int score = 0, best = INT_MAX, bestj = 0, fail_row = 0;
    uint32_t msg = 1u << 0 | 0u;                 // {score << 2 | hout + 1}

    for (int t = 1; t <= t_end; ++t) {
        int j = t - s_cur;
        if (j > hi) {                            // window finished: take this lane's next superblock
            s_cur += PBA_WAVE;
            open_superblock();
            j = t - s_cur;
            tx.seek(colsF, j - 1);
        }
        if (j == lo) {                           // window opens: column lo-1 is "+1 per row"
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) { Pv[nb] = ~0u; Mv[nb] = 0u; }
        }
        if ((t & 15) == 1) tx.seek(colsF, j - 1);          // wave-uniform refill of the text window
        const int c = tx.next(colsF.dir);
        const uint32_t clo = 0u - (uint32_t)(c & 1), chi = 0u - (uint32_t)(c >> 1);

        // what the superblock above produced for this same column one step ago
        const uint32_t m_in = wave_ror1(msg);
        int hin = 1;                             // row above the window: "+1 per column"
        if (s_cur > 0 && j <= base_row + w) hin = (int)(m_in & 3u) - 1;

        const int rr = j - base_row - 1;         // row of the diagonal cell inside this superblock
        uint32_t d0_sel = 0, ph_m = 0, mh_m = 0;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            uint32_t Eq = ~(Plo[nb] ^ clo) & ~(Phi[nb] ^ chi);
            const uint32_t pv = Pv[nb], mv = Mv[nb];
            const uint32_t hneg = hin < 0 ? 1u : 0u, hpos = hin > 0 ? 1u : 0u;
            const uint32_t Xv = Eq | mv;
            Eq |= hneg;
            const uint32_t Xh = (((Eq & pv) + pv) ^ pv) | Eq;
            uint32_t Ph = mv | ~(Xh | pv);
            uint32_t Mh = pv & Xh;
            if ((rr >> 5) == nb) d0_sel = Xh | mv;          // D0: bit r set iff D(i,j) == D(i-1,j-1)
            if (nb == nb_m) { ph_m = Ph; mh_m = Mh; }
            hin = (int)(Ph >> 31) - (int)(Mh >> 31);         // hout of this block = hin of the next
            Ph = (Ph << 1) | hpos;
            Mh = (Mh << 1) | hneg;
            Pv[nb] = Mh | ~(Xv | Ph);
            Mv[nb] = Ph & Xv;
        }

        if (t <= t1) {
            // diagonal cell (j,j): D(j,j) = D(j-1,j-1) + 1 - D0 bit; early failure seq_aligner.h:185
            if ((unsigned)rr < (unsigned)RB && j <= m) {
                if (rr == 0) score = s_cur == 0 ? 0 : (int)(m_in >> 2);
                score += 1 - (int)((d0_sel >> (rr & 31)) & 1u);
                if (j > 10 && (double)score > (double)j * R && fail_row == 0) fail_row = j;
                if (j == m) { best = score; bestj = m; }
            }
            if (__builtin_amdgcn_ballot_w64(fail_row != 0)) break;
        } else if (s_cur == s_m && j > m && j <= hi) {
            // goal row m right of the diagonal: D(m,j) = D(m,j-1) + horizontal delta (seq_aligner.h:202-211)
            score += (int)((ph_m >> r_m) & 1u) - (int)((mh_m >> r_m) & 1u);
            if (score < best) { best = score; bestj = j; }
        }
        msg = ((uint32_t)score << 2) | (uint32_t)(hin + 1);
    }

    // rows fail in increasing order of step, so the smallest recorded row is the first failing row
    int fr = fail_row ? fail_row : INT_MAX;
#pragma unroll
    for (int d = 1; d < PBA_WAVE; d <<= 1) fr = min(fr, __shfl_xor(fr, d, PBA_WAVE));
    if (fr != INT_MAX) return fr;
    const int owner = s_m & (PBA_WAVE - 1);
    best_out = __shfl(best, owner, PBA_WAVE);
    bestj_out = __shfl(bestj, owner, PBA_WAVE);
    return 0;
}

// true when the bit-vector kernel can take a pair with this max_dst (else: row sweep)
__device__ __host__ inline bool bitvec_supports(int max_dst) { return bv_nb_for(max_dst) != 0; }
// first-pass half width for a given max_dst
__device__ __host__ inline int bv_first_w(int md) {
    const int w = (md / 2 > (int)((long long)md * PBA_BV_BAND_NUM / PBA_BV_BAND_DEN)
                       ? md / 2 : (int)((long long)md * PBA_BV_BAND_NUM / PBA_BV_BAND_DEN)) + 1;
    return w > md ? md : w;
}

#define PBA_RC_UNCERTIFIED (-3)   // narrow pass could not certify the goal row: re-run with full_band

// One pair.  NB is chosen by the host from the largest max_dst in the launch (any NB >= the pair's own
// need is valid).  full_band = false runs the narrow first pass and reports PBA_RC_UNCERTIFIED when the
// goal row cannot be certified; the host then re-launches those pairs with full_band = true.
// There is deliberately no device function call in here: everything inlines into the kernel.
// lds: >= 256 bytes (only the m <= 10 corner uses it, through the row sweep)
template <int NB>
__device__ __forceinline__ void align_bitvec(const PackedFetch &fa, int la, const PackedFetch &fb, int lb, double R,
                                             int maxn, int maxm, bool full_band, uint16_t *lds, int lds_cells,
                                             AlnOut &o) {
    aln_params(la, lb, R, o);
    const int len_a = o.len_a, len_b = o.len_b, md = o.max_dst;
    if (maxn > 0 && (len_a >= maxn + maxm || md >= maxm)) return;      // seq_aligner.h:104-107
    const bool swap = len_a > len_b;            // the DP is symmetric under transposition: rows = shorter side
    const int m = swap ? len_b : len_a, n = swap ? len_a : len_b;
    if (m <= 10) {                              // no diagonal check ever fires: plain DP on a <= 23-cell band
        align_rowsweep(fa, la, fb, lb, R, maxn, maxm, lds, lds_cells, o);
        return;
    }
    const int w = full_band ? md : bv_first_w(md);
    if (w > bv_max_w(NB)) { o.rc = -2; return; }   // host sizes NB for the launch: cannot happen
    const PackedFetch rowsF = swap ? fb : fa, colsF = swap ? fa : fb;
    int best = 0, bestj = 0;
    const int fr = bitvec_pass<NB>(rowsF, m, colsF, n, w, R, best, bestj);
    if (fr) { o.fail_row = fr; return; }
    if (w < md && best > w) { o.rc = PBA_RC_UNCERTIFIED; return; }     // header comment: goal row not certified
    o.cost = best;
    o.matlen_a = swap ? bestj : m;
    o.matlen_b = swap ? m : bestj;
    o.rc = ((double)o.matlen_b < (double)len_b * (1.0 - R)) ? -1 : o.matlen_b;   // seq_aligner.h:114
}

#endif
