// align_rowsweep.h -- full-band row-sweep kernel body: one candidate pair per wavefront.
//
// Computes exactly the DP of seq_aligner<>::align (/root/reference/src/seq_aligner.h:92-213,
// canonical reading SURVEY.md A.4) with every band cell evaluated, as the reference does:
//   D(i,j) = min( D(i-1,j-1) + (a[i-1] != b[j-1]),  D(i,j-1)+1 if i-j < max_dst,
//                 D(i-1,j)+1 if j-i < max_dst )                    (seq_aligner.h:164-173)
// Layout: the band row (2*max_dst+1 cells, u16 costs) lives in LDS in diagonal-stripe
// coordinates c = j - i + max_dst (seq_aligner.h:131).  In these coordinates the diagonal
// predecessor is the same c of the previous row and the vertical one is c+1, so the row is
// updated in place, 64 cells per step; the horizontal (same-row) dependency is resolved with
// a wavefront min-plus prefix scan plus a carry between 64-cell groups.
// Works for any element type (2-bit codes or raw bytes): this is the general, reference-
// shaped kernel; align_bitvec.h is the fast path for packed ACGT.
#ifndef PBA_ALIGN_ROWSWEEP_H
#define PBA_ALIGN_ROWSWEEP_H

#include "dev_common.h"

__device__ __forceinline__ int wave_prefix_min(int y, int lane) {
#pragma unroll
    for (int d = 1; d < PBA_WAVE; d <<= 1) {
        const int t = __shfl_up(y, d, PBA_WAVE);
        if (lane >= d) y = min(y, t);
    }
    return y;
}

// par (nullable): one parent code per band cell (seq_aligner.h:165-175: 1 MATCH, 2 INSERT, 3 DELETE) at
// par[i * (2*max_dst+1) + (j - i + max_dst)], for the traceback of seq_aligner.h:214-233.
// cst (nullable, needs par): the cell's cost at the same index -- the reference's `mat` (seq_aligner.h:81,131-134) for the
// callers that read it (pba_align_text_matrix); rows 1.., column 0 included (init_cell's DELETE border), row 0 is the host's.
template <class FA, class FB>
__device__ void align_rowsweep(const FA &fa, int la, const FB &fb, int lb, double R, int maxn, int maxm,
                               uint16_t *row, int row_cap, AlnOut &o, uint8_t *par = nullptr, uint16_t *cst = nullptr) {
    const int lane = threadIdx.x & (PBA_WAVE - 1);
    aln_params(la, lb, R, o);
    const int len_a = o.len_a, len_b = o.len_b, m = o.max_dst;
    if (maxn > 0 && (len_a >= maxn + maxm || m >= maxm)) return;   // seq_aligner.h:104-107
    const int W = 2 * m + 1;
    if (W > row_cap) { o.rc = -2; return; }                         // engine limit; host checks first

    // row 0: D(0,j) = j for j in [0, max_dst] (seq_aligner.h:144-149); cells left of column 0 do not exist
    for (int c = lane; c < W; c += PBA_WAVE) row[c] = (uint16_t)(c >= m ? c - m : 0xFFFF);
    __builtin_amdgcn_wave_barrier();

    int col_best = 0, col_ml = 0;   // running goal along column len_b when a is the longer side
    for (int i = 1; i <= len_a; ++i) {
        const int sa = fa(i - 1);
        const int c_lo = max(0, m - i);                 // j = 0 while i <= max_dst, else j = i - max_dst
        const int c_hi = min(W - 1, len_b - i + m);     // j = min(len_b, i + max_dst)
        int carry = PBA_INF;                            // D(i, j-1) of the cell left of the group
        for (int c0 = c_lo & ~(PBA_WAVE - 1); c0 <= c_hi; c0 += PBA_WAVE) {
            const int c = c0 + lane;
            const int j = i + c - m;
            const bool inb = c >= c_lo && c <= c_hi;
            int v = PBA_INF, vm = PBA_INF, vd = PBA_INF;   // best of {match, delete}; the match / delete candidates
            if (inb) {
                if (j == 0) {
                    v = i;                              // D(i,0) = i, seq_aligner.h:140-143
                } else {
                    int od = row[c];                    // D(i-1, j-1)
                    int ou = c + 1 < W ? row[c + 1] : 0xFFFF;   // D(i-1, j), only if j-i < max_dst
                    od = od == 0xFFFF ? PBA_INF : od;
                    ou = ou == 0xFFFF ? PBA_INF : ou;
                    vm = od + (sa != fb(j - 1));
                    vd = ou + 1;
                    v = min(vm, vd);
                }
            }
            // D(i,j) = min over k <= j of v_k + (j-k): prefix-min of v_k - k, shifted back
            int y = wave_prefix_min(v - lane, lane);
            y = min(y, carry + 1);
            const int nv = y + lane;
            if (par) {                                  // parent, in the reference's order of strict comparisons
                int left = __shfl_up(nv, 1, PBA_WAVE);  // D(i, j-1)
                if (lane == 0) left = carry;
                int cost = vm, src = 1;
                if (c > 0 && left + 1 < cost) { cost = left + 1; src = 2; }   // i-j < max_dst  <=>  c > 0
                if (vd < cost) src = 3;                                       // j-i < max_dst is in vd (INF otherwise)
                if (inb && j > 0) par[(size_t)i * W + c] = (uint8_t)src;
                if (cst && inb) {
                    cst[(size_t)i * W + c] = (uint16_t)min(nv, 0xFFFF);
                    if (j == 0) par[(size_t)i * W + c] = 3;       // set_parent(i, 0, DELETE), seq_aligner.h:142
                }
            }
            carry = __builtin_amdgcn_readlane(nv, PBA_WAVE - 1);
            __builtin_amdgcn_wave_barrier();
            if (inb) row[c] = (uint16_t)min(nv, 0xFFFF);
        }
        __builtin_amdgcn_wave_barrier();
        // early failure on the diagonal cell, seq_aligner.h:185 (skipped when (i,i) is not a cell, A.4)
        if (i > 10 && i <= len_b) {
            const int d = __builtin_amdgcn_readfirstlane((int)row[m]);
            if ((double)d > (double)i * R) { o.fail_row = i; return; }
        }
        if (i == min(len_a, len_b)) o.diag = __builtin_amdgcn_readfirstlane((int)row[m]);   // D(i,i), the end of the diagonal
        if (len_a > len_b && i >= len_b) {              // goal_cell, seq_aligner.h:192-201
            const int v = __builtin_amdgcn_readfirstlane((int)row[len_b - i + m]);
            if (i == len_b || v < col_best) { col_best = v; col_ml = i; }
        }
    }

    if (len_a > len_b) {
        o.matlen_a = col_ml; o.matlen_b = len_b; o.cost = col_best;
    } else {
        // goal_cell, seq_aligner.h:202-211: first strict minimum of D(len_a, j), j = len_a..len_b
        int bv = PBA_INF, bj = 0x7FFFFFFF;
        for (int j = len_a + lane; j <= len_b; j += PBA_WAVE) {
            const int v = row[j - len_a + m];
            if (v < bv) { bv = v; bj = j; }
        }
#define PBA_RS_STEP(d) { const int ov = PBA_SWZ_XOR(bv, d), oj = PBA_SWZ_XOR(bj, d); \
                         if (ov < bv || (ov == bv && oj < bj)) { bv = ov; bj = oj; } }
        PBA_RS_STEP(1) PBA_RS_STEP(2) PBA_RS_STEP(4) PBA_RS_STEP(8) PBA_RS_STEP(16)       // (dev_common.h on why not __shfl_xor)
#undef PBA_RS_STEP
        const int v0 = __builtin_amdgcn_readlane(bv, 0), j0 = __builtin_amdgcn_readlane(bj, 0);
        const int v1 = __builtin_amdgcn_readlane(bv, 32), j1 = __builtin_amdgcn_readlane(bj, 32);
        const bool hi = v1 < v0 || (v1 == v0 && j1 < j0);
        o.matlen_a = len_a; o.matlen_b = hi ? j1 : j0; o.cost = hi ? v1 : v0;
    }
    // acceptance, seq_aligner.h:114
    o.rc = ((double)o.matlen_b < (double)len_b * (1.0 - R)) ? -1 : o.matlen_b;
}

#endif
