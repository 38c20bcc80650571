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
//     values agree wherever the comparison could go either way, and every goal cell (last row or
//     column past the diagonal) that could be a strict minimum costs less than D(m,m) <= max_dst-1.
// U is evaluated in Myers/Hyyro vertical-delta encoding: a 32-row block holds +1/-1 bit masks
// (Pv, Mv); one text column updates the block with ~20 integer ops, i.e. 32 DP cells per
// ~20 lane-ops instead of ~8 ops per cell.
//
// Mapping to the wavefront.  Rows are the LONGER sequence (as far down as the window reaches: m + w
// of them), columns the shorter one (m): the free end of the alignment then runs down the last
// column, whose cells are produced by the last ~w/RB superblocks while the lanes above still work --
// with the shorter sequence as rows the same cells cost w extra steps at the end with most of the
// array idle.  Rows are cut into superblocks of NB*32 rows; superblock s lives in lane s mod 64 and
// processes column j at step t = j + s, so a column flows down the lanes one lane per step.  The
// horizontal deltas leaving the bottom row of a superblock are carry-outs of v_addc_co_u32, i.e. lane
// masks in SGPR pairs, and reach the next lane through a scalar 64-bit rotate of those masks (no DPP,
// no VALU).  Only the columns of a superblock's window are processed (lo..hi); cells left of the
// window are taken as "+1 per row" and the row above the window as "+1 per column", which are real
// (if expensive) paths, so everything computed is an upper bound W >= U that equals U whenever U's
// optimal path stays inside the windows.
// The window is asymmetric: row i sees the columns [i - w, i + wl]: w on the side of the free end
// (below the diagonal, where the goal cells (i, m), i >= m, lie), wl on the other.  A path that leaves
// on the wl side must come back across the diagonal: it costs >= 2*wl+2; one that leaves on the w
// side costs >= w+1.  Hence:
//   * a PASSING diagonal verdict is always exact (W >= U);  a FAILING one at row i is exact when
//     floor(i*R) < 2*wl+2 (no unseen path back to the diagonal is cheap enough to change it);
//   * the goal column is exact when its minimum is <= min(w, 2*wl+1).
// Anything else answers "uncertified" and the pair is re-run with w = max_dst, the reference's own
// band, and wl = max_dst/2 + 1: every row's threshold floor(i*R) <= max_dst - 1 < 2*wl + 2 and every
// goal minimum of a pair that passed its checks is <= max_dst - 1, so that sweep certifies
// everything.  Results are bit-identical to the reference in all cases.  First pass: w = 9/16
// max_dst, wl = w/2: 25 % fewer cells than a symmetric window and, more to the point, narrow enough
// for one block less per lane at BASELINE sizes.
#ifndef PBA_ALIGN_BITVEC_H
#define PBA_ALIGN_BITVEC_H

#include <limits.h>

#include "align_rowsweep.h"
#include "dev_common.h"

#define PBA_BV_MAX_NB 8
#define PBA_BV_FIN_WORDS(nb) ((2 * (nb) + 1) * 64)     // u32 of LDS per wavefront for the last column's deltas (bitvec_pass: fin)
#define PBA_BV_BAND_NUM 9      // first-pass half width = max(max_dst/2, 9/16 * max_dst) + 1
#define PBA_BV_BAND_DEN 16

// widest window (wl + w) a superblock of NB blocks supports: lane s must be done with superblock s
// before superblock s+64 starts (wl + w <= 63*32*NB + 64)
__device__ __host__ inline int bv_max_span(int nb) { return 2016 * nb + 64; }
__device__ __host__ inline int bv_nb_for_span(int span) {
    const int need = (span - 64 + 2015) / 2016;
    const int nb = need < 1 ? 1 : need;
    return nb <= 4 ? nb : (nb <= 6 ? 6 : (nb <= 8 ? 8 : 0));   // instantiated: 1,2,3,4,6,8; 0 = too wide
}
__device__ __host__ inline int bv_nb_for(int w) { return bv_nb_for_span(2 * w); }   // symmetric window

// 64 bits of an accessor's 2-bit stream starting at base index `idx` (may be unaligned / slightly out of
// range: sequence sets carry 1 KB of readable slack on both sides): bit 63:62 = base idx, ..., bit 1:0 = base idx+31
__device__ __forceinline__ uint64_t load_bases32(const uint8_t *seq, int idx) {
    const uint8_t *p = seq + (idx >> 2);
    const uint64_t hi = __builtin_bswap64(ld_u64(p)), nx = __builtin_bswap64(ld_u64(p + 8));
    const int sh = 2 * (idx & 3);
    return sh ? (hi << sh) | (nx >> (64 - sh)) : hi;
}

// bit q of the result = bit 2q of x (q = 0..31)
__device__ __forceinline__ uint32_t compress_even64(uint64_t x) {
    uint32_t lo = (uint32_t)x & 0x55555555u, hi = (uint32_t)(x >> 32) & 0x55555555u;
    lo = (lo | (lo >> 1)) & 0x33333333u; hi = (hi | (hi >> 1)) & 0x33333333u;
    lo = (lo | (lo >> 2)) & 0x0F0F0F0Fu; hi = (hi | (hi >> 2)) & 0x0F0F0F0Fu;
    lo = (lo | (lo >> 4)) & 0x00FF00FFu; hi = (hi | (hi >> 4)) & 0x00FF00FFu;
    lo = (lo | (lo >> 8)) & 0x0000FFFFu; hi = (hi | (hi >> 8)) & 0x0000FFFFu;
    return lo | (hi << 16);
}

// bit planes of 32 consecutive accessor elements starting at element e: bit r of plo/phi = low/high bit of
// element e+r.  From the set's precomputed planes, stored word by word side by side (low word w, high word w, low word
// w+1, ...): four consecutive dwords -- one 16-byte fetch, one cache line for both planes -- and one funnel shift per
// plane (the index may run a few hundred bases before / past the sequence: neighbours or zero slack, never used).
__device__ __forceinline__ void load_planes32(const PackedFetch &f, int e, uint32_t &plo, uint32_t &phi) {
    const int idx = f.dir > 0 ? f.org + e : f.org - e - 31;          // lowest base index of the 32, in memory order
    const uint32_t *p = f.pl + 2 * (ptrdiff_t)(idx >> 5);             // (arithmetic shift: negative indices reach the slack)
    const uint32_t sh = (uint32_t)idx & 31u;
    const uint32_t lo = __builtin_amdgcn_alignbit(p[2], p[0], sh);    // ({low[w+1], low[w]} >> sh)[31:0]
    const uint32_t hi = __builtin_amdgcn_alignbit(p[3], p[1], sh);
    plo = f.dir > 0 ? lo : __builtin_bitreverse32(lo);                // backward: element r is base idx + 31 - r
    phi = f.dir > 0 ? hi : __builtin_bitreverse32(hi);
}

// the planes of 32 bases from the packed bytes (k_make_planes builds a set's planes with this, once)
__device__ __forceinline__ void planes_from_packed(const uint8_t *seq, int idx, uint32_t &plo, uint32_t &phi) {
    const uint64_t x = load_bases32(seq, idx);                        // base idx + r at bits 63-2r : 62-2r
    plo = __builtin_bitreverse32(compress_even64(x));
    phi = __builtin_bitreverse32(compress_even64(x >> 1));
}

// d = a + b + carry-in, carry-out to a lane mask: v_addc_co_u32 with SGPR-pair carries.  One instruction does
// "shift left by one, OR in the delta entering at the top row, hand out the delta leaving at the bottom row".
__device__ __forceinline__ uint32_t addc_mask(uint32_t a, uint32_t b, uint64_t cin, uint64_t &cout) {
    uint32_t d;
    asm("v_addc_co_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(cout) : "v"(a), "v"(b), "s"(cin));
    return d;
}

// a | (b & c) in one full-rate v_bitop3_b32 (the compiler would pick the half-rate v_and_or_b32)
__device__ __forceinline__ uint32_t or_of_and(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t d;
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xf8" : "=v"(d) : "v"(a), "v"(b), "v"(c));   // 0xF0 | (0xCC & 0xAA)
    return d;
}
// ~a & ~(b ^ c): the match mask of a block from one xor and one bitop3 (left to itself the compiler rebuilds
// Eq & Pv from the two xors separately, one instruction more per block)
__device__ __forceinline__ uint32_t eq_mask(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t d;
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x09" : "=v"(d) : "v"(a), "v"(b), "v"(c));   // ~0xF0 & ~(0xCC ^ 0xAA)
    return d;
}
// sign-extended bit k of x (k wave-uniform): 0 or ~0
__device__ __forceinline__ uint32_t bit_mask(uint32_t x, int k) {
    uint32_t d;
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(d) : "v"(x), "s"(k));
    return d;
}

// One sweep over the window [i - wleft, i + w].  Returns 0 when all diagonal checks pass (then best / besti hold the
// minimum of the last column at or below the diagonal and its row), else the first failing row.
//
// Instruction budget (measured on MI355X, tools/ubench_ops: and/or/xor/add/sub/not/bitop3/ashr issue every
// ~2.5 cycles per SIMD, shifts, bfe, alignbit, lshl_or, add3, addc and DPP moves every ~4.3): a Myers block is
// 9 full-rate ops + 3 v_addc_co (the carries are the horizontal deltas entering / leaving the block, kept as lane
// masks in SGPR pairs), and the lane-to-lane hand-off of those deltas is a SCALAR rotate of the two masks -- no
// DPP, no VALU.  The diagonal's D0 bits are collected with one bitop3 per block and a rotating one-hot; text
// elements come from two 32-step bit-plane registers.  Everything rare sits behind one compare (t == t_next).
//
// TRACE: also store, for every step and block, the two words the traceback needs (bv_trace_walk below):
//   word 0 = Eq | ~D0   bit r set: the cell's parent is the diagonal one (seq_aligner.h:164-166: MATCH wins ties)
//   word 1 = Ph (horizontal +1 entering the cell from the left), or the new Pv (vertical +1) when the rows are
//            the reference's b (`swap_roles`): set = the reference's INSERT, clear = its DELETE, for cells whose
//            parent is not diagonal (INSERT is tried first and wins the tie, seq_aligner.h:167-173)
// at tr[((t-1) * NB + nb) * 128 + 2 * lane + word]: one 8-byte store per lane, 512 contiguous bytes per instruction.
// TRACE == 2: instead, checkpoints -- at the start of every 32-step chunk the lane's Pv / Mv words, its superblock and window
// state and the two hand-off masks go to tr (bv_ck_words per chunk, 1/25 of the streamed words); the walk re-runs one
// chunk at a time from its checkpoint into LDS (align_bvtrace.h: bitvec_rerun).
#define PBA_BV_CK_WORDS(nb) ((nb) * 128 + 128)
#define PBA_BV_BAIL 0x40000000                         // bitvec_pass: "not worth finishing in this window" (above any row number)
template <int NB>
__device__ __forceinline__ void bv_ckpt_store(uint32_t *tr, int tb, int lane, const uint32_t *Pv, const uint32_t *Mv, int s_cur,
                                              uint32_t opened, uint64_t hp_last, uint64_t hn_last) {
    uint32_t *ck = tr + (size_t)((tb - 1) >> 5) * PBA_BV_CK_WORDS(NB);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) { ck[nb * 128 + lane] = Pv[nb]; ck[nb * 128 + 64 + lane] = Mv[nb]; }
    ck[NB * 128 + lane] = (uint32_t)s_cur | (opened << 31);
    if (lane < 4)
        ck[NB * 128 + 64 + lane] = lane == 0 ? (uint32_t)hp_last : lane == 1 ? (uint32_t)(hp_last >> 32) : lane == 2 ? (uint32_t)hn_last : (uint32_t)(hn_last >> 32);
}

template <int NB, int TRACE = 0, bool BAIL = false>
__device__ __forceinline__ int bitvec_pass(const PackedFetch &rowsF, int nr, const PackedFetch &colsF, int m, int wleft, int w,
                                           double R, int &best_out, int &besti_out, int &diag_out, uint32_t *fin,
                                           uint32_t *tr = nullptr, bool swap_roles = false, int bail_w = 0) {
    // BAIL and bail_w > 0: give the sweep up (return PBA_BV_BAIL) once the diagonal says where the pair is heading -- at row i >= 1024,
    // (D(i,i) - 4 sqrt(D(i,i))) * m / i > bail_w (four standard deviations of a count of D(i,i) errors below the projection:
    // a pair that will come in under bail_w is not given up by accident; the check runs at every 32nd row): a cost the
    // window could not certify anyway (align_bitvec: the caller re-runs the pair with the
    // reference's band either way; this only decides how many rows the narrow sweep spends finding out).  Never changes a result.
    // rowsF / nr: the longer sequence and how many of its rows are swept (m <= nr <= m + wleft); colsF / m: the shorter
    // one.  Row i sees the columns [i - wleft, i + w] (wleft: the wide side, towards the free end; w: the narrow one).
    constexpr int RB = 32 * NB;                 // rows per superblock
    const int lane = threadIdx.x & (PBA_WAVE - 1);
    m = __builtin_amdgcn_readfirstlane(m); nr = __builtin_amdgcn_readfirstlane(nr);   // wave-uniform by construction
    w = __builtin_amdgcn_readfirstlane(w); wleft = __builtin_amdgcn_readfirstlane(wleft);
    const int S = (nr + RB - 1) / RB;           // superblocks
    const int s_m = (m - 1) / RB;               // superblock, block and bit of row m (the end of the diagonal)
    const int nb_m = ((m - 1) - s_m * RB) >> 5, r_m = (m - 1) & 31;
    const int t1 = m + s_m;                     // step at which cell (m,m) is produced
    const int t_end = m + S - 1;                // step at which the last superblock takes the last column
    (void)nb_m; (void)r_m;

    uint32_t Pv[NB], Mv[NB], Plo[NB], Phi[NB], acc[NB];
    int s_cur = lane;
    int t_evt;         // next step at which this lane's window opens or (after that) has just closed
    // functions of the superblock the lane holds, recomputed where an event needs them (three registers fewer to carry
    // through the step loop: at 64 registers per lane they were what the event handler spilled to scratch memory):
    auto t_close1 = [&]() { return s_cur < S ? min(m, s_cur * RB + RB + w) + s_cur + 1 : INT_MAX; };          // first step after the window
    auto t_hin_end = [&]() { return (s_cur < S && s_cur > 0) ? min(m, s_cur * RB + w) + s_cur : INT_MIN; };   // last step at which the lane above still delivers hout for this lane's column
    auto t_diag0 = [&]() { return (s_cur < S && s_cur * RB < m) ? s_cur * RB + 1 + s_cur : INT_MAX - RB; };   // step at which this lane's first row is on the diagonal
    int t_dstart;      // = t_diag0 until the diagonal has entered this lane's rows
    int t_seg;         // step at which the current 32-row diagonal segment of this lane is complete
    uint32_t opened = 0;   // 0 / 1 in a VGPR (a bool would live in an SGPR pair and be re-merged with exec every step)

    auto open_superblock = [&]() {              // s_cur names the superblock this lane now owns
        const int base_row = s_cur * RB;
        opened = 0;
        if (s_cur < S) {
            const int lo = max(1, base_row + 1 - wleft), hi = min(m, base_row + RB + w);
            t_evt = lo + s_cur;
            (void)hi;
            if (base_row < m) {                 // the diagonal (rows 1..m) crosses this superblock
                t_dstart = base_row + 1 + s_cur;
                t_seg = min(t_dstart + 31, m + s_cur);
            } else { t_dstart = INT_MAX; t_seg = INT_MAX - 1; }
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) load_planes32(rowsF, base_row + 32 * nb, Plo[nb], Phi[nb]);
        } else {                                // nothing left for this lane
            t_evt = INT_MAX; t_dstart = INT_MAX;
            t_seg = INT_MAX - 1;
        }
    };
    open_superblock();
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) { Pv[nb] = ~0u; Mv[nb] = 0u; acc[nb] = 0u; }

    uint32_t wl = 0, wh = 0;                     // text bit planes: bit k = low / high bit of the element of step tb+k
    int score = 0, best = INT_MAX, fail_row = 0;
    // the vertical deltas of the LAST column as this lane's last superblock at or below row m left them (a lane keeps
    // stepping on garbage after its window has closed, so they are put aside at the close); fin_s: that superblock, or -1
    // They live in LDS (fin: PBA_BV_FIN_WORDS(NB) u32 of this wavefront's own: [2*nb][lane] Pv, [2*nb+1][lane] Mv, [2*NB][lane]
    // the superblock): written once per superblock that reaches the last column, read once at the end -- as registers
    // they were live through the whole step loop, and at 64 registers per lane the compiler spilled them to scratch memory.
    fin[2 * NB * PBA_WAVE + lane] = 0xFFFFFFFFu;
    uint32_t dmw = 0;                            // one-hot: bit of the diagonal cell in its block's word (0: not in this lane)
    uint64_t hp_last = ~0ull, hn_last = 0ull;    // lane masks: delta +1 / -1 leaving each lane's last block in the previous step
    // lanes whose first block still receives the lane above's hout (the others see "+1 per column")
    uint64_t valid = __builtin_amdgcn_ballot_w64(1 <= t_hin_end());
    // the one rare-event compare of the step loop
    int t_next = min(min(t_evt, t_seg + 1), min(t_dstart, 1 <= t_hin_end() ? t_hin_end() + 1 : INT_MAX));

    // text planes for the 32 steps starting at the wave-uniform step tb (element of step t is t - s_cur - 1)
    auto load_text = [&](int tb) { load_planes32(colsF, tb - s_cur - 1, wl, wh); };

    // A 32-row diagonal segment ended at step t-1 (row i = t-1-s_cur, or row m): check its rows against
    // seq_aligner.h:185 and carry the diagonal score on.
    // `above`: the score register of the lane above, fetched by the caller with all lanes enabled
    auto segment_done = [&](int t, int above) {
        const int rr = t - 1 - t_diag0(), i = t - 1 - s_cur, cnt = (rr & 31) + 1, i0 = i - cnt, q = rr >> 5;
        if (q == 0) score = s_cur == 0 ? 0 : above;      // D(i0,i0): the lane above finished its rows >= 2 steps ago
        uint32_t dw = 0;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) dw = q == nb ? acc[nb] : dw;
        dw &= 0xFFFFFFFFu >> (32 - cnt);
        // D(i0+k, i0+k) = score + k - popcount(low k bits of dw); non-decreasing in k
        const int end = score + cnt - __builtin_popcount(dw);
        const int ifirst = max(i0 + 1, 11);                                   // rows <= 10 are never checked
        if (i >= ifirst && (double)end > (double)ifirst * R && fail_row == 0) {
            for (int k = ifirst - i0; k <= cnt; ++k) {                        // row by row
                const int d = score + k - __builtin_popcount(dw & (0xFFFFFFFFu >> (32 - k)));
                if ((double)d > (double)(i0 + k) * R) { fail_row = i0 + k; break; }
            }
        }
        if constexpr (BAIL) {
            if (bail_w > 0 && i >= 1024 && fail_row == 0 &&
                ((float)end - 4.0f * __builtin_sqrtf((float)end)) * (float)m > (float)i * (float)bail_w) fail_row = PBA_BV_BAIL;
        }
        score = end;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = 0;     // the next block's word starts clean
        if (i == m) best = score;                        // D(m,m): where the scan down the last column starts
        if (i == m || rr == RB - 1) { t_seg = INT_MAX - 1; dmw = 0; }         // the diagonal leaves this lane's rows
        else { t_seg = min(t_seg + 32, m + s_cur); dmw = 1u; }                // ... or enters the lane's next block at its bit 0
    };

    // rare, divergent: segment finished last step / window opens now / window closed last step / diagonal enters
#define PBA_BV_RARE(ON_EVENT)                                                         \
    if (__builtin_amdgcn_ballot_w64(t == t_next)) {      /* wave-uniform: all lanes enabled for the shuffle */ \
      const int above = __shfl(score, (lane + PBA_WAVE - 1) & (PBA_WAVE - 1), PBA_WAVE); \
      if (t == t_next) {                                                              \
        if (t == t_seg + 1) segment_done(t, above);                                   \
        if (t == t_evt) {                                                             \
            /* a lane that moves on to its next superblock needs that superblock's slice of the text; a lane that \
               opens its first window already got its planes at the start of the chunk (no load, no wait: during  \
               the ramp one lane opens per step, and false candidates are all ramp) */                              \
            if (opened) {                                                             \
                if (s_cur >= s_m) {          /* its window ended with the last column: keep that column */ \
                    fin[2 * NB * PBA_WAVE + lane] = (uint32_t)s_cur;                  \
                    _Pragma("unroll") for (int nb = 0; nb < NB; ++nb) { fin[2 * nb * PBA_WAVE + lane] = Pv[nb]; fin[(2 * nb + 1) * PBA_WAVE + lane] = Mv[nb]; } \
                }                                                                     \
                s_cur += PBA_WAVE; open_superblock(); load_text(t - ((t - 1) & 31));  \
            }                                                                         \
            if (t == t_evt) {                                                         \
                _Pragma("unroll") for (int nb = 0; nb < NB; ++nb) { Pv[nb] = ~0u; Mv[nb] = 0u; } \
                opened = 1;                                                           \
                t_evt = t_close1();                                                   \
            }                                                                         \
        }                                                                             \
        if (t == t_dstart) {                                                          \
            dmw = 1u; t_dstart = INT_MAX;                                             \
            _Pragma("unroll") for (int nb = 0; nb < NB; ++nb) acc[nb] = 0;            \
        }                                                                             \
        { const int he = t_hin_end();                                                 \
          t_next = min(min(t_evt, t_seg + 1), min(t_dstart, t <= he ? he + 1 : INT_MAX)); } \
      }                                                                               \
      valid = __builtin_amdgcn_ballot_w64(t <= t_hin_end());   /* changes only at events: kept as a scalar mask */ \
      if constexpr (TRACE == 1) st_on = ((__builtin_amdgcn_ballot_w64(opened != 0) >> (lane & 48)) & 0xFFFFull) != 0; \
      ON_EVENT;                                                                       \
    }

    // hin of each lane's first block: the lane above's hout of the previous step, rotated one lane up, where that
    // lane was still inside its window; elsewhere the row above the window counts "+1 per column"
#define PBA_BV_HIN()                                                                  \
    uint64_t hp = (((hp_last << 1) | (hp_last >> 63)) & valid) | ~valid;              \
    uint64_t hn = ((hn_last << 1) | (hn_last >> 63)) & valid;

    // one block update (hp / hn: lane masks of the delta entering at the block's top row, replaced by the one leaving)
    // D0 (bit r set iff D(i,j) == D(i-1,j-1)) doubles as Myers' Xv in the two vertical updates (Hyyro's form of
    // the recurrences: wherever D0 and Eq | Mv differ a carry came in from the row above, whose horizontal delta is
    // then -1, and both forms give Pv' = 1, Mv' = 0)
#define PBA_BV_BLOCK(nb, PH_PRE, MH_PRE, D0, EQ)                                     \
    {                                                                                \
        const uint32_t Eq = eq_mask(Plo[nb] ^ clo, Phi[nb], chi);                    \
        EQ = Eq;                                                                     \
        const uint32_t pv = Pv[nb], mv = Mv[nb];                                     \
        uint64_t unused;                                                             \
        const uint32_t sum = addc_mask(Eq & pv, pv, hn, unused);   /* hn as carry-in == Eq |= 1 at the top row */ \
        const uint32_t Xh = (sum ^ pv) | Eq;                                         \
        const uint32_t Ph = mv | ~(Xh | pv);                                         \
        const uint32_t Mh = pv & Xh;                                                 \
        D0 = Xh | mv;                                                                \
        PH_PRE = Ph; MH_PRE = Mh;                                                    \
        const uint32_t Ph2 = addc_mask(Ph, Ph, hp, hp);                              \
        const uint32_t Mh2 = addc_mask(Mh, Mh, hn, hn);                              \
        Pv[nb] = Mh2 | ~(D0 | Ph2);                                                  \
        Mv[nb] = Ph2 & D0;                                                           \
    }

    // TRACE: this lane's slot of step t, block 0, word 0.  Recomputed from t every step: a pointer carried through
    // the loop loses its increment on the text-refill path of the step loop (ROCm 7.2 clang, seen in the ISA)
    uint2 *const tr_lane = (uint2 *)tr + lane;
    (void)tr_lane;
    bool st_on = false;          // TRACE: some lane of this lane's 16-lane group is inside its window (updated at events)
    (void)st_on;
#define PBA_BV_TRP() (tr_lane + (size_t)(t - 1) * (NB * 64))
    // ------------------------------------------------------------------ phase 1: down to cell (m,m)
    // The step loop is cut into chunks of 32 steps (one pair of text planes): the scalar unit is shared by the CU's
    // four SIMDs, so every SALU instruction of the step costs four issue cycles -- the inner loop carries nothing
    // but its counter.  Everything in it is wave-uniform and stays in SGPRs.
    bool failed = false;
    for (int tbv = 1; tbv <= t1; tbv += 32) {
        const int tb = __builtin_amdgcn_readfirstlane(tbv);   // (the early exit below makes the compiler treat tbv as divergent)
        if constexpr (TRACE == 2) bv_ckpt_store<NB>(tr, tb, lane, Pv, Mv, s_cur, opened, hp_last, hn_last);
        load_text(tb);                           // next text planes
        if (failed) break;
        int kend = min(32, t1 - tb + 1);
        for (int k = 0; k < kend; ++k) {
        const int t = tb + k;
        // a segment that just ended may have failed the reference's check: leave right after this step (false
        // candidates die on their first segment, so they cost 33 steps, not the 64 of a poll per chunk)
        PBA_BV_RARE(if (__builtin_amdgcn_ballot_w64(fail_row != 0)) { failed = true; kend = k; });
        const uint32_t clo = bit_mask(wl, k), chi = bit_mask(wh, k);
        PBA_BV_HIN();
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            uint32_t d0, php, mhp, eq;
            PBA_BV_BLOCK(nb, php, mhp, d0, eq);
            (void)php; (void)mhp; (void)eq;
            if constexpr (TRACE == 1) {      // whole 128-byte lines only: a 16-lane group stores when any of its lanes is
                if (st_on)                   // inside its window (masking lane by lane was measured 1.5x SLOWER: partial lines)
                    PBA_BV_TRP()[nb * 64] = make_uint2(eq | ~d0, swap_roles ? Pv[nb] : php);
            }
            // keep the diagonal cell's D0 bit (garbage while the diagonal is in another block: the word is
            // cleared when the diagonal enters)
            acc[nb] = or_of_and(acc[nb], d0, dmw);
        }
        hp_last = hp; hn_last = hn;
        dmw += dmw;                              // next row of the block (full-rate add; segment_done re-arms it every 32 rows)
        }
    }
    int t = t1 + 1;
    if (!failed) {                               // segments that ended in the last step (the one holding row m does)
        const int above = __shfl(score, (lane + PBA_WAVE - 1) & (PBA_WAVE - 1), PBA_WAVE);
        if (t == t_seg + 1) segment_done(t, above);
        failed = __builtin_amdgcn_ballot_w64(fail_row != 0) != 0;
    }
    if (failed) {
        int fr = fail_row ? fail_row : INT_MAX;  // rows fail in increasing order of step: the smallest is the first
        return wave_min_i32(fr);
    }
    t_seg = INT_MAX - 1; t_dstart = INT_MAX;
    t_next = min(t_evt, t <= t_hin_end() ? t_hin_end() + 1 : INT_MAX);

    // ------------------------------------------------------------------ phase 2: the superblocks below row m take the last column
    for (int tb = t1 + 1; tb <= t_end;) {
        const int k0 = (tb - 1) & 31;            // phase 2 starts inside a chunk whose planes are already loaded
        if (k0 == 0) {
            if constexpr (TRACE == 2) bv_ckpt_store<NB>(tr, tb, lane, Pv, Mv, s_cur, opened, hp_last, hn_last);
            load_text(tb);
        }
        const int kend = min(32, k0 + (t_end - tb + 1));
        for (int k = k0; k < kend; ++k) {
        const int t = tb - k0 + k;
        PBA_BV_RARE((void)0);
        const uint32_t clo = bit_mask(wl, k), chi = bit_mask(wh, k);
        PBA_BV_HIN();
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            uint32_t d0, php, mhp, eq;
            PBA_BV_BLOCK(nb, php, mhp, d0, eq);
            (void)d0; (void)eq; (void)php; (void)mhp;
            if constexpr (TRACE == 1) {      // whole 128-byte lines only: a 16-lane group stores when any of its lanes is
                if (st_on)                   // inside its window (masking lane by lane was measured 1.5x SLOWER: partial lines)
                    PBA_BV_TRP()[nb * 64] = make_uint2(eq | ~d0, swap_roles ? Pv[nb] : php);
            }
        }
        hp_last = hp; hn_last = hn;
        }
        tb += kend - k0;
    }
#undef PBA_BV_BLOCK
#undef PBA_BV_TRP
#undef PBA_BV_HIN
#undef PBA_BV_RARE
    // ------------------------------------------------------------------ the free end: first strict minimum down the last column
    // D(i,m) = D(m,m) + the vertical deltas of column m over rows m+1..i (seq_aligner.h:192-211 scans exactly these cells,
    // upward from the diagonal, and keeps the first strict minimum).  A lane whose window is still open holds the
    // column in Pv / Mv, the others put it aside when their window closed.
    if (opened && s_cur >= s_m && s_cur < S) {
        fin[2 * NB * PBA_WAVE + lane] = (uint32_t)s_cur;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) { fin[2 * nb * PBA_WAVE + lane] = Pv[nb]; fin[(2 * nb + 1) * PBA_WAVE + lane] = Mv[nb]; }
    }
    const int fin_s = (int)fin[2 * NB * PBA_WAVE + lane];      // (a lane reads what it wrote itself: no barrier)
    int b_tot[NB], b_min[NB], b_pos[NB];          // per block of this lane's last superblock: sum, lowest prefix sum, its bit
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int row0 = fin_s * RB + nb * 32;    // bit r is row row0 + r + 1
        const uint32_t fpv = fin_s >= 0 ? fin[2 * nb * PBA_WAVE + lane] : 0u, fmv = fin_s >= 0 ? fin[(2 * nb + 1) * PBA_WAVE + lane] : 0u;
        int v = 0, pm = INT_MAX, pp = 0;
        for (int r = 0; r < 32; ++r) {
            const int row = row0 + r + 1;
            if (fin_s >= 0 && row > m && row <= nr) {
                v += (int)((fpv >> r) & 1u) - (int)((fmv >> r) & 1u);
                if (v < pm) { pm = v; pp = r; }
            }
        }
        b_tot[nb] = v; b_min[nb] = pm; b_pos[nb] = pp;
    }
    // v_readlane: the results are wave-uniform and the compiler must know it, or every loop that depends on
    // them (the callers' candidate walks, and through their lengths this function's own step loop) is treated
    // as divergent and its counters and bounds move from SGPRs into VGPRs
    int gbest = __builtin_amdgcn_readlane(best, s_m & (PBA_WAVE - 1)), gi = m, run = gbest;   // D(m,m)
    diag_out = gbest;
    for (int sb = s_m; sb < S; ++sb) {
        const int L = sb & (PBA_WAVE - 1);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int pm = __builtin_amdgcn_readlane(b_min[nb], L);
            if (pm != INT_MAX && run + pm < gbest) {
                gbest = run + pm;
                gi = sb * RB + nb * 32 + __builtin_amdgcn_readlane(b_pos[nb], L) + 1;
            }
            run += __builtin_amdgcn_readlane(b_tot[nb], L);
        }
    }
    best_out = gbest;
    besti_out = gi;
    return 0;
}

// true when the bit-vector kernel can take a pair with this max_dst (else: row sweep)
// window of the certifying ("full band") sweep
__device__ __host__ inline int bv_full_wl(int md) { return md / 2 + 1 < md ? md / 2 + 1 : md; }
__device__ __host__ inline bool bitvec_supports(int max_dst) { return bv_nb_for_span(bv_full_wl(max_dst) + max_dst) != 0; }
// first-pass window for a given max_dst: w columns right of the diagonal, wl left of it
__device__ __host__ inline int bv_first_w(int md) {
    const int w = (md / 2 > (int)((long long)md * PBA_BV_BAND_NUM / PBA_BV_BAND_DEN)
                       ? md / 2 : (int)((long long)md * PBA_BV_BAND_NUM / PBA_BV_BAND_DEN)) + 1;
    return w > md ? md : w;
}
__device__ __host__ inline int bv_first_wl(int md) {
    const int wl = bv_first_w(md) / 2 + 1;
    return wl > md ? md : wl;
}
// The first-pass window of a launch whose lanes hold NB blocks: with NB fixed the number of steps does not depend on the
// window (lanes that would idle fill up), so it is as wide as the ring lets it be -- the whole band for short pairs,
// 18 % instead of 16.9 % divergence certified at 15 kb -- at no cost.
__device__ __host__ inline int bv_pass1_w(int md, int nb) {
    int w = bv_first_w(md);
    const int room = (bv_max_span(nb) - 4) * 2 / 3;      // w + (w/2 + 1) <= span
    if (room > w) w = room;
    return w > md ? md : w;
}
__device__ __host__ inline int bv_pass1_wl(int md, int nb) {
    const int wl = bv_pass1_w(md, nb) / 2 + 1;
    return wl > md ? md : wl;
}
// the verdicts of a sweep over [i - wl, i + w] (header comment): a failure at row fr / a goal minimum `best`
__device__ __forceinline__ bool bv_fail_certified(int fr, double R, int wl, int md) {
    return 2 * wl + 2 > md || (double)fr * R < (double)(2 * wl + 2);      // floor(i*R) <= md - 1 for every row
}
__device__ __forceinline__ bool bv_goal_certified(int best, int wl, int w, int md) {
    return (w >= md && 2 * wl + 1 >= md) || (best <= w && best <= 2 * wl + 1);   // a passing pair has best <= md - 1
}

#define PBA_RC_UNCERTIFIED (-3)   // narrow pass could not certify the goal row: re-run with full_band

// One pair.  NB is chosen by the host from the largest max_dst in the launch (any NB >= the pair's own
// need is valid).  full_band = false runs the narrow first pass and reports PBA_RC_UNCERTIFIED when the
// goal row cannot be certified; the host then re-launches those pairs with full_band = true.
// There is deliberately no device function call in here: everything inlines into the kernel.
// lds: PBA_BV_FIN_WORDS(NB) u32 of this wavefront's own (>= 256 bytes: the m <= 10 corner runs the row sweep in it)
// need_diag: the caller reports D(m,m) (locator.cpp:86): a narrow pass whose window cannot vouch for that cell too
// answers PBA_RC_UNCERTIFIED; without it o.diag is -1 in that case.
// BAIL: give a narrow first pass up early when the pair is heading for a cost its window cannot certify (bitvec_pass:
// bail_w).  For callers whose true pairs are expected beyond the first-pass window (the all-vs-all walk: two noisy reads);
// compiled out elsewhere -- the test costs the step loop three scalar registers, 6 % of k_locate's time at 8 waves per SIMD.
template <int NB, bool BAIL = false>
__device__ __forceinline__ void align_bitvec(const PackedFetch &fa, int la, const PackedFetch &fb, int lb, double R,
                                             int maxn, int maxm, bool full_band, uint16_t *lds, int lds_cells,
                                             AlnOut &o, bool need_diag = false) {
    aln_params(la, lb, R, o);
    const int len_a = o.len_a, len_b = o.len_b, md = o.max_dst;
    if (maxn > 0 && (len_a >= maxn + maxm || md >= maxm)) return;      // seq_aligner.h:104-107
    const bool a_rows = len_a > len_b;          // the DP is symmetric under transposition: rows = the LONGER side
    const int m = a_rows ? len_b : len_a, n = a_rows ? len_a : len_b;
    if (m <= 10) {                              // no diagonal check ever fires: plain DP on a <= 23-cell band
        align_rowsweep(fa, la, fb, lb, R, maxn, maxm, lds, lds_cells, o);
        return;
    }
    // w: towards the free end (below the diagonal now that the longer side runs down the rows), wl: the other side
    const int w = full_band ? md : bv_pass1_w(md, NB), wl = full_band ? bv_full_wl(md) : bv_pass1_wl(md, NB);
    if (wl + w > bv_max_span(NB)) { o.rc = -2; return; }   // host sizes NB for the launch: cannot happen
    const PackedFetch rowsF = a_rows ? fa : fb, colsF = a_rows ? fb : fa;
    int best = 0, besti = 0, diag = 0;
    // A narrow window certifies a goal minimum <= w only: a pair whose diagonal is heading past that (two noisy reads of an
    // all-vs-all run differ by ~28 %, against a window sized for a read against its genome) is given up after ~1000 rows
    // instead of being swept to the end for an answer nobody can use.
    const int bail_w = (BAIL && !full_band && w < md) ? w : 0;
    const int fr = bitvec_pass<NB, 0, BAIL>(rowsF, min(n, m + w), colsF, m, w, wl, R, best, besti, diag, (uint32_t *)lds, nullptr, false, bail_w);
    if (fr) {
        bool gave_up = false;
        if constexpr (BAIL) gave_up = fr == PBA_BV_BAIL;
        if (!gave_up && bv_fail_certified(fr, R, wl, md)) o.fail_row = fr; else o.rc = PBA_RC_UNCERTIFIED;
        return;
    }
    if (!bv_goal_certified(best, wl, w, md)) { o.rc = PBA_RC_UNCERTIFIED; return; }   // header comment
    // D(m,m) is a cell like the goal cells: exact when it is small enough for its path to lie inside the windows
    const bool diag_ok = bv_goal_certified(diag, wl, w, md);
    if (need_diag && !diag_ok) { o.rc = PBA_RC_UNCERTIFIED; return; }
    o.cost = best;
    o.diag = diag_ok ? diag : -1;
    o.matlen_a = a_rows ? besti : m;
    o.matlen_b = a_rows ? m : besti;
    o.rc = ((double)o.matlen_b < (double)len_b * (1.0 - R)) ? -1 : o.matlen_b;   // seq_aligner.h:114
}

#endif
