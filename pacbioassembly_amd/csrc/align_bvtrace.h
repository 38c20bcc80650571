// align_bvtrace.h -- edit scripts (seq_aligner<>::find_path, /root/reference/src/seq_aligner.h:214-233) on the
// bit-vector array: the HBM-bound tier of the aligner.
//
// The forward pass is bitvec_pass<NB, TRACE = true> (align_bitvec.h): the same sweep as the score-only kernel,
// which also streams two words per (step, block, lane) into a per-wavefront scratch area -- 2 bits per DP
// cell, 512 contiguous bytes per store instruction.  The parent of a cell, decided by the reference with strict
// comparisons in the order MATCH, INSERT, DELETE (seq_aligner.h:164-173), is a function of the cell's delta bits:
//   * Eq                     -> diagonal (cost D(i-1,j-1) is a lower bound of the cell, MATCH is tried first)
//   * !Eq, D0 = 0            -> diagonal (a substitution; the cell is D(i-1,j-1)+1 and nothing is strictly cheaper)
//   * !Eq, D0 = 1            -> the cell equals D(i-1,j-1) < the diagonal candidate; INSERT (from D(i,j-1)) if the
//                               horizontal delta into the cell is +1, else DELETE -- INSERT wins when both are
//                               optimal because DELETE must be strictly cheaper than INSERT
// With the rows holding the reference's b (len_a > len_b: the array works on the transposed matrix) the
// reference's "horizontal" is the array's vertical: the stored bit is then the new Pv instead of Ph.
// Every cell of the traced path costs <= final_cost <= w (a certified pass) or < max_dst (the reference-band
// pass), so its three predecessors are cells where the array's values equal the reference's and
// the path never leaves the processed windows.
//
// The walk runs on the same wavefront right after the forward pass: scalar control, 64 steps of one
// (lane, block) column of the scratch area fetched per round trip (one step per lane), ops collected 64 at a
// time and written goal-first to a temporary, then copied out reversed.
#ifndef PBA_ALIGN_BVTRACE_H
#define PBA_ALIGN_BVTRACE_H

#include "align_bitvec.h"

// scratch words a traced pass needs: steps * NB blocks * 2 words * 64 lanes (m columns, the longer side swept down to
// row min(n, m + w); the last superblock takes the last column at step m + S - 1)
__device__ __host__ inline uint64_t bv_trace_words(int nb, int m, int n, int w) {
    const int rb = 32 * nb;
    const long long nr = (long long)m + w < n ? (long long)m + w : n;
    const long long S = (nr + rb - 1) / rb;
    const long long t_end = m + S - 1;
    return (uint64_t)(t_end > 0 ? t_end : 0) * (uint64_t)nb * 128u;
}

// ... and of the checkpoint form (TRACE == 2): one checkpoint per 32-step chunk
__device__ __host__ inline uint64_t bv_ck_words(int nb, int m, int n, int w) {
    const int rb = 32 * nb;
    const long long nr = (long long)m + w < n ? (long long)m + w : n;
    const long long S = (nr + rb - 1) / rb;
    const long long t_end = m + S - 1;
    return (uint64_t)((t_end > 0 ? t_end : 0) / 32 + 2) * (uint64_t)PBA_BV_CK_WORDS(nb);
}
#define PBA_BV_TILE_LANES 4                     // lanes of a re-run chunk whose words are kept (the path's lane and the three above it)
#define PBA_BV_TILE_WORDS(nb) (32 * PBA_BV_TILE_LANES * (nb))     // uint2 per wavefront

// The walk reads what OTHER LANES OF THE SAME WAVEFRONT stored: same CU, same vector L1 (write-through, shared
// by the CU), so plain loads behind a workgroup-scope fence (the stores have completed) are coherent.  Agent-scope
// (sc1) loads are not the tool here: they are served past this XCD's L2, which still holds the lines dirty.
__device__ __forceinline__ void wave_mem_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }
__device__ __forceinline__ uint2 ld_coherent2(const uint32_t *p) {
    const unsigned long long v = __hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
}
__device__ __forceinline__ uint32_t ld_coherent_u8(const uint8_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// A sink receives the path goal-first: put(op, i, j) = the reference's op at cell (i, j) of ITS matrix (a along i).
// Everything passed to put() is wave-uniform.

// ops into memory: 64 at a time (lane (k & 63) keeps op k until the group is full), goal-first into a temporary,
// copied out reversed by finish()
struct OpSink {
    uint8_t *tmp;
    int k;
    uint32_t pend;
    __device__ __forceinline__ void put(int op, int, int) {
        const int lane = threadIdx.x & (PBA_WAVE - 1);
        if (lane == (k & (PBA_WAVE - 1))) pend = (uint32_t)op;
        ++k;
        if ((k & (PBA_WAVE - 1)) == 0) tmp[k - PBA_WAVE + lane] = (uint8_t)pend;
    }
    // n MATCH ops down a diagonal from cell (i, j): op k + r is the MATCH at (i - r, j - r)
    __device__ __forceinline__ void put_run(int n, int, int) {
        const int lane = threadIdx.x & (PBA_WAVE - 1);
        while (n > 0) {
            const int at = k & (PBA_WAVE - 1), take = min(n, PBA_WAVE - at);
            if (lane >= at && lane < at + take) pend = 1u;
            k += take; n -= take;
            if ((k & (PBA_WAVE - 1)) == 0) tmp[k - PBA_WAVE + lane] = (uint8_t)pend;
        }
    }
    // ops_out receives min(k, ops_cap) ops in the reference's order (origin first); returns nedit
    __device__ __forceinline__ int finish(uint8_t *ops_out, uint64_t ops_cap) {
        const int lane = threadIdx.x & (PBA_WAVE - 1);
        if (lane < (k & (PBA_WAVE - 1))) tmp[(k & ~(PBA_WAVE - 1)) + lane] = (uint8_t)pend;
        wave_mem_fence();
        for (int x = lane; x < k; x += PBA_WAVE)
            if ((uint64_t)x < ops_cap) ops_out[x] = (uint8_t)ld_coherent_u8(tmp + (k - 1 - x));
        return k;
    }
};

// (ri, cj): the goal cell in array coordinates (rows = the shorter sequence).  Returns with the ops of the path
// down to a border cell in `sink`; ri / cj hold that border cell.
template <int NB, class Sink>
__device__ __forceinline__ void bv_trace_walk(const uint32_t *tr, int &ri, int &cj, bool swap_roles, Sink &sink) {
    constexpr int RB = 32 * NB;
    const int lane = threadIdx.x & (PBA_WAVE - 1);
    while (ri > 0 && cj > 0) {
        const int s = (ri - 1) / RB, ln = s & (PBA_WAVE - 1), nb = ((ri - 1) - s * RB) >> 5;
        const int row_lo = s * RB + nb * 32;               // this word holds rows row_lo+1 .. row_lo+32
        const int tt = cj + s - lane;                      // lane d looks at the step d before the cell's
        uint32_t wm = 0, wh = 0;
        if (tt >= 1) {
            const uint2 v = ld_coherent2(tr + ((size_t)(tt - 1) * NB + nb) * 128 + 2 * ln);
            wm = v.x;
            wh = v.y;
        }
        int d = 0;
        do {
            const int bit = (ri - 1) & 31;
            const uint32_t mbit = ((uint32_t)__builtin_amdgcn_readlane((int)wm, d) >> bit) & 1u;
            const uint32_t hbit = ((uint32_t)__builtin_amdgcn_readlane((int)wh, d) >> bit) & 1u;
            const int oi = swap_roles ? cj : ri, oj = swap_roles ? ri : cj;   // the cell in the reference's coordinates
            if (mbit) {                                    // MATCH: (i-1, j-1)
                sink.put(1, oi, oj); --ri; --cj; ++d;
            } else {
                sink.put(hbit ? 2 : 3, oi, oj);            // INSERT : DELETE
                if ((hbit != 0) != swap_roles) { --cj; ++d; }   // the array's column moves
                else --ri;                                      // the array's row moves (same step, next bit down)
            }
        } while (ri > row_lo && cj > 0 && d < PBA_WAVE);
    }
}

// ---- checkpoint form -------------------------------------------------------------------------------------------------
// Re-run chunk `chunk` (steps 32*chunk + 1 .. t_last) of the sweep bitvec_pass<NB, 2> made, from its checkpoint: the same
// steps, the same window opens and closes (a lane's superblock and window state at a step are functions of the geometry;
// the checkpoint holds where the lane stood), none of the diagonal bookkeeping -- and the two traceback words of every
// (step, block) of the lanes lb, lb-1, lb-2, lb-3 (mod 64) go to `tile` in LDS:
//   tile[((t - t0) * PBA_BV_TILE_LANES + ((lb - lane) & 63)) * NB + nb] = { Eq | ~D0,  Ph (or the new Pv when swap_roles) }
template <int NB>
__device__ __forceinline__ void bitvec_rerun(const PackedFetch &rowsF, int nr, const PackedFetch &colsF, int m, int wleft, int w,
                                             const uint32_t *ck_base, int chunk, int t_last, int lb, uint2 *tile, bool swap_roles) {
    constexpr int RB = 32 * NB;
    const int lane = threadIdx.x & (PBA_WAVE - 1);
    m = __builtin_amdgcn_readfirstlane(m); nr = __builtin_amdgcn_readfirstlane(nr);
    w = __builtin_amdgcn_readfirstlane(w); wleft = __builtin_amdgcn_readfirstlane(wleft);
    chunk = __builtin_amdgcn_readfirstlane(chunk); t_last = __builtin_amdgcn_readfirstlane(t_last);
    const int S = (nr + RB - 1) / RB;
    const int t0 = 32 * chunk + 1;
    const uint32_t *ck = ck_base + (size_t)chunk * PBA_BV_CK_WORDS(NB);
    uint32_t Pv[NB], Mv[NB], Plo[NB], Phi[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        Pv[nb] = __hip_atomic_load(ck + nb * 128 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        Mv[nb] = __hip_atomic_load(ck + nb * 128 + 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    const uint32_t meta = __hip_atomic_load(ck + NB * 128 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const uint32_t mk = __hip_atomic_load(ck + NB * 128 + 64 + (lane & 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    uint64_t hp_last = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)mk, 0) | (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)mk, 1) << 32;
    uint64_t hn_last = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)mk, 2) | (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)mk, 3) << 32;
    int s_cur = (int)(meta & 0x7FFFFFFFu);
    uint32_t opened = meta >> 31;
    int t_evt, t_close1 = 0, t_hin_end;
    auto open_superblock = [&]() {              // bitvec_pass: open_superblock, without the diagonal
        const int base_row = s_cur * RB;
        if (s_cur < S) {
            const int lo = max(1, base_row + 1 - wleft), hi = min(m, base_row + RB + w);
            t_evt = lo + s_cur;
            t_close1 = hi + s_cur + 1;
            t_hin_end = s_cur > 0 ? min(m, base_row + w) + s_cur : INT_MIN;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) load_planes32(rowsF, base_row + 32 * nb, Plo[nb], Phi[nb]);
        } else {
            t_evt = INT_MAX; t_close1 = INT_MAX; t_hin_end = INT_MIN;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) { Plo[nb] = 0; Phi[nb] = 0; }
        }
    };
    open_superblock();
    if (opened) t_evt = t_close1;               // inside its window: the next event is the close
    uint32_t wl = 0, wh = 0;
    auto load_text = [&]() { load_planes32(colsF, t0 - s_cur - 1, wl, wh); };
    load_text();
    uint64_t valid = __builtin_amdgcn_ballot_w64(t0 <= t_hin_end);
    int t_next = min(t_evt, t0 <= t_hin_end ? t_hin_end + 1 : INT_MAX);
    const int slot = (lb - lane) & (PBA_WAVE - 1);
    const bool keep = slot < PBA_BV_TILE_LANES;
    const int kend = min(32, t_last - t0 + 1);
    for (int k = 0; k < kend; ++k) {
        const int t = t0 + k;
        if (__builtin_amdgcn_ballot_w64(t == t_next)) {
            if (t == t_next) {
                if (t == t_evt) {
                    if (opened) { s_cur += PBA_WAVE; opened = 0; open_superblock(); load_text(); }
                    if (t == t_evt) {
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) { Pv[nb] = ~0u; Mv[nb] = 0u; }
                        opened = 1;
                        t_evt = t_close1;
                    }
                }
                t_next = min(t_evt, t <= t_hin_end ? t_hin_end + 1 : INT_MAX);
            }
            valid = __builtin_amdgcn_ballot_w64(t <= t_hin_end);
        }
        const uint32_t clo = bit_mask(wl, k), chi = bit_mask(wh, k);
        uint64_t hp = (((hp_last << 1) | (hp_last >> 63)) & valid) | ~valid;
        uint64_t hn = ((hn_last << 1) | (hn_last >> 63)) & valid;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const uint32_t Eq = eq_mask(Plo[nb] ^ clo, Phi[nb], chi);
            const uint32_t pv = Pv[nb], mv = Mv[nb];
            uint64_t unused;
            const uint32_t sum = addc_mask(Eq & pv, pv, hn, unused);
            const uint32_t Xh = (sum ^ pv) | Eq;
            const uint32_t Ph = mv | ~(Xh | pv);
            const uint32_t Mh = pv & Xh;
            const uint32_t D0 = Xh | mv;
            const uint32_t Ph2 = addc_mask(Ph, Ph, hp, hp);
            const uint32_t Mh2 = addc_mask(Mh, Mh, hn, hn);
            Pv[nb] = Mh2 | ~(D0 | Ph2);
            Mv[nb] = Ph2 & D0;
            if (keep) tile[(k * PBA_BV_TILE_LANES + slot) * NB + nb] = make_uint2(Eq | ~D0, swap_roles ? Pv[nb] : Ph);
        }
        hp_last = hp; hn_last = hn;
    }
}

// The walk of the checkpoint form: like bv_trace_walk, but the words of the step a cell sits at come from the LDS tile of
// its chunk, re-run on demand.  The step of the path never grows on the way back (a diagonal or horizontal move costs one
// or two steps, a vertical one none), so every chunk is re-run once -- unless the path runs vertically through more
// lanes than the tile keeps, which re-runs the chunk with the tile moved up.
template <int NB, class Sink>
__device__ __forceinline__ void bv_trace_walk_ck(const PackedFetch &rowsF, int nr, const PackedFetch &colsF, int m, int wleft, int w,
                                                 const uint32_t *ck_base, int t_end, uint2 *tile, int &ri, int &cj, bool swap_roles,
                                                 Sink &sink) {
    constexpr int RB = 32 * NB;
    const int lane = threadIdx.x & (PBA_WAVE - 1);
    int have_chunk = -1, have_lb = 0;
    while (ri > 0 && cj > 0) {
        const int s = (ri - 1) / RB, ln = s & (PBA_WAVE - 1), nb = ((ri - 1) - s * RB) >> 5;
        const int row_lo = s * RB + nb * 32;               // this word holds rows row_lo+1 .. row_lo+32
        const int tcur = cj + s;                           // the step the cell was computed at
        const int chunk = (tcur - 1) >> 5, t0 = 32 * chunk + 1;
        if (chunk != have_chunk || ((have_lb - ln) & (PBA_WAVE - 1)) >= PBA_BV_TILE_LANES) {
            __builtin_amdgcn_wave_barrier();               // (earlier reads of the tile are done)
            bitvec_rerun<NB>(rowsF, nr, colsF, m, wleft, w, ck_base, chunk, min(t0 + 31, t_end), ln, tile, swap_roles);
            __builtin_amdgcn_wave_barrier();
            have_chunk = chunk; have_lb = ln;
        }
        const int slot = (have_lb - ln) & (PBA_WAVE - 1);
        const int tt = tcur - lane;                        // lane d looks at the step d before the cell's
        uint32_t wm = 0, wh = 0;
        if (tt >= t0) {
            const uint2 v = tile[((tt - t0) * PBA_BV_TILE_LANES + slot) * NB + nb];
            wm = v.x;
            wh = v.y;
        }
        const int dmax = tcur - t0;                        // lanes 0 .. dmax hold words of this chunk
        int d = 0;
        do {
            const int bit = (ri - 1) & 31;
            const int oi = swap_roles ? cj : ri, oj = swap_roles ? ri : cj;   // the cell in the reference's coordinates
            // the diagonal from here: the cell r moves on sits one step earlier and one bit lower, i.e. in lane d + r at bit
            // `bit - r` -- one ballot tells how many MATCH moves follow each other (six on average at 15 % error), and they
            // go to the sink together
            const int r = lane - d;
            const bool on_diag = r >= 0 && r <= bit && lane <= dmax && r < cj;
            const uint64_t mm = __builtin_amdgcn_ballot_w64(on_diag && ((wm >> (bit - (on_diag ? r : 0))) & 1u)) >> d;
            const int run = (int)__builtin_ctzll(~mm | (1ull << 63));
            if (run) {                                     // MATCH x run: (i-1, j-1) each
                sink.put_run(run, oi, oj); ri -= run; cj -= run; d += run;
            } else {
                const uint32_t hbit = ((uint32_t)__builtin_amdgcn_readlane((int)wh, d) >> bit) & 1u;
                sink.put(hbit ? 2 : 3, oi, oj);            // INSERT : DELETE
                if ((hbit != 0) != swap_roles) { --cj; ++d; }   // the array's column moves
                else --ri;                                      // the array's row moves (same step, next bit down)
            }
        } while (ri > row_lo && cj > 0 && d <= dmax);
    }
}

// One pair with its path.  full_band = false sweeps the narrow first-pass window and answers PBA_RC_UNCERTIFIED
// (nothing reaches the sink) when its verdict cannot be certified; the host re-launches those pairs with
// full_band = true, like the score-only kernels do.  min_matlen_a: the path is walked only when
// matlen_a >= min_matlen_a (ref_seq::try_align's OVERLAP_MIN gate, ref_seq.h:265; 0 for plain scripts).
// scratch: cap_words u32 of this wavefront's own.  Returns true when the path went to the sink.
// CK: the checkpoint form (scratch holds bv_ck_words, the walk re-runs chunks into `tile`: PBA_BV_TILE_WORDS(NB) uint2 of
// LDS of this wavefront's own); else every step's words are streamed to scratch (bv_trace_words).
template <int NB, bool CK, class Sink>
__device__ __forceinline__ bool align_bitvec_trace(const PackedFetch &fa, int la, const PackedFetch &fb, int lb, double R,
                                                   int maxn, int maxm, bool full_band, uint16_t *lds, int lds_cells,
                                                   uint32_t *scratch, uint64_t cap_words, int min_matlen_a, Sink &sink,
                                                   AlnOut &o, uint2 *tile = nullptr) {
    aln_params(la, lb, R, o);
    const int len_a = o.len_a, len_b = o.len_b, md = o.max_dst;
    if (maxn > 0 && (len_a >= maxn + maxm || md >= maxm)) return false;      // seq_aligner.h:104-107
    const bool a_rows = len_a > len_b;          // rows = the longer side (align_bitvec.h)
    const bool swap = !a_rows;                  // the array's rows are the reference's b
    const int m = a_rows ? len_b : len_a, n = a_rows ? len_a : len_b;
    if (m <= 10) {
        // the row sweep's corner (align_bitvec.h): one parent code per band cell in the scratch area
        const int W = 2 * md + 1;
        if ((uint64_t)(len_a + 1) * (uint64_t)W > cap_words * 4) { o.rc = -2; return false; }
        uint8_t *par = (uint8_t *)scratch;
        align_rowsweep(fa, la, fb, lb, R, maxn, maxm, lds, lds_cells, o, par);
        if (o.rc < 0 || o.matlen_a < min_matlen_a) return false;
        wave_mem_fence();
        int i = o.matlen_a, j = o.matlen_b;
        while (i > 0 || j > 0) {
            // init_cell: (i,0) has parent DELETE, (0,j) INSERT (seq_aligner.h:140-147)
            const int src = j == 0 ? 3 : (i == 0 ? 2 : (int)ld_coherent_u8(par + (size_t)i * W + (j - i + md)));
            const int u = __builtin_amdgcn_readfirstlane(src);
            sink.put(u, i, j);
            if (u == 1) { --i; --j; } else if (u == 2) --j; else --i;
        }
        return true;
    }
    const PackedFetch rowsF = a_rows ? fa : fb, colsF = a_rows ? fb : fa;
    const int w = full_band ? md : bv_pass1_w(md, NB), wl = full_band ? bv_full_wl(md) : bv_pass1_wl(md, NB);
    int best = 0, besti = 0, diag = 0;
    if ((CK ? bv_ck_words(NB, m, n, w) : bv_trace_words(NB, m, n, w)) > cap_words || wl + w > bv_max_span(NB)) { o.rc = -2; return false; }   // host sizes both
    const int nr = min(n, m + w);
    const int fr = bitvec_pass<NB, CK ? 2 : 1>(rowsF, nr, colsF, m, w, wl, R, best, besti, diag, (uint32_t *)lds, scratch, swap);
    if (fr) {
        if (bv_fail_certified(fr, R, wl, md)) o.fail_row = fr; else o.rc = PBA_RC_UNCERTIFIED;
        return false;
    }
    if (!bv_goal_certified(best, wl, w, md)) { o.rc = PBA_RC_UNCERTIFIED; return false; }
    o.cost = best;
    o.diag = bv_goal_certified(diag, wl, w, md) ? diag : -1;
    o.matlen_a = a_rows ? besti : m;
    o.matlen_b = a_rows ? m : besti;
    o.rc = ((double)o.matlen_b < (double)len_b * (1.0 - R)) ? -1 : o.matlen_b;   // seq_aligner.h:114
    if (o.rc < 0 || o.matlen_a < min_matlen_a) return false;
    wave_mem_fence();                                   // the walk reads what other lanes stored
    int ri = besti, cj = m;                             // the goal cell in array coordinates: (row of the minimum, last column)
    if constexpr (CK) bv_trace_walk_ck<NB>(rowsF, nr, colsF, m, w, wl, scratch, m + (nr + 32 * NB - 1) / (32 * NB) - 1, tile, ri, cj, swap, sink);
    else bv_trace_walk<NB>(scratch, ri, cj, swap, sink);
    // border cells (init_cell): row 0 of the reference's matrix is INSERTs, column 0 DELETEs
    if (swap) {
        for (; ri > 0; --ri) sink.put(2, 0, ri);        // array rows are the reference's j
        for (; cj > 0; --cj) sink.put(3, cj, 0);
    } else {
        for (; ri > 0; --ri) sink.put(3, ri, 0);
        for (; cj > 0; --cj) sink.put(2, 0, cj);
    }
    return true;
}

#endif
