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

// One pair with its path.  full_band = false sweeps the narrow first-pass window and answers PBA_RC_UNCERTIFIED
// (nothing reaches the sink) when its verdict cannot be certified; the host re-launches those pairs with
// full_band = true, like the score-only kernels do.  min_matlen_a: the path is walked only when
// matlen_a >= min_matlen_a (ref_seq::try_align's OVERLAP_MIN gate, ref_seq.h:265; 0 for plain scripts).
// scratch: cap_words u32 of this wavefront's own.  Returns true when the path went to the sink.
template <int NB, class Sink>
__device__ __forceinline__ bool align_bitvec_trace(const PackedFetch &fa, int la, const PackedFetch &fb, int lb, double R,
                                                   int maxn, int maxm, bool full_band, uint16_t *lds, int lds_cells,
                                                   uint32_t *scratch, uint64_t cap_words, int min_matlen_a, Sink &sink,
                                                   AlnOut &o) {
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
    if (bv_trace_words(NB, m, n, w) > cap_words || wl + w > bv_max_span(NB)) { o.rc = -2; return false; }   // host sizes both
    const int fr = bitvec_pass<NB, true>(rowsF, min(n, m + w), colsF, m, w, wl, R, best, besti, diag, scratch, swap);
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
    bv_trace_walk<NB>(scratch, ri, cj, swap, sink);
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
