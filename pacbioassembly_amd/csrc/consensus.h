// consensus.h -- vote boxes, elect and evolve of the unlocked ref_seq (/root/reference/src/ref_seq.h:25-188,
// 207-242, 317-362) on the device.  SURVEY.md 8f-3.
//
// Layout: like ref_seq::txt_buf the object spans 3*max_len positions with the origin (`beg`) at max_len, and
// the vote box of a text position sits at the same index -- the reference's std::list<vote_box> is positional
// (one box per character of [pre, post)), so append / prepend are range fills and need no shifting.
//   sel[p], sup[p] : 4 x u16 counters (base_vote::acgt, ref_seq.h:49) packed in one u64, A in bits 15:0
//   tot[p]         : vote_box::total
// Votes are commutative, so elect runs one wavefront per edit script with atomics; the counters are bumped with
// 32-bit atomics on the half-word's dword (they wrap like the reference's unsigned short only below 65 536 votes
// per counter: far beyond any coverage this code meets).
// evolve (ref_seq.h:317-349) is a stream compaction: every box yields 0..2 output boxes (itself if its selection
// wins a majority, a box split from its suppliment if that does), a deleted box's selection is absorbed by the
// suppliment of the last box kept before it.  One workgroup sweeps the list in chunks of 1024 with an LDS scan:
// it runs once per assembly round on at most a few million boxes.
#ifndef PBA_CONSENSUS_H
#define PBA_CONSENSUS_H

#include "dev_common.h"

struct ConsDev {
    unsigned long long *sel, *sup;   // 4 x u16 each
    int *tot;
    char *txt;
};

__device__ __forceinline__ int cons_c2i(int ch) { return ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : 3; }   // dna_seq.h:21
__device__ __forceinline__ int cons_max4(unsigned long long v) {          // base_vote::max_vote, ref_seq.h:88-91
    const int a = (int)(v & 0xFFFF), b = (int)((v >> 16) & 0xFFFF), c = (int)((v >> 32) & 0xFFFF), d = (int)(v >> 48);
    return max(max(a, b), max(c, d));
}
__device__ __forceinline__ char cons_winner(unsigned long long v) {        // base_vote::winner, ref_seq.h:96-100
    const int mv = cons_max4(v);
    return mv == (int)(v & 0xFFFF) ? 'A' : (mv == (int)((v >> 16) & 0xFFFF) ? 'C' : (mv == (int)((v >> 32) & 0xFFFF) ? 'G' : 'T'));
}
__device__ __forceinline__ void cons_bump(unsigned long long *box, int c, unsigned n) {   // counter c += n
    atomicAdd((unsigned *)box + (c >> 1), n << (16 * (c & 1)));
}

// boxes [first, first+len) <- vote_box(text[k], weight) (ref_seq.h:118: selection(c, n), total(1)); text copied too
static __global__ void __launch_bounds__(256)
k_cons_fill(ConsDev C, int first, int len, const char *text, int weight) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= len) return;
    const char ch = text[k];
    C.txt[first + k] = ch;
    C.sel[first + k] = (unsigned long long)(unsigned)(weight & 0xFFFF) << (16 * cons_c2i(ch));
    C.sup[first + k] = 0ull;
    C.tot[first + k] = 1;
}

// apply_edits (ref_seq.h:25-41) for n scripts, one wavefront each.  pos is relative to `beg`.
static __global__ void __launch_bounds__(PBA_WAVE)
k_cons_elect(ConsDev C, int beg, int pre, int post, uint32_t n, const int *pos, const uint8_t *fwd, const uint8_t *ops,
             const char *vals, const unsigned long long *ops_off, const int *nedit) {
    const uint32_t q = blockIdx.x;
    if (q >= n) return;
    const int lane = threadIdx.x;
    const bool forward = fwd[q] != 0;
    const int it0 = beg + pos[q], ne = nedit[q];
    const uint8_t *o = ops + ops_off[q];
    const char *v = vals + ops_off[q];
    int done = 0;                                   // boxes consumed by the ops of earlier chunks
    for (int k0 = 0; k0 < ne; k0 += PBA_WAVE) {
        const int k = k0 + lane;
        const int op = k < ne ? o[k] : 0;
        const bool adv = op == 1 || op == 3;        // MATCH and DELETE move the iterator
        const unsigned long long m = __builtin_amdgcn_ballot_w64(adv);
        const int before = done + __builtin_popcountll(m & ((1ull << lane) - 1ull));
        int at = forward ? it0 + before : it0 - before;
        if (op == 2 && forward) at -= 1;            // INSERT: `--it; supply; ++it` forward, the box itself backward
        if (op != 0 && at >= pre && at < post) {
            if (op == 1) { cons_bump(C.sel + at, cons_c2i(v[k]), 1u); atomicAdd(C.tot + at, 1); }   // select
            else if (op == 3) atomicAdd(C.tot + at, 1);                                                // ignore
            else cons_bump(C.sup + at, cons_c2i(v[k]), 1u);                                            // supply
        }
        done += __builtin_popcountll(m);
    }
}

// Votes straight from the traceback walk (align_bvtrace.h), no script in memory: a sink that turns the reference's
// op at cell (i, j) into its vote.  The cell says it all: a MATCH / DELETE at (i, j) consumes a-element i-1, i.e. the
// box `it0 +/- (i-1)`; an INSERT at (i, j) comes after i consumed elements, i.e. supplies box `it0 + i - 1` forward
// (`--it; supply; ++it`) or `it0 - i` backward (apply_edits, ref_seq.h:25-41); the value is b-element j-1.
// 64 votes are gathered (one per lane) and applied together, each lane fetching its own b element.
struct VoteSink {
    ConsDev C;
    int it0, pre, post;         // box of the start position; the live range of boxes
    bool fwd;
    PackedFetch fb;             // the segment (b)
    int k;
    int p_at, p_j;              // this lane's pending vote: box, b element + 1 (0: none)
    uint32_t p_op;
    __device__ __forceinline__ void apply() {
        if (p_op && p_at >= pre && p_at < post) {
            if (p_op == 3) atomicAdd(C.tot + p_at, 1);                                  // ignore
            else {
                const int c = fb(p_j - 1);                                              // 2-bit code == C2I of the base
                if (p_op == 1) { cons_bump(C.sel + p_at, c, 1u); atomicAdd(C.tot + p_at, 1); }   // select
                else cons_bump(C.sup + p_at, c, 1u);                                    // supply
            }
        }
        p_op = 0;
    }
    __device__ __forceinline__ void put(int op, int i, int j) {
        const int lane = threadIdx.x & (PBA_WAVE - 1);
        const int at = op == 2 ? (fwd ? it0 + i - 1 : it0 - i) : (fwd ? it0 + (i - 1) : it0 - (i - 1));
        if (lane == (k & (PBA_WAVE - 1))) { p_op = (uint32_t)op; p_at = at; p_j = j; }
        ++k;
        if ((k & (PBA_WAVE - 1)) == 0) apply();
    }
    // n MATCH votes down a diagonal from cell (i, j): vote k + r is the MATCH at (i - r, j - r)
    __device__ __forceinline__ void put_run(int n, int i, int j) {
        const int lane = threadIdx.x & (PBA_WAVE - 1);
        while (n > 0) {
            const int at = k & (PBA_WAVE - 1), take = min(n, PBA_WAVE - at);
            if (lane >= at && lane < at + take) {
                const int r = lane - at;
                p_op = 1u; p_at = fwd ? it0 + (i - r - 1) : it0 - (i - r - 1); p_j = j - r;
            }
            k += take; n -= take; i -= take; j -= take;
            if ((k & (PBA_WAVE - 1)) == 0) apply();
        }
    }
    __device__ __forceinline__ void finish() { apply(); }
};

// evolve: [pre, post) of `in` -> boxes and text from index `beg` of `out`; *n_out = boxes kept
static __global__ void __launch_bounds__(1024)
k_cons_evolve(ConsDev in, ConsDev out, int pre, int post, int beg, int *n_out) {
    __shared__ int scan[1024];
    __shared__ int carry;
    const int tid = threadIdx.x;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int base = pre; base < post; base += 1024) {
        const int i = base + tid;
        const bool live = i < post;
        unsigned long long sel = 0, sup = 0;
        int tot = 0;
        if (live) { sel = in.sel[i]; sup = in.sup[i]; tot = in.tot[i]; }
        const bool S = live && (double)cons_max4(sup) > 0.5 * (double)tot;    // has_supply(0.5), ref_seq.h:327
        const bool V = live && (double)cons_max4(sel) > 0.5 * (double)tot;    // is_valid(0.5),   ref_seq.h:336
        const int cnt = (V ? 1 : 0) + (S ? 1 : 0);
        scan[tid] = cnt;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {       // inclusive scan (Hillis-Steele; once per round, not hot)
            const int t = tid >= d ? scan[tid - d] : 0;
            __syncthreads();
            scan[tid] += t;
            __syncthreads();
        }
        const int off = beg + carry + scan[tid] - cnt;          // first output slot of this box
        if (V) {                                                // kept; its suppliment moves out if it was split
            out.sel[off] = sel; out.sup[off] = S ? 0ull : sup; out.tot[off] = tot;
            out.txt[off] = cons_winner(sel);
        }
        if (S) {                                                // the split box, inserted right behind (ref_seq.h:328-334)
            out.sel[off + (V ? 1 : 0)] = sup; out.sup[off + (V ? 1 : 0)] = 0ull; out.tot[off + (V ? 1 : 0)] = tot;
            out.txt[off + (V ? 1 : 0)] = cons_winner(sup);
        }
        __threadfence();
        __syncthreads();                                        // every box of this chunk is in place ...
        if (live && !V && off - 1 >= beg) {                     // ... before a deleted one is absorbed by its predecessor
            for (int c = 0; c < 4; ++c) {
                const unsigned add = (unsigned)((sel >> (16 * c)) & 0xFFFF);
                if (add) cons_bump(out.sup + (off - 1), c, add);
            }
        }
        __syncthreads();
        if (tid == 1023) carry += scan[1023];
        __syncthreads();
    }
    if (tid == 0) *n_out = carry;
}

#endif
