// overlap.h -- all-vs-all overlap driver: every read is a target in the
// reference role (ref_seq::get_seedmap index, /root/reference/src/ref_seq.h:291-311) and every other read a
// query walked like one read of a locked spaced_seed round (spaced_seed.cpp:262-298, 420-437): 2*max_trial
// probes (forward at j, backward at slen-j-16), hits in the seedmap's list order, first success per
// (target, query) wins.  All successful (target, query) pairs are reported.
//
// Inverted form: the PROBES are indexed (2*max_trial keys per read: small, and the object a
// multi-GPU run all-gathers), the targets' positions are scanned against that table:
//   k_probe_emit : one probe entry (key << 32 | probe id) per (query, j, direction), key 0 dropped
//   k_pt_*       : the entries into a direct-address probe table (bucket offsets + entries + one presence bit per key)
//   k_ovl_scan   : (bit-vector kernels) one workgroup per target, 16 positions per thread from one 8-byte load; every
//                  candidate -- (position, matching probe of another read) -- gets a lane, which has the probe's record
//                  (query, its length, the 32 elements its alignment starts with: 16 bytes beside the probe id in the
//                  table), takes the target's 32 elements at the hit and runs the FIRST 32 ROWS of the reference's sweep
//                  right there (prefilter.h).  98.6 % of the candidates fail the reference's diagonal check in those
//                  rows: they are pairs the reference aligned and dropped -- counted, never written.  The survivors
//                  (candidate = query << 23 | (2j + backward) << 16 | ordinal) go to the target's slice.
//   (per-target sort of the survivors in LDS; 64-bit order = query, then j, forward before backward, then the
//    seedmap's list order -- exactly the order spaced_seed tries them in)
//   k_ovl_walk   : persistent wavefronts walk a target's survivors, align until the first success per query;
//                  (target, query) runs whose narrow-window verdict is not certified (or whose narrow pass was given up: it
//                  was heading past what the window certifies) are parked and resumed by later launches in wider rings
//   k_ovl_after  : per success, the candidates of its (target, query) run the reference never got to (it stops at the first
//                  success): pairs = candidates - those
//   k_ovl_count / k_ovl_fill : the row-sweep kernel's form (the cross-check: no prefilter, no windows): every candidate is
//                  written, sorted and walked
#ifndef PBA_OVERLAP_H
#define PBA_OVERLAP_H

#include "align_bitvec.h"
#include "align_rowsweep.h"
#include "dev_common.h"
#include "pba.h"
#include "prefilter.h"
#include "seed_index.h"

#define PBA_OVL_ORD_BITS 16
#define PBA_OVL_JD_BITS 7
#define PBA_OVL_Q_SHIFT (PBA_OVL_ORD_BITS + PBA_OVL_JD_BITS)

// probe id = query * t2 + 2*j + (backward ? 1 : 0); t2 = 2 * max_trial
static __global__ void __launch_bounds__(256)
k_probe_emit(SeqSetDev Rd, uint32_t q_lo, uint32_t n_queries, uint32_t t2, uint32_t mask, uint64_t *out,
             unsigned long long cap, unsigned long long *counter) {
    const uint64_t lid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (lid >= (uint64_t)n_queries * t2) return;
    const uint64_t gid = lid + (uint64_t)q_lo * t2;                       // global probe id
    const uint32_t q = (uint32_t)(gid / t2), jd = (uint32_t)(gid % t2);
    const int slen = (int)Rd.len[q];
    const int j = (int)(jd >> 1);
    const int pos = (jd & 1) ? slen - j - 16 : j;                       // spaced_seed.cpp:426
    if (pos < 0 || pos + 16 > slen) return;
    const uint32_t key = window_key(Rd.packed + Rd.off[q], (uint32_t)pos, (uint32_t)slen) & mask;
    if (!key) return;                                                   // a zero key is never in a seedmap
    const unsigned long long o = atomicAdd(counter, 1ull);
    if (o < cap) out[o] = (uint64_t)key << 32 | (uint32_t)gid;
}

// visiting order of ref_seq::get_seedmap for a sequence of `len` bases
struct HeadTail {
    int nhead, visited, tail_top;
    __device__ __forceinline__ HeadTail(int len) {
        const int nh = min(len - 16, 20000), nt = min(len - 20000 - 16, 20000);
        nhead = nh > 0 ? nh : 0;
        visited = nhead + (nt > 0 ? nt : 0);
        tail_top = len - 16;
    }
    __device__ __forceinline__ int pos_of(int ord) const { return ord < nhead ? ord : tail_top - (ord - nhead); }
};

// ---------------------------------------------------------------------------------------------------------------
// The probe table: a direct-address CSR over the probe keys ("LDS-staged hash buckets" of the north star, sized for HBM:
// the table of a million reads is 64 MB of bucket offsets + 1 GB of entry records and stays resident).
//   bucket(key) = the key's care bits gathered into one number when the mask has <= PBA_PT_MAX_BITS of them (every key
//                 its own bucket: a lookup is ONE 8-byte load of two adjacent offsets, no search, no key compare), else a
//                 multiplicative hash into 2^PBA_PT_MAX_BITS buckets (HASHED: the entry's key is kept and compared)
//   start[b] = first entry of bucket b in pid[] / prec[] (and pkey[] when HASHED); start[b + 1] ends it
//   pid[i]   = query << 7 | (2j + backward)
//   prec[i]  = { pid[i], length of the query, low / high bit plane of the 32 ELEMENTS the probe's alignment starts with }:
//              element r = base j + r of the query forward, base slen - j - 1 - r backward (spaced_seed.cpp:274-275), i.e.
//              the columns of the first 32 rows of the reference's sweep.  With it a candidate's first diagonal checks
//              need nothing of the query but this record, which arrives coalesced with its bucket -- the scan used to
//              write every candidate and a later pass fetched one scattered line of the query per candidate.
//              Filled from the read set on first use (k_pt_ctx): the exchanged object stays the 8-byte probe entry.
//   presence = one bit per bucket (2 MB for the weight-12 masks of seeds.txt: resident in every XCD's L2), consulted
//              first -- with 20 k reads 12 of 13 positions stop there, with a million reads none do
#define PBA_PT_MAX_BITS 26
struct ProbeTab {
    uint32_t *start, *pid, *pkey, *presence;
    uint4 *prec;
    uint32_t mask, mv[5];
    int bits;
};
template <bool HASHED>
__device__ __forceinline__ uint32_t pt_bucket(const ProbeTab &T, uint32_t key) {
    if (HASHED) return (key * 0x9E3779B1u) >> (32 - T.bits);
    uint32_t x = key;                                   // compress(key, mask): Hacker's Delight 7-4, move masks from the host
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const uint32_t t = x & T.mv[i];
        x = (x ^ t) | (t >> (1 << i));
    }
    return x;
}

// entries (key << 32 | probe id; all-ones = padding of an all-gathered buffer) -> bucket sizes at start[b + 1], presence bits
template <bool HASHED>
static __global__ void __launch_bounds__(256)
k_pt_count(const uint64_t *in, uint64_t n, ProbeTab T) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t e = in[i];
    if (e == ~0ull) return;
    const uint32_t b = pt_bucket<HASHED>(T, (uint32_t)(e >> 32));
    atomicAdd(T.start + b + 1, 1u);
    atomicOr(T.presence + (b >> 5), 1u << (b & 31));
}
// ... and, once start[] has been scanned, the entries into their buckets (cursor = a copy of start)
template <bool HASHED>
static __global__ void __launch_bounds__(256)
k_pt_fill(const uint64_t *in, uint64_t n, ProbeTab T, uint32_t *cursor, uint32_t t2) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t e = in[i];
    if (e == ~0ull) return;
    const uint32_t key = (uint32_t)(e >> 32), gid = (uint32_t)e;
    const uint32_t slot = atomicAdd(cursor + pt_bucket<HASHED>(T, key), 1u);
    T.pid[slot] = (gid / t2) << PBA_OVL_JD_BITS | (gid % t2);
    if (HASHED) T.pkey[slot] = key;
}

// prec[i] once the entries are in place and the read set is at hand (ProbeTab above)
static __global__ void __launch_bounds__(256)
k_pt_ctx(ProbeTab T, SeqSetDev Rd, uint32_t n_entries) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_entries) return;
    const uint32_t pe = T.pid[i], q = pe >> PBA_OVL_JD_BITS, jd = pe & ((1u << PBA_OVL_JD_BITS) - 1);
    const int slen = (int)Rd.len[q], j = (int)(jd >> 1);
    const bool fwd = (jd & 1u) == 0;
    const int s_off = fwd ? j : slen - j - 1;                           // spaced_seed.cpp:274 (pos + 15 backward, pos = slen - j - 16)
    uint32_t lo = 0, hi = 0;
    if (slen - j >= PBA_PRE_ROWS) load_planes32(fetch_of(Rd, q, s_off, fwd ? 1 : -1), 0, lo, hi);   // (shorter: the prefilter does not apply)
    T.prec[i] = make_uint4(pe, (uint32_t)slen, lo, hi);
}

// ---------------------------------------------------------------------------------------------------------------
// The scan: one workgroup per target, every thread takes 16 consecutive positions from ONE 8-byte load of packed bases
// (0.25 B per position, the algorithmic read) and looks their keys up with all 16 loads of a stage in flight at once --
// the scan is bound by the rate at which the memory system serves random lines, not by arithmetic.
//   k_ovl_scan  : the form of the bit-vector kernels: every candidate gets a lane, its first 32 rows run there, only the
//                 survivors are written (below)
//   k_ovl_count : the row-sweep form, pass 1: candidates a target will produce at most (probe entries in the buckets its keys
//                 reach) -> its slice of the candidate array.  One presence bit and one offset pair per position.
//   k_ovl_fill  : pass 2: candidate = query << 23 | (2j + backward) << 16 | ordinal for every entry of another read (and,
//                 HASHED, of the same key); the target's own probes and foreign keys leave all-ones slots, which sort to the
//                 end of the slice.  valid[tl] = real candidates.
// Positions that found entries ("runs") are compacted into LDS per wavefront: the 16-positions-per-thread walk stays
// convergent and the emission is balanced whatever the hit density (0.08 per position at 20 k reads, 3.8 at a million).
#define PBA_OVL_PPT 16                                         // positions per thread and step
#define PBA_OVL_WAVES 4                                        // wavefronts per workgroup
#ifndef PBA_OVL_HALF
#define PBA_OVL_HALF 8                                         // positions per thread whose runs are compacted together (tuning hook; the scan
#endif                                                         // of 400 k reads: 16 -> 347 ms, 8 -> 296 ms, 4 -> 300 ms: LDS per workgroup, so occupancy)
#define PBA_OVL_RUNS (PBA_WAVE * PBA_OVL_HALF)                 // runs a wavefront can meet in one round

struct TargetWalk {          // visiting order of ref_seq::get_seedmap as a function of the position
    int nhead, tail_lo, tail_top, nchunks;
    __device__ __forceinline__ TargetWalk(int len) {
        const HeadTail ht(len);
        nhead = ht.nhead; tail_top = ht.tail_top;
        tail_lo = tail_top - (ht.visited - ht.nhead) + 1;
        nchunks = ht.visited > 0 ? (tail_top >> 4) + 1 : 0;  // positions 0 .. len - 16
    }
    __device__ __forceinline__ int ord_of(int pos) const {   // -1: not visited (ref_seq.h:297-308)
        return pos < nhead ? pos : (pos >= tail_lo && pos <= tail_top ? nhead + (tail_top - pos) : -1);
    }
};

template <bool HASHED>
static __global__ void __launch_bounds__(PBA_WAVE * PBA_OVL_WAVES)
k_ovl_count(ProbeTab T, SeqSetDev Rd, uint32_t t_lo, uint32_t n_targets, uint32_t *slice) {
    __shared__ uint32_t wsum[PBA_OVL_WAVES];
    const uint32_t tl = blockIdx.x;
    const int len = (int)Rd.len[t_lo + tl];
    const uint8_t *seq = Rd.packed + Rd.off[t_lo + tl];
    const TargetWalk tw(len);
    uint32_t sum = 0;
    for (int c = (int)threadIdx.x; c < tw.nchunks; c += PBA_WAVE * PBA_OVL_WAVES) {
        const uint64_t be = chunk_bits(seq, (uint32_t)c);
        uint32_t b[PBA_OVL_PPT], pw[PBA_OVL_PPT];
#pragma unroll
        for (int k = 0; k < PBA_OVL_PPT; ++k) {
            const uint32_t key = __builtin_bswap32((uint32_t)((be << (2 * k)) >> 32)) & T.mask;
            const bool ok = key != 0u && tw.ord_of(16 * c + k) >= 0;          // ref_seq.h:300,307
            b[k] = ok ? pt_bucket<HASHED>(T, key) : 0xFFFFFFFFu;
            pw[k] = T.presence[ok ? b[k] >> 5 : 0u];
        }
#pragma unroll
        for (int k = 0; k < PBA_OVL_PPT; ++k) {
            const bool hit = b[k] != 0xFFFFFFFFu && ((pw[k] >> (b[k] & 31)) & 1u);
            uint2 se;                                                   // the bucket's offset pair in one 8-byte load
            __builtin_memcpy(&se, T.start + (hit ? b[k] : 0u), 8);
            sum += hit ? se.y - se.x : 0u;
        }
    }
#pragma unroll
    for (int d = 1; d < PBA_WAVE; d <<= 1) sum += __shfl_xor(sum, d, PBA_WAVE);
    if ((threadIdx.x & (PBA_WAVE - 1)) == 0) wsum[threadIdx.x / PBA_WAVE] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t s = 0;
        for (int w = 0; w < PBA_OVL_WAVES; ++w) s += wsum[w];
        slice[tl] = s;
    }
}

// the runs of one round of a wavefront's step, compacted into its part of LDS: first entry, first slot (relative to the round's
// reservation; r_rel[R] = the round's total), ordinal of the position, key (HASHED)
#define PBA_OVL_RUN_LISTS(HASHEDV)                                                         \
    __shared__ uint32_t r_s0[PBA_OVL_WAVES][PBA_OVL_RUNS];                                 \
    __shared__ uint32_t r_rel[PBA_OVL_WAVES][PBA_OVL_RUNS + 1];                            \
    __shared__ uint16_t r_ord[PBA_OVL_WAVES][PBA_OVL_RUNS];                                \
    __shared__ uint32_t r_key[(HASHEDV) ? PBA_OVL_WAVES : 1][(HASHEDV) ? PBA_OVL_RUNS : 1]

// One step of a thread: its 16 positions' keys -> presence -> bucket extents (s0[k], n[k]; n = 0: nothing there)
#define PBA_OVL_LOOKUP16()                                                                                     \
    uint32_t b[PBA_OVL_PPT], pw[PBA_OVL_PPT], key[PBA_OVL_PPT];                                                \
    _Pragma("unroll") for (int k = 0; k < PBA_OVL_PPT; ++k) {                                                  \
        key[k] = __builtin_bswap32((uint32_t)((be << (2 * k)) >> 32)) & T.mask;                                \
        const bool ok = live && key[k] != 0u && tw.ord_of(16 * c + k) >= 0;          /* ref_seq.h:300,307 */   \
        b[k] = ok ? pt_bucket<HASHED>(T, key[k]) : 0xFFFFFFFFu;                                                \
        pw[k] = T.presence[ok ? b[k] >> 5 : 0u];                                                               \
    }                                                                                                          \
    uint32_t s0[PBA_OVL_PPT], n[PBA_OVL_PPT];                                                                  \
    _Pragma("unroll") for (int k = 0; k < PBA_OVL_PPT; ++k) {                                                  \
        const bool hit = b[k] != 0xFFFFFFFFu && ((pw[k] >> (b[k] & 31)) & 1u);                                 \
        uint2 se;                                                   /* the bucket's offset pair in one 8-byte load */ \
        __builtin_memcpy(&se, T.start + (hit ? b[k] : 0u), 8);                                                 \
        s0[k] = se.x; n[k] = hit ? se.y - se.x : 0u;                                                           \
    }

// ... and the runs of positions [h0, h0 + PBA_OVL_HALF) of every lane into the wavefront's lists: R runs, Tot entries
// (both wave-uniform); `continue`s the enclosing loop when there is none
#define PBA_OVL_COMPACT()                                                                                      \
    uint32_t tot_h = 0, nruns_h = 0;                                                                           \
    _Pragma("unroll") for (int k = h0; k < h0 + PBA_OVL_HALF; ++k) { tot_h += n[k]; nruns_h += n[k] != 0u; }   \
    uint32_t run_at = nruns_h, slot_at = tot_h;        /* exclusive prefix over the lanes */                   \
    _Pragma("unroll") for (int d = 1; d < PBA_WAVE; d <<= 1) {                                                 \
        const uint32_t a_ = __shfl_up(run_at, d, PBA_WAVE), s_ = __shfl_up(slot_at, d, PBA_WAVE);              \
        if (lane >= d) { run_at += a_; slot_at += s_; }                                                        \
    }                                                                                                          \
    const uint32_t R = (uint32_t)__builtin_amdgcn_readlane((int)run_at, PBA_WAVE - 1);                         \
    const uint32_t Tot = (uint32_t)__builtin_amdgcn_readlane((int)slot_at, PBA_WAVE - 1);                      \
    if (R == 0) continue;                                                                                      \
    run_at -= nruns_h; slot_at -= tot_h;                                                                       \
    _Pragma("unroll") for (int k = h0; k < h0 + PBA_OVL_HALF; ++k) {                                           \
        if (n[k]) {                                                                                            \
            r_s0[w][run_at] = s0[k]; r_rel[w][run_at] = slot_at; r_ord[w][run_at] = (uint16_t)tw.ord_of(16 * c + k); \
            if (HASHED) r_key[w][run_at] = key[k];                                                             \
            ++run_at; slot_at += n[k];                                                                         \
        }                                                                                                      \
    }                                                                                                          \
    if (lane == PBA_WAVE - 1) r_rel[w][R] = Tot;                                                               \
    __builtin_amdgcn_wave_barrier();                 /* the lists are this wavefront's own: LDS keeps its order */

template <bool HASHED>
static __global__ void __launch_bounds__(PBA_WAVE * PBA_OVL_WAVES)
k_ovl_fill(ProbeTab T, SeqSetDev Rd, uint32_t t_lo, uint32_t n_targets, const uint32_t *cand_off, uint64_t *cand, uint32_t *valid) {
    PBA_OVL_RUN_LISTS(HASHED);
    __shared__ uint32_t cursor, nvalid;
    const uint32_t tl = blockIdx.x, t = t_lo + tl;
    const int lane = threadIdx.x & (PBA_WAVE - 1), w = threadIdx.x / PBA_WAVE;
    if (threadIdx.x == 0) { cursor = 0; nvalid = 0; }
    __syncthreads();
    const int len = (int)Rd.len[t];
    const uint8_t *seq = Rd.packed + Rd.off[t];
    const TargetWalk tw(len);
    uint64_t *out = cand + cand_off[tl];
    uint32_t myvalid = 0;
    // every wavefront runs the same number of steps (the ballots and shuffles below need all its lanes; a lane past the
    // end just has nothing to contribute)
    const int steps = (tw.nchunks + PBA_WAVE * PBA_OVL_WAVES - 1) / (PBA_WAVE * PBA_OVL_WAVES);
    for (int st = 0; st < steps; ++st) {
        const int c = st * PBA_WAVE * PBA_OVL_WAVES + (int)threadIdx.x;
        const bool live = c < tw.nchunks;
        const uint64_t be = live ? chunk_bits(seq, (uint32_t)c) : 0ull;
        PBA_OVL_LOOKUP16();
#pragma unroll
        for (int h0 = 0; h0 < PBA_OVL_PPT; h0 += PBA_OVL_HALF) {
            PBA_OVL_COMPACT();
            // (every lane calls the atomic, lane 0 adds: the lane-0-only form is what ROCm 7.2's clang miscompiles in loops,
            // tools/ubench_queue.hip)
            const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane((int)atomicAdd(&cursor, lane == 0 ? Tot : 0u));
            for (uint32_t r = (uint32_t)lane; r < R; r += PBA_WAVE) {           // one run per lane
                const uint32_t e0 = r_s0[w][r], rel = r_rel[w][r], cnt = r_rel[w][r + 1] - rel;
                const uint64_t lowbits = (uint64_t)r_ord[w][r];
                const uint32_t want = HASHED ? r_key[w][r] : 0u;
                uint64_t *o = out + base + rel;
                for (uint32_t h = 0; h < cnt; ++h) {
                    const uint32_t pe = T.pid[e0 + h];
                    const uint32_t q = pe >> PBA_OVL_JD_BITS;
                    bool ok = q != t;
                    if (HASHED) ok = ok && T.pkey[e0 + h] == want;
                    o[h] = ok ? ((uint64_t)q << PBA_OVL_Q_SHIFT | (uint64_t)(pe & ((1u << PBA_OVL_JD_BITS) - 1)) << PBA_OVL_ORD_BITS | lowbits)
                              : ~0ull;
                    myvalid += ok;
                }
            }
            __builtin_amdgcn_wave_barrier();                             // before the next round overwrites the lists
        }
    }
#pragma unroll
    for (int d = 1; d < PBA_WAVE; d <<= 1) myvalid += __shfl_xor(myvalid, d, PBA_WAVE);
    if (lane == 0 && myvalid) atomicAdd(&nvalid, myvalid);
    __syncthreads();
    if (threadIdx.x == 0) valid[tl] = nvalid;
}

struct OvlCfg {
    double R;
    int overlap_min;
    int row_cap;      // u16 cells of LDS per wavefront
    uint32_t t2;
    uint32_t chunk;   // work items a wavefront of the walk takes at a time when there are >= 2^20 of them (0: PBA_OVL_CHUNK)
    int fused;        // the lists hold survivors of the scan's first 32 rows only (k_ovl_scan); pairs are counted by the scan and k_ovl_after
};

// ---------------------------------------------------------------------------------------------------------------
// The scan of the bit-vector kernels.  Same walk over the target's positions, same runs -- but every candidate of a round
// gets a LANE (the wavefront takes 64 consecutive slots of the concatenated runs at a time, a lane finds the run of its slot
// in the prefix sums that are in LDS anyway: whole-line loads of the records whatever the run lengths) and the lane settles
// it as far as 32 rows can: decode (spaced_seed.cpp:274-286, ref_seq.h:282-286), the OVERLAP_MIN gate, the target's 32
// elements at the hit against the 32 elements of the probe's record through one Myers block (prefilter.h: exactly the
// reference's diagonal checks of rows 11..32, seq_aligner.h:185).  A candidate that fails is a pair the reference aligned
// and dropped at that row: it is COUNTED and never exists in memory.  A survivor (1.4 % at R = 0.30, plus every true
// overlap) is written to the target's slice.  What the reference would not have aligned at all -- candidates of a (target,
// query) run behind that run's first success -- is taken off the count afterwards, per success (k_ovl_after).
//   surv_off == nullptr: nothing is written (the first range of a table: how much room a target needs, from a sample of its targets)
//   slice_cap          : slots a target may write (equal room from an earlier range's census, or its exact need)
//   needed[tl]         : survivors the target produced, written or not (> slice_cap: the host runs the range again, exact)
//   totals[0] += seed matches (entries of other reads under the same key), totals[1] += candidates past the OVERLAP_MIN
//                 gate (= pairs the reference's loop tries if no run ever succeeded)
// Before this form the scan wrote all 57 G candidates of a million reads (8 B each), a second pass fetched each with one
// scattered 64-byte line of its query to do these 32 rows, and two more passes packed the 4 % worth keeping: scan + packing
// + sort 2.57 s of a 5.4 s run; each false candidate now costs its share of one coalesced record line and ~9 instructions.
#ifndef PBA_SCAN_OCC
#define PBA_SCAN_OCC 1            // waves per SIMD the register allocator must leave room for (tuning hook)
#endif
template <bool HASHED>
static __global__ void __launch_bounds__(PBA_WAVE * PBA_OVL_WAVES, PBA_SCAN_OCC)
k_ovl_scan(ProbeTab T, SeqSetDev Rd, uint32_t t_lo, uint32_t t_stride, const uint32_t *surv_off, uint64_t *surv, uint32_t slice_cap,
           uint32_t *needed, OvlCfg cfg, PreChecks pre_c, unsigned long long *totals) {
    PBA_OVL_RUN_LISTS(HASHED);
    __shared__ uint32_t r_mark[PBA_OVL_WAVES][2];
    __shared__ uint32_t cursor, s_cand, s_ok;
    extern __shared__ __align__(16) uint32_t s_planes[];           // the target's bit planes: pairs 0 .. len / 32 + 1 (the host sizes it)
    const uint32_t tl = blockIdx.x * t_stride, t = t_lo + tl;      // (t_stride > 1: a census over every t_stride-th target, needed[] by workgroup)
    const int lane = threadIdx.x & (PBA_WAVE - 1), w = threadIdx.x / PBA_WAVE;
    if (threadIdx.x == 0) { cursor = 0; s_cand = 0; s_ok = 0; }
    const int len = (int)Rd.len[t];
    const uint8_t *seq = Rd.packed + Rd.off[t];
    {   // every candidate's rows come from here: 16 bytes at a random place of the target per candidate
        const uint32_t *gp = Rd.plane + 2 * Rd.poff[t];
        const int nw = 2 * (len / 32 + 2);
        for (int k = (int)threadIdx.x; k < nw; k += PBA_WAVE * PBA_OVL_WAVES) s_planes[k] = gp[k];
    }
    __syncthreads();
    const TargetWalk tw(len);
    const HeadTail ht(len);
    uint64_t *out = surv_off ? surv + surv_off[tl] : nullptr;
    uint32_t ncand = 0, nok = 0;
    const int steps = (tw.nchunks + PBA_WAVE * PBA_OVL_WAVES - 1) / (PBA_WAVE * PBA_OVL_WAVES);
    for (int st = 0; st < steps; ++st) {
        const int c = st * PBA_WAVE * PBA_OVL_WAVES + (int)threadIdx.x;
        const bool live = c < tw.nchunks;
        const uint64_t be = live ? chunk_bits(seq, (uint32_t)c) : 0ull;
        PBA_OVL_LOOKUP16();
#pragma unroll
        for (int h0 = 0; h0 < PBA_OVL_PPT; h0 += PBA_OVL_HALF) {
            PBA_OVL_COMPACT();
            // 64 consecutive slots of the concatenated runs per iteration.  Which run a slot belongs to: `ra` is the run of
            // slot0 (wave-uniform, carried along); lane l looks at the start of run ra + 1 + l -- at most 64 runs can begin in 64
            // slots -- and marks it in a 64-bit mask in LDS; a lane's run is ra + the marks at or below its slot.  (A binary
            // search per lane over the prefix sums cost ten dependent LDS round trips per iteration.)
            // The records of the NEXT 64 slots are requested before this iteration's 32 rows are swept.
            uint32_t ra = 0;
            auto locate = [&](uint32_t slot0, uint32_t &run, uint32_t &ra_next) {
                const uint32_t nxt = ra + 1u + (uint32_t)lane;
                const uint32_t d = (nxt <= R ? r_rel[w][nxt] : 0xFFFFFFFFu) - slot0;      // >= 1: run ra holds slot0
                if (lane == 0) { r_mark[w][0] = 0u; r_mark[w][1] = 0u; }
                __builtin_amdgcn_wave_barrier();
                if (d < PBA_WAVE) atomicOr(&r_mark[w][d >> 5], 1u << (d & 31u));
                __builtin_amdgcn_wave_barrier();
                const uint64_t M = (uint64_t)r_mark[w][1] << 32 | r_mark[w][0];
                run = ra + (uint32_t)__builtin_popcountll(M & ((2ull << lane) - 1ull));
                ra_next = ra + (uint32_t)__builtin_popcountll(M) + (__builtin_amdgcn_ballot_w64(d == PBA_WAVE) ? 1u : 0u);
                __builtin_amdgcn_wave_barrier();
            };
            uint32_t run_n, ra_n;
            locate(0u, run_n, ra_n);
            uint32_t e_n = r_s0[w][min(run_n, R - 1)] + ((uint32_t)lane < Tot ? (uint32_t)lane - r_rel[w][min(run_n, R - 1)] : 0u);
            if ((uint32_t)lane >= Tot) { run_n = R - 1; e_n = r_s0[w][R - 1]; }           // (a lane without a slot reads its wavefront's last run's first record)
            uint4 rec_n = T.prec[e_n];
            for (uint32_t slot0 = 0; slot0 < Tot; slot0 += PBA_WAVE) {
                const uint32_t slot = slot0 + (uint32_t)lane;
                const bool have = slot < Tot;
                const uint32_t lo_r = run_n, e = e_n;
                const uint4 rec = rec_n;
                ra = ra_n;
                if (slot0 + PBA_WAVE < Tot) {                            // the next iteration's records, in flight under this one's arithmetic
                    locate(slot0 + PBA_WAVE, run_n, ra_n);
                    const uint32_t s2 = slot + PBA_WAVE;
                    if (s2 < Tot) e_n = r_s0[w][run_n] + (s2 - r_rel[w][run_n]);
                    else { run_n = R - 1; e_n = r_s0[w][R - 1]; }
                    rec_n = T.prec[e_n];
                }
                const uint32_t q = rec.x >> PBA_OVL_JD_BITS, jd = rec.x & ((1u << PBA_OVL_JD_BITS) - 1);
                bool valid = have && q != t;
                if (HASHED) valid = valid && T.pkey[e] == r_key[w][lo_r];
                const int j = (int)(jd >> 1);
                const bool fwd = (jd & 1u) == 0;
                const int s_len = (int)rec.y - j;                        // spaced_seed.cpp:274-275: slen - j in both directions
                const bool ok = valid && s_len >= cfg.overlap_min;       // spaced_seed.cpp:280
                const uint32_t ord = r_ord[w][lo_r];
                const int hit = ht.pos_of((int)ord);
                const int r_off = fwd ? hit : hit + 15;                  // spaced_seed.cpp:285
                const int r_len = fwd ? len - r_off : r_off + 1;         // ref_seq.h:284-285
                bool failed = false;
                // (the prefilter applies when both clipped lengths reach 32, prefilter32_applies: with max_dst >= 1 that is
                // r_len >= 32 and s_len >= 32, seq_aligner.h:94-102)
                if (ok && r_len >= PBA_PRE_ROWS && s_len >= PBA_PRE_ROWS) {
                    // rows: the target's 32 elements from its hit, out of its planes in LDS (load_planes32, written out so that
                    // the direction is a select at the end: left to itself the compiler made two copies of everything below,
                    // one per direction, and a wavefront with both directions in it ran both)
                    const int idx = fwd ? r_off : r_off - 31;             // lowest base index of the 32, in memory order
                    const uint32_t *pp = s_planes + 2 * (idx >> 5);
                    const uint32_t sh = (uint32_t)idx & 31u;
                    const uint32_t lo32 = __builtin_amdgcn_alignbit(pp[2], pp[0], sh), hi32 = __builtin_amdgcn_alignbit(pp[3], pp[1], sh);
                    const uint32_t rlo = __builtin_bitreverse32(lo32), rhi = __builtin_bitreverse32(hi32);
                    const uint32_t alo = fwd ? lo32 : rlo, ahi = fwd ? hi32 : rhi;
#ifdef PBA_SCAN_NOPRE                                                                 // (timing experiment: what the scan costs without its 32 rows)
                    failed = ((alo ^ rec.z) & (ahi ^ rec.w)) != 0x12345u;
#else
                    failed = prefilter32_fails(alo, ahi, rec.z, rec.w, pre_c);       // columns: the probe's record
#endif
                }
                const bool survivor = ok && !failed;
                ncand += valid; nok += ok;
                const uint64_t sm = __builtin_amdgcn_ballot_w64(survivor);
                if (sm) {
                    // (every lane calls the atomic, lane 0 adds: tools/ubench_queue.hip)
                    const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane(
                        (int)atomicAdd(&cursor, lane == 0 ? (uint32_t)__builtin_popcountll(sm) : 0u));
                    const uint32_t at = base + (uint32_t)__builtin_popcountll(sm & ((1ull << lane) - 1ull));
                    if (survivor && out && at < slice_cap)
                        out[at] = (uint64_t)q << PBA_OVL_Q_SHIFT | (uint64_t)jd << PBA_OVL_ORD_BITS | (uint64_t)ord;
                }
            }
            __builtin_amdgcn_wave_barrier();                             // before the next round overwrites the lists
        }
    }
#pragma unroll
    for (int d = 1; d < PBA_WAVE; d <<= 1) { ncand += __shfl_xor(ncand, d, PBA_WAVE); nok += __shfl_xor(nok, d, PBA_WAVE); }
    if (lane == 0) { atomicAdd(&s_cand, ncand); atomicAdd(&s_ok, nok); }
    __syncthreads();
    if (threadIdx.x == 0) {
        needed[blockIdx.x] = cursor;
        atomicAdd(&totals[0], (unsigned long long)s_cand);
        atomicAdd(&totals[1], (unsigned long long)s_ok);
    }
}

// A target whose slice outgrows one LDS sort (a million reads put 57 000 candidates on a target) is cut into pieces of
// consecutive queries first: PBA_OVL_SUB fine buckets (bucket = umulhi(q, 2^32 * SUB / n_reads), monotone in q; the
// all-ones slots go last), neighbours merged greedily into pieces that fit, the slice scattered piece by piece into a
// second buffer -- the pieces in order are the list in query order, and each is then sorted on its own back into the
// candidate array (k_seg_sort).  A piece that still does not fit (one query with tens of thousands of candidates on one
// target: tandem repeats) is reported for the global bitonic pass.
#define PBA_OVL_SUB 256
typedef SegRef OvlPiece;       // {off, n}: a piece of a target's slice
static __global__ void __launch_bounds__(1024)
k_ovl_split(const uint32_t *big, const uint32_t *cand_off, const uint64_t *cand, uint64_t *tmp, uint32_t sub_mul,
            OvlPiece *pieces, uint32_t *n_pieces, uint32_t *max_piece) {
    __shared__ uint32_t hist[PBA_OVL_SUB], piece_of[PBA_OVL_SUB], pstart[PBA_OVL_SUB], pcur[PBA_OVL_SUB];
    __shared__ uint32_t np_s, pbase_s;
    const uint32_t tl = big[blockIdx.x];
    const uint32_t lo = cand_off[tl], n = cand_off[tl + 1] - lo;
    for (uint32_t i = threadIdx.x; i < PBA_OVL_SUB; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    auto bucket = [&](uint64_t cd) {
        return cd == ~0ull ? (uint32_t)PBA_OVL_SUB - 1 : min((uint32_t)PBA_OVL_SUB - 1, __umulhi((uint32_t)(cd >> PBA_OVL_Q_SHIFT), sub_mul));
    };
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) atomicAdd(&hist[bucket(cand[lo + i])], 1u);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t np = 0, acc = 0, at = 0;
        pstart[0] = 0;
        for (uint32_t b = 0; b < PBA_OVL_SUB; ++b) {
            if (acc && acc + hist[b] > PBA_IX_LDS_SORT_CAP) { pstart[++np] = at; acc = 0; }   // close the piece before bucket b
            piece_of[b] = np;
            acc += hist[b]; at += hist[b];
        }
        np_s = np + 1;
        pbase_s = atomicAdd(n_pieces, np + 1);
    }
    __syncthreads();
    const uint32_t np = np_s;
    for (uint32_t p = threadIdx.x; p < np; p += blockDim.x) {
        const uint32_t end = p + 1 < np ? pstart[p + 1] : n;
        pieces[pbase_s + p] = OvlPiece{lo + pstart[p], end - pstart[p]};
        if (end - pstart[p] <= PBA_IX_LDS_SORT_CAP) atomicMax(max_piece, end - pstart[p]);
        pcur[p] = pstart[p];
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        const uint64_t cd = cand[lo + i];
        tmp[lo + atomicAdd(&pcur[piece_of[bucket(cd)]], 1u)] = cd;
    }
}
// work item i of a call = (target, first candidate of its group of 64): item_pre[t] = items of the targets before t
static __global__ void __launch_bounds__(256)
k_ovl_items(const uint32_t *item_pre, const uint32_t *cand_off, uint32_t n_targets, uint32_t n_items, uint2 *items) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_items) return;
    uint32_t lo = 0, hi = n_targets;                // last target t with item_pre[t] <= i
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (item_pre[mid] <= i) lo = mid; else hi = mid;
    }
    items[i] = make_uint2(lo, cand_off[lo] + (i - item_pre[lo]) * PBA_WAVE);
}

// waves per SIMD of the walk (tuning hooks, as in align_common.h).  Measured at 100 k reads (walk<2> then walk<3>): 6 / 6 ->
// 288 ms, 8 / 6 -> 277 ms, 8 / 8 -> 270 ms.
#ifndef PBA_OVL_OCC12
#define PBA_OVL_OCC12 8
#endif
#ifndef PBA_OVL_OCC34
#define PBA_OVL_OCC34 8
#endif
#define PBA_OVL_CHUNK 16          // work items a wavefront takes at a time (big inputs)
#define PBA_OVL_S2_CAP 256        // survivors of the first prefilter stage a chunk can hand to the second


// One candidate of target `ref` (length ref_len, visiting order ht) set up like spaced_seed.cpp:274-286 /
// ref_seq.h:282-286.  ok = false: skipped before the aligner (segment shorter than OVERLAP_MIN).
struct OvlCand {
    bool ok, fwd;
    uint32_t q;
    int j, hit, r_off, r_len, s_off, s_len;
};
__device__ __forceinline__ OvlCand ovl_decode_len(int slen, int ref_len, const HeadTail &ht, uint64_t cd, const OvlCfg &cfg) {
    OvlCand c;
    c.q = (uint32_t)(cd >> PBA_OVL_Q_SHIFT);
    const uint32_t jd = (uint32_t)(cd >> PBA_OVL_ORD_BITS) & ((1u << PBA_OVL_JD_BITS) - 1);
    c.hit = ht.pos_of((int)(cd & ((1u << PBA_OVL_ORD_BITS) - 1)));
    c.j = (int)(jd >> 1);
    c.fwd = (jd & 1) == 0;
    const int pos = c.fwd ? c.j : slen - c.j - 16;
    c.s_off = c.fwd ? pos : pos + 15;                           // spaced_seed.cpp:274
    c.s_len = c.fwd ? slen - c.s_off : c.s_off + 1;             // spaced_seed.cpp:275
    c.ok = c.s_len >= cfg.overlap_min;                          // spaced_seed.cpp:280
    c.r_off = c.fwd ? c.hit : c.hit + 15;                       // spaced_seed.cpp:285
    c.r_len = c.fwd ? ref_len - c.r_off : c.r_off + 1;          // ref_seq.h:284-285
    return c;
}
__device__ __forceinline__ OvlCand ovl_decode(const SeqSetDev &Rd, int ref_len, const HeadTail &ht, uint64_t cd,
                                              const OvlCfg &cfg) {
    return ovl_decode_len((int)Rd.len[(uint32_t)(cd >> PBA_OVL_Q_SHIFT)], ref_len, ht, cd, cfg);
}

__device__ __forceinline__ void ovl_emit(pba_overlap *out, unsigned long long cap, unsigned long long *n_out, uint32_t t,
                                         uint32_t q, int j, bool fwd, int hit, const AlnOut &o) {
    // every lane calls the atomic (lane 0 adds 1): the lane-0-only form inside a persistent loop is what
    // ROCm 7.2's clang miscompiles (tools/ubench_queue.hip)
    const bool l0 = (threadIdx.x & (PBA_WAVE - 1)) == 0;
    const unsigned long long slot = atomicAdd(n_out, l0 ? 1ull : 0ull);
    if (l0 && slot < cap) {
        pba_overlap *r = out + slot;
        r->target = (int32_t)t; r->query = (int32_t)q; r->j = j; r->dir = fwd ? 1 : -1; r->ref_pos = hit;
        r->cost = o.cost; r->matlen_a = o.matlen_a; r->matlen_b = o.matlen_b;
    }
}

// First launch (redo_in == nullptr): persistent wavefronts pull work items = (target, group of 64 consecutive
// candidates) and walk them with the narrow window (NB = 0: row sweep).  Candidates of one (target, query) are
// consecutive and must be tried in order, so a group owns the runs that START in it: it skips a leading run begun in
// the previous group and follows its last run past its own end.  (Whole targets as work items left the chip idle at
// the end: a target costs as much as its few true overlaps, which vary a lot.)
// A candidate the narrow window cannot certify parks its (target, query): the rest of that query's candidates are
// skipped and (target - t_lo, candidate index) goes to redo_out.
// Second launch (redo_in != nullptr, full_band): one parked (target, query) per work item, resumed at the parked
// candidate with the reference band until the first success or the end of the query's candidates.
template <int NB>
static __global__ void __launch_bounds__(PBA_WAVE * (NB ? 4 : 1), NB == 0 ? 1 : (NB <= 2 ? PBA_OVL_OCC12 : (NB <= 4 ? PBA_OVL_OCC34 : 3)))
k_ovl_walk(SeqSetDev Rd, uint32_t t_lo, uint32_t n_items, const uint2 *items, const uint32_t *cand_off, const uint32_t *cand_cnt,
           const uint64_t *cand, OvlCfg cfg, int full_band, const uint2 *redo_in, uint2 *redo_out, unsigned long long redo_cap,
           unsigned long long *n_redo_out, pba_overlap *out, unsigned long long cap, unsigned long long *n_out,
           unsigned long long *n_pairs, uint32_t *queue) {
    extern __shared__ __align__(16) uint8_t lds_all[];
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / PBA_WAVE));
    uint16_t *lds = (uint16_t *)(lds_all + (size_t)wave * cfg.row_cap * 2);
    const bool l0 = (threadIdx.x & (PBA_WAVE - 1)) == 0;
    const uint32_t lane_id = threadIdx.x & (PBA_WAVE - 1);
    const PreThresholds pre_t(cfg.R);
    // A million reads make hundreds of millions of light items (a group of 64 mostly false candidates): one atomic on
    // the queue and one on the pair counter per item is then what the walk waits for (every wavefront on the same two
    // addresses).  Items are taken up to 16 at a time and the pairs are added up per wavefront.
    // (below a million items chunks cost more in load balance at the end of the launch than they save: an item with a true
    // overlap is ~500 x a group of false candidates)
    // (after the pre-sort prefilter stage an item is 64 candidates of runs that hold a survivor -- a true overlap in every
    // fourth: the host then asks for single items, cfg.chunk = 1)
    const uint32_t chunk = (redo_in || n_items < (1u << 20)) ? 1u : (cfg.chunk ? min(cfg.chunk, (uint32_t)PBA_OVL_CHUNK) : (uint32_t)PBA_OVL_CHUNK);
    // Two-stage prefilter over a chunk (first launch, bit-vector kernels): the first 32 rows of every candidate of the
    // chunk's groups (one candidate per lane, prefilter32), the survivors of ALL its groups listed in LDS and put
    // through rows 33..64 together (prefilter64, one survivor per lane) -- and only what passes both reaches the
    // wavefront-wide array.  fail[k]: lanes of group k that failed in either stage.
    constexpr int WPB = NB ? 4 : 1;
    __shared__ uint32_t s2_fail[WPB][PBA_OVL_CHUNK][2];
    __shared__ uint16_t s2_list[WPB][PBA_OVL_S2_CAP];
    __shared__ uint64_t s_cd[WPB][PBA_WAVE];           // the group in flight: kept here, not in registers, while the array has the wavefront
    // cfg.fused: every listed candidate has passed its first 32 rows in the scan (k_ovl_scan), which also counted the pairs;
    // a chunk then goes through rows 33..64 directly (no first stage, no survivor list) and nothing is counted here
    const bool fused = NB != 0 && cfg.fused != 0;
    const bool two_stage = NB != 0 && !redo_in && !fused;
    unsigned long long pairs = 0;
    for (;;) {
        const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane(
            (int)atomicAdd(queue, l0 ? chunk : 0u));                    // see next_slot() in pba_device.hip
        if (base >= n_items) break;
        const uint32_t item_end = min(n_items, base + chunk);
        if constexpr (NB != 0) {
          if (fused && !redo_in) {
            for (uint32_t k = 0; k < item_end - base; ++k) {
                const uint2 it = items[base + k];
                const uint32_t c_end_k = cand_off[it.x] + cand_cnt[it.x];
                const bool act = lane_id < min((uint32_t)PBA_WAVE, c_end_k - it.y);
                const uint64_t cd = act ? cand[it.y + lane_id] : 0ull;
                const uint32_t t = t_lo + it.x;
                const int ref_len = (int)Rd.len[t];
                const HeadTail ht(ref_len);
                const OvlCand m = ovl_decode(Rd, ref_len, ht, cd, cfg);
                const bool fail2 = prefilter64(act && m.ok, fetch_of(Rd, t, m.r_off, m.fwd ? 1 : -1), m.r_len,
                                               fetch_of(Rd, act ? m.q : 0u, m.s_off, m.fwd ? 1 : -1), m.s_len, cfg.R);
                const uint64_t f2 = __builtin_amdgcn_ballot_w64(fail2);
                if (l0) { s2_fail[wave][k][0] = (uint32_t)f2; s2_fail[wave][k][1] = (uint32_t)(f2 >> 32); }
            }
            __builtin_amdgcn_wave_barrier();
          }
          if (two_stage) {
            uint32_t ns = 0;                                            // survivors of stage 1 listed so far (wave-uniform)
            for (uint32_t k = 0; k < item_end - base; ++k) {
                const uint2 it = items[base + k];
                const uint32_t c_end_k = cand_off[it.x] + cand_cnt[it.x];
                const bool act = lane_id < min((uint32_t)PBA_WAVE, c_end_k - it.y);
                const uint64_t cd = act ? cand[it.y + lane_id] : 0ull;
                const uint32_t t = t_lo + it.x;
                const int ref_len = (int)Rd.len[t];
                const HeadTail ht(ref_len);
                const OvlCand m = ovl_decode(Rd, ref_len, ht, cd, cfg);
                AlnOut po;
                const int fr = prefilter32(act && m.ok, fetch_of(Rd, t, m.r_off, m.fwd ? 1 : -1), m.r_len,
                                           fetch_of(Rd, act ? m.q : 0u, m.s_off, m.fwd ? 1 : -1), m.s_len, cfg.R, 0, 0, pre_t, po);
                const uint64_t f1 = __builtin_amdgcn_ballot_w64(fr != 0);
                const uint64_t s1 = __builtin_amdgcn_ballot_w64(act && m.ok && fr == 0);
                if (l0) { s2_fail[wave][k][0] = (uint32_t)f1; s2_fail[wave][k][1] = (uint32_t)(f1 >> 32); }
                if (s1) {
                    const uint32_t at = ns + (uint32_t)__builtin_popcountll(s1 & ((1ull << lane_id) - 1ull));
                    if (((s1 >> lane_id) & 1ull) && at < PBA_OVL_S2_CAP) s2_list[wave][at] = (uint16_t)(k << 6 | lane_id);
                    ns += (uint32_t)__builtin_popcountll(s1);
                }
            }
            ns = min(ns, (uint32_t)PBA_OVL_S2_CAP);                     // (survivors beyond the list go to the array untested: still exact)
            __builtin_amdgcn_wave_barrier();
            for (uint32_t b0 = 0; b0 < ns; b0 += PBA_WAVE) {
                const bool have = b0 + lane_id < ns;
                const uint32_t e = have ? s2_list[wave][b0 + lane_id] : 0u, k = e >> 6, ln = e & 63u;
                const uint2 it = items[base + (have ? k : 0u)];
                const uint64_t cd = have ? cand[it.y + ln] : 0ull;
                const uint32_t t = t_lo + it.x;
                const int ref_len = (int)Rd.len[t];
                const HeadTail ht(ref_len);
                const OvlCand m = ovl_decode(Rd, ref_len, ht, cd, cfg);
                const bool fail2 = prefilter64(have, fetch_of(Rd, t, m.r_off, m.fwd ? 1 : -1), m.r_len,
                                               fetch_of(Rd, have ? m.q : 0u, m.s_off, m.fwd ? 1 : -1), m.s_len, cfg.R);
                if (fail2) atomicOr(&s2_fail[wave][k][ln >> 5], 1u << (ln & 31u));
            }
            __builtin_amdgcn_wave_barrier();
          }
        }
        for (uint32_t item = base; item < item_end; ++item) {
        const uint32_t NONE = 0xFFFFFFFFu;
        uint32_t tl, c_begin, c_end, own_end, skip_q = NONE, only_q = NONE;
        // (an item is the same in every lane: saying so keeps what describes it in scalar registers)
        auto uni = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
        if (redo_in) {
            tl = uni(redo_in[item].x); c_begin = uni(redo_in[item].y); c_end = own_end = uni(cand_off[tl] + cand_cnt[tl]);
            only_q = uni((uint32_t)(cand[c_begin] >> PBA_OVL_Q_SHIFT));
        } else {
            tl = uni(items[item].x); c_begin = uni(items[item].y); c_end = uni(cand_off[tl] + cand_cnt[tl]);
            own_end = min(c_begin + (uint32_t)PBA_WAVE, c_end);
            if (c_begin > uni(cand_off[tl])) skip_q = uni((uint32_t)(cand[c_begin - 1] >> PBA_OVL_Q_SHIFT));   // a run begun in the previous group
        }
        const uint32_t t = t_lo + tl;
        const PackedFetch ref = fetch_of_uniform(Rd, t, 0, 1);
        const int ref_len = __builtin_amdgcn_readfirstlane((int)Rd.len[t]);
        const HeadTail ht(ref_len);
        uint32_t done_q = NONE, last_q = NONE;
        bool stop = false;
        // 64 candidates at a time: every lane decodes its candidate and runs its first 32 rows (prefilter.h); then the
        // group is walked in order and only the candidates that survived get the wavefront
        for (uint32_t c0 = c_begin; c0 < c_end && !stop; c0 += PBA_WAVE) {
            if (c0 >= own_end) {                                            // past the own group: only to finish its last run
                const uint32_t nq = uni((uint32_t)(cand[c0] >> PBA_OVL_Q_SHIFT));
                if (nq != last_q || done_q == last_q || skip_q == last_q) break;
            }
            const uint32_t lane = threadIdx.x & (PBA_WAVE - 1), ng = min((uint32_t)PBA_WAVE, c_end - c0);
            const bool act = lane < ng;
            const uint64_t mycd = act ? cand[c0 + lane] : 0ull;
            const uint32_t myq = (uint32_t)(mycd >> PBA_OVL_Q_SHIFT);
            const OvlCand m = ovl_decode(Rd, ref_len, ht, mycd, cfg);
            int myfr = 0;
            if constexpr (NB != 0) {
                if ((two_stage || (fused && !redo_in)) && c0 == c_begin) {   // the stage(s) ran on this group with the rest of the chunk
                    const uint32_t w32 = s2_fail[wave][item - base][lane >> 5];
                    myfr = (int)((w32 >> (lane & 31u)) & 1u);
                } else if (fused) {
                    myfr = 0;                                            // passed its first 32 rows in the scan: the array decides
                } else {
                    AlnOut po;
                    myfr = prefilter32(act && m.ok, ref.at(m.r_off, m.fwd ? 1 : -1), m.r_len,
                                       fetch_of(Rd, act ? myq : 0u, m.s_off, m.fwd ? 1 : -1), m.s_len, cfg.R, 0, 0, pre_t, po);
                }
            }
            // Nearly every candidate has failed by now, so the group is not walked lane by lane: lane masks say where the
            // walk of this group ends, which failed candidates count as pairs, and only the survivors are visited.
            const uint32_t up_q = (uint32_t)__shfl_up((int)myq, 1, PBA_WAVE);
            __builtin_amdgcn_wave_barrier();                                    // (the previous group's reads are done)
            s_cd[wave][lane] = mycd;
            __builtin_amdgcn_wave_barrier();
            const uint32_t prev_q = lane == 0 ? last_q : up_q;
            uint64_t brk = 0;                                                   // first lane the walk does not reach
            if (redo_in) brk = __builtin_amdgcn_ballot_w64(act && myq != only_q);           // the parked query's candidates are contiguous
            if (c0 >= own_end) brk |= __builtin_amdgcn_ballot_w64(act && myq != prev_q);    // the run that started in the own group has ended
            const uint32_t lim = brk ? (uint32_t)__builtin_ctzll(brk) : ng;
            const uint64_t in = lim >= PBA_WAVE ? ~0ull : (1ull << lim) - 1ull;
            const uint64_t live = __builtin_amdgcn_ballot_w64(act && myq != skip_q) & in;   // (a leading run of the previous group is not ours)
            const uint64_t failed = __builtin_amdgcn_ballot_w64(myfr != 0) & live;          // failed within their first 32 rows
            uint64_t surv = __builtin_amdgcn_ballot_w64(myfr == 0 && m.ok) & live;
            uint32_t from = 0;
            // failed candidates of lanes [from, to): pairs the reference aligned, unless their query was done already
            auto count_failed = [&](uint32_t to) {
                if (to > from) {
                    const uint64_t f = failed & (to >= PBA_WAVE ? ~0ull : (1ull << to) - 1ull) & ~((1ull << from) - 1ull);
                    if (f) pairs += (unsigned long long)__builtin_popcountll(
                               f & __builtin_amdgcn_ballot_w64((uint32_t)(s_cd[wave][lane] >> PBA_OVL_Q_SHIFT) != done_q));
                }
            };
            while (surv) {
                const uint32_t k = (uint32_t)__builtin_ctzll(surv);
                surv &= surv - 1ull;
                count_failed(k);
                from = k + 1;
                const uint32_t c = c0 + k;
                const uint64_t cd = uniform_u64(s_cd[wave][k]);
                const uint32_t q = (uint32_t)(cd >> PBA_OVL_Q_SHIFT);
                if (q == done_q) continue;                                  // first success per (target, query) already taken
                const OvlCand mk = ovl_decode_len(__builtin_amdgcn_readfirstlane((int)Rd.len[q]), ref_len, ht, cd, cfg);
                const PackedFetch fa = ref.at(mk.r_off, mk.fwd ? 1 : -1);    // a = the target in the reference role (ref_seq.h:264)
                const PackedFetch fb = fetch_of_uniform(Rd, q, mk.s_off, mk.fwd ? 1 : -1);
                AlnOut o;
                if constexpr (NB == 0) align_rowsweep(fa, mk.r_len, fb, mk.s_len, cfg.R, 0, 0, lds, cfg.row_cap, o);
                else align_bitvec<NB, true>(fa, mk.r_len, fb, mk.s_len, cfg.R, 0, 0, full_band != 0, lds, cfg.row_cap, o);
                if (o.rc == PBA_RC_UNCERTIFIED) {                           // park the (target, query): resumed in a wider ring
                    done_q = q;
                    const unsigned long long slot = atomicAdd(n_redo_out, l0 ? 1ull : 0ull);
                    if (l0 && slot < redo_cap) redo_out[slot] = make_uint2(tl, c);
                    continue;
                }
                ++pairs;
                if (o.rc < 0 || o.matlen_a < cfg.overlap_min) continue;     // ref_seq.h:264-265
                done_q = q;
                ovl_emit(out, cap, n_out, t, q, mk.j, mk.fwd, mk.hit, o);
            }
            count_failed(lim);
            if (lim) last_q = uni((uint32_t)(s_cd[wave][lim - 1] >> PBA_OVL_Q_SHIFT));
            if (brk) stop = true;
        }
        }
    }
    atomicAdd(n_pairs, l0 ? pairs : 0ull);
}

// ---------------------------------------------------------------------------------------------------------------
// What a success spares.  The reference walks a (target, query) run in order and stops at the first success
// (spaced_seed.cpp:424-432, 288-297): the run's candidates behind it are never handed to align().  The scan has counted
// every candidate past the OVERLAP_MIN gate as a pair (k_ovl_scan: totals[1]); this kernel takes one wavefront per reported
// overlap (target t, query q, probe jd* = 2j + backward, hit position -> ordinal ord*) and counts the candidates of the run
// that sort behind (jd*, ord*): for every probe jd > jd* of q that exists (spaced_seed.cpp:426, key != 0) and passes the gate,
// the visited positions of t with the same masked key, plus the positions behind ord* under the success's own key.
// pairs = totals[1] - sum over the overlaps.  (~10 k instructions per success, against ~700 k for the alignment itself.)
#define PBA_OVL_AFTER_SLOTS 512                    // hash slots per wavefront for <= 126 probe keys
static __global__ void __launch_bounds__(PBA_WAVE * 4)
k_ovl_after(SeqSetDev Rd, const pba_overlap *ov, uint32_t n_ov, uint32_t mask, uint32_t t2, int overlap_min, unsigned long long *after) {
    __shared__ uint32_t h_key[4][PBA_OVL_AFTER_SLOTS], h_cnt[4][PBA_OVL_AFTER_SLOTS];
    const uint32_t lane = threadIdx.x & (PBA_WAVE - 1), wave = threadIdx.x / PBA_WAVE;
    const uint32_t i = blockIdx.x * 4 + wave;
    if (i >= n_ov) return;                                               // (wave-uniform; no workgroup barrier below)
    const pba_overlap o = ov[i];
    const uint32_t t = (uint32_t)o.target, q = (uint32_t)o.query;
    const uint32_t jd0 = 2u * (uint32_t)o.j + (o.dir < 0 ? 1u : 0u);
    const int tlen = (int)Rd.len[t], slen = (int)Rd.len[q];
    const TargetWalk tw(tlen);
    const int ord0 = tw.ord_of(o.ref_pos);
    const uint8_t *tseq = Rd.packed + Rd.off[t], *qseq = Rd.packed + Rd.off[q];
    for (uint32_t k = lane; k < PBA_OVL_AFTER_SLOTS; k += PBA_WAVE) { h_key[wave][k] = 0u; h_cnt[wave][k] = 0u; }   // key 0 = empty: no probe has it
    __builtin_amdgcn_wave_barrier();
    // the probes of q from jd* on: the success's own key, and the later ones into the table with their multiplicity
    uint32_t key0 = 0;
    for (uint32_t jb = 0; jb < t2; jb += PBA_WAVE) {
        const uint32_t jd = jb + lane;
        const int j = (int)(jd >> 1);
        const int pos = (jd & 1u) ? slen - j - 16 : j;                   // spaced_seed.cpp:426
        const bool exists = jd < t2 && jd >= jd0 && pos >= 0 && pos + 16 <= slen && slen - j >= overlap_min;
        const uint32_t key = exists ? window_key(qseq, (uint32_t)pos, (uint32_t)slen) & mask : 0u;
        const uint64_t own = __builtin_amdgcn_ballot_w64(jd == jd0);
        if (own) key0 = (uint32_t)__builtin_amdgcn_readlane((int)key, __builtin_ctzll(own));
        if (key != 0u && jd != jd0) {
            uint32_t h = (key * 0x9E3779B1u) >> (32 - 9);
            for (;;) {
                const uint32_t old = atomicCAS(&h_key[wave][h], 0u, key);
                if (old == 0u || old == key) { atomicAdd(&h_cnt[wave][h], 1u); break; }
                h = (h + 1) & (PBA_OVL_AFTER_SLOTS - 1);
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    uint32_t cnt = 0;
    for (int c = (int)lane; c < tw.nchunks; c += PBA_WAVE) {
        const uint64_t be = chunk_bits(tseq, (uint32_t)c);
#pragma unroll 4
        for (int k = 0; k < 16; ++k) {
            const uint32_t key = __builtin_bswap32((uint32_t)((be << (2 * k)) >> 32)) & mask;
            const int ord = tw.ord_of(16 * c + k);
            if (key == 0u || ord < 0) continue;                          // ref_seq.h:300,307
            if (key == key0 && ord > ord0) ++cnt;
            uint32_t h = (key * 0x9E3779B1u) >> (32 - 9);
            for (;;) {
                const uint32_t kk = h_key[wave][h];
                if (kk == 0u) break;
                if (kk == key) { cnt += h_cnt[wave][h]; break; }
                h = (h + 1) & (PBA_OVL_AFTER_SLOTS - 1);
            }
        }
    }
#pragma unroll
    for (int d = 1; d < PBA_WAVE; d <<= 1) cnt += __shfl_xor(cnt, d, PBA_WAVE);
    if (lane == 0 && cnt) atomicAdd(after, (unsigned long long)cnt);
}

#endif
