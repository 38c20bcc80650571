// overlap.h -- all-vs-all overlap driver: every read is a target in the
// reference role (ref_seq::get_seedmap index, /root/reference/src/ref_seq.h:291-311) and every other read a
// query walked like one read of a locked spaced_seed round (spaced_seed.cpp:262-298, 420-437): 2*max_trial
// probes (forward at j, backward at slen-j-16), hits in the seedmap's list order, first success per
// (target, query) wins.  All successful (target, query) pairs are reported.
//
// Inverted form: the PROBES are indexed (2*max_trial keys per read: small, and the object a
// multi-GPU run all-gathers), the targets' positions are scanned against that table:
//   k_probe_emit : one probe entry (key << 32 | probe id) per (query, j, direction), key 0 dropped
//   (partition + sort of the entries: the seed-index builder, seed_index.h)
//   k_ovl_scan   : one workgroup per target; every indexed position of the target in get_seedmap order looks
//                  its key up and emits one candidate per matching probe of another read,
//                  candidate = query << 23 | (2j + backward) << 16 | ordinal   (count pass, then fill pass)
//   (per-target sort of the candidates: k_part_sort; 64-bit order = query, then j, forward before backward,
//    then the seedmap's list order -- exactly the order spaced_seed tries them in)
//   k_ovl_walk   : persistent wavefronts walk a target's candidates, align until the first success per query;
//                  (target, query) runs whose narrow-window verdict is not certified are parked and resumed by a
//                  second launch at the reference band
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

// Presence filter of the probe table: one bit per hashed key (2^26 bits = 8 MB, L2 / MALL resident).  Only ~1 in 13
// positions of a target carries a key some probe has (2*max_trial probes per read against 4^12 keys), so one bit
// load spares the scan most of its dependent binary searches.
#define PBA_OVL_PRES_LOG 26
__device__ __forceinline__ uint32_t ovl_pres_slot(uint32_t key) { return (key * 0x9E3779B1u) >> (32 - PBA_OVL_PRES_LOG); }
static __global__ void __launch_bounds__(256)
k_ovl_presence(const uint64_t *ent, uint64_t n, uint32_t *bits) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t h = ovl_pres_slot((uint32_t)(ent[i] >> 32));
    atomicOr(bits + (h >> 5), 1u << (h & 31));
}

// A target's candidates are kept in PBA_OVL_SUB buckets of consecutive queries (bucket = umulhi(q, 2^32 * SUB / n_reads),
// monotone in q, so the buckets in order are the list in query order): each bucket is counted, filled and sorted on its own, which keeps
// every sorted piece inside the LDS sort even when a target has far more than 16 384 candidates (57 000 at a million
// reads).  `shift` merges 2^shift neighbouring buckets (the host picks the coarsest split whose pieces still fit).
// FILL = false: cnt[(t - t_lo) * SUB + bucket] = candidates.  FILL = true: piece p = bucket >> shift of target t is written
// from cursor[(t - t_lo) * (SUB >> shift) + p] on.
#define PBA_OVL_SUB 64
template <bool FILL>
static __global__ void __launch_bounds__(256)
k_ovl_scan(IndexDev probes, KeyDir kd, const uint32_t *presence, SeqSetDev Rd, uint32_t t_lo, uint32_t n_targets, uint32_t t2,
           uint32_t sub_mul, int shift, uint32_t *cnt_or_cursor, uint64_t *cand) {
    const uint32_t tl = blockIdx.x;
    if (tl >= n_targets) return;
    // one workgroup owns a target, so its buckets' counters / write cursors live in LDS (a global atomic per candidate on
    // per-bucket addresses cannot be folded into one per wavefront and was the slowest part of the scan)
    __shared__ uint32_t sub[PBA_OVL_SUB];
    if (threadIdx.x < PBA_OVL_SUB)
        sub[threadIdx.x] = FILL ? (threadIdx.x < (PBA_OVL_SUB >> shift) ? cnt_or_cursor[tl * (PBA_OVL_SUB >> shift) + threadIdx.x] : 0u) : 0u;
    __syncthreads();
    const uint32_t t = t_lo + tl;
    const int len = (int)Rd.len[t];
    const uint8_t *seq = Rd.packed + Rd.off[t];
    const HeadTail ht(len);
    for (int ord = (int)threadIdx.x; ord < ht.visited; ord += (int)blockDim.x) {
        const int pos = ht.pos_of(ord);
        const uint32_t key = window_key(seq, (uint32_t)pos, (uint32_t)len) & probes.mask;
        if (!key) continue;                                             // ref_seq.h:300,307
        const uint32_t h = ovl_pres_slot(key);
        if (!((presence[h >> 5] >> (h & 31)) & 1u)) continue;           // no probe has this key
        uint32_t beg, n;
        if (kd.dir) dir_find(kd, probes.ent, key, beg, n);
        else ix_find(probes, key, beg, n);
        for (uint32_t h = 0; h < n; ++h) {
            const uint32_t pid = (uint32_t)probes.ent[beg + h];
            const uint32_t q = pid / t2;
            if (q == t) continue;
            const uint32_t b = min((uint32_t)PBA_OVL_SUB - 1, __umulhi(q, sub_mul));   // ~ q * SUB / n_reads, monotone in q
            if (FILL) {
                const uint32_t slot = atomicAdd(&sub[b >> shift], 1u);
                cand[slot] = (uint64_t)q << PBA_OVL_Q_SHIFT | (uint64_t)(pid % t2) << PBA_OVL_ORD_BITS | (uint32_t)ord;
            } else {
                atomicAdd(&sub[b], 1u);
            }
        }
    }
    if (!FILL) {
        __syncthreads();
        if (threadIdx.x < PBA_OVL_SUB) cnt_or_cursor[tl * PBA_OVL_SUB + threadIdx.x] = sub[threadIdx.x];
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

struct OvlCfg {
    double R;
    int overlap_min;
    int row_cap;      // u16 cells of LDS per wavefront
    uint32_t t2;
};

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
static __global__ void __launch_bounds__(PBA_WAVE * (NB ? 4 : 1), NB == 0 ? 1 : (NB <= 4 ? 6 : 3))
k_ovl_walk(SeqSetDev Rd, uint32_t t_lo, uint32_t n_items, const uint2 *items, const uint32_t *cand_off,
           const uint64_t *cand, OvlCfg cfg, int full_band, const uint2 *redo_in, uint2 *redo_out, unsigned long long redo_cap,
           unsigned long long *n_redo_out, pba_overlap *out, unsigned long long cap, unsigned long long *n_out,
           unsigned long long *n_pairs, uint32_t *queue) {
    extern __shared__ __align__(16) uint8_t lds_all[];
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / PBA_WAVE));
    uint16_t *lds = (uint16_t *)(lds_all + (size_t)wave * cfg.row_cap * 2);
    const bool l0 = (threadIdx.x & (PBA_WAVE - 1)) == 0;
    const PreThresholds pre_t(cfg.R);
    // A million reads make hundreds of millions of light items (a group of 64 mostly false candidates): one atomic on
    // the queue and one on the pair counter per item is then what the walk waits for (every wavefront on the same two
    // addresses).  Items are taken 16 at a time there and the pairs are added up per wavefront.
    const uint32_t chunk = (redo_in || n_items < (1u << 20)) ? 1u : 16u;
    unsigned long long pairs = 0;
    for (;;) {
        const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane(
            (int)atomicAdd(queue, l0 ? chunk : 0u));                    // see next_slot() in pba_device.hip
        if (base >= n_items) break;
        const uint32_t item_end = min(n_items, base + chunk);
        for (uint32_t item = base; item < item_end; ++item) {
        const uint32_t NONE = 0xFFFFFFFFu;
        uint32_t tl, c_begin, c_end, own_end, skip_q = NONE, only_q = NONE;
        if (redo_in) {
            tl = redo_in[item].x; c_begin = redo_in[item].y; c_end = own_end = cand_off[tl + 1];
            only_q = (uint32_t)(cand[c_begin] >> PBA_OVL_Q_SHIFT);
        } else {
            tl = items[item].x; c_begin = items[item].y; c_end = cand_off[tl + 1];
            own_end = min(c_begin + (uint32_t)PBA_WAVE, c_end);
            if (c_begin > cand_off[tl]) skip_q = (uint32_t)(cand[c_begin - 1] >> PBA_OVL_Q_SHIFT);   // a run begun in the previous group
        }
        const uint32_t t = t_lo + tl;
        const PackedFetch ref = fetch_of(Rd, t, 0, 1);
        const int ref_len = (int)Rd.len[t];
        const HeadTail ht(ref_len);
        uint32_t done_q = NONE, last_q = NONE;
        bool stop = false;
        // 64 candidates at a time: every lane decodes its candidate and runs its first 32 rows (prefilter.h); then the
        // group is walked in order and only the candidates that survived get the wavefront
        for (uint32_t c0 = c_begin; c0 < c_end && !stop; c0 += PBA_WAVE) {
            if (c0 >= own_end) {                                            // past the own group: only to finish its last run
                const uint32_t nq = (uint32_t)(cand[c0] >> PBA_OVL_Q_SHIFT);
                if (nq != last_q || done_q == last_q || skip_q == last_q) break;
            }
            const uint32_t lane = threadIdx.x & (PBA_WAVE - 1), ng = min((uint32_t)PBA_WAVE, c_end - c0);
            const bool act = lane < ng;
            const uint64_t mycd = act ? cand[c0 + lane] : 0ull;
            const uint32_t myq = (uint32_t)(mycd >> PBA_OVL_Q_SHIFT);
            const OvlCand m = ovl_decode(Rd, ref_len, ht, mycd, cfg);
            int myfr = 0;
            if constexpr (NB != 0) {
                AlnOut po;
                myfr = prefilter32(act && m.ok, ref.at(m.r_off, m.fwd ? 1 : -1), m.r_len,
                                   fetch_of(Rd, act ? myq : 0u, m.s_off, m.fwd ? 1 : -1), m.s_len, cfg.R, 0, 0, pre_t, po);
            }
            // Nearly every candidate has failed by now, so the group is not walked lane by lane: lane masks say where the
            // walk of this group ends, which failed candidates count as pairs, and only the survivors are visited.
            const uint32_t up_q = (uint32_t)__shfl_up((int)myq, 1, PBA_WAVE);
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
                    if (f) pairs += (unsigned long long)__builtin_popcountll(f & __builtin_amdgcn_ballot_w64(myq != done_q));
                }
            };
            while (surv) {
                const uint32_t k = (uint32_t)__builtin_ctzll(surv);
                surv &= surv - 1ull;
                count_failed(k);
                from = k + 1;
                const uint32_t c = c0 + k;
                const uint64_t cd = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(mycd >> 32), (int)k) << 32) |
                                    (uint32_t)__builtin_amdgcn_readlane((int)mycd, (int)k);
                const uint32_t q = (uint32_t)(cd >> PBA_OVL_Q_SHIFT);
                if (q == done_q) continue;                                  // first success per (target, query) already taken
                const OvlCand mk = ovl_decode(Rd, ref_len, ht, cd, cfg);
                const PackedFetch fa = ref.at(mk.r_off, mk.fwd ? 1 : -1);    // a = the target in the reference role (ref_seq.h:264)
                const PackedFetch fb = fetch_of(Rd, q, mk.s_off, mk.fwd ? 1 : -1);
                AlnOut o;
                if constexpr (NB == 0) align_rowsweep(fa, mk.r_len, fb, mk.s_len, cfg.R, 0, 0, lds, cfg.row_cap, o);
                else align_bitvec<NB>(fa, mk.r_len, fb, mk.s_len, cfg.R, 0, 0, full_band != 0, lds, cfg.row_cap, o);
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
            if (lim) last_q = (uint32_t)__builtin_amdgcn_readlane((int)myq, (int)(lim - 1));
            if (brk) stop = true;
        }
        }
    }
    atomicAdd(n_pairs, l0 ? pairs : 0ull);
}

#endif
