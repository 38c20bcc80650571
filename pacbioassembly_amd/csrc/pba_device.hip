// pba_device.hip -- gfx950 kernels and the device half of the C ABI (include/pba.h).
//
// One process per GPU, one pba_ctx per process, one HIP stream per ctx.  Everything here fails
// loudly (PBA_E_NODEVICE / PBA_E_HIP): there is no CPU path behind these entry points.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <functional>
#include <new>
#include <vector>

#include "align_bitvec.h"
#include "align_bvtrace.h"
#include "align_rowsweep.h"
#include "consensus.h"
#include "dev_common.h"
#include "pba.h"
#include "pba_internal.h"
#include "prefilter.h"
#include "overlap.h"
#include "seed_index.h"

// ---------------------------------------------------------------------------------------------
// host-side objects
// ---------------------------------------------------------------------------------------------
struct pba_ctx {
    int device;
    hipStream_t own_stream, stream;
    hipDeviceProp_t prop;
    hipEvent_t ev[6];        // index begin/end, align begin/end, redo begin/end
    uint32_t *d_queue;       // work-queue counters of the persistent aligning kernels (one per launch in flight)
    void *d_scratch;         // parent-bit scratch of the trace / vote kernels, kept between calls (tens of GB: mapping
    size_t scratch_bytes;    // it anew on every call cost seconds); grown on demand, freed with the ctx
    pba_profile prof;
    char err[512];
};

static const size_t kPlaneSlack = 64;           // zero words before the first and after the last sequence of a bit plane
struct pba_seqs {
    pba_ctx *ctx;
    uint32_t n, max_len;
    uint64_t packed_bytes;   // packed payload resident in HBM (incl. alignment padding)
    bool non_acgt;           // some byte outside ACGT was packed as code 3 (C2I): the packed DP would match it against T
    uint8_t *d_alloc;        // allocation; d_packed = d_alloc + kSlack
    uint8_t *d_packed;
    uint64_t *d_off;
    uint32_t *d_len;
    uint32_t *d_planes;      // allocation of the two bit planes, interleaved word by word, kPlaneSlack zero word pairs around them
    uint64_t *d_poff;        // word offset of every sequence inside a plane
    uint64_t plane_words;    // words of one plane incl. its slack
    std::vector<uint64_t> h_off;
    std::vector<uint32_t> h_len;
    SeqSetDev dev() const { return SeqSetDev{d_packed, d_off, d_len, d_planes + 2 * kPlaneSlack, d_poff}; }
};

struct pba_index {
    pba_ctx *ctx;
    uint32_t mask, seq_len, visited, nhead;
    int32_t tail_top;
    int mode, logP;
    uint64_t n_entries;
    uint64_t *d_ent;
    uint32_t *d_part_off;
    IndexDev dev() const { return IndexDev{d_ent, d_part_off, logP, mask, nhead, tail_top}; }
};

static int ctx_fail(pba_ctx *ctx, int st, const char *what, hipError_t e) {
    if (ctx)
        snprintf(ctx->err, sizeof ctx->err, "%s: %s", what, e == hipSuccess ? pba_strerror(st) : hipGetErrorString(e));
    return st;
}
#define HIPCHK(call)                                                        \
    do {                                                                    \
        hipError_t e__ = (call);                                            \
        if (e__ != hipSuccess) return ctx_fail(ctx, PBA_E_HIP, #call, e__); \
    } while (0)
#define PBA_FAIL(st, what) return ctx_fail(ctx, (st), (what), hipSuccess)

// engine limits
static const int kMaxSeqLen = 65000;            // u16 DP costs: D(i,j) <= max(i,j) < 65535
static const int kRowSweepLdsCap = 96 * 1024;   // LDS bytes one wavefront may take for its band row
static const size_t kSlack = 1024;              // readable bytes before the first and after the last packed byte
                                                // (the bit-vector kernel streams a few hundred bases past an accessor)

// RAII for temporaries so early returns do not leak device memory
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    template <class T> T *as() const { return (T *)p; }
};

// ---------------------------------------------------------------------------------------------
// kernels: packing
// ---------------------------------------------------------------------------------------------
// ASCII -> 2-bit, 16 chars per thread into one packed dword.  Thread t owns packed dword t of the
// whole set; its sequence is found by bisection over the (16-byte aligned) packed offsets.
__global__ void __launch_bounds__(256)
k_pack_text(const uint8_t *text, const uint64_t *text_off, const uint64_t *pk_off, const uint32_t *len, uint32_t n,
            uint64_t total_dwords, uint8_t *packed, int strict, uint32_t *bad) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total_dwords) return;
    const uint64_t byte = t * 4;
    uint32_t lo = 0, hi = n;            // last s with pk_off[s] <= byte
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (pk_off[mid] <= byte) lo = mid; else hi = mid;
    }
    const uint32_t s = lo;
    const uint32_t L = len[s];
    const uint64_t w = (byte - pk_off[s]) >> 2;          // dword index inside the sequence
    if (w * 16 >= L) return;                             // alignment padding
    const uint8_t *src = text + text_off[s] + w * 16;
    const uint32_t nb = (uint32_t)min((uint64_t)16, (uint64_t)L - w * 16);
    uint32_t word = 0, notacgt = 0;
    for (uint32_t k = 0; k < nb; ++k) {
        const uint32_t ch = src[k];
        const uint32_t code = ch == 'A' ? 0u : ch == 'C' ? 1u : ch == 'G' ? 2u : 3u;   // C2I, dna_seq.h:21
        notacgt |= (code == 3u && ch != 'T');
        word |= code << (8 * (k >> 2) + 6 - 2 * (k & 3));   // byte k/4, first base in bits 7:6
    }
    *reinterpret_cast<uint32_t *>(packed + byte) = word;    // padding bytes of the last dword stay 0
    if (notacgt) atomicOr(bad, 1u);
}

// Bit planes of a packed set (dev_common.h: SeqSetDev::plane): thread w owns plane word w of the whole set.
__global__ void __launch_bounds__(256)
k_make_planes(const uint8_t *packed, const uint64_t *off, const uint32_t *len, const uint64_t *poff, uint32_t n,
              uint64_t total_words, uint32_t *plane) {
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= total_words) return;
    uint32_t lo = 0, hi = n;            // last s with poff[s] <= w
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (poff[mid] <= w) lo = mid; else hi = mid;
    }
    const uint32_t s = lo;
    const uint64_t k = w - poff[s];     // word index inside the sequence
    const uint32_t L = len[s];
    uint32_t plo = 0, phi = 0;
    if (k * 32 < L) {
        planes_from_packed(packed + off[s], (int)(k * 32), plo, phi);
        const uint32_t valid = L - (uint32_t)(k * 32);
        if (valid < 32) { plo &= (1u << valid) - 1u; phi &= (1u << valid) - 1u; }   // nothing of the neighbour's bytes
    }
    plane[2 * w] = plo;                  // the two planes side by side: one line serves both (align_bitvec.h: load_planes32)
    plane[2 * w + 1] = phi;
}


// ---------------------------------------------------------------------------------------------
// kernels: alignment of explicit pairs
// ---------------------------------------------------------------------------------------------
// Every aligning kernel is a template on NB, the number of 32-row blocks a lane of the bit-vector
// array holds (align_bitvec.h); NB = 0 is the row-sweep kernel.  The host picks NB per launch from
// the widest band in the batch.  Nothing below calls a device function: the bodies inline.
struct AlignCfg {
    double R;
    int maxn, maxm;
    int row_cap;     // u16 cells of LDS per wavefront
    int full_band;   // bit-vector kernel: 0 = narrow first pass (may answer PBA_RC_UNCERTIFIED), 1 = reference band
};

// Wavefronts per workgroup: the CU admits only 16 workgroups, so single-wave workgroups cap the bit-vector
// kernel at 4 waves/SIMD; four independent waves per workgroup (one pair / read each, no barrier, own LDS
// slice) lift that.  The row sweep keeps one wave per workgroup because its band row can take most of the LDS.
template <int NB> struct Wpb {
    static constexpr int v = NB ? 4 : 1;
    // waves per SIMD the register allocator must leave room for (2nd __launch_bounds__ argument)
    static constexpr int occ = NB == 0 ? 1 : (NB <= 4 ? 6 : 3);   // measured on configs[1]: 5 -> 123 ms, 6 -> 105 ms, 8 (spills in the step loop) -> 113 ms
};

template <int NB>
__device__ __forceinline__ void align_dispatch(const PackedFetch &fa, int la, const PackedFetch &fb, int lb,
                                               const AlignCfg &cfg, void *lds, AlnOut &o) {
    if constexpr (NB == 0)
        align_rowsweep(fa, la, fb, lb, cfg.R, cfg.maxn, cfg.maxm, (uint16_t *)lds, cfg.row_cap, o);
    else
        align_bitvec<NB>(fa, la, fb, lb, cfg.R, cfg.maxn, cfg.maxm, cfg.full_band != 0, (uint16_t *)lds, cfg.row_cap, o);
}

__device__ __forceinline__ void store_result(pba_result *out, const AlnOut &o) {
    if ((threadIdx.x & (PBA_WAVE - 1)) == 0) {
        const bool ok = o.rc >= 0;
        out->rc = ok ? o.rc : (o.rc == PBA_RC_UNCERTIFIED ? PBA_RC_UNCERTIFIED : -1);
        out->cost = ok ? o.cost : 0;
        out->matlen_a = ok ? o.matlen_a : 0;
        out->matlen_b = ok ? o.matlen_b : 0;
        out->len_a = o.len_a; out->len_b = o.len_b; out->max_dst = o.max_dst;
    }
}

// Work distribution: every aligning kernel is launched with just enough workgroups to fill the chip and each
// wavefront pulls work items (pairs / reads) from a global counter until it runs dry.  A true 15 kb pair costs
// ~500x a false candidate, so a fixed item-per-wavefront mapping leaves most of a workgroup idle while its
// slowest wave finishes; the queue keeps every wavefront busy to the end.  Exit: the counter only grows, so every
// wave eventually reads a value >= n and leaves.
// NOTE: every lane calls atomicAdd (lane 0 adds 1, the others 0; the compiler folds that into one wave-level
// atomic).  The obvious `if (lane == 0) v = atomicAdd(q, 1)` inside a persistent loop is miscompiled by ROCm 7.2's
// clang (the loop's exit mask ends up covering every lane but lane 0 and the wave spins forever);
// tools/ubench_queue.hip reproduces both forms.
__device__ __forceinline__ uint32_t next_slot(uint32_t *queue) {
    const uint32_t v = atomicAdd(queue, (threadIdx.x & (PBA_WAVE - 1)) == 0 ? 1u : 0u);
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}

// ids (nullable): the subset of pairs / reads to process (second, full-band launch)
template <int NB>
__global__ void __launch_bounds__(PBA_WAVE * Wpb<NB>::v, Wpb<NB>::occ)
k_align_pairs(SeqSetDev A, SeqSetDev B, const pba_pair *pairs, const uint32_t *ids, uint32_t n, AlignCfg cfg,
              pba_result *out, uint32_t *queue) {
    extern __shared__ __align__(16) uint8_t lds_all[];
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / PBA_WAVE));   // wave-uniform on purpose: keeps the walk in SGPRs
    uint8_t *lds = lds_all + (size_t)wave * cfg.row_cap * 2;
    for (;;) {                                // persistent wavefront: pull the next pair until the queue is dry
        const uint32_t slot = next_slot(queue);
        if (slot >= n) break;
        const uint32_t q = ids ? ids[slot] : slot;
        const pba_pair pr = pairs[q];
        const PackedFetch fa = fetch_of(A, pr.a_seq, pr.a_pos, (pr.flags & PBA_A_BACKWARD) ? -1 : 1);
        const PackedFetch fb = fetch_of(B, pr.b_seq, pr.b_pos, (pr.flags & PBA_B_BACKWARD) ? -1 : 1);
        AlnOut o;
        align_dispatch<NB>(fa, pr.a_len, fb, pr.b_len, cfg, lds, o);
        store_result(out + q, o);
    }
}

__global__ void __launch_bounds__(PBA_WAVE)
k_align_bytes(const uint8_t *a, int a_dir, int la, const uint8_t *b, int b_dir, int lb, AlignCfg cfg,
              pba_result *out) {
    extern __shared__ __align__(16) uint8_t lds[];
    ByteFetch fa{a, a_dir}, fb{b, b_dir};
    AlnOut o;
    align_rowsweep(fa, la, fb, lb, cfg.R, cfg.maxn, cfg.maxm, (uint16_t *)lds, cfg.row_cap, o);
    store_result(out, o);
}

// ---- traceback: full-band row sweep that also stores one parent code per band cell, then a backward walk
__global__ void __launch_bounds__(PBA_WAVE)
k_align_pairs_trace(SeqSetDev A, SeqSetDev B, const pba_pair *pairs, uint32_t n, AlignCfg cfg, pba_result *out,
                    uint8_t *par, const uint64_t *par_off) {
    extern __shared__ __align__(16) uint8_t lds[];
    const uint32_t q = blockIdx.x;
    if (q >= n) return;
    const pba_pair pr = pairs[q];
    const PackedFetch fa = fetch_of(A, pr.a_seq, pr.a_pos, (pr.flags & PBA_A_BACKWARD) ? -1 : 1);
    const PackedFetch fb = fetch_of(B, pr.b_seq, pr.b_pos, (pr.flags & PBA_B_BACKWARD) ? -1 : 1);
    AlnOut o;
    align_rowsweep(fa, pr.a_len, fb, pr.b_len, cfg.R, cfg.maxn, cfg.maxm, (uint16_t *)lds, cfg.row_cap, o, par + par_off[q]);
    store_result(out + q, o);
}

__global__ void __launch_bounds__(PBA_WAVE)
k_align_bytes_trace(const uint8_t *a, int a_dir, int la, const uint8_t *b, int b_dir, int lb, AlignCfg cfg,
                    pba_result *out, uint8_t *par) {
    extern __shared__ __align__(16) uint8_t lds[];
    ByteFetch fa{a, a_dir}, fb{b, b_dir};
    AlnOut o;
    align_rowsweep(fa, la, fb, lb, cfg.R, cfg.maxn, cfg.maxm, (uint16_t *)lds, cfg.row_cap, o, par);
    store_result(out, o);
}

// ---- traceback on the bit-vector array (align_bvtrace.h): persistent wavefronts, each with its own scratch area
// of wave_words u32 (cap_words of parent bits, then the goal-first ops of the pair in flight).
// ids (nullable): the subset of pairs to process (second, full-band launch)
template <int NB>
__global__ void __launch_bounds__(PBA_WAVE * 4, NB <= 4 ? 4 : 2)
k_trace_pairs(SeqSetDev A, SeqSetDev B, const pba_pair *pairs, const uint32_t *ids, uint32_t n, AlignCfg cfg,
              pba_result *out, uint32_t *scratch, uint64_t wave_words, uint64_t cap_words, uint8_t *ops,
              const uint64_t *ops_off, int32_t *nedit, uint32_t *queue) {
    extern __shared__ __align__(16) uint8_t lds_all[];
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / PBA_WAVE));
    uint8_t *lds = lds_all + (size_t)wave * cfg.row_cap * 2;
    uint32_t *mine = scratch + ((uint64_t)blockIdx.x * 4 + wave) * wave_words;
    for (;;) {
        const uint32_t slot = next_slot(queue);
        if (slot >= n) break;
        const uint32_t q = ids ? ids[slot] : slot;
        const pba_pair pr = pairs[q];
        const PackedFetch fa = fetch_of(A, pr.a_seq, pr.a_pos, (pr.flags & PBA_A_BACKWARD) ? -1 : 1);
        const PackedFetch fb = fetch_of(B, pr.b_seq, pr.b_pos, (pr.flags & PBA_B_BACKWARD) ? -1 : 1);
        AlnOut o;
        int ne = 0;
        const uint64_t o0 = ops_off[q], o1 = ops_off[q + 1];
        OpSink sink{(uint8_t *)(mine + cap_words), 0, 0u};
        if (align_bitvec_trace<NB>(fa, pr.a_len, fb, pr.b_len, cfg.R, cfg.maxn, cfg.maxm, cfg.full_band != 0, (uint16_t *)lds,
                                   cfg.row_cap, mine, cap_words, 0, sink, o))
            ne = sink.finish(ops + o0, o1 - o0);
        store_result(out + q, o);
        if ((threadIdx.x & (PBA_WAVE - 1)) == 0) nedit[q] = ne;
    }
}

// The same sweep and walk, but the path goes straight into the vote boxes of an unlocked reference (consensus.h:
// VoteSink) -- ref_seq::try_align's align + OVERLAP_MIN gate + elect (ref_seq.h:264-267) for a batch, no script in
// memory.  a is the reference: pair.a_pos is the position the votes start at.
template <int NB>
__global__ void __launch_bounds__(PBA_WAVE * 4, NB <= 4 ? 4 : 2)
k_vote_pairs(SeqSetDev A, SeqSetDev B, const pba_pair *pairs, const uint32_t *ids, uint32_t n, AlignCfg cfg, int overlap_min,
             pba_result *out, uint32_t *scratch, uint64_t wave_words, uint64_t cap_words, ConsDev C, int beg, int pre, int post,
             uint32_t *queue) {
    extern __shared__ __align__(16) uint8_t lds_all[];
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / PBA_WAVE));
    uint8_t *lds = lds_all + (size_t)wave * cfg.row_cap * 2;
    uint32_t *mine = scratch + ((uint64_t)blockIdx.x * 4 + wave) * wave_words;
    for (;;) {
        const uint32_t slot = next_slot(queue);
        if (slot >= n) break;
        const uint32_t q = ids ? ids[slot] : slot;
        const pba_pair pr = pairs[q];
        const bool fwd = !(pr.flags & PBA_A_BACKWARD);
        const PackedFetch fa = fetch_of(A, pr.a_seq, pr.a_pos, fwd ? 1 : -1);
        const PackedFetch fb = fetch_of(B, pr.b_seq, pr.b_pos, (pr.flags & PBA_B_BACKWARD) ? -1 : 1);
        AlnOut o;
        VoteSink sink{C, beg + pr.a_pos, pre, post, fwd, fb, 0, 0, 0, 0u};
        if (align_bitvec_trace<NB>(fa, pr.a_len, fb, pr.b_len, cfg.R, cfg.maxn, cfg.maxm, cfg.full_band != 0, (uint16_t *)lds,
                                   cfg.row_cap, mine, cap_words, overlap_min, sink, o))
            sink.finish();
        store_result(out + q, o);
    }
}

// find_path (seq_aligner.h:214-233) walked iteratively from the goal cell; one thread per pair
__global__ void k_trace_walk(const pba_result *res, const uint8_t *par, const uint64_t *par_off, uint8_t *ops,
                             const uint64_t *ops_off, int32_t *nedit, uint32_t n) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    const pba_result r = res[q];
    if (r.rc < 0) { nedit[q] = 0; return; }
    const uint8_t *p = par + par_off[q];
    uint8_t *o = ops + ops_off[q];
    const uint64_t capq = ops_off[q + 1] - ops_off[q];
    const int md = r.max_dst, W = 2 * md + 1;
    int i = r.matlen_a, j = r.matlen_b;
    uint64_t k = 0;
    while (i > 0 || j > 0) {
        int src;
        if (j == 0) src = 3;                   // init_cell: (i,0) has parent DELETE, (0,j) INSERT (seq_aligner.h:140-147)
        else if (i == 0) src = 2;
        else src = p[(size_t)i * W + (j - i + md)];
        if (k < capq) o[k] = (uint8_t)src;
        ++k;
        if (src == 1) { --i; --j; } else if (src == 2) --j; else --i;
    }
    const uint64_t m = k < capq ? k : capq;
    for (uint64_t x = 0, y = m; x + 1 < y; ++x) { --y; const uint8_t t = o[x]; o[x] = o[y]; o[y] = t; }   // goal-first -> origin-first
    nedit[q] = (int32_t)k;
}

// ---------------------------------------------------------------------------------------------
// kernels: drivers.  One wavefront per read walks the reference's ordered candidate loop and
// stops at the first success, so the pairs it aligns are exactly the pairs the reference aligns.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ long long pair_cells(const AlnOut &o) {
    return band_cells(o.len_b, o.max_dst, o.fail_row ? o.fail_row : o.len_a);
}

// per-read side outputs of k_locate
struct LocAux {
    long long cells;     // band cells the reference would evaluate for this read
    int probe_hits;      // probes that found their key
    int redo;            // 1: a pair came back PBA_RC_UNCERTIFIED, the read must be re-run at full band
};

// locator.cpp:70-92
template <int NB>
__global__ void __launch_bounds__(PBA_WAVE * Wpb<NB>::v, Wpb<NB>::occ)
k_locate(IndexDev ix, SeqSetDev T, uint32_t tseq, SeqSetDev Rd, const uint32_t *ids, uint32_t n, int trials,
         int min_len, AlignCfg cfg, pba_loc_row *rows, LocAux *aux, uint32_t *queue) {
    extern __shared__ __align__(16) uint8_t lds_all[];
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / PBA_WAVE));   // wave-uniform on purpose: keeps the walk in SGPRs
    uint8_t *lds = lds_all + (size_t)wave * cfg.row_cap * 2;
    const PreThresholds pre_t(cfg.R);
    for (;;) {                                // persistent wavefront: pull the next read until the queue is dry
    const uint32_t slot = next_slot(queue);
    if (slot >= n) break;
    const uint32_t r = ids ? ids[slot] : slot;
    const int len = (int)Rd.len[r];
    int found = 0, fj = -1, fpos = -1, fcost = -1, fma = 0, fmb = 0, npairs = 0, nhit = 0, redo = 0;
    long long ncell = 0;
    if (len >= min_len) {                                                   // locator.cpp:72
        const PackedFetch rbase = fetch_of(Rd, r, 0, 1), tbase = fetch_of(T, tseq, 0, 1);
        const uint8_t *rseq = rbase.seq;
        const int clen = (int)T.len[tseq];
        for (int j = 0; j < trials && j < len && !found && !redo; ++j) {    // locator.cpp:74
            const uint32_t key = window_key(rseq, (uint32_t)j, (uint32_t)len) & ix.mask;   // locator.cpp:75
            if (key == 0) continue;                                         // never inserted, locator.cpp:64
            uint32_t beg, cnt;
            ix_find(ix, key, beg, cnt);                                     // locator.cpp:76
            if (cnt == 0) continue;
            ++nhit;
            // locator.cpp:79, 64 hits at a time: every lane runs the first 32 rows of its hit (prefilter.h), then the
            // hits are walked in list order -- the ones that failed there are done, the others get the wavefront
            for (uint32_t h0 = 0; h0 < cnt && !found && !redo; h0 += PBA_WAVE) {
                const uint32_t lane = threadIdx.x & (PBA_WAVE - 1), ng = min((uint32_t)PBA_WAVE, cnt - h0);
                const bool act = lane < ng;
                const int mypos = act ? ix_pos_of(ix, (uint32_t)ix.ent[beg + h0 + lane]) : 0;
                int myfr = 0;
                long long mycells = 0;
                if constexpr (NB != 0) {
                    AlnOut po;
                    myfr = prefilter32(act, rbase.at(j, 1), len - j, tbase.at(mypos, 1), clen - mypos, cfg.R,
                                       cfg.maxn, cfg.maxm, pre_t, po);
                    mycells = myfr ? band_cells(po.len_b, po.max_dst, myfr) : 0;
                }
                for (uint32_t hh = 0; hh < ng; ++hh) {
                    const int fr = __builtin_amdgcn_readlane(myfr, (int)hh);
                    if (fr) {                                               // failed at row fr <= 32: seq_aligner.h:185
                        ++npairs;
                        ncell += ((long long)__builtin_amdgcn_readlane((int)(mycells >> 32), (int)hh) << 32) |
                                 (unsigned)__builtin_amdgcn_readlane((int)mycells, (int)hh);
                        continue;
                    }
                    const int pos = __builtin_amdgcn_readlane(mypos, (int)hh);
                    const PackedFetch fa = rbase.at(j, 1);                  // a = read from j   (locator.cpp:78)
                    const PackedFetch fb = tbase.at(pos, 1);                // b = contig from pos (locator.cpp:80)
                    AlnOut o;
                    align_dispatch<NB>(fa, len - j, fb, clen - pos, cfg, lds, o);
                    if (o.rc == PBA_RC_UNCERTIFIED) { redo = 1; break; }
                    ++npairs;
                    ncell += pair_cells(o);
                    if (o.rc > 0) {                                         // locator.cpp:82
                        found = 1; fj = j; fpos = pos; fcost = o.cost; fma = o.matlen_a; fmb = o.matlen_b;
                        break;
                    }
                }
            }
        }
    }
    if ((threadIdx.x & (PBA_WAVE - 1)) == 0) {
        pba_loc_row *row = rows + r;          // read / nseq are filled by the host
        row->found = found; row->j = fj; row->pos = fpos; row->cost = fcost;
        row->seglen = found ? len - fj : 0; row->matlen_a = fma; row->matlen_b = fmb; row->n_pairs = npairs;
        aux[r].cells = ncell; aux[r].probe_hits = nhit; aux[r].redo = redo;
    }
    }
}

// one locked round of spaced_seed.cpp:420-437 (try_align :262-298, ref_seq::try_align ref_seq.h:259-265)
__device__ __forceinline__ uint32_t seed_at_dev(const uint8_t *payload, int pos, uint32_t len, int buggy) {
    if (buggy && (pos & 3) == 0) return ld_u32(payload + pos);   // dna_seq.h:64: pos used as a byte offset
    return window_key(payload, (uint32_t)pos, len);
}

struct SsState {
    int found, dir, ref_pos, cost, ma, mb, ntrials, npairs, redo;
    int touch;      // 1: a forward candidate whose reference accessor ends within reach of the alignment (its outcome depends
                    // on where the reference text ends: ref_seq.h:268 growth), 2: a backward one (where it begins)
};

template <int NB>
__device__ __forceinline__ bool ss_try(const IndexDev &ix, const PackedFetch &refb, int ref_len, int ref_org, const PackedFetch &readb,
                                       int slen, int pos, int dir, int overlap_min, int buggy, const AlignCfg &cfg,
                                       const PreThresholds &pre_t, void *lds, SsState &st) {
    const uint8_t *rseq = readb.seq;
    if (pos < 0 || pos + 16 > slen) return false;   // the reference only keeps reads > 500 bases
    const uint32_t key = seed_at_dev(rseq, pos, (uint32_t)slen, buggy) & ix.mask;   // spaced_seed.cpp:265
    if (key == 0) return false;
    uint32_t beg, cnt;
    ix_find(ix, key, beg, cnt);
    if (cnt == 0) return false;
    ++st.ntrials;
    const bool fwd = dir == 1;
    const int s_off = fwd ? pos : pos + 15;                        // spaced_seed.cpp:274
    const int s_len = fwd ? slen - s_off : s_off + 1;              // spaced_seed.cpp:275
    if (s_len < overlap_min) return false;                         // spaced_seed.cpp:280
    for (uint32_t h0 = 0; h0 < cnt; h0 += PBA_WAVE) {                 // 64 hits at a time through the prefilter, then in list order
        const uint32_t lane = threadIdx.x & (PBA_WAVE - 1), ng = min((uint32_t)PBA_WAVE, cnt - h0);
        const bool act = lane < ng;
        const int myhit = act ? ix_pos_of(ix, (uint32_t)ix.ent[beg + h0 + lane]) + ref_org : 0;   // index positions count from `beg`
        int myfr = 0;
        {   // seq_aligner.h:94-102: the accessor's own length matters only below len_b + max_dst
            const int my_rlen = fwd ? ref_len - myhit : myhit + 16;
            if (__builtin_amdgcn_ballot_w64(act && my_rlen <= s_len + 1 + (int)((double)s_len * cfg.R)) != 0ull) st.touch |= fwd ? 1 : 2;
        }
        if constexpr (NB != 0) {
            const int r_off = fwd ? myhit : myhit + 15;
            AlnOut po;
            myfr = prefilter32(act, refb.at(r_off, fwd ? 1 : -1), fwd ? ref_len - r_off : r_off + 1,
                               readb.at(s_off, fwd ? 1 : -1), s_len, cfg.R, cfg.maxn, cfg.maxm, pre_t, po);
        }
        for (uint32_t hh = 0; hh < ng; ++hh) {
            if (__builtin_amdgcn_readlane(myfr, (int)hh)) { ++st.npairs; continue; }   // failed within its first 32 rows
            const int hit = __builtin_amdgcn_readlane(myhit, (int)hh);
            const int r_off = fwd ? hit : hit + 15;                    // spaced_seed.cpp:285
            const int r_len = fwd ? ref_len - r_off : r_off + 1;       // ref_seq.h:284-285
            const PackedFetch fa = refb.at(r_off, fwd ? 1 : -1);       // a = reference (ref_seq.h:264)
            const PackedFetch fb = readb.at(s_off, fwd ? 1 : -1);
            AlnOut o;
            align_dispatch<NB>(fa, r_len, fb, s_len, cfg, lds, o);
            if (o.rc == PBA_RC_UNCERTIFIED) { st.redo = 1; return true; }
            ++st.npairs;
            if (o.rc < 0) continue;                                    // ref_seq.h:264
            if (o.matlen_a < overlap_min) continue;                    // ref_seq.h:265
            st.found = 1; st.dir = dir; st.ref_pos = hit - ref_org; st.cost = o.cost; st.ma = o.matlen_a; st.mb = o.matlen_b;
            return true;
        }
    }
    return false;
}

// ref_org: where position 0 of the index (ref_seq's `beg`) sits inside Rf[rseq_id] -- 0 for a locked reference; beg - pre
// when the text is an unlocked reference that has grown before its origin (ref_seq.h:235-242).
// redo[r]: bit 0 = re-run at the reference band, bits 2:1 = SsState::touch.
template <int NB>
__global__ void __launch_bounds__(PBA_WAVE * Wpb<NB>::v, Wpb<NB>::occ)
k_spaced_round(IndexDev ix, SeqSetDev Rf, uint32_t rseq_id, int ref_org, SeqSetDev Rd, const uint32_t *ids, uint32_t n,
               int max_trial, int overlap_min, int buggy, AlignCfg cfg, pba_ss_row *rows, int *redo, uint32_t *queue) {
    extern __shared__ __align__(16) uint8_t lds_all[];
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / PBA_WAVE));   // wave-uniform on purpose: keeps the walk in SGPRs
    uint8_t *lds = lds_all + (size_t)wave * cfg.row_cap * 2;
    const PreThresholds pre_t(cfg.R);
    for (;;) {
    const uint32_t slot = next_slot(queue);
    if (slot >= n) break;
    const uint32_t r = ids ? ids[slot] : slot;
    const PackedFetch ref = fetch_of(Rf, rseq_id, 0, 1), rseq = fetch_of(Rd, r, 0, 1);
    const int ref_len = (int)Rf.len[rseq_id];
    const int slen = (int)Rd.len[r];
    SsState st = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int fj = -1;
    for (int j = 0; j < max_trial; ++j) {                          // spaced_seed.cpp:424-426
        if (ss_try<NB>(ix, ref, ref_len, ref_org, rseq, slen, j, 1, overlap_min, buggy, cfg, pre_t, lds, st) ||
            ss_try<NB>(ix, ref, ref_len, ref_org, rseq, slen, slen - j - 16, -1, overlap_min, buggy, cfg, pre_t, lds, st)) {
            fj = j;
            break;
        }
    }
    if ((threadIdx.x & (PBA_WAVE - 1)) == 0) {
        pba_ss_row *row = rows + r;
        row->read = (int32_t)r; row->found = st.found; row->j = st.found ? fj : -1; row->dir = st.dir;
        row->ref_pos = st.ref_pos; row->cost = st.cost; row->matlen_a = st.ma; row->matlen_b = st.mb;
        row->n_trials = st.ntrials; row->n_pairs = st.npairs;
        redo[r] = st.redo | (st.touch << 1);
    }
    }
}

// NB -> template instantiation.  K(NB) must expand to a statement launching the kernel.
#define PBA_DISPATCH_NB(nb, K) \
    switch (nb) {              \
        case 0: K(0); break;   \
        case 1: K(1); break;   \
        case 2: K(2); break;   \
        case 3: K(3); break;   \
        case 4: K(4); break;   \
        case 6: K(6); break;   \
        default: K(8); break;  \
    }

// ---------------------------------------------------------------------------------------------
// host API: context
// ---------------------------------------------------------------------------------------------
extern "C" {

int pba_ctx_create(int device_id, pba_ctx **out) {
    if (!out) return PBA_E_INVALID;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return PBA_E_NODEVICE;
    if (device_id < 0 || device_id >= ndev) return PBA_E_NODEVICE;
    pba_ctx *ctx = new (std::nothrow) pba_ctx();
    if (!ctx) return PBA_E_NOMEM;
    ctx->device = device_id;
    ctx->err[0] = 0;
    if (hipSetDevice(device_id) != hipSuccess || hipGetDeviceProperties(&ctx->prop, device_id) != hipSuccess) {
        delete ctx;
        return PBA_E_NODEVICE;
    }
    if (strncmp(ctx->prop.gcnArchName, "gfx950", 6) != 0) {   // the code object holds gfx950 ISA only
        delete ctx;
        return PBA_E_NODEVICE;
    }
    if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return PBA_E_HIP;
    }
    ctx->stream = ctx->own_stream;
    for (int i = 0; i < 6; ++i)
        if (hipEventCreate(&ctx->ev[i]) != hipSuccess) { delete ctx; return PBA_E_HIP; }
    memset(&ctx->prof, 0, sizeof ctx->prof);
    if (hipMalloc((void **)&ctx->d_queue, 64) != hipSuccess) { delete ctx; return PBA_E_NOMEM; }
    ctx->d_scratch = nullptr; ctx->scratch_bytes = 0;
    // kernels that take more than the default 64 KB of dynamic LDS
    const int big = 160 * 1024;
    (void)hipFuncSetAttribute((const void *)k_part_sort, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    (void)hipFuncSetAttribute((const void *)k_align_pairs<0>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    (void)hipFuncSetAttribute((const void *)k_align_bytes, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    (void)hipFuncSetAttribute((const void *)k_align_bytes_trace, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    (void)hipFuncSetAttribute((const void *)k_align_pairs_trace, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    (void)hipFuncSetAttribute((const void *)k_locate<0>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    (void)hipFuncSetAttribute((const void *)k_spaced_round<0>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    *out = ctx;
    return PBA_OK;
}

void pba_ctx_destroy(pba_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipStreamDestroy(ctx->own_stream);
    for (int i = 0; i < 6; ++i) (void)hipEventDestroy(ctx->ev[i]);
    (void)hipFree(ctx->d_queue);
    if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
    delete ctx;
}

const char *pba_ctx_error(const pba_ctx *ctx) { return ctx ? ctx->err : "null ctx"; }

int pba_ctx_set_stream(pba_ctx *ctx, void *hip_stream) {
    if (!ctx) return PBA_E_INVALID;
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return PBA_OK;
}

int pba_ctx_sync(pba_ctx *ctx) {
    if (!ctx) return PBA_E_INVALID;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return PBA_OK;
}

int pba_ctx_last_profile(const pba_ctx *ctx, pba_profile *out) {
    if (!ctx || !out) return PBA_E_INVALID;
    *out = ctx->prof;
    return PBA_OK;
}

int pba_ctx_device_info(const pba_ctx *ctx, char *name, size_t cap, int *n_cu, int *clock_mhz, uint64_t *hbm) {
    if (!ctx) return PBA_E_INVALID;
    if (name && cap) snprintf(name, cap, "%s (%s)", ctx->prop.name, ctx->prop.gcnArchName);
    if (n_cu) *n_cu = ctx->prop.multiProcessorCount;
    if (clock_mhz) *clock_mhz = ctx->prop.clockRate / 1000;
    if (hbm) *hbm = (uint64_t)ctx->prop.totalGlobalMem;
    return PBA_OK;
}

// ---------------------------------------------------------------------------------------------
// host API: sequence sets
// ---------------------------------------------------------------------------------------------
static int seqs_alloc(pba_ctx *ctx, pba_seqs *s, uint64_t packed_bytes) {
    s->packed_bytes = packed_bytes;
    HIPCHK(hipMalloc((void **)&s->d_alloc, packed_bytes + 2 * kSlack));
    HIPCHK(hipMemsetAsync(s->d_alloc, 0, packed_bytes + 2 * kSlack, ctx->stream));
    s->d_packed = s->d_alloc + kSlack;
    HIPCHK(hipMalloc((void **)&s->d_off, sizeof(uint64_t) * (s->n + 1)));
    HIPCHK(hipMalloc((void **)&s->d_len, sizeof(uint32_t) * (s->n + 1)));
    HIPCHK(hipMemcpyAsync(s->d_off, s->h_off.data(), sizeof(uint64_t) * s->n, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(s->d_len, s->h_len.data(), sizeof(uint32_t) * s->n, hipMemcpyHostToDevice, ctx->stream));
    return PBA_OK;
}

// the bit planes of a set whose packed bytes, offsets and lengths are on the device (enqueued on the ctx's stream)
static int seqs_planes(pba_ctx *ctx, pba_seqs *s) {
    std::vector<uint64_t> poff(s->n + 1);
    uint64_t w = 0;
    for (uint32_t i = 0; i < s->n; ++i) { poff[i] = w; w += ((uint64_t)s->h_len[i] + 31) / 32; }
    poff[s->n] = w;
    s->plane_words = w + 2 * kPlaneSlack;
    HIPCHK(hipMalloc((void **)&s->d_planes, s->plane_words * 2 * sizeof(uint32_t)));
    HIPCHK(hipMemsetAsync(s->d_planes, 0, s->plane_words * 2 * sizeof(uint32_t), ctx->stream));
    HIPCHK(hipMalloc((void **)&s->d_poff, sizeof(uint64_t) * (s->n + 1)));
    HIPCHK(hipMemcpyAsync(s->d_poff, poff.data(), sizeof(uint64_t) * (s->n + 1), hipMemcpyHostToDevice, ctx->stream));
    if (w) {
        const uint64_t blocks = (w + 255) / 256;
        if (blocks > 0x7FFFFFFFull) PBA_FAIL(PBA_E_TOOLONG, "sequence set too large");
        hipLaunchKernelGGL(k_make_planes, dim3((uint32_t)blocks), dim3(256), 0, ctx->stream, s->d_packed, s->d_off, s->d_len,
                           s->d_poff, s->n, w, s->d_planes + 2 * kPlaneSlack);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));      // poff (host vector) must outlive the copy
    return PBA_OK;
}

void pba_seqs_destroy(pba_seqs *s) {
    if (!s) return;
    (void)hipSetDevice(s->ctx->device);
    if (s->d_planes) (void)hipFree(s->d_planes);
    if (s->d_poff) (void)hipFree(s->d_poff);
    if (s->d_alloc) (void)hipFree(s->d_alloc);
    if (s->d_off) (void)hipFree(s->d_off);
    if (s->d_len) (void)hipFree(s->d_len);
    delete s;
}

// shared tail of the two text constructors: d_text / d_toff are on the device, h_toff on the host
static int seqs_pack(pba_ctx *ctx, const uint8_t *d_text, const uint64_t *d_toff, const uint64_t *h_toff, uint32_t n,
                     int strict, pba_seqs **out) {
    pba_seqs *s = new (std::nothrow) pba_seqs();
    if (!s) PBA_FAIL(PBA_E_NOMEM, "pba_seqs");
    s->ctx = ctx; s->n = n; s->max_len = 0; s->non_acgt = false; s->d_alloc = nullptr; s->d_packed = nullptr; s->d_off = nullptr; s->d_len = nullptr; s->d_planes = nullptr; s->d_poff = nullptr; s->plane_words = 0;
    s->h_off.resize(n + 1); s->h_len.resize(n + 1);
    uint64_t pk = 0;
    for (uint32_t i = 0; i < n; ++i) {
        if (h_toff[i + 1] < h_toff[i] || h_toff[i + 1] - h_toff[i] > 0x7FFFFFF0ull) {
            delete s;
            PBA_FAIL(PBA_E_INVALID, "offsets must be non-decreasing and each sequence < 2^31 bases");
        }
        const uint32_t L = (uint32_t)(h_toff[i + 1] - h_toff[i]);
        s->h_off[i] = pk; s->h_len[i] = L;
        s->max_len = std::max(s->max_len, L);
        pk += (((uint64_t)L + 3) / 4 + 15) & ~15ull;     // every sequence starts 16-byte aligned
    }
    s->h_off[n] = pk; s->h_len[n] = 0;
    int st = seqs_alloc(ctx, s, pk);
    if (st != PBA_OK) { pba_seqs_destroy(s); return st; }
    DevBuf bad;
    if (hipMalloc(&bad.p, 4) != hipSuccess) { pba_seqs_destroy(s); PBA_FAIL(PBA_E_NOMEM, "hipMalloc"); }
    (void)hipMemsetAsync(bad.p, 0, 4, ctx->stream);
    // the kernel bisects over n+1 offsets: upload the end offset too
    (void)hipMemcpyAsync(s->d_off + n, &s->h_off[n], sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream);
    const uint64_t total_dwords = pk / 4;
    if (total_dwords) {
        const uint64_t blocks = (total_dwords + 255) / 256;
        if (blocks > 0x7FFFFFFFull) { pba_seqs_destroy(s); PBA_FAIL(PBA_E_TOOLONG, "sequence set too large"); }
        hipLaunchKernelGGL(k_pack_text, dim3((uint32_t)blocks), dim3(256), 0, ctx->stream, d_text, d_toff, s->d_off,
                           s->d_len, n, total_dwords, s->d_packed, strict, bad.as<uint32_t>());
    }
    uint32_t h_bad = 0;
    hipError_t e = hipMemcpyAsync(&h_bad, bad.p, 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) { pba_seqs_destroy(s); return ctx_fail(ctx, PBA_E_HIP, "k_pack_text", e); }
    if (strict && h_bad) { pba_seqs_destroy(s); PBA_FAIL(PBA_E_ALPHABET, "pba_seqs_from_text"); }
    s->non_acgt = h_bad != 0;
    st = seqs_planes(ctx, s);
    if (st != PBA_OK) { pba_seqs_destroy(s); return st; }
    *out = s;
    return PBA_OK;
}

int pba_seqs_from_text(pba_ctx *ctx, const char *text, const uint64_t *offsets, uint32_t n, int strict_acgt,
                       pba_seqs **out) {
    if (!ctx || !offsets || !out || (!text && n && offsets[n] > offsets[0])) return PBA_E_INVALID;
    *out = nullptr;
    HIPCHK(hipSetDevice(ctx->device));
    const uint64_t total = n ? offsets[n] : 0;
    DevBuf d_text, d_toff;
    HIPCHK(hipMalloc(&d_text.p, total + kSlack));
    HIPCHK(hipMalloc(&d_toff.p, sizeof(uint64_t) * (n + 1)));
    if (total) HIPCHK(hipMemcpyAsync(d_text.p, text, total, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d_toff.p, offsets, sizeof(uint64_t) * (n + 1), hipMemcpyHostToDevice, ctx->stream));
    return seqs_pack(ctx, d_text.as<uint8_t>(), d_toff.as<uint64_t>(), offsets, n, strict_acgt, out);
}

int pba_seqs_from_device_text(pba_ctx *ctx, const void *d_text, const void *d_offsets, uint32_t n, uint64_t total_bytes,
                              uint32_t max_len, pba_seqs **out) {
    (void)total_bytes; (void)max_len;
    if (!ctx || !d_offsets || !out || (!d_text && n)) return PBA_E_INVALID;
    *out = nullptr;
    HIPCHK(hipSetDevice(ctx->device));
    std::vector<uint64_t> h_toff(n + 1);
    HIPCHK(hipMemcpyAsync(h_toff.data(), d_offsets, sizeof(uint64_t) * (n + 1), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return seqs_pack(ctx, (const uint8_t *)d_text, (const uint64_t *)d_offsets, h_toff.data(), n, 0, out);
}

int pba_seqs_from_records(pba_ctx *ctx, const uint8_t *file, size_t file_len, uint32_t min_excl, uint32_t max_excl,
                          pba_seqs **out) {
    if (!ctx || !out || (!file && file_len)) return PBA_E_INVALID;
    *out = nullptr;
    HIPCHK(hipSetDevice(ctx->device));
    size_t total = 0;
    const size_t kept = pba_open_binary(file, file_len, min_excl, max_excl, nullptr, 0, &total);
    if (kept > 0x7FFFFFFFull) PBA_FAIL(PBA_E_TOOLONG, "too many records");
    std::vector<uint64_t> recs(kept + 1);
    pba_open_binary(file, file_len, min_excl, max_excl, recs.data(), kept, nullptr);
    pba_seqs *s = new (std::nothrow) pba_seqs();
    if (!s) PBA_FAIL(PBA_E_NOMEM, "pba_seqs");
    s->ctx = ctx; s->n = (uint32_t)kept; s->max_len = 0; s->non_acgt = false; s->d_alloc = nullptr; s->d_packed = nullptr; s->d_off = nullptr; s->d_len = nullptr; s->d_planes = nullptr; s->d_poff = nullptr; s->plane_words = 0;
    s->h_off.resize(kept + 1); s->h_len.resize(kept + 1);
    for (size_t i = 0; i < kept; ++i) {
        uint32_t L;
        memcpy(&L, file + recs[i], 4);
        if (recs[i] + 4 + ((uint64_t)L + 3) / 4 > file_len) { delete s; PBA_FAIL(PBA_E_INVALID, "truncated record"); }
        s->h_off[i] = recs[i] + 4;      // payload follows the u32 length (dna_seq.h:119-121)
        s->h_len[i] = L;
        s->max_len = std::max(s->max_len, L);
    }
    s->h_off[kept] = file_len; s->h_len[kept] = 0;
    // the file image goes up as it is (no re-packing); the slack after it is large enough for
    // seed_at's byte-offset reads to stay inside the allocation and read zeros
    const uint64_t slack = 65536;
    s->packed_bytes = file_len;
    hipError_t e = hipMalloc((void **)&s->d_alloc, file_len + slack + kSlack);
    if (e == hipSuccess) e = hipMemsetAsync(s->d_alloc, 0, file_len + slack + kSlack, ctx->stream);
    if (e == hipSuccess) s->d_packed = s->d_alloc + kSlack;
    if (e == hipSuccess && file_len) e = hipMemcpyAsync(s->d_packed, file, file_len, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_off, sizeof(uint64_t) * (kept + 1));
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_len, sizeof(uint32_t) * (kept + 1));
    if (e == hipSuccess) e = hipMemcpyAsync(s->d_off, s->h_off.data(), sizeof(uint64_t) * (kept + 1), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(s->d_len, s->h_len.data(), sizeof(uint32_t) * (kept + 1), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { pba_seqs_destroy(s); return ctx_fail(ctx, PBA_E_HIP, "pba_seqs_from_records", e); }
    const int stp = seqs_planes(ctx, s);
    if (stp != PBA_OK) { pba_seqs_destroy(s); return stp; }
    *out = s;
    return PBA_OK;
}

uint32_t pba_seqs_count(const pba_seqs *s) { return s ? s->n : 0; }
uint32_t pba_seqs_max_len(const pba_seqs *s) { return s ? s->max_len : 0; }
uint64_t pba_seqs_packed_bytes(const pba_seqs *s) { return s ? s->packed_bytes : 0; }

int pba_seqs_lengths(const pba_seqs *s, uint32_t *lengths, uint32_t cap) {
    if (!s || !lengths) return PBA_E_INVALID;
    for (uint32_t i = 0; i < s->n && i < cap; ++i) lengths[i] = s->h_len[i];
    return PBA_OK;
}

int pba_seqs_get_text(pba_ctx *ctx, const pba_seqs *s, uint32_t i, char *text, size_t cap) {
    if (!ctx || !s || !text || i >= s->n) return PBA_E_INVALID;
    const uint32_t L = s->h_len[i];
    if (cap <= L) return PBA_E_INVALID;
    HIPCHK(hipSetDevice(ctx->device));
    std::vector<uint8_t> pk(((size_t)L + 3) / 4 + 1);
    if (L) HIPCHK(hipMemcpyAsync(pk.data(), s->d_packed + s->h_off[i], ((size_t)L + 3) / 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    static const char base[4] = {'A', 'C', 'G', 'T'};
    for (uint32_t k = 0; k < L; ++k) text[k] = base[(pk[k >> 2] >> (6 - 2 * (k & 3))) & 3];
    text[L] = 0;
    return PBA_OK;
}

// ---------------------------------------------------------------------------------------------
// host API: seed index
// ---------------------------------------------------------------------------------------------
void pba_index_destroy(pba_index *ix) {
    if (!ix) return;
    (void)hipSetDevice(ix->ctx->device);
    if (ix->d_ent) (void)hipFree(ix->d_ent);
    if (ix->d_part_off) (void)hipFree(ix->d_part_off);
    delete ix;
}

uint64_t pba_index_entries(const pba_index *ix) { return ix ? ix->n_entries : 0; }
uint32_t pba_index_visited(const pba_index *ix) { return ix ? ix->visited : 0; }

// sort one oversize partition in global memory
static int sort_partition_global(pba_ctx *ctx, uint64_t *d_part, uint32_t n) {
    uint32_t N = 2;
    while (N < n) N <<= 1;
    DevBuf tmp;
    HIPCHK(hipMalloc(&tmp.p, sizeof(uint64_t) * N));
    HIPCHK(hipMemcpyAsync(tmp.p, d_part, sizeof(uint64_t) * n, hipMemcpyDeviceToDevice, ctx->stream));
    if (N > n)
        hipLaunchKernelGGL(k_fill_u64, dim3((N - n + 255) / 256), dim3(256), 0, ctx->stream, tmp.as<uint64_t>(), n, N,
                           ~0ull);
    for (uint32_t k = 2; k <= N; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1)
            hipLaunchKernelGGL(k_bitonic_step, dim3((N / 2 + 255) / 256), dim3(256), 0, ctx->stream, tmp.as<uint64_t>(),
                               N, k, j);
    HIPCHK(hipMemcpyAsync(d_part, tmp.p, sizeof(uint64_t) * n, hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return PBA_OK;
}

// the reference's visiting order as at most two position segments (+ what get_seedmap returns)
struct VisitPlan {
    ScanSeg segs[2];
    int nseg;
    uint32_t visited, nhead;
    int32_t tail_top;
};

static VisitPlan visit_plan(uint32_t len, int mode) {
    VisitPlan v;
    v.nseg = 0; v.visited = 0; v.nhead = 0xFFFFFFFFu; v.tail_top = 0;
    if (mode == PBA_INDEX_ALL) {                       // locator.cpp:62: for i in [0, len)
        if (len) v.segs[v.nseg++] = ScanSeg{0, len, 0, 0};
        v.visited = len;
    } else {                                           // ref_seq.h:291-311, MAX_READ_LEN = 20000, N_SEQ_WORD = 16
        const long long L = len, nmax = L - 16;
        const long long nh = std::min(nmax, 20000ll);
        const long long nt = std::min(L - 20000 - 16, 20000ll);
        v.nhead = nh > 0 ? (uint32_t)nh : 0;
        v.tail_top = (int32_t)(L - 16);
        if (nh > 0) v.segs[v.nseg++] = ScanSeg{0, (uint32_t)nh, 0, 0};
        if (nt > 0) v.segs[v.nseg++] = ScanSeg{(uint32_t)(L - 16 - nt + 1), (uint32_t)(L - 16 + 1), v.nhead, 1};
        v.visited = (uint32_t)(nh + (nt < 0 ? 0 : nt));   // ref_seq.h:310 (a negative nhead is added as it is)
    }
    return v;
}

static uint32_t seg_grid(const ScanSeg &sg) {
    const uint64_t chunks = ((uint64_t)sg.hi + 15) / 16 - sg.lo / 16;
    return (uint32_t)((chunks + PBA_IX_TILE_THREADS * PBA_IX_TILE_ITERS - 1) / (PBA_IX_TILE_THREADS * PBA_IX_TILE_ITERS));
}

static int index_logp(uint64_t n) {
    int logP = 0;
    while (logP < PBA_IX_MAX_LOGP && (n >> logP) > 1024) ++logP;
    return logP;
}

static pba_index *index_new(pba_ctx *ctx, uint32_t mask, uint32_t len, int mode, const VisitPlan &v) {
    pba_index *ix = new (std::nothrow) pba_index();
    if (!ix) return nullptr;
    ix->ctx = ctx; ix->mask = mask; ix->seq_len = len; ix->visited = v.visited; ix->nhead = v.nhead;
    ix->tail_top = v.tail_top; ix->mode = mode; ix->n_entries = 0; ix->d_ent = nullptr; ix->d_part_off = nullptr;
    ix->logP = 0;
    return ix;
}

// counts are in cnt (device, P+1 u32): turn them into offsets, allocate the entry array, let `scatter`
// fill it (cnt then holds the cursors), sort every partition
static int index_finish(pba_ctx *ctx, pba_index *ix, DevBuf &cnt, const std::function<void()> &scatter) {
    const uint32_t P = 1u << ix->logP;
    std::vector<uint32_t> h_cnt(P + 1, 0), h_off(P + 1, 0);
    HIPCHK(hipMemcpyAsync(h_cnt.data(), cnt.p, sizeof(uint32_t) * P, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipGetLastError());
    uint64_t total = 0;
    for (uint32_t p = 0; p < P; ++p) { h_off[p] = (uint32_t)total; total += h_cnt[p]; }
    if (total > 0xFFFFFFF0ull) PBA_FAIL(PBA_E_TOOLONG, "more than 2^32 index entries");
    h_off[P] = (uint32_t)total;
    ix->n_entries = total;
    HIPCHK(hipMalloc((void **)&ix->d_ent, sizeof(uint64_t) * (total + 1)));
    HIPCHK(hipMalloc((void **)&ix->d_part_off, sizeof(uint32_t) * (P + 1)));
    HIPCHK(hipMemcpyAsync(ix->d_part_off, h_off.data(), sizeof(uint32_t) * (P + 1), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(cnt.p, h_off.data(), sizeof(uint32_t) * (P + 1), hipMemcpyHostToDevice, ctx->stream));   // cursors
    if (total) {
        scatter();
        // LDS by need, not by capacity: a 2 048-entry partition takes 16 KB, so ten workgroups share a CU
        // instead of one (k_part_sort was 0.61 ms of a 0.77 ms build at 5 Mb with the full 128 KB request)
        uint32_t biggest = 2;
        for (uint32_t p = 0; p < P; ++p)
            if (h_cnt[p] <= PBA_IX_LDS_SORT_CAP) biggest = std::max(biggest, h_cnt[p]);
        uint32_t pow2 = 2;
        while (pow2 < biggest) pow2 <<= 1;
        hipLaunchKernelGGL(k_part_sort, dim3(P), dim3(256), sizeof(uint64_t) * pow2, ctx->stream, ix->d_ent,
                           ix->d_part_off);
        HIPCHK(hipStreamSynchronize(ctx->stream));
        HIPCHK(hipGetLastError());
        for (uint32_t p = 0; p < P; ++p)
            if (h_cnt[p] > PBA_IX_LDS_SORT_CAP) {
                int st = sort_partition_global(ctx, ix->d_ent + h_off[p], h_cnt[p]);
                if (st != PBA_OK) return st;
            }
    }
    (void)hipEventRecord(ctx->ev[1], ctx->stream);
    (void)hipEventSynchronize(ctx->ev[1]);
    (void)hipEventElapsedTime(&ctx->prof.index_ms, ctx->ev[0], ctx->ev[1]);
    return PBA_OK;
}

int pba_index_build(pba_ctx *ctx, const pba_seqs *target, uint32_t seq, uint32_t mask, int mode, pba_index **out) {
    if (!ctx || !target || !out || seq >= target->n) return PBA_E_INVALID;
    if (mode != PBA_INDEX_ALL && mode != PBA_INDEX_HEAD_TAIL) return PBA_E_INVALID;
    *out = nullptr;
    HIPCHK(hipSetDevice(ctx->device));
    const uint32_t len = target->h_len[seq];
    if (len > 0x7FFFFFF0u) PBA_FAIL(PBA_E_TOOLONG, "target sequence");
    const VisitPlan v = visit_plan(len, mode);
    pba_index *ix = index_new(ctx, mask, len, mode, v);
    if (!ix) PBA_FAIL(PBA_E_NOMEM, "pba_index");
    uint64_t npos = 0;
    for (int s = 0; s < v.nseg; ++s) npos += v.segs[s].hi - v.segs[s].lo;
    ix->logP = index_logp(npos);
    const int logP = ix->logP;
    const uint32_t P = 1u << logP;
    const uint8_t *d_seq = target->d_packed + target->h_off[seq];
    DevBuf cnt;
    hipError_t e = hipMalloc(&cnt.p, sizeof(uint32_t) * (P + 1));
    if (e == hipSuccess) e = hipMemsetAsync(cnt.p, 0, sizeof(uint32_t) * (P + 1), ctx->stream);
    if (e != hipSuccess) { pba_index_destroy(ix); return ctx_fail(ctx, PBA_E_HIP, "index alloc", e); }
    (void)hipEventRecord(ctx->ev[0], ctx->stream);
    for (int s = 0; s < v.nseg; ++s)
        hipLaunchKernelGGL(k_seed_count, dim3(seg_grid(v.segs[s])), dim3(PBA_IX_TILE_THREADS), 0, ctx->stream, d_seq, len,
                           mask, v.segs[s], logP, cnt.as<uint32_t>());
    int st = index_finish(ctx, ix, cnt, [&]() {
        for (int s = 0; s < v.nseg; ++s)
            hipLaunchKernelGGL(k_seed_scatter, dim3(seg_grid(v.segs[s])), dim3(PBA_IX_TILE_THREADS), 0, ctx->stream, d_seq,
                               len, mask, v.segs[s], logP, cnt.as<uint32_t>(), ix->d_ent);
    });
    if (st != PBA_OK) { pba_index_destroy(ix); return st; }
    *out = ix;
    return PBA_OK;
}

int pba_index_scan(pba_ctx *ctx, const pba_seqs *target, uint32_t seq, uint32_t mask, int mode, uint32_t part,
                   uint32_t nparts, void *d_entries, uint64_t cap, uint64_t *n_out) {
    if (!ctx || !target || !d_entries || !n_out || seq >= target->n || nparts == 0 || part >= nparts) return PBA_E_INVALID;
    if (mode != PBA_INDEX_ALL && mode != PBA_INDEX_HEAD_TAIL) return PBA_E_INVALID;
    HIPCHK(hipSetDevice(ctx->device));
    const uint32_t len = target->h_len[seq];
    const VisitPlan v = visit_plan(len, mode);
    // this rank's contiguous slice of the ordinal space [0, visited)
    const uint64_t nv = v.nseg ? (uint64_t)v.segs[v.nseg - 1].ord0 + (v.segs[v.nseg - 1].hi - v.segs[v.nseg - 1].lo) : 0;
    const uint64_t o_lo = nv * part / nparts, o_hi = nv * (part + 1) / nparts;
    DevBuf counter;
    HIPCHK(hipMalloc(&counter.p, 8));
    HIPCHK(hipMemsetAsync(counter.p, 0, 8, ctx->stream));
    const uint8_t *d_seq = target->d_packed + target->h_off[seq];
    for (int s = 0; s < v.nseg; ++s) {
        const ScanSeg &g = v.segs[s];
        const uint64_t g_lo = g.ord0, g_hi = (uint64_t)g.ord0 + (g.hi - g.lo);
        const uint64_t a = std::max(o_lo, g_lo), b = std::min(o_hi, g_hi);
        if (a >= b) continue;
        ScanSeg c;
        c.descending = g.descending; c.ord0 = (uint32_t)a;
        if (!g.descending) { c.lo = g.lo + (uint32_t)(a - g_lo); c.hi = g.lo + (uint32_t)(b - g_lo); }
        else { c.lo = g.hi - (uint32_t)(b - g_lo); c.hi = g.hi - (uint32_t)(a - g_lo); }
        hipLaunchKernelGGL(k_seed_emit, dim3(seg_grid(c)), dim3(PBA_IX_TILE_THREADS), 0, ctx->stream, d_seq, len, mask, c,
                           (uint64_t *)d_entries, (unsigned long long)cap, counter.as<unsigned long long>());
    }
    HIPCHK(hipGetLastError());
    unsigned long long h_n = 0;
    HIPCHK(hipMemcpyAsync(&h_n, counter.p, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (h_n > cap) PBA_FAIL(PBA_E_INVALID, "pba_index_scan: entry buffer too small");
    *n_out = h_n;
    return PBA_OK;
}

int pba_index_from_entries(pba_ctx *ctx, const void *d_entries, uint64_t n, uint32_t mask, int mode, uint32_t seq_len,
                           pba_index **out) {
    if (!ctx || !out || (!d_entries && n)) return PBA_E_INVALID;
    if (mode != PBA_INDEX_ALL && mode != PBA_INDEX_HEAD_TAIL) return PBA_E_INVALID;
    *out = nullptr;
    HIPCHK(hipSetDevice(ctx->device));
    const VisitPlan v = visit_plan(seq_len, mode);
    pba_index *ix = index_new(ctx, mask, seq_len, mode, v);
    if (!ix) PBA_FAIL(PBA_E_NOMEM, "pba_index");
    ix->logP = index_logp(n);
    const int logP = ix->logP;
    const uint32_t P = 1u << logP;
    DevBuf cnt;
    hipError_t e = hipMalloc(&cnt.p, sizeof(uint32_t) * (P + 1));
    if (e == hipSuccess) e = hipMemsetAsync(cnt.p, 0, sizeof(uint32_t) * (P + 1), ctx->stream);
    if (e != hipSuccess) { pba_index_destroy(ix); return ctx_fail(ctx, PBA_E_HIP, "index alloc", e); }
    const uint32_t grid = (uint32_t)((n + PBA_IX_TILE_POS - 1) / PBA_IX_TILE_POS);
    (void)hipEventRecord(ctx->ev[0], ctx->stream);
    if (grid)
        hipLaunchKernelGGL(k_ent_count, dim3(grid), dim3(PBA_IX_TILE_THREADS), 0, ctx->stream, (const uint64_t *)d_entries, n,
                           logP, cnt.as<uint32_t>());
    int st = index_finish(ctx, ix, cnt, [&]() {
        hipLaunchKernelGGL(k_ent_scatter, dim3(grid), dim3(PBA_IX_TILE_THREADS), 0, ctx->stream, (const uint64_t *)d_entries,
                           n, logP, cnt.as<uint32_t>(), ix->d_ent);
    });
    if (st != PBA_OK) { pba_index_destroy(ix); return st; }
    *out = ix;
    return PBA_OK;
}

int pba_index_dump(pba_ctx *ctx, const pba_index *ix, uint32_t *keys, int32_t *pos, uint64_t cap, uint64_t *n) {
    if (!ctx || !ix || !n) return PBA_E_INVALID;
    HIPCHK(hipSetDevice(ctx->device));
    *n = ix->n_entries;
    if (!keys || !pos) return PBA_OK;
    std::vector<uint64_t> ent(ix->n_entries + 1);
    if (ix->n_entries)
        HIPCHK(hipMemcpyAsync(ent.data(), ix->d_ent, sizeof(uint64_t) * ix->n_entries, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ent.resize(ix->n_entries);
    std::sort(ent.begin(), ent.end());          // partitions are sorted; this only merges them by key
    for (uint64_t i = 0; i < ent.size() && i < cap; ++i) {
        const uint32_t ord = (uint32_t)ent[i];
        keys[i] = (uint32_t)(ent[i] >> 32);
        pos[i] = ord < ix->nhead ? (int32_t)ord : ix->tail_top - (int32_t)(ord - ix->nhead);
    }
    return PBA_OK;
}

int pba_index_find(pba_ctx *ctx, const pba_index *ix, const uint32_t *keys, uint32_t n_keys, uint64_t *hit_off,
                   int32_t *hit_pos, uint64_t hit_cap) {
    if (!ctx || !ix || !hit_off || (!keys && n_keys)) return PBA_E_INVALID;
    HIPCHK(hipSetDevice(ctx->device));
    hit_off[0] = 0;
    if (!n_keys) return PBA_OK;
    DevBuf d_keys, d_beg, d_cnt, d_off, d_pos;
    HIPCHK(hipMalloc(&d_keys.p, 4ull * n_keys));
    HIPCHK(hipMalloc(&d_beg.p, 4ull * n_keys));
    HIPCHK(hipMalloc(&d_cnt.p, 4ull * n_keys));
    HIPCHK(hipMemcpyAsync(d_keys.p, keys, 4ull * n_keys, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_find_count, dim3((n_keys + 255) / 256), dim3(256), 0, ctx->stream, ix->dev(),
                       d_keys.as<uint32_t>(), n_keys, d_beg.as<uint32_t>(), d_cnt.as<uint32_t>());
    std::vector<uint32_t> cnt(n_keys);
    HIPCHK(hipMemcpyAsync(cnt.data(), d_cnt.p, 4ull * n_keys, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (uint32_t q = 0; q < n_keys; ++q) hit_off[q + 1] = hit_off[q] + cnt[q];
    const uint64_t total = hit_off[n_keys];
    if (!hit_pos || !total) return PBA_OK;
    const uint64_t ncopy = std::min(total, hit_cap);
    HIPCHK(hipMalloc(&d_off.p, 8ull * (n_keys + 1)));
    HIPCHK(hipMalloc(&d_pos.p, 4ull * (ncopy + 1)));
    HIPCHK(hipMemcpyAsync(d_off.p, hit_off, 8ull * (n_keys + 1), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_find_fill, dim3((n_keys + 255) / 256), dim3(256), 0, ctx->stream, ix->dev(), d_beg.as<uint32_t>(),
                       d_off.as<uint64_t>(), n_keys, d_pos.as<int32_t>(), ncopy);
    HIPCHK(hipMemcpyAsync(hit_pos, d_pos.p, 4ull * ncopy, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipGetLastError());
    return PBA_OK;
}

// ---------------------------------------------------------------------------------------------
// host API: alignment
// ---------------------------------------------------------------------------------------------
static int max_dst_of(int la, int lb, double R) {      // seq_aligner.h:94-102
    return 1 + (int)((lb >= la ? la : lb) * R);
}

// Launch plan for a batch whose widest band is max_dst_max.
struct Plan {
    AlignCfg cfg;
    size_t lds;
    int nb1;     // first launch: 0 = row sweep, else bit-vector array with nb1 blocks per lane (narrow band)
    int nb2;     // second launch (uncertified pairs only): bit-vector array at the reference band
};

static int make_plan(pba_ctx *ctx, double R, int maxn, int maxm, int kernel, int max_dst_max, Plan *pl) {
    if (!(R > 0.0) || !(R < 1.0)) PBA_FAIL(PBA_E_INVALID, "R must be in (0,1)");
    if (kernel != PBA_KERNEL_AUTO && kernel != PBA_KERNEL_ROWSWEEP && kernel != PBA_KERNEL_BITVEC)
        PBA_FAIL(PBA_E_INVALID, "unknown kernel");
    const bool bv = kernel != PBA_KERNEL_ROWSWEEP && bitvec_supports(max_dst_max);
    if (kernel == PBA_KERNEL_BITVEC && !bv) PBA_FAIL(PBA_E_TOOLONG, "band too wide for the bit-vector kernel");
    // the bit-vector kernel needs LDS only for its m <= 10 corner (a 23-cell row at most); the row sweep
    // needs the whole band row
    const long long W = bv ? 127 : 2ll * max_dst_max + 1;
    const long long bytes = ((W * 2 + 15) / 16) * 16;
    if (bytes > kRowSweepLdsCap) PBA_FAIL(PBA_E_TOOLONG, "band row does not fit the per-wavefront LDS budget");
    pl->cfg.R = R; pl->cfg.maxn = maxn; pl->cfg.maxm = maxm; pl->cfg.full_band = 0;
    pl->cfg.row_cap = (int)(bytes / 2);
    pl->lds = (size_t)bytes;
    pl->nb1 = bv ? bv_nb_for_span(bv_first_wl(max_dst_max) + bv_first_w(max_dst_max)) : 0;
    pl->nb2 = bv ? bv_nb_for_span(bv_full_wl(max_dst_max) + max_dst_max) : 0;
    return PBA_OK;
}

// Workgroups for a persistent launch: enough to fill every CU (up to 8 waves per SIMD, as many as the LDS
// allows), never more than there are work items.  Any residency works: the queue needs no co-residency.
static uint32_t persistent_grid(const pba_ctx *ctx, uint32_t n_items, int waves_per_wg, size_t lds_per_wave) {
    const uint32_t cus = (uint32_t)ctx->prop.multiProcessorCount;
    uint32_t wg_per_cu = 32u / (uint32_t)waves_per_wg;
    const size_t lds_wg = lds_per_wave * (size_t)waves_per_wg;
    if (lds_wg) wg_per_cu = std::min<uint32_t>(wg_per_cu, (uint32_t)std::max<size_t>(1, (160 * 1024) / lds_wg));
    const uint32_t need = (n_items + (uint32_t)waves_per_wg - 1) / (uint32_t)waves_per_wg;
    return std::max(1u, std::min(need, cus * wg_per_cu));
}

static void prof_finish(pba_ctx *ctx) {      // all launches of the call have completed (stream synchronised)
    (void)hipEventElapsedTime(&ctx->prof.align_ms, ctx->ev[2], ctx->ev[3]);
    if (ctx->prof.n_redo) (void)hipEventElapsedTime(&ctx->prof.align_redo_ms, ctx->ev[4], ctx->ev[5]);
}

static bool pair_ok(const pba_seqs *S, uint32_t seq, int pos, int len, bool backward) {
    if (seq >= S->n || len < 0 || len > kMaxSeqLen || pos < 0) return false;
    const long long L = S->h_len[seq];
    if (len == 0) return pos <= L;
    return backward ? (pos < L && pos - (len - 1) >= 0) : ((long long)pos + len <= L);
}

int pba_align_batch(pba_ctx *ctx, const pba_seqs *A, const pba_seqs *B, const pba_pair *pairs, size_t n, double R,
                    int maxn, int maxm, int kernel, pba_result *out) {
    if (!ctx || !A || !B || (!pairs && n) || (!out && n)) return PBA_E_INVALID;
    if (n == 0) return PBA_OK;
    if (n > 0x7FFFFFFFull) PBA_FAIL(PBA_E_INVALID, "too many pairs in one batch");
    if (A->non_acgt || B->non_acgt)
        PBA_FAIL(PBA_E_ALPHABET, "a sequence set holds bytes outside ACGT: the reference compares raw bytes, use pba_align_text");
    HIPCHK(hipSetDevice(ctx->device));
    int mdmax = 1;
    for (size_t q = 0; q < n; ++q) {
        const pba_pair &p = pairs[q];
        if (!pair_ok(A, p.a_seq, p.a_pos, p.a_len, p.flags & PBA_A_BACKWARD) ||
            !pair_ok(B, p.b_seq, p.b_pos, p.b_len, p.flags & PBA_B_BACKWARD))
            PBA_FAIL(PBA_E_INVALID, "pair outside its sequence (or longer than the engine limit)");
        if (R > 0.0 && R < 1.0) mdmax = std::max(mdmax, max_dst_of(p.a_len, p.b_len, R));
    }
    Plan pl;
    int st = make_plan(ctx, R, maxn, maxm, kernel, mdmax, &pl);
    if (st != PBA_OK) return st;
    DevBuf d_pairs, d_out, d_ids;
    HIPCHK(hipMalloc(&d_pairs.p, sizeof(pba_pair) * n));
    HIPCHK(hipMalloc(&d_out.p, sizeof(pba_result) * n));
    HIPCHK(hipMemcpyAsync(d_pairs.p, pairs, sizeof(pba_pair) * n, hipMemcpyHostToDevice, ctx->stream));
#define K_PAIRS(NBV)                                                                                               \
    (void)hipMemsetAsync(ctx->d_queue, 0, 4, ctx->stream);                                                         \
    hipLaunchKernelGGL(k_align_pairs<NBV>, dim3(persistent_grid(ctx, cnt, Wpb<NBV>::v, pl.lds)),                    \
                       dim3(PBA_WAVE * Wpb<NBV>::v), pl.lds * Wpb<NBV>::v, ctx->stream, A->dev(), B->dev(),         \
                       d_pairs.as<pba_pair>(), ids, cnt, pl.cfg, d_out.as<pba_result>(), ctx->d_queue)
    {
        const uint32_t cnt = (uint32_t)n;
        const uint32_t *ids = nullptr;
        (void)hipEventRecord(ctx->ev[2], ctx->stream);
        PBA_DISPATCH_NB(pl.nb1, K_PAIRS);
        (void)hipEventRecord(ctx->ev[3], ctx->stream);
        ctx->prof.nb_first = (uint32_t)pl.nb1; ctx->prof.n_first = cnt; ctx->prof.nb_redo = 0; ctx->prof.n_redo = 0;
        ctx->prof.align_redo_ms = 0.f;
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, d_out.p, sizeof(pba_result) * n, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    // pairs whose narrow pass could not certify the goal row go round again at the reference band
    std::vector<uint32_t> redo;
    for (size_t q = 0; q < n; ++q)
        if (out[q].rc == PBA_RC_UNCERTIFIED) redo.push_back((uint32_t)q);
    if (!redo.empty()) {
        HIPCHK(hipMalloc(&d_ids.p, sizeof(uint32_t) * redo.size()));
        HIPCHK(hipMemcpyAsync(d_ids.p, redo.data(), sizeof(uint32_t) * redo.size(), hipMemcpyHostToDevice, ctx->stream));
        pl.cfg.full_band = 1;
        const uint32_t cnt = (uint32_t)redo.size();
        const uint32_t *ids = d_ids.as<uint32_t>();
        (void)hipEventRecord(ctx->ev[4], ctx->stream);
        PBA_DISPATCH_NB(pl.nb2, K_PAIRS);
        (void)hipEventRecord(ctx->ev[5], ctx->stream);
        ctx->prof.nb_redo = (uint32_t)pl.nb2; ctx->prof.n_redo = cnt;
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(out, d_out.p, sizeof(pba_result) * n, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
#undef K_PAIRS
    prof_finish(ctx);
    return PBA_OK;
}

int pba_align_text(pba_ctx *ctx, const char *a, int a_fwd, int la, const char *b, int b_fwd, int lb, double R,
                   int maxn, int maxm, pba_result *out) {
    if (!ctx || !out || la < 0 || lb < 0 || (!a && la) || (!b && lb)) return PBA_E_INVALID;
    if (la > kMaxSeqLen || lb > kMaxSeqLen) PBA_FAIL(PBA_E_TOOLONG, "pba_align_text");
    HIPCHK(hipSetDevice(ctx->device));
    Plan pl;
    int st = make_plan(ctx, R, maxn, maxm, PBA_KERNEL_ROWSWEEP, max_dst_of(la, lb, R), &pl);
    if (st != PBA_OK) return st;
    // element k of a backward accessor is p[-k]: ship [p-(len-1), p] and point at its last byte
    const size_t oa = 0, ob = ((size_t)la + 31) & ~(size_t)15;
    DevBuf buf, d_out;
    HIPCHK(hipMalloc(&buf.p, ob + lb + 32));
    HIPCHK(hipMalloc(&d_out.p, sizeof(pba_result)));
    if (la) HIPCHK(hipMemcpyAsync(buf.as<uint8_t>() + oa, a_fwd ? a : a - (la - 1), la, hipMemcpyHostToDevice, ctx->stream));
    if (lb) HIPCHK(hipMemcpyAsync(buf.as<uint8_t>() + ob, b_fwd ? b : b - (lb - 1), lb, hipMemcpyHostToDevice, ctx->stream));
    const uint8_t *da = buf.as<uint8_t>() + oa + (a_fwd || !la ? 0 : la - 1);
    const uint8_t *db = buf.as<uint8_t>() + ob + (b_fwd || !lb ? 0 : lb - 1);
    hipLaunchKernelGGL(k_align_bytes, dim3(1), dim3(PBA_WAVE), pl.lds, ctx->stream, da, a_fwd ? 1 : -1, la, db,
                       b_fwd ? 1 : -1, lb, pl.cfg, d_out.as<pba_result>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, d_out.p, sizeof(pba_result), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return PBA_OK;
}

// ---- traceback
static const uint64_t kTraceBudget = 96ull << 30;      // parent codes / bits resident for one call (and at most 80 % of free HBM)

static uint64_t par_bytes_of(int la, int lb, double R) {           // (len_a + 1) * (2*max_dst + 1), seq_aligner.h:94-102
    const int md = max_dst_of(la, lb, R);
    const int len_a = lb >= la ? la : std::min(la, lb + md);
    return ((uint64_t)len_a + 1) * (2ull * md + 1);
}

int pba_align_text_trace(pba_ctx *ctx, const char *a, int a_fwd, int la, const char *b, int b_fwd, int lb, double R,
                         int maxn, int maxm, pba_result *out, uint8_t *ops, int32_t ops_cap, int32_t *nedit) {
    if (!ctx || !out || !nedit || la < 0 || lb < 0 || (!a && la) || (!b && lb) || (!ops && ops_cap) || ops_cap < 0)
        return PBA_E_INVALID;
    if (la > kMaxSeqLen || lb > kMaxSeqLen) PBA_FAIL(PBA_E_TOOLONG, "pba_align_text_trace");
    HIPCHK(hipSetDevice(ctx->device));
    Plan pl;
    int st = make_plan(ctx, R, maxn, maxm, PBA_KERNEL_ROWSWEEP, max_dst_of(la, lb, R), &pl);
    if (st != PBA_OK) return st;
    const uint64_t pb = par_bytes_of(la, lb, R);
    if (pb > kTraceBudget) PBA_FAIL(PBA_E_NOMEM, "parent codes exceed the traceback budget");
    const size_t oa = 0, ob = ((size_t)la + 31) & ~(size_t)15;
    DevBuf buf, d_out, d_par, d_ops, d_off, d_ne;
    HIPCHK(hipMalloc(&buf.p, ob + lb + 32));
    HIPCHK(hipMalloc(&d_out.p, sizeof(pba_result)));
    HIPCHK(hipMalloc(&d_par.p, pb + 16));
    HIPCHK(hipMalloc(&d_ops.p, (size_t)ops_cap + 16));
    HIPCHK(hipMalloc(&d_off.p, 4 * sizeof(uint64_t)));
    HIPCHK(hipMalloc(&d_ne.p, sizeof(int32_t)));
    if (la) HIPCHK(hipMemcpyAsync(buf.as<uint8_t>() + oa, a_fwd ? a : a - (la - 1), la, hipMemcpyHostToDevice, ctx->stream));
    if (lb) HIPCHK(hipMemcpyAsync(buf.as<uint8_t>() + ob, b_fwd ? b : b - (lb - 1), lb, hipMemcpyHostToDevice, ctx->stream));
    const uint64_t offs[4] = {0, (uint64_t)ops_cap, 0, 0};          // ops_off[0..1], par_off[0]
    HIPCHK(hipMemcpyAsync(d_off.p, offs, sizeof offs, hipMemcpyHostToDevice, ctx->stream));
    const uint8_t *da = buf.as<uint8_t>() + oa + (a_fwd || !la ? 0 : la - 1);
    const uint8_t *db = buf.as<uint8_t>() + ob + (b_fwd || !lb ? 0 : lb - 1);
    hipLaunchKernelGGL(k_align_bytes_trace, dim3(1), dim3(PBA_WAVE), pl.lds, ctx->stream, da, a_fwd ? 1 : -1, la, db,
                       b_fwd ? 1 : -1, lb, pl.cfg, d_out.as<pba_result>(), d_par.as<uint8_t>());
    hipLaunchKernelGGL(k_trace_walk, dim3(1), dim3(64), 0, ctx->stream, d_out.as<pba_result>(), d_par.as<uint8_t>(),
                       d_off.as<uint64_t>() + 2, d_ops.as<uint8_t>(), d_off.as<uint64_t>(), d_ne.as<int32_t>(), 1u);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, d_out.p, sizeof(pba_result), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(nedit, d_ne.p, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    const int32_t ncopy = std::min(*nedit, ops_cap);
    if (ncopy > 0) HIPCHK(hipMemcpyAsync(ops, d_ops.p, (size_t)ncopy, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return PBA_OK;
}

// scratch the traced bit-vector pass of one pair needs (u32 words): narrow first pass or reference band
static uint64_t trace_words_of(int la, int lb, double R, int nb, bool full_band) {
    const int md = max_dst_of(la, lb, R);
    const int len_a = lb >= la ? la : std::min(la, lb + md), len_b = lb >= la ? std::min(lb, la + md) : lb;
    const int m = std::min(len_a, len_b), n = std::max(len_a, len_b);
    if (m <= 10) return (((uint64_t)len_a + 1) * (2ull * md + 1) + 3) / 4;       // the row sweep's corner: byte codes
    return bv_trace_words(nb, m, n, full_band ? md : bv_pass1_w(md, nb));
}

struct pba_cons;
static int cons_vote_view(const pba_cons *c, ConsDev *dev, int *beg, int *pre, int *post);

// Edit scripts of a batch (vote == nullptr: ops / ops_off / nedit receive them) or their votes (vote != nullptr: the
// paths go straight into its boxes, gated by overlap_min; ops / ops_off / nedit unused).
static int trace_batch(pba_ctx *ctx, const pba_seqs *A, const pba_seqs *B, const pba_pair *pairs, size_t n, double R,
                       int maxn, int maxm, int kernel, pba_result *out, uint8_t *ops, const uint64_t *ops_off,
                       int32_t *nedit, const pba_cons *vote, int overlap_min) {
    if (!ctx || !A || !B || (!pairs && n) || (!out && n) || (!vote && ((!ops_off && n) || (!nedit && n)))) return PBA_E_INVALID;
    if (n == 0) return PBA_OK;
    if (n > 0x7FFFFFFFull) PBA_FAIL(PBA_E_INVALID, "too many pairs in one batch");
    if (A->non_acgt || B->non_acgt) PBA_FAIL(PBA_E_ALPHABET, "a sequence set holds bytes outside ACGT: use pba_align_text_trace");
    HIPCHK(hipSetDevice(ctx->device));
    int mdmax = 1;
    uint64_t ops_max = 0;
    for (size_t q = 0; q < n; ++q) {
        const pba_pair &p = pairs[q];
        if (!pair_ok(A, p.a_seq, p.a_pos, p.a_len, p.flags & PBA_A_BACKWARD) ||
            !pair_ok(B, p.b_seq, p.b_pos, p.b_len, p.flags & PBA_B_BACKWARD))
            PBA_FAIL(PBA_E_INVALID, "pair outside its sequence (or longer than the engine limit)");
        if (!vote && (ops_off[q + 1] < ops_off[q] || ops_off[q + 1] - ops_off[q] < (uint64_t)p.a_len + p.b_len))
            PBA_FAIL(PBA_E_INVALID, "ops_off must leave a_len + b_len slots per pair");
        if (R > 0.0 && R < 1.0) mdmax = std::max(mdmax, max_dst_of(p.a_len, p.b_len, R));
        ops_max = std::max(ops_max, (uint64_t)p.a_len + p.b_len);
    }
    Plan pl;
    int st = make_plan(ctx, R, maxn, maxm, kernel, mdmax, &pl);
    if (st != PBA_OK) return st;
    ConsDev vdev = {nullptr, nullptr, nullptr, nullptr};
    int vbeg = 0, vpre = 0, vpost = 0;
    if (vote) {
        if (pl.nb1 == 0) PBA_FAIL(PBA_E_TOOLONG, "votes from the walk need the bit-vector kernel (band too wide)");
        st = cons_vote_view(vote, &vdev, &vbeg, &vpre, &vpost);
        if (st != PBA_OK) return st;
        ops_max = 0;                                             // no goal-first temporary
    }
    const uint64_t ops_total = vote ? 0 : ops_off[n] - ops_off[0];
    DevBuf d_pairs, d_out, d_par, d_poff, d_ops, d_ooff, d_ne;
    HIPCHK(hipMalloc(&d_pairs.p, sizeof(pba_pair) * n));
    HIPCHK(hipMalloc(&d_out.p, sizeof(pba_result) * n));
    HIPCHK(hipMalloc(&d_ops.p, ops_total + 16));
    HIPCHK(hipMalloc(&d_ooff.p, sizeof(uint64_t) * (n + 1)));
    HIPCHK(hipMalloc(&d_ne.p, sizeof(int32_t) * n));
    std::vector<uint64_t> rel(n + 1, 0);
    if (!vote) for (size_t q = 0; q <= n; ++q) rel[q] = ops_off[q] - ops_off[0];
    HIPCHK(hipMemcpyAsync(d_pairs.p, pairs, sizeof(pba_pair) * n, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d_ooff.p, rel.data(), sizeof(uint64_t) * (n + 1), hipMemcpyHostToDevice, ctx->stream));
    size_t free_b = 0, total_b = 0;
    HIPCHK(hipMemGetInfo(&free_b, &total_b));
    free_b += ctx->scratch_bytes;                            // the scratch kept from earlier calls is ours to reuse
    uint64_t budget = std::min<uint64_t>(kTraceBudget, (uint64_t)(free_b / 10) * 8);
    if (const char *e = getenv("PBA_TRACE_BUDGET_GB"))       // tuning aid: HBM the parent bits / codes of one call may take
        budget = std::min<uint64_t>((uint64_t)atoll(e) << 30, (uint64_t)(free_b / 10) * 9);
    (void)hipEventRecord(ctx->ev[2], ctx->stream);
    if (pl.nb1 == 0) {
        // row sweep: one parent byte per band cell, every pair's codes resident at once
        std::vector<uint64_t> par_off(n + 1, 0);
        for (size_t q = 0; q < n; ++q)
            par_off[q + 1] = par_off[q] + ((par_bytes_of(pairs[q].a_len, pairs[q].b_len, R) + 15) & ~15ull);
        if (par_off[n] > budget) PBA_FAIL(PBA_E_NOMEM, "parent codes of this batch exceed the traceback budget: split it");
        HIPCHK(hipMalloc(&d_par.p, par_off[n] + 16));
        HIPCHK(hipMalloc(&d_poff.p, sizeof(uint64_t) * (n + 1)));
        HIPCHK(hipMemcpyAsync(d_poff.p, par_off.data(), sizeof(uint64_t) * (n + 1), hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(k_align_pairs_trace, dim3((uint32_t)n), dim3(PBA_WAVE), pl.lds, ctx->stream, A->dev(), B->dev(),
                           d_pairs.as<pba_pair>(), (uint32_t)n, pl.cfg, d_out.as<pba_result>(), d_par.as<uint8_t>(),
                           d_poff.as<uint64_t>());
        hipLaunchKernelGGL(k_trace_walk, dim3((uint32_t)((n + 63) / 64)), dim3(64), 0, ctx->stream, d_out.as<pba_result>(),
                           d_par.as<uint8_t>(), d_poff.as<uint64_t>(), d_ops.as<uint8_t>(), d_ooff.as<uint64_t>(),
                           d_ne.as<int32_t>(), (uint32_t)n);
        (void)hipEventRecord(ctx->ev[3], ctx->stream);
        ctx->prof.nb_first = 0; ctx->prof.n_first = (uint32_t)n; ctx->prof.nb_redo = 0; ctx->prof.n_redo = 0;
        ctx->prof.align_redo_ms = 0.f;
    } else {
        // bit-vector array: 2 bits per processed cell in a per-wavefront scratch area, walked by the same wavefront.
        // First launch: every pair, narrow window, scratch sized for it (so more wavefronts fit the budget); second
        // launch: the pairs that came back uncertified, reference band.
        ctx->prof.nb_first = (uint32_t)pl.nb1; ctx->prof.n_first = (uint32_t)n;
        ctx->prof.nb_redo = 0; ctx->prof.n_redo = 0; ctx->prof.align_redo_ms = 0.f;
        std::vector<uint32_t> redo;
        for (int pass = 0; pass < 2; ++pass) {
            const int nb = pass ? pl.nb2 : pl.nb1;
            const uint32_t cnt = pass ? (uint32_t)redo.size() : (uint32_t)n;
            uint64_t cap_words = 128;
            for (uint32_t k = 0; k < cnt; ++k) {
                const pba_pair &p = pairs[pass ? redo[k] : k];
                cap_words = std::max(cap_words, trace_words_of(p.a_len, p.b_len, R, nb, pass != 0));
            }
            cap_words = (cap_words + 63) & ~63ull;
            const uint64_t wave_words = cap_words + ((ops_max + 64 + 255) & ~255ull) / 4;
            uint32_t grid = persistent_grid(ctx, cnt, 4, pl.lds);
            grid = (uint32_t)std::min<uint64_t>(grid, budget / (wave_words * 4 * 4));
            if (grid == 0) PBA_FAIL(PBA_E_NOMEM, "one wavefront's parent bits exceed the traceback budget");
            DevBuf d_ids;
            const size_t need = (size_t)grid * 4 * wave_words * 4;
            if (need > ctx->scratch_bytes) {
                if (ctx->d_scratch) { HIPCHK(hipStreamSynchronize(ctx->stream)); (void)hipFree(ctx->d_scratch); }
                ctx->d_scratch = nullptr; ctx->scratch_bytes = 0;
                HIPCHK(hipMalloc(&ctx->d_scratch, need));
                ctx->scratch_bytes = need;
            }
            uint32_t *const d_scr = (uint32_t *)ctx->d_scratch;
            const uint32_t *ids = nullptr;
            if (pass) {
                HIPCHK(hipMalloc(&d_ids.p, sizeof(uint32_t) * cnt));
                HIPCHK(hipMemcpyAsync(d_ids.p, redo.data(), sizeof(uint32_t) * cnt, hipMemcpyHostToDevice, ctx->stream));
                ids = d_ids.as<uint32_t>();
                pl.cfg.full_band = 1;
            }
            HIPCHK(hipMemsetAsync(ctx->d_queue, 0, sizeof(uint32_t), ctx->stream));
            (void)hipEventRecord(ctx->ev[pass ? 4 : 2], ctx->stream);
#define K_TRACE(NBV)                                                                                                  \
    if (vote)                                                                                                         \
        hipLaunchKernelGGL(k_vote_pairs<NBV>, dim3(grid), dim3(PBA_WAVE * 4), pl.lds * 4, ctx->stream, A->dev(), B->dev(), \
                           d_pairs.as<pba_pair>(), ids, cnt, pl.cfg, overlap_min, d_out.as<pba_result>(),              \
                           d_scr, wave_words, cap_words, vdev, vbeg, vpre, vpost, ctx->d_queue);        \
    else                                                                                                              \
        hipLaunchKernelGGL(k_trace_pairs<NBV>, dim3(grid), dim3(PBA_WAVE * 4), pl.lds * 4, ctx->stream, A->dev(), B->dev(), \
                           d_pairs.as<pba_pair>(), ids, cnt, pl.cfg, d_out.as<pba_result>(), d_scr,     \
                           wave_words, cap_words, d_ops.as<uint8_t>(), d_ooff.as<uint64_t>(), d_ne.as<int32_t>(),      \
                           ctx->d_queue)
            switch (nb) {
                case 1: K_TRACE(1); break;
                case 2: K_TRACE(2); break;
                case 3: K_TRACE(3); break;
                case 4: K_TRACE(4); break;
                case 6: K_TRACE(6); break;
                default: K_TRACE(8); break;
            }
#undef K_TRACE
            (void)hipEventRecord(ctx->ev[pass ? 5 : 3], ctx->stream);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(out, d_out.p, sizeof(pba_result) * n, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipStreamSynchronize(ctx->stream));           // the scratch area is freed at the end of this pass
            if (pass) { ctx->prof.nb_redo = (uint32_t)nb; ctx->prof.n_redo = cnt; break; }
            for (size_t q = 0; q < n; ++q)
                if (out[q].rc == PBA_RC_UNCERTIFIED) redo.push_back((uint32_t)q);
            if (redo.empty()) break;
        }
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, d_out.p, sizeof(pba_result) * n, hipMemcpyDeviceToHost, ctx->stream));
    if (!vote) {
        HIPCHK(hipMemcpyAsync(nedit, d_ne.p, sizeof(int32_t) * n, hipMemcpyDeviceToHost, ctx->stream));
        if (ops_total) HIPCHK(hipMemcpyAsync(ops + ops_off[0], d_ops.p, ops_total, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    prof_finish(ctx);
    return PBA_OK;
}

int pba_align_batch_trace(pba_ctx *ctx, const pba_seqs *A, const pba_seqs *B, const pba_pair *pairs, size_t n, double R,
                          int maxn, int maxm, int kernel, pba_result *out, uint8_t *ops, const uint64_t *ops_off,
                          int32_t *nedit) {
    return trace_batch(ctx, A, B, pairs, n, R, maxn, maxm, kernel, out, ops, ops_off, nedit, nullptr, 0);
}

// ---------------------------------------------------------------------------------------------
// host API: drivers
// ---------------------------------------------------------------------------------------------
int pba_locate(pba_ctx *ctx, const pba_index *ix, const pba_seqs *target, uint32_t target_seq, const pba_seqs *reads,
               double R, int trials, int min_len, int maxn, int maxm, int kernel, pba_loc_row *rows,
               pba_loc_stats *stats) {
    if (!ctx || !ix || !target || !reads || !rows || target_seq >= target->n || trials < 0) return PBA_E_INVALID;
    if (ix->mode != PBA_INDEX_ALL || ix->seq_len != target->h_len[target_seq])
        PBA_FAIL(PBA_E_INVALID, "pba_locate needs a PBA_INDEX_ALL index of the target sequence");
    if (reads->max_len > (uint32_t)kMaxSeqLen) PBA_FAIL(PBA_E_TOOLONG, "read longer than the engine limit");
    if (target->non_acgt || reads->non_acgt)   // locator.cpp compares raw bytes (an 'N' only matches an 'N'); codes would match it to T
        PBA_FAIL(PBA_E_ALPHABET, "pba_locate: a sequence set holds bytes outside ACGT");
    HIPCHK(hipSetDevice(ctx->device));
    const uint32_t n = reads->n;
    Plan pl;
    int st = make_plan(ctx, R, maxn, maxm, kernel, 1 + (int)(reads->max_len * R), &pl);
    if (st != PBA_OK) return st;
    DevBuf d_rows, d_aux, d_ids;
    HIPCHK(hipMalloc(&d_rows.p, sizeof(pba_loc_row) * (n + 1)));
    HIPCHK(hipMalloc(&d_aux.p, sizeof(LocAux) * (n + 1)));
    std::vector<LocAux> aux(n + 1);
#define K_LOC(NBV)                                                                                                   \
    (void)hipMemsetAsync(ctx->d_queue, 0, 4, ctx->stream);                                                           \
    hipLaunchKernelGGL(k_locate<NBV>, dim3(persistent_grid(ctx, cnt, Wpb<NBV>::v, pl.lds)),                           \
                       dim3(PBA_WAVE * Wpb<NBV>::v), pl.lds * Wpb<NBV>::v, ctx->stream, ix->dev(), target->dev(),     \
                       target_seq, reads->dev(), ids, cnt, trials, min_len, pl.cfg, d_rows.as<pba_loc_row>(),        \
                       d_aux.as<LocAux>(), ctx->d_queue)
    if (n) {
        const uint32_t cnt = n;
        const uint32_t *ids = nullptr;
        (void)hipEventRecord(ctx->ev[2], ctx->stream);
        PBA_DISPATCH_NB(pl.nb1, K_LOC);
        (void)hipEventRecord(ctx->ev[3], ctx->stream);
        ctx->prof.nb_first = (uint32_t)pl.nb1; ctx->prof.n_first = cnt; ctx->prof.nb_redo = 0; ctx->prof.n_redo = 0;
        ctx->prof.align_redo_ms = 0.f;
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(aux.data(), d_aux.p, sizeof(LocAux) * n, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        std::vector<uint32_t> redo;
        for (uint32_t r = 0; r < n; ++r)
            if (aux[r].redo) redo.push_back(r);
        if (!redo.empty()) {      // reads with an uncertified pair: walk them again at the reference band
            HIPCHK(hipMalloc(&d_ids.p, sizeof(uint32_t) * redo.size()));
            HIPCHK(hipMemcpyAsync(d_ids.p, redo.data(), sizeof(uint32_t) * redo.size(), hipMemcpyHostToDevice, ctx->stream));
            pl.cfg.full_band = 1;
            const uint32_t cnt = (uint32_t)redo.size();
            const uint32_t *ids = d_ids.as<uint32_t>();
            (void)hipEventRecord(ctx->ev[4], ctx->stream);
            PBA_DISPATCH_NB(pl.nb2, K_LOC);
            (void)hipEventRecord(ctx->ev[5], ctx->stream);
            ctx->prof.nb_redo = (uint32_t)pl.nb2; ctx->prof.n_redo = cnt;
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(aux.data(), d_aux.p, sizeof(LocAux) * n, hipMemcpyDeviceToHost, ctx->stream));
        }
        HIPCHK(hipMemcpyAsync(rows, d_rows.p, sizeof(pba_loc_row) * n, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
#undef K_LOC
    if (n) prof_finish(ctx);
    pba_loc_stats s = {0, 0, 0, 0, 0};
    int nseq = 0;
    for (uint32_t r = 0; r < n; ++r) {
        rows[r].read = (int32_t)r;
        rows[r].nseq = (int)reads->h_len[r] < min_len ? -1 : nseq++;      // locator.cpp:72,91
        if (rows[r].nseq >= 0) ++s.n_reads_kept;
        s.n_pairs += rows[r].n_pairs;
        s.n_located += rows[r].found;
        s.n_probe_hits += aux[r].probe_hits;
        s.n_cells += aux[r].cells;
    }
    if (stats) *stats = s;
    return PBA_OK;
}

// One locked round over the reads `subset` (host ids; nullptr = every read).  rows is indexed by read id: rows of
// reads outside the subset are left untouched.
// Unlocked rounds (pba_cons_round) pass the grown text: ref_org = the index's position 0 inside it, maxn / maxm = the
// size guard of the caller's t_aligner, touch[read id] = SsState::touch of the reads walked.
static int spaced_round_subset(pba_ctx *ctx, const pba_index *ix, const pba_seqs *ref, uint32_t ref_seq, const pba_seqs *reads,
                               double R, int max_trial, int overlap_min, int buggy_seed_at, int kernel,
                               const uint32_t *subset, uint32_t n_subset, pba_ss_row *rows, int ref_org = 0, int maxn = 0,
                               int maxm = 0, uint8_t *touch = nullptr) {
    if (!ctx || !ix || !ref || !reads || !rows || ref_seq >= ref->n || max_trial < 0) return PBA_E_INVALID;
    if (ix->mode != PBA_INDEX_HEAD_TAIL || (!touch && ix->seq_len != ref->h_len[ref_seq]) || ref_org < 0 ||
        (uint64_t)ref_org + ix->seq_len > ref->h_len[ref_seq])
        PBA_FAIL(PBA_E_INVALID, "pba_spaced_round needs a PBA_INDEX_HEAD_TAIL index of the reference sequence");
    if (reads->max_len > (uint32_t)kMaxSeqLen || ref->h_len[ref_seq] > 0x7FFFFFF0u)
        PBA_FAIL(PBA_E_TOOLONG, "sequence longer than the engine limit");
    if (ref->non_acgt || reads->non_acgt) PBA_FAIL(PBA_E_ALPHABET, "pba_spaced_round: a sequence set holds bytes outside ACGT");
    HIPCHK(hipSetDevice(ctx->device));
    const uint32_t n = reads->n, n_first = subset ? n_subset : n;
    Plan pl;
    // a = reference window, b = read window: the shorter side bounds max_dst (seq_aligner.h:94-102)
    int st = make_plan(ctx, R, maxn, maxm, kernel, 1 + (int)(reads->max_len * R), &pl);
    if (st != PBA_OK) return st;
    DevBuf d_rows, d_redo, d_ids, d_sub;
    HIPCHK(hipMalloc(&d_rows.p, sizeof(pba_ss_row) * (n + 1)));
    HIPCHK(hipMalloc(&d_redo.p, sizeof(int) * (n + 1)));
    HIPCHK(hipMemsetAsync(d_redo.p, 0, sizeof(int) * (n + 1), ctx->stream));
    if (subset && n_subset) {
        HIPCHK(hipMalloc(&d_sub.p, sizeof(uint32_t) * n_subset));
        HIPCHK(hipMemcpyAsync(d_sub.p, subset, sizeof(uint32_t) * n_subset, hipMemcpyHostToDevice, ctx->stream));
    }
#define K_SS(NBV)                                                                                                    \
    (void)hipMemsetAsync(ctx->d_queue, 0, 4, ctx->stream);                                                           \
    hipLaunchKernelGGL(k_spaced_round<NBV>, dim3(persistent_grid(ctx, cnt, Wpb<NBV>::v, pl.lds)),                     \
                       dim3(PBA_WAVE * Wpb<NBV>::v), pl.lds * Wpb<NBV>::v, ctx->stream, ix->dev(), ref->dev(),       \
                       ref_seq, ref_org, reads->dev(), ids, cnt, max_trial, overlap_min, buggy_seed_at, pl.cfg,      \
                       d_rows.as<pba_ss_row>(), d_redo.as<int>(), ctx->d_queue)
    if (n_first) {
        std::vector<int> h_redo(n);
        std::vector<pba_ss_row> h_rows(n);
        const uint32_t cnt = n_first;
        const uint32_t *ids = subset ? d_sub.as<uint32_t>() : nullptr;
        (void)hipEventRecord(ctx->ev[2], ctx->stream);
        PBA_DISPATCH_NB(pl.nb1, K_SS);
        (void)hipEventRecord(ctx->ev[3], ctx->stream);
        ctx->prof.nb_first = (uint32_t)pl.nb1; ctx->prof.n_first = cnt; ctx->prof.nb_redo = 0; ctx->prof.n_redo = 0;
        ctx->prof.align_redo_ms = 0.f;
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(h_redo.data(), d_redo.p, sizeof(int) * n, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        std::vector<uint32_t> redo;
        for (uint32_t r = 0; r < n; ++r)
            if (h_redo[r] & 1) redo.push_back(r);
        if (!redo.empty()) {
            HIPCHK(hipMalloc(&d_ids.p, sizeof(uint32_t) * redo.size()));
            HIPCHK(hipMemcpyAsync(d_ids.p, redo.data(), sizeof(uint32_t) * redo.size(), hipMemcpyHostToDevice, ctx->stream));
            pl.cfg.full_band = 1;
            const uint32_t cnt = (uint32_t)redo.size();
            const uint32_t *ids = d_ids.as<uint32_t>();
            (void)hipEventRecord(ctx->ev[4], ctx->stream);
            PBA_DISPATCH_NB(pl.nb2, K_SS);
            (void)hipEventRecord(ctx->ev[5], ctx->stream);
            ctx->prof.nb_redo = (uint32_t)pl.nb2; ctx->prof.n_redo = cnt;
            HIPCHK(hipGetLastError());
        }
        HIPCHK(hipMemcpyAsync(h_rows.data(), d_rows.p, sizeof(pba_ss_row) * n, hipMemcpyDeviceToHost, ctx->stream));
        if (touch && !redo.empty()) HIPCHK(hipMemcpyAsync(h_redo.data(), d_redo.p, sizeof(int) * n, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        if (touch) for (uint32_t r = 0; r < n; ++r) touch[r] = (uint8_t)((h_redo[r] >> 1) & 3);
        if (!subset) memcpy(rows, h_rows.data(), sizeof(pba_ss_row) * n);
        else for (uint32_t k = 0; k < n_subset; ++k) rows[subset[k]] = h_rows[subset[k]];
        prof_finish(ctx);
    }
#undef K_SS
    return PBA_OK;
}

int pba_spaced_round(pba_ctx *ctx, const pba_index *ix, const pba_seqs *ref, uint32_t ref_seq, const pba_seqs *reads,
                     double R, int max_trial, int overlap_min, int buggy_seed_at, int kernel, pba_ss_row *rows) {
    return spaced_round_subset(ctx, ix, ref, ref_seq, reads, R, max_trial, overlap_min, buggy_seed_at, kernel, nullptr, 0, rows);
}

// spaced_seed.cpp:409-452 for a locked reference (-l): rounds over the reads not found yet, the seed of a round drawn
// like the reference draws it (a fresh draw after a round that found something, else the seeds in file order), stop
// when every seed has failed in a row or after max_round.  picks[] stands in for the values rand() returns.
int pba_spaced_multi(pba_ctx *ctx, const pba_seqs *ref, uint32_t ref_seq, const pba_seqs *reads, double R, int max_trial,
                     int overlap_min, int buggy_seed_at, int kernel, const uint32_t *masks, int n_masks,
                     const uint32_t *picks, int n_picks, int max_round, pba_ss_row *rows, int32_t *found_round,
                     pba_ss_round_log *log, int log_cap, int *n_rounds) {
    if (!ctx || !ref || !reads || !masks || n_masks < 1 || !picks || n_picks < 1 || max_round < 0 || !rows || !found_round ||
        !n_rounds || log_cap < 0 || (!log && log_cap) || ref_seq >= ref->n)
        return PBA_E_INVALID;
    const uint32_t n = reads->n;
    std::vector<uint32_t> pool(n);
    for (uint32_t r = 0; r < n; ++r) { pool[r] = r; found_round[r] = 0; memset(&rows[r], 0, sizeof rows[r]); rows[r].read = (int32_t)r; rows[r].j = -1; }
    int nfailure = 0, draws = 0, done = 0;
    for (int nround = 1; nround <= max_round; ++nround) {
        const uint32_t mask = nfailure == 0 ? masks[picks[draws++ % n_picks] % (uint32_t)n_masks] : masks[nfailure - 1];   // :412
        pba_index *ix = nullptr;
        int st = pba_index_build(ctx, ref, ref_seq, mask, PBA_INDEX_HEAD_TAIL, &ix);        // get_seedmap, :415
        if (st != PBA_OK) return st;
        st = spaced_round_subset(ctx, ix, ref, ref_seq, reads, R, max_trial, overlap_min, buggy_seed_at, kernel, pool.data(),
                                 (uint32_t)pool.size(), rows);
        pba_index_destroy(ix);
        if (st != PBA_OK) return st;
        int nmatches = 0;
        std::vector<uint32_t> rest;
        rest.reserve(pool.size());
        for (uint32_t r : pool) {
            if (rows[r].found) { found_round[r] = nround; ++nmatches; }                     // erased from the pool, :443
            else rest.push_back(r);
        }
        if (done < log_cap) { log[done].round = nround; log[done].mask = mask; log[done].n_tried = (int32_t)pool.size(); log[done].n_found = nmatches; }
        ++done;
        pool.swap(rest);
        if (nmatches != 0) nfailure = 0;                                                    // :448-451
        else if (++nfailure == n_masks) break;
    }
    *n_rounds = done;
    return PBA_OK;
}

// ---------------------------------------------------------------------------------------------
// host API: all-vs-all overlap
// ---------------------------------------------------------------------------------------------
#define PBA_OVL_WALK(NBV)                                                                                             \
    hipLaunchKernelGGL((k_ovl_walk<NBV>), dim3(persistent_grid(ctx, n_items, (NBV) ? 4 : 1, lds)),                        \
                       dim3(PBA_WAVE * ((NBV) ? 4 : 1)), lds * ((NBV) ? 4 : 1), ctx->stream, reads->dev(), t_lo, n_items,  \
                       items, d_off.as<uint32_t>(), d_cand.as<uint64_t>(), ocfg, full_band, redo_in,                     \
                       d_redo.as<uint2>(),                                                                              \
                       (unsigned long long)redo_cap, d_cnt64.as<unsigned long long>() + 2, d_out.as<pba_overlap>(),     \
                       (unsigned long long)cap, d_cnt64.as<unsigned long long>(), d_cnt64.as<unsigned long long>() + 1,  \
                       ctx->d_queue)

int pba_overlap_probes(pba_ctx *ctx, const pba_seqs *reads, uint32_t q_lo, uint32_t q_hi, uint32_t mask, int max_trial,
                       void *d_entries, uint64_t cap, uint64_t *n_out) {
    if (!ctx || !reads || !d_entries || !n_out || q_lo > q_hi || q_hi > reads->n) return PBA_E_INVALID;
    if (max_trial < 1 || 2 * max_trial >= (1 << PBA_OVL_JD_BITS)) PBA_FAIL(PBA_E_INVALID, "max_trial must be in [1, 63]");
    if (reads->n >= (1u << 24)) PBA_FAIL(PBA_E_TOOLONG, "at most 2^24 reads");
    HIPCHK(hipSetDevice(ctx->device));
    const uint32_t t2 = 2u * (uint32_t)max_trial;
    const uint64_t slots = (uint64_t)(q_hi - q_lo) * t2;
    DevBuf counter;
    HIPCHK(hipMalloc(&counter.p, 8));
    HIPCHK(hipMemsetAsync(counter.p, 0, 8, ctx->stream));
    if (slots)
        hipLaunchKernelGGL(k_probe_emit, dim3((uint32_t)((slots + 255) / 256)), dim3(256), 0, ctx->stream, reads->dev(), q_lo,
                           q_hi - q_lo, t2, mask, (uint64_t *)d_entries, (unsigned long long)cap,
                           counter.as<unsigned long long>());
    unsigned long long h_n = 0;
    HIPCHK(hipMemcpyAsync(&h_n, counter.p, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipGetLastError());
    if (h_n > cap) PBA_FAIL(PBA_E_INVALID, "pba_overlap_probes: entry buffer too small");
    *n_out = h_n;
    return PBA_OK;
}

int pba_overlap_all(pba_ctx *ctx, const pba_seqs *reads, uint32_t t_lo, uint32_t t_hi, uint32_t mask, double R,
                    int max_trial, int overlap_min, int kernel, pba_overlap *out, uint64_t cap, uint64_t *n_out,
                    pba_overlap_stats *stats) {
    if (!ctx || !reads || !n_out) return PBA_E_INVALID;
    if (max_trial < 1 || 2 * max_trial >= (1 << PBA_OVL_JD_BITS)) PBA_FAIL(PBA_E_INVALID, "max_trial must be in [1, 63]");
    HIPCHK(hipSetDevice(ctx->device));
    // 1. probe table of every read, built here (single-GPU form)
    const uint64_t pcap = (uint64_t)reads->n * 2u * (uint32_t)max_trial;
    DevBuf d_pent;
    HIPCHK(hipMalloc(&d_pent.p, sizeof(uint64_t) * (pcap + 1)));
    uint64_t n_pent = 0;
    int rc = pba_overlap_probes(ctx, reads, 0, reads->n, mask, max_trial, d_pent.p, pcap, &n_pent);
    if (rc != PBA_OK) return rc;
    return pba_overlap_all_probes(ctx, reads, t_lo, t_hi, d_pent.p, n_pent, mask, R, max_trial, overlap_min, kernel, out, cap,
                                  n_out, stats);
}

int pba_overlap_all_probes(pba_ctx *ctx, const pba_seqs *reads, uint32_t t_lo, uint32_t t_hi, const void *d_probe_entries,
                           uint64_t n_probe_slots, uint32_t mask, double R, int max_trial, int overlap_min, int kernel,
                           pba_overlap *out, uint64_t cap, uint64_t *n_out, pba_overlap_stats *stats) {
    if (!ctx || !reads || !n_out || (!out && cap) || t_lo > t_hi || t_hi > reads->n || (!d_probe_entries && n_probe_slots))
        return PBA_E_INVALID;
    if (max_trial < 1 || 2 * max_trial >= (1 << PBA_OVL_JD_BITS)) PBA_FAIL(PBA_E_INVALID, "max_trial must be in [1, 63]");
    if (reads->n >= (1u << 24)) PBA_FAIL(PBA_E_TOOLONG, "pba_overlap_all: at most 2^24 reads");
    if (reads->max_len > (uint32_t)kMaxSeqLen) PBA_FAIL(PBA_E_TOOLONG, "read longer than the engine limit");
    if (reads->non_acgt) PBA_FAIL(PBA_E_ALPHABET, "pba_overlap_all: the read set holds bytes outside ACGT");
    HIPCHK(hipSetDevice(ctx->device));
    *n_out = 0;
    pba_overlap_stats st;
    memset(&st, 0, sizeof st);
    const uint32_t n = reads->n, nt = t_hi - t_lo, t2 = 2u * (uint32_t)max_trial;
    if (nt == 0 || n < 2) { if (stats) *stats = st; return PBA_OK; }
    Plan pl;
    int rc = make_plan(ctx, R, 0, 0, kernel, 1 + (int)(reads->max_len * R), &pl);
    if (rc != PBA_OK) return rc;

    // 1. probe table: the (gathered) probe entries, partitioned and sorted like a seed index
    DevBuf d_cnt64;
    HIPCHK(hipMalloc(&d_cnt64.p, 32));
    (void)hipEventRecord(ctx->ev[2], ctx->stream);
    pba_index *pix = nullptr;
    rc = pba_index_from_entries(ctx, d_probe_entries, n_probe_slots, mask, PBA_INDEX_ALL, 0, &pix);   // identity ordinal -> value: the probe id
    if (rc != PBA_OK) return rc;
    struct IxGuard { pba_index *p; ~IxGuard() { pba_index_destroy(p); } } guard{pix};
    st.n_probe_entries = pix->n_entries;

    // 2. scan the targets' positions against the probe table: count, offsets, fill
    DevBuf d_off, d_cur, d_cand, d_out, d_pres;
    HIPCHK(hipMalloc(&d_pres.p, (size_t)1 << (PBA_OVL_PRES_LOG - 3)));
    HIPCHK(hipMemsetAsync(d_pres.p, 0, (size_t)1 << (PBA_OVL_PRES_LOG - 3), ctx->stream));
    if (pix->n_entries)
        hipLaunchKernelGGL(k_ovl_presence, dim3((uint32_t)((pix->n_entries + 255) / 256)), dim3(256), 0, ctx->stream,
                           (const uint64_t *)pix->d_ent, (uint64_t)pix->n_entries, d_pres.as<uint32_t>());
    // a direct-address directory of the probe keys (seed_index.h: KeyDir), when the mask's care bits allow one
    KeyDir kd;
    memset(&kd, 0, sizeof kd);
    DevBuf d_dir;
    const int care = __builtin_popcount(mask);
    if (pix->n_entries && pix->n_entries < 0xFFFFFFFFull && care <= PBA_DIR_MAX_BITS) {
        kd.mask = mask; kd.n_entries = (uint32_t)pix->n_entries;
        uint32_t m = mask, mk = ~m << 1;                         // Hacker's Delight 7-4: the move masks of compress(x, m)
        for (int i = 0; i < 5; ++i) {
            uint32_t mp = mk ^ (mk << 1);
            mp ^= mp << 2; mp ^= mp << 4; mp ^= mp << 8; mp ^= mp << 16;
            const uint32_t mv = mp & m;
            kd.mv[i] = mv;
            m = (m ^ mv) | (mv >> (1 << i));
            mk &= ~mp;
        }
        HIPCHK(hipMalloc(&d_dir.p, sizeof(uint32_t) << care));
        HIPCHK(hipMemsetAsync(d_dir.p, 0xFF, sizeof(uint32_t) << care, ctx->stream));
        hipLaunchKernelGGL(k_dir_build, dim3((kd.n_entries + 255) / 256), dim3(256), 0, ctx->stream, (const uint64_t *)pix->d_ent,
                           kd.n_entries, kd, d_dir.as<uint32_t>());
        kd.dir = d_dir.as<uint32_t>();
    }
    HIPCHK(hipMalloc(&d_off.p, sizeof(uint32_t) * (nt + 1)));
    // count per (target, bucket of consecutive queries): PBA_OVL_SUB buckets per target
    const uint64_t nsub = (uint64_t)nt * PBA_OVL_SUB;
    const uint32_t sub_mul = (uint32_t)std::min<uint64_t>(0xFFFFFFFFull, ((uint64_t)PBA_OVL_SUB << 32) / n);   // bucket = umulhi(q, sub_mul)
    DevBuf d_sub;
    HIPCHK(hipMalloc(&d_sub.p, sizeof(uint32_t) * (nsub + 1)));
    hipLaunchKernelGGL(k_ovl_scan<false>, dim3(nt), dim3(256), 0, ctx->stream, pix->dev(), kd, d_pres.as<uint32_t>(), reads->dev(),
                       t_lo, nt, t2, sub_mul, 0, d_sub.as<uint32_t>(), (uint64_t *)nullptr);
    std::vector<uint32_t> h_sub(nsub + 1), h_cnt(nt + 1), h_off(nt + 1);
    HIPCHK(hipMemcpyAsync(h_sub.data(), d_sub.p, sizeof(uint32_t) * nsub, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipGetLastError());
    // the coarsest split (2^shift neighbouring buckets merged) whose pieces all fit the LDS sort
    int shift = 0;
    for (int sh = 6; sh >= 0; --sh) {                            // PBA_OVL_SUB = 2^6
        bool fits = true;
        for (uint64_t p0 = 0; p0 < nsub && fits; p0 += (1ull << sh)) {
            uint64_t c = 0;
            for (uint64_t k = 0; k < (1ull << sh); ++k) c += h_sub[p0 + k];
            fits = c <= PBA_IX_LDS_SORT_CAP;
        }
        if (fits) { shift = sh; break; }
    }
    const uint32_t per_t = PBA_OVL_SUB >> shift;                  // sorted pieces per target
    const uint64_t npiece = (uint64_t)nt * per_t;
    std::vector<uint32_t> h_poff(npiece + 1);
    uint64_t total = 0;
    uint32_t biggest = 2;
    std::vector<uint32_t> oversize;                               // pieces that outgrow the LDS sort even at the finest split
    for (uint32_t i = 0; i < nt; ++i) {
        h_off[i] = (uint32_t)total;
        for (uint32_t pc = 0; pc < per_t; ++pc) {
            uint64_t c = 0;
            for (uint32_t k = 0; k < (1u << shift); ++k) c += h_sub[(uint64_t)i * PBA_OVL_SUB + ((uint64_t)pc << shift) + k];
            h_poff[(uint64_t)i * per_t + pc] = (uint32_t)total;
            if (c <= PBA_IX_LDS_SORT_CAP) biggest = std::max<uint32_t>(biggest, (uint32_t)c);
            else oversize.push_back((uint32_t)((uint64_t)i * per_t + pc));
            total += c;
            if (total > 0xFFFFFFF0ull) PBA_FAIL(PBA_E_TOOLONG, "pba_overlap_all: more than 2^32 candidates in one target range; shard it");
        }
        h_cnt[i] = (uint32_t)(total - h_off[i]);
    }
    h_off[nt] = (uint32_t)total;
    h_poff[npiece] = (uint32_t)total;
    st.n_candidates = total;
    DevBuf d_poff;
    HIPCHK(hipMalloc(&d_cand.p, sizeof(uint64_t) * (total + 1)));
    HIPCHK(hipMalloc(&d_poff.p, sizeof(uint32_t) * (npiece + 1)));
    HIPCHK(hipMalloc(&d_cur.p, sizeof(uint32_t) * (npiece + 1)));
    HIPCHK(hipMemcpyAsync(d_off.p, h_off.data(), sizeof(uint32_t) * (nt + 1), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d_poff.p, h_poff.data(), sizeof(uint32_t) * (npiece + 1), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d_cur.p, h_poff.data(), sizeof(uint32_t) * (npiece + 1), hipMemcpyHostToDevice, ctx->stream));
    if (total) {
        hipLaunchKernelGGL(k_ovl_scan<true>, dim3(nt), dim3(256), 0, ctx->stream, pix->dev(), kd, d_pres.as<uint32_t>(),
                           reads->dev(), t_lo, nt, t2, sub_mul, shift, d_cur.as<uint32_t>(), d_cand.as<uint64_t>());
        (void)hipEventRecord(ctx->ev[3], ctx->stream);
        // 3. sort every piece = the reference's try order inside every (target, query); the pieces of a target in
        //    order are its list in query order
        uint32_t pow2 = 2;
        while (pow2 < biggest) pow2 <<= 1;
        for (uint64_t p0 = 0; p0 < npiece; p0 += 0x40000000ull) {  // (grid dimension limit)
            const uint32_t chunk = (uint32_t)std::min<uint64_t>(npiece - p0, 0x40000000ull);
            // (a big piece takes most of a CU's LDS, so its workgroup is the only one there: 1 024 threads keep the CU busy)
            hipLaunchKernelGGL(k_part_sort, dim3(chunk), dim3(pow2 >= 4096 ? 1024 : 256), sizeof(uint64_t) * pow2, ctx->stream, d_cand.as<uint64_t>(),
                               d_poff.as<uint32_t>() + p0);
        }
        HIPCHK(hipStreamSynchronize(ctx->stream));
        HIPCHK(hipGetLastError());
        for (uint32_t pc : oversize) {                            // e.g. one query with > 16 384 candidates on a target
            rc = sort_partition_global(ctx, d_cand.as<uint64_t>() + h_poff[pc], h_poff[pc + 1] - h_poff[pc]);
            if (rc != PBA_OK) return rc;
        }
    } else {
        (void)hipEventRecord(ctx->ev[3], ctx->stream);
    }
    (void)hipEventRecord(ctx->ev[4], ctx->stream);

    // 4. walk: persistent wavefronts, one target at a time, narrow window; then the parked (target, query) runs
    //    at the reference band
    HIPCHK(hipMalloc(&d_out.p, sizeof(pba_overlap) * (cap + 1)));
    HIPCHK(hipMemsetAsync(d_cnt64.p, 0, 32, ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->d_queue, 0, 4, ctx->stream));
    OvlCfg ocfg;
    ocfg.R = R; ocfg.overlap_min = overlap_min; ocfg.row_cap = pl.cfg.row_cap; ocfg.t2 = t2;
    const size_t lds = pl.lds;
    DevBuf d_redo, d_items;
    const uint64_t redo_cap = std::max<uint64_t>(1024, total / 4);
    HIPCHK(hipMalloc(&d_redo.p, sizeof(uint2) * redo_cap));
    // work items: (target, first candidate of a group of 64), expanded on the device from the per-target item counts
    // (a million reads make 22 M items per call: building and copying them from the host took longer than a scan pass)
    std::vector<uint32_t> h_ipre(nt + 1);
    uint64_t n_items64 = 0;
    for (uint32_t i = 0; i < nt; ++i) { h_ipre[i] = (uint32_t)n_items64; n_items64 += (h_cnt[i] + PBA_WAVE - 1) / PBA_WAVE; }
    if (n_items64 >= 0xFFFFFFFFull) PBA_FAIL(PBA_E_TOOLONG, "pba_overlap_all: too many work items per call, use a smaller target range");
    h_ipre[nt] = (uint32_t)n_items64;
    DevBuf d_ipre;
    HIPCHK(hipMalloc(&d_ipre.p, sizeof(uint32_t) * (nt + 1)));
    HIPCHK(hipMemcpyAsync(d_ipre.p, h_ipre.data(), sizeof(uint32_t) * (nt + 1), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMalloc(&d_items.p, sizeof(uint2) * (n_items64 + 1)));
    if (n_items64)
        hipLaunchKernelGGL(k_ovl_items, dim3((uint32_t)((n_items64 + 255) / 256)), dim3(256), 0, ctx->stream, d_ipre.as<uint32_t>(),
                           d_off.as<uint32_t>(), nt, (uint32_t)n_items64, d_items.as<uint2>());
    HIPCHK(hipStreamSynchronize(ctx->stream));                   // h_ipre must outlive its copy
    // one launch of the walk: items [lo, hi) of the group list (redo_in == nullptr) or n_redo parked runs
    auto walk = [&](int nb, const uint2 *items, uint32_t n_items, int full_band, const uint2 *redo_in) -> int {
        HIPCHK(hipMemsetAsync(ctx->d_queue, 0, 4, ctx->stream));
        HIPCHK(hipMemsetAsync(d_cnt64.as<unsigned long long>() + 2, 0, 8, ctx->stream));
        PBA_DISPATCH_NB(nb, PBA_OVL_WALK);
        HIPCHK(hipGetLastError());
        return PBA_OK;
    };
    // narrow window for items [lo, hi), then the runs it parked at the reference band; returns the number parked
    uint64_t parked_total = 0;
    // items [lo, hi) in the ring nb_first with its first-pass window, then what that parked in the widest ring below the
    // reference band's (its window takes all the room the ring has, bv_pass1_w: at 15 kb NB = 3 holds 4 072 of max_dst
    // 4 501), then what is still parked at the reference band; returns the number parked by the first stage
    int nb_mid = 0;
    for (int nb : {1, 2, 3, 4, 6})
        if (nb > pl.nb1 && nb < pl.nb2) nb_mid = nb;
    auto narrow_then_redo = [&](int nb_first, size_t lo, size_t hi, uint64_t *parked) -> int {
        *parked = 0;
        if (hi <= lo) return PBA_OK;
        int rc2 = walk(nb_first, d_items.as<uint2>() + lo, (uint32_t)(hi - lo), 0, nullptr);
        if (rc2 != PBA_OK) return rc2;
        for (int stage = 0; stage < 2; ++stage) {
            unsigned long long h_redo = 0;
            HIPCHK(hipMemcpyAsync(&h_redo, d_cnt64.as<unsigned long long>() + 2, 8, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipStreamSynchronize(ctx->stream));
            if (h_redo > redo_cap) PBA_FAIL(PBA_E_NOMEM, "pba_overlap_all: more uncertified (target, query) runs than the redo list holds");
            if (stage == 0) *parked = h_redo;
            if (!h_redo) return PBA_OK;
            if (stage == 0 && nb_mid <= nb_first) continue;      // no ring between this one and the reference band's
            DevBuf d_in;
            HIPCHK(hipMalloc(&d_in.p, sizeof(uint2) * h_redo));
            HIPCHK(hipMemcpyAsync(d_in.p, d_redo.p, sizeof(uint2) * h_redo, hipMemcpyDeviceToDevice, ctx->stream));
            rc2 = stage == 0 ? walk(nb_mid, nullptr, (uint32_t)h_redo, 0, d_in.as<uint2>())
                             : walk(pl.nb2, nullptr, (uint32_t)h_redo, 1, d_in.as<uint2>());
            if (rc2 != PBA_OK) return rc2;
            HIPCHK(hipStreamSynchronize(ctx->stream));
        }
        return PBA_OK;
    };
    // Whether the narrow window pays depends on how far the reads are from each other (two 15 % reads differ by ~27 %:
    // nothing certifies below the reference band), which only the data tells: a sample of the items goes through
    // narrow-then-redo, and if most of its successful runs had to be parked the rest starts wider: in the widest ring
    // below the reference band's (its first-pass window takes all the room that ring has, bv_pass1_w -- at 15 kb NB = 3
    // holds a window of 4 072, which certifies every overlap but the longest), or straight at the reference band.
    const size_t n_all = (size_t)n_items64;
    size_t sample_min = 4096;
    if (const char *e = getenv("PBA_OVL_SAMPLE_MIN")) sample_min = (size_t)std::max(1L, atol(e));   // test hook: small inputs through the sampled decision
    const size_t n_sample = pl.nb1 == 0 ? n_all : std::min(n_all, std::max<size_t>(sample_min, n_all / 32));
    uint64_t parked = 0;
    rc = narrow_then_redo(pl.nb1, 0, n_sample, &parked);
    if (rc != PBA_OK) return rc;
    parked_total += parked;
    if (n_sample < n_all) {
        unsigned long long h_ov = 0;
        HIPCHK(hipMemcpyAsync(&h_ov, d_cnt64.p, 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        if (2 * parked > h_ov) {                                 // most overlaps of the sample needed more than the narrow window
            st.wide_first = 1;
            if (nb_mid) {
                rc = narrow_then_redo(nb_mid, n_sample, n_all, &parked);
                parked_total += parked;
            } else rc = walk(pl.nb2, d_items.as<uint2>() + n_sample, (uint32_t)(n_all - n_sample), 1, nullptr);
        } else {
            rc = narrow_then_redo(pl.nb1, n_sample, n_all, &parked);
            parked_total += parked;
        }
        if (rc != PBA_OK) return rc;
    }
    st.n_redo = parked_total;
    (void)hipEventRecord(ctx->ev[5], ctx->stream);
    HIPCHK(hipGetLastError());
    unsigned long long h_cnt2[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(h_cnt2, d_cnt64.p, 16, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    const uint64_t got = std::min<uint64_t>(h_cnt2[0], cap);
    if (got) HIPCHK(hipMemcpyAsync(out, d_out.p, sizeof(pba_overlap) * got, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    std::sort(out, out + got, [](const pba_overlap &x, const pba_overlap &y) {
        return x.target != y.target ? x.target < y.target : x.query < y.query;
    });
    *n_out = h_cnt2[0];
    st.n_overlaps = h_cnt2[0];
    st.n_pairs = h_cnt2[1];
    (void)hipEventElapsedTime(&st.scan_ms, ctx->ev[2], ctx->ev[3]);
    (void)hipEventElapsedTime(&st.sort_ms, ctx->ev[3], ctx->ev[4]);
    (void)hipEventElapsedTime(&st.walk_ms, ctx->ev[4], ctx->ev[5]);
    if (stats) *stats = st;
    return PBA_OK;
}

// ---------------------------------------------------------------------------------------------
// host API: consensus voting and reference growth (ref_seq.h, the unlocked half)
// ---------------------------------------------------------------------------------------------
struct pba_cons {
    int device;                           // (not the ctx: the object may outlive it, like pba_seqs / pba_index)
    int max_len, beg, end, pre, post;     // as in ref_seq (ref_seq.h:364-368), indices into the 3*max_len arrays
    int cur;                              // which of the two array sets is live (evolve ping-pongs)
    ConsDev set[2];
    int *d_n;
    int vote_ext;                         // votes of the running batch address the text of [pre, post) (pba_cons_round), not [beg, end)
};

static void cons_free_sets(pba_cons *c) {
    for (int k = 0; k < 2; ++k) {
        if (c->set[k].sel) (void)hipFree(c->set[k].sel);
        if (c->set[k].sup) (void)hipFree(c->set[k].sup);
        if (c->set[k].tot) (void)hipFree(c->set[k].tot);
        if (c->set[k].txt) (void)hipFree(c->set[k].txt);
    }
    if (c->d_n) (void)hipFree(c->d_n);
}

static int cons_fill(pba_ctx *ctx, pba_cons *c, int first, const char *text, int len, int weight) {
    if (len <= 0) return PBA_OK;
    DevBuf d_text;
    HIPCHK(hipMalloc(&d_text.p, (size_t)len));
    HIPCHK(hipMemcpyAsync(d_text.p, text, (size_t)len, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_cons_fill, dim3((uint32_t)((len + 255) / 256)), dim3(256), 0, ctx->stream, c->set[c->cur], first,
                       len, d_text.as<char>(), weight);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return PBA_OK;
}

int pba_cons_create(pba_ctx *ctx, const char *text, int len, int weight, int max_len, pba_cons **out) {
    if (!ctx || !out || len < 0 || (!text && len) || max_len < 1 || len > max_len || weight < 0 || weight > 0xFFFF)
        return PBA_E_INVALID;
    *out = nullptr;
    HIPCHK(hipSetDevice(ctx->device));
    pba_cons *c = new (std::nothrow) pba_cons();
    if (!c) return PBA_E_NOMEM;
    memset(c, 0, sizeof *c);
    c->device = ctx->device; c->max_len = max_len;
    c->beg = c->pre = max_len; c->end = c->post = max_len + len;
    const size_t cap = (size_t)3 * max_len + 64;
    bool ok = hipMalloc((void **)&c->d_n, sizeof(int)) == hipSuccess;
    for (int k = 0; k < 2 && ok; ++k)
        ok = hipMalloc((void **)&c->set[k].sel, cap * 8) == hipSuccess && hipMalloc((void **)&c->set[k].sup, cap * 8) == hipSuccess &&
             hipMalloc((void **)&c->set[k].tot, cap * 4) == hipSuccess && hipMalloc((void **)&c->set[k].txt, cap) == hipSuccess;
    if (!ok) { cons_free_sets(c); delete c; PBA_FAIL(PBA_E_NOMEM, "pba_cons_create"); }
    int st = cons_fill(ctx, c, c->beg, text, len, weight);                    // ref_seq.h:218-225
    if (st != PBA_OK) { cons_free_sets(c); delete c; return st; }
    *out = c;
    return PBA_OK;
}

void pba_cons_destroy(pba_cons *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    cons_free_sets(c);                    // hipFree waits for the work that uses the buffers
    delete c;
}

int pba_cons_extent(const pba_cons *c, int32_t *extent) {
    if (!c || !extent) return PBA_E_INVALID;
    extent[0] = c->pre - c->beg; extent[1] = c->post - c->beg; extent[2] = c->end - c->beg;
    return PBA_OK;
}

int pba_cons_append(pba_ctx *ctx, pba_cons *c, const char *seg, int len) {        // ref_seq.h:227-233
    if (!ctx || !c || len < 0 || (!seg && len)) return PBA_E_INVALID;
    if ((long long)c->post + len > 3ll * c->max_len) PBA_FAIL(PBA_E_TOOLONG, "pba_cons_append: reference grew past 2*max_len");
    HIPCHK(hipSetDevice(ctx->device));
    int st = cons_fill(ctx, c, c->post, seg, len, 1);
    if (st == PBA_OK) c->post += len;
    return st;
}

int pba_cons_prepend(pba_ctx *ctx, pba_cons *c, const char *seg, int len) {       // ref_seq.h:235-242
    if (!ctx || !c || len < 0 || (!seg && len)) return PBA_E_INVALID;
    if (c->pre - len < 0) PBA_FAIL(PBA_E_TOOLONG, "pba_cons_prepend: reference grew past max_len before its origin");
    HIPCHK(hipSetDevice(ctx->device));
    int st = cons_fill(ctx, c, c->pre - len, seg, len, 1);
    if (st == PBA_OK) c->pre -= len;
    return st;
}

int pba_cons_elect(pba_ctx *ctx, pba_cons *c, uint32_t n, const int32_t *pos, const uint8_t *fwd, const uint8_t *ops,
                   const char *vals, const uint64_t *ops_off, const int32_t *nedit) {
    if (!ctx || !c || (n && (!pos || !fwd || !ops || !vals || !ops_off || !nedit))) return PBA_E_INVALID;
    if (n == 0) return PBA_OK;
    for (uint32_t q = 0; q < n; ++q)
        if (nedit[q] < 0 || ops_off[q + 1] < ops_off[q] || (uint64_t)nedit[q] > ops_off[q + 1] - ops_off[q] ||
            c->beg + pos[q] < c->pre || c->beg + pos[q] >= c->post)              // "pos should be contained", ref_seq.h:351
            PBA_FAIL(PBA_E_INVALID, "pba_cons_elect: script outside its slot or position outside the reference");
    HIPCHK(hipSetDevice(ctx->device));
    const uint64_t total = ops_off[n] - ops_off[0];
    DevBuf d_pos, d_fwd, d_ops, d_vals, d_off, d_ne;
    HIPCHK(hipMalloc(&d_pos.p, sizeof(int32_t) * n));
    HIPCHK(hipMalloc(&d_fwd.p, n));
    HIPCHK(hipMalloc(&d_ops.p, total + 16));
    HIPCHK(hipMalloc(&d_vals.p, total + 16));
    HIPCHK(hipMalloc(&d_off.p, sizeof(uint64_t) * (n + 1)));
    HIPCHK(hipMalloc(&d_ne.p, sizeof(int32_t) * n));
    std::vector<uint64_t> rel(n + 1);
    for (uint32_t q = 0; q <= n; ++q) rel[q] = ops_off[q] - ops_off[0];
    HIPCHK(hipMemcpyAsync(d_pos.p, pos, sizeof(int32_t) * n, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d_fwd.p, fwd, n, hipMemcpyHostToDevice, ctx->stream));
    if (total) {
        HIPCHK(hipMemcpyAsync(d_ops.p, ops + ops_off[0], total, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(d_vals.p, vals + ops_off[0], total, hipMemcpyHostToDevice, ctx->stream));
    }
    HIPCHK(hipMemcpyAsync(d_off.p, rel.data(), sizeof(uint64_t) * (n + 1), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d_ne.p, nedit, sizeof(int32_t) * n, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_cons_elect, dim3(n), dim3(PBA_WAVE), 0, ctx->stream, c->set[c->cur], c->beg, c->pre, c->post, n,
                       d_pos.as<int>(), d_fwd.as<uint8_t>(), d_ops.as<uint8_t>(), d_vals.as<char>(),
                       d_off.as<unsigned long long>(), d_ne.as<int>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return PBA_OK;
}

static int cons_vote_view(const pba_cons *c, ConsDev *dev, int *beg, int *pre, int *post) {
    if (!c) return PBA_E_INVALID;
    *dev = c->set[c->cur]; *beg = c->vote_ext ? c->pre : c->beg; *pre = c->pre; *post = c->post;
    return PBA_OK;
}

// ref_seq::try_align's align + gate + elect (ref_seq.h:264-267) for a batch of pairs whose `a` is the reference:
// sweep, walk and vote on the device, no script in memory.  No growth (append / prepend are the caller's, after the
// batch): the batch form of a round of interior reads.
int pba_cons_vote_pairs(pba_ctx *ctx, pba_cons *c, const pba_seqs *A, uint32_t ref_seq, const pba_seqs *B,
                        const pba_pair *pairs, size_t n, double R, int maxn, int maxm, int overlap_min, pba_result *out) {
    if (!ctx || !c || !A || !B || (!pairs && n) || (!out && n) || ref_seq >= A->n) return PBA_E_INVALID;
    if ((int)A->h_len[ref_seq] != c->end - c->beg) PBA_FAIL(PBA_E_INVALID, "pba_cons_vote_pairs: A[ref_seq] is not the reference of these boxes");
    for (size_t q = 0; q < n; ++q) {
        const bool ab = (pairs[q].flags & PBA_A_BACKWARD) != 0, bb = (pairs[q].flags & PBA_B_BACKWARD) != 0;
        if (pairs[q].a_seq != ref_seq || ab != bb)               // try_align walks both accessors the same way (ref_seq.h:260-261)
            PBA_FAIL(PBA_E_INVALID, "pba_cons_vote_pairs: a must be the reference, both accessors in one direction");
    }
    return trace_batch(ctx, A, B, pairs, n, R, maxn, maxm, PBA_KERNEL_BITVEC, out, nullptr, nullptr, nullptr, c, overlap_min);
}

// the text of boxes [first, first+len) as a one-sequence set (packed on the device from the object's own text array)
static int cons_text_seqs(pba_ctx *ctx, const pba_cons *c, int first, int len, pba_seqs **out) {
    DevBuf d_offs;
    const uint64_t offs[2] = {0, (uint64_t)len};
    HIPCHK(hipMalloc(&d_offs.p, sizeof offs));
    HIPCHK(hipMemcpyAsync(d_offs.p, offs, sizeof offs, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return pba_seqs_from_device_text(ctx, c->set[c->cur].txt + first, d_offs.p, 1, (uint64_t)len, (uint32_t)len, out);
}

// One round of spaced_seed.cpp:420-446 against an UNLOCKED reference: the reads of `pool`, in order, each stopping at its
// first success -- and every success votes (elect) and may grow the text (ref_seq.h:259-276), which the reads after it
// then see.  Votes never change an alignment inside a round (the text changes in evolve), growth does, but only for a
// candidate whose reference accessor is shorter than len_b + max_dst (seq_aligner.h:94-102): it `touches` an end.
// So the round runs as a few batches: all pending reads are walked at once against the text as it stands
// (k_spaced_round reports which ends each read's candidates touched); the rows are then taken in pool order, and a row
// is the reference's as long as no earlier read of the batch has grown -- or been put back for -- an end it touches.
// The accepted successes vote from their traceback walk (k_vote_pairs) against the batch's text, then the growths are
// applied, and what was put back is the next batch.  The first pending read is always accepted, so it ends; a batch
// takes at most one growth per end.
int pba_cons_round(pba_ctx *ctx, pba_cons *c, const pba_seqs *reads, const uint32_t *pool, uint32_t n_pool, uint32_t mask,
                   double R, int max_trial, int overlap_min, int buggy_seed_at, int kernel, int maxn, int maxm,
                   pba_ss_row *rows, pba_cons_round_stats *stats) {
    if (!ctx || !c || !reads || (!pool && n_pool) || !rows || max_trial < 0) return PBA_E_INVALID;
    for (uint32_t k = 0; k < n_pool; ++k)
        if (pool[k] >= reads->n) return PBA_E_INVALID;
    HIPCHK(hipSetDevice(ctx->device));
    pba_cons_round_stats S;
    memset(&S, 0, sizeof S);
    pba_seqs *base = nullptr;
    pba_index *ix = nullptr;
    int st = cons_text_seqs(ctx, c, c->beg, c->end - c->beg, &base);                 // get_seedmap reads [beg, end), ref_seq.h:291-311
    if (st == PBA_OK) st = pba_index_build(ctx, base, 0, mask, PBA_INDEX_HEAD_TAIL, &ix);
    if (st != PBA_OK) { pba_seqs_destroy(base); return st; }
    S.n_index = (uint32_t)pba_index_entries(ix);
    std::vector<uint32_t> pending(pool, pool + n_pool), deferred;
    std::vector<uint8_t> touch(reads->n);
    std::vector<char> rtext;
    struct Growth { uint32_t read; bool fwd; int j, matlen_b; };
    while (st == PBA_OK && !pending.empty()) {
        ++S.n_batches;
        pba_seqs *ext = base;
        if (c->pre != c->beg || c->post != c->end) st = cons_text_seqs(ctx, c, c->pre, c->post - c->pre, &ext);
        if (st != PBA_OK) break;
        const int org = c->beg - c->pre, post_rel = c->post - c->beg, pre_rel = c->pre - c->beg;
        st = spaced_round_subset(ctx, ix, ext, 0, reads, R, max_trial, overlap_min, buggy_seed_at, kernel, pending.data(),
                                 (uint32_t)pending.size(), rows, org, maxn, maxm, touch.data());
        std::vector<pba_pair> vp;
        std::vector<uint32_t> vread;
        std::vector<Growth> grow;
        bool dirty_post = false, dirty_pre = false;
        deferred.clear();
        for (size_t k = 0; st == PBA_OK && k < pending.size(); ++k) {
            const uint32_t r = pending[k];
            if (((touch[r] & 1) && dirty_post) || ((touch[r] & 2) && dirty_pre)) {
                deferred.push_back(r);               // what it does once it is re-walked is unknown: it may grow either end
                dirty_post = dirty_pre = true;
                continue;
            }
            const pba_ss_row &w = rows[r];
            if (!w.found) continue;
            const bool fwd = w.dir == 1;
            const int slen = (int)reads->h_len[r], s_len = slen - w.j;                  // spaced_seed.cpp:274-275, both directions
            const int r_off = fwd ? w.ref_pos : w.ref_pos + 15;                         // spaced_seed.cpp:285
            const int la = fwd ? post_rel - r_off : r_off - pre_rel + 1;                // get_accessor, ref_seq.h:284-285
            pba_pair pr;
            memset(&pr, 0, sizeof pr);
            pr.a_seq = 0; pr.a_pos = r_off + org; pr.a_len = la;
            pr.b_seq = r; pr.b_pos = fwd ? w.j : slen - w.j - 1; pr.b_len = s_len;
            pr.flags = fwd ? 0u : (PBA_A_BACKWARD | PBA_B_BACKWARD);
            vp.push_back(pr); vread.push_back(r);
            ++S.n_found;
            if (w.matlen_a == la) {                                                     // ref_seq.h:268
                grow.push_back(Growth{r, fwd, w.j, w.matlen_b});
                if (fwd) dirty_post = true; else dirty_pre = true;
            }
        }
        if (st == PBA_OK && !vp.empty()) {                                              // elect, ref_seq.h:267
            std::vector<pba_result> out(vp.size());
            c->vote_ext = 1;
            st = trace_batch(ctx, ext, reads, vp.data(), vp.size(), R, maxn, maxm, PBA_KERNEL_BITVEC, out.data(), nullptr, nullptr,
                             nullptr, c, overlap_min);
            c->vote_ext = 0;
            for (size_t q = 0; st == PBA_OK && q < vp.size(); ++q) {
                const pba_ss_row &w = rows[vread[q]];
                if (out[q].rc < 0 || out[q].cost != w.cost || out[q].matlen_a != w.matlen_a || out[q].matlen_b != w.matlen_b) {
                    snprintf(ctx->err, sizeof ctx->err, "pba_cons_round: the voting walk of read %u disagrees with its round row", vread[q]);
                    st = PBA_E_HIP;
                }
            }
        }
        for (size_t g = 0; st == PBA_OK && g < grow.size(); ++g) {                      // ref_seq.h:268-275
            const Growth &G = grow[g];
            const int slen = (int)reads->h_len[G.read], add = (slen - G.j) - G.matlen_b;
            rtext.resize((size_t)slen + 1);
            st = pba_seqs_get_text(ctx, reads, G.read, rtext.data(), rtext.size());
            if (st != PBA_OK) break;
            if (G.fwd) { st = pba_cons_append(ctx, c, rtext.data() + G.j + G.matlen_b, add); ++S.n_grown_fwd; }
            else { st = pba_cons_prepend(ctx, c, rtext.data(), add); ++S.n_grown_bwd; }
        }
        if (ext != base) pba_seqs_destroy(ext);
        S.n_deferred += (uint32_t)deferred.size();
        pending.swap(deferred);
    }
    pba_index_destroy(ix);
    pba_seqs_destroy(base);
    if (stats) *stats = S;
    return st;
}

// spaced_seed's main loop (spaced_seed.cpp:409-452) without -l: rounds of pba_cons_round over the reads not found yet,
// seeds drawn as in pba_spaced_multi, evolve after every round that does not end the loop.
int pba_cons_assemble(pba_ctx *ctx, pba_cons *c, const pba_seqs *reads, double R, int max_trial, int overlap_min,
                      int buggy_seed_at, int kernel, int maxn, int maxm, const uint32_t *masks, int n_masks,
                      const uint32_t *picks, int n_picks, int max_round, pba_ss_row *rows, int32_t *found_round,
                      pba_ss_round_log *log, int32_t *ref_len_log, int log_cap, int *n_rounds) {
    if (!ctx || !c || !reads || !masks || n_masks < 1 || !picks || n_picks < 1 || max_round < 0 || !rows || !found_round ||
        !n_rounds || log_cap < 0 || ((!log || !ref_len_log) && log_cap))
        return PBA_E_INVALID;
    const uint32_t n = reads->n;
    std::vector<uint32_t> pool(n);
    for (uint32_t r = 0; r < n; ++r) { pool[r] = r; found_round[r] = 0; memset(&rows[r], 0, sizeof rows[r]); rows[r].read = (int32_t)r; rows[r].j = -1; }
    int nfailure = 0, draws = 0, done = 0;
    for (int nround = 1; nround <= max_round; ++nround) {
        const uint32_t mask = nfailure == 0 ? masks[picks[draws++ % n_picks] % (uint32_t)n_masks] : masks[nfailure - 1];   // :412
        pba_cons_round_stats S;
        int st = pba_cons_round(ctx, c, reads, pool.data(), (uint32_t)pool.size(), mask, R, max_trial, overlap_min, buggy_seed_at,
                                kernel, maxn, maxm, rows, &S);
        if (st != PBA_OK) return st;
        std::vector<uint32_t> rest;
        rest.reserve(pool.size());
        for (uint32_t r : pool) {
            if (rows[r].found) found_round[r] = nround;                                     // erased from the pool, :443
            else rest.push_back(r);
        }
        if (done < log_cap) { log[done].round = nround; log[done].mask = mask; log[done].n_tried = (int32_t)pool.size(); log[done].n_found = S.n_found; }
        pool.swap(rest);
        bool last = false;
        if (S.n_found != 0) nfailure = 0;                                                   // :448-449
        else if (++nfailure == n_masks) last = true;                                        // :450: break before evolve
        if (!last) {
            int32_t new_len = 0;
            st = pba_cons_evolve(ctx, c, nullptr, 0, &new_len);                             // :451
            if (st != PBA_OK) return st;
            if (done < log_cap) ref_len_log[done] = new_len;
        } else if (done < log_cap) ref_len_log[done] = c->post - c->pre;
        ++done;
        if (last) break;
    }
    *n_rounds = done;
    return PBA_OK;
}

int pba_cons_evolve(pba_ctx *ctx, pba_cons *c, char *text_out, int cap, int32_t *new_len) {   // ref_seq.h:317-349
    if (!ctx || !c || !new_len || cap < 0 || (!text_out && cap)) return PBA_E_INVALID;
    HIPCHK(hipSetDevice(ctx->device));
    const int nxt = c->cur ^ 1;
    hipLaunchKernelGGL(k_cons_evolve, dim3(1), dim3(1024), 0, ctx->stream, c->set[c->cur], c->set[nxt], c->pre, c->post,
                       c->max_len, c->d_n);
    HIPCHK(hipGetLastError());
    int n = 0;
    HIPCHK(hipMemcpyAsync(&n, c->d_n, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    c->cur = nxt;
    c->beg = c->pre = c->max_len;
    c->end = c->post = c->max_len + n;
    *new_len = n;
    const int ncopy = std::min(n, cap);
    if (ncopy > 0) {
        HIPCHK(hipMemcpyAsync(text_out, c->set[c->cur].txt + c->beg, (size_t)ncopy, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    return PBA_OK;
}

int pba_cons_dump(pba_ctx *ctx, const pba_cons *c, uint16_t *sel, uint16_t *sup, int32_t *tot, int cap, int32_t *n) {
    if (!ctx || !c || !n || cap < 0 || (cap && (!sel || !sup || !tot))) return PBA_E_INVALID;
    HIPCHK(hipSetDevice(ctx->device));
    *n = c->post - c->pre;
    const int k = std::min(*n, cap);
    if (k > 0) {
        const ConsDev &d = c->set[c->cur];
        HIPCHK(hipMemcpyAsync(sel, d.sel + c->pre, (size_t)k * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipMemcpyAsync(sup, d.sup + c->pre, (size_t)k * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipMemcpyAsync(tot, d.tot + c->pre, (size_t)k * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    return PBA_OK;
}

int pba_cons_text(pba_ctx *ctx, const pba_cons *c, char *out, int cap, int32_t *n) {
    if (!ctx || !c || !n || cap < 0 || (cap && !out)) return PBA_E_INVALID;
    HIPCHK(hipSetDevice(ctx->device));
    *n = c->post - c->pre;
    const int k = std::min(*n, cap);
    if (k > 0) {
        HIPCHK(hipMemcpyAsync(out, c->set[c->cur].txt + c->pre, (size_t)k, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    return PBA_OK;
}

}  // extern "C"
