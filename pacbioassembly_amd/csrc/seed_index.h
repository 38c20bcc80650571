// seed_index.h -- device side of the seed-hit index.
//
// Replaces hash_table = hash_map<unsigned, list<int>> (/root/reference/src/common.h:54) as built
// by locator.cpp:62-66 and ref_seq::get_seedmap (ref_seq.h:291-311) and probed by
// locator.cpp:76 / spaced_seed.cpp:265.
//
// Shape: one entry per indexed position, entry = (key << 32) | ord, where key = window & mask
// (entries with key == 0 are dropped, ref_seq.h:300 / locator.cpp:64) and ord is the position's
// ordinal in the reference's insertion sequence.  Entries are split into 2^logP hash partitions
// (multiplicative hash of the key); a partition is sorted by the 64-bit entry, i.e. by key and,
// inside a key, by insertion order -- exactly the order a list<int> hands back.  A probe hashes
// to its partition and binary-searches the key's run.
//
// Build = coalesced scan of the 2-bit packed bases (0.25 B/base read), then an MSD partition on the bits of the key's
// hash, at most PBA_IX_LVL_BITS per level: per-workgroup LDS histogram of the level's bins, one global reservation per
// (workgroup, bin), scatter -- a workgroup's 16 384-entry tile leaves >= 64 entries (512 contiguous bytes) in every bin of a
// level, so the partition writes whole lines whatever the size of the index (one level with 4 096 bins wrote 32-byte
// pieces: 3.8x write amplification, and nothing beyond 2^12 partitions: a 500 Mb target took 1.7 s) -- and finally one
// workgroup per partition sorts its ~2 048 entries in LDS.
#ifndef PBA_SEED_INDEX_H
#define PBA_SEED_INDEX_H

#include "dev_common.h"

#define PBA_IX_MAX_LOGP 24                 // partitions of ~2 048 entries for up to 2^32 entries (minus the average)
#ifndef PBA_IX_LVL_BITS
#define PBA_IX_LVL_BITS 8                  // hash bits one partition level resolves (256 bins: two u32 LDS tables = 2 KB); tuning hook
#endif
#ifndef PBA_IX_PART_AVG
#define PBA_IX_PART_AVG 2048               // entries per partition the builder aims at (1 024 .. 2 048); tuning hook
#endif
#define PBA_IX_LDS_SORT_CAP 16384          // entries of the largest segment one k_seg_sort workgroup takes (1 024 threads x 16)
#define PBA_IX_TILE_THREADS 256
#ifndef PBA_IX_TILE_ITERS
#define PBA_IX_TILE_ITERS 4
#endif
#define PBA_IX_TILE_POS (PBA_IX_TILE_THREADS * 16 * PBA_IX_TILE_ITERS)   // positions per workgroup

struct IndexDev {
    const uint64_t *ent;        // entries, partition by partition, each partition sorted
    const uint32_t *part_off;   // 2^logP + 1 offsets into ent
    int logP;
    uint32_t mask;              // spaced-seed mask the index was built under
    // ord -> position (ref_seq.h:297-308): ord < nhead -> ord, else tail_top - (ord - nhead)
    uint32_t nhead;
    int32_t tail_top;
};

__device__ __forceinline__ uint32_t ix_part(uint32_t key, int logP) {
    return logP ? (key * 0x9E3779B1u) >> (32 - logP) : 0u;
}

__device__ __forceinline__ int32_t ix_pos_of(const IndexDev &ix, uint32_t ord) {
    return ord < ix.nhead ? (int32_t)ord : ix.tail_top - (int32_t)(ord - ix.nhead);
}

__device__ __forceinline__ uint32_t ix_lower_bound(const uint64_t *ent, uint32_t lo, uint32_t hi, uint64_t x) {
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (ent[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// end of the run of `key` that starts at entry `beg` (entries up to `hi` may belong to it): almost always 0-2 entries
// further on, so gallop from `beg` (1, 2, 4, ... entries) and finish with a binary search inside the last stride -- a
// second full-depth search would double the dependent loads of a probe
__device__ __forceinline__ uint32_t ix_run_end(const uint64_t *ent, uint32_t beg, uint32_t hi, uint32_t key) {
    uint32_t a = beg, step = 1;
    while (a < hi && (uint32_t)(ent[a] >> 32) == key) {
        const uint32_t nxt = a + step < hi ? a + step : hi;
        if (nxt < hi && (uint32_t)(ent[nxt] >> 32) == key) { a = nxt; step <<= 1; }
        else {                                   // the run ends in (a, nxt]
            uint32_t l = a + 1, h = nxt;
            while (l < h) {
                const uint32_t mid = l + ((h - l) >> 1);
                if ((uint32_t)(ent[mid] >> 32) == key) l = mid + 1; else h = mid;
            }
            a = l;
            break;
        }
    }
    return a;
}

// hash_table::find: run [beg, beg+cnt) of `key` in reference list order
__device__ __forceinline__ void ix_find(const IndexDev &ix, uint32_t key, uint32_t &beg, uint32_t &cnt) {
    const uint32_t p = ix_part(key, ix.logP);
    const uint32_t lo = ix.part_off[p], hi = ix.part_off[p + 1];
    beg = ix_lower_bound(ix.ent, lo, hi, (uint64_t)key << 32);
    cnt = ix_run_end(ix.ent, beg, hi, key) - beg;
}

// One scan segment: positions [lo, hi) of a sequence, visited ascending (ord = ord0 + pos - lo)
// or descending (ord = ord0 + hi - 1 - pos).
struct ScanSeg {
    uint32_t lo, hi, ord0;
    int descending;
};

// keys of the 16 positions [16*chunk, 16*chunk+16) from one 8-byte load of packed bases
__device__ __forceinline__ uint64_t chunk_bits(const uint8_t *seq, uint32_t chunk) {
    return __builtin_bswap64(ld_u64(seq + 4 * (size_t)chunk));
}
__device__ __forceinline__ uint32_t chunk_key(uint64_t be, uint32_t k, uint32_t pos, uint32_t len, uint32_t mask) {
    uint32_t w = (uint32_t)((be << (2 * k)) >> 32);
    const uint32_t valid = len - pos;
    if (valid < 16) w |= 0xFFFFFFFFu >> (2 * valid);
    return __builtin_bswap32(w) & mask;
}

// level 1 from the packed bases: bin = the top `bits` bits of the key's hash
// cnt1 points one slot past the bin's offset slot (counts are scanned in place into offsets)
// ITERS: chunks of 16 positions a thread takes (a workgroup's tile = 256 x 16 x ITERS positions): PBA_IX_TILE_ITERS, or 1 for
// inputs so small that tiles of 16 384 would leave most of the chip without a workgroup (a 5 Mb target is 305 of them)
template <int ITERS>
static __global__ void __launch_bounds__(PBA_IX_TILE_THREADS)
k_seed_count(const uint8_t *seq, uint32_t len, uint32_t mask, ScanSeg sg, int bits, uint32_t *cnt1) {
    __shared__ uint32_t hist[1 << PBA_IX_LVL_BITS];
    const uint32_t P = 1u << bits;
    for (uint32_t p = threadIdx.x; p < P; p += blockDim.x) hist[p] = 0;
    __syncthreads();
    const uint32_t chunk0 = sg.lo >> 4;
    for (int it = 0; it < ITERS; ++it) {
        const uint32_t chunk = chunk0 + (blockIdx.x * ITERS + it) * PBA_IX_TILE_THREADS + threadIdx.x;
        if ((uint64_t)chunk * 16 >= sg.hi) break;
        const uint64_t be = chunk_bits(seq, chunk);
#pragma unroll
        for (uint32_t k = 0; k < 16; ++k) {
            const uint32_t pos = chunk * 16 + k;
            if (pos < sg.lo || pos >= sg.hi) continue;
            const uint32_t key = chunk_key(be, k, pos, len, mask);
            if (key) atomicAdd(&hist[ix_part(key, bits)], 1u);
        }
    }
    __syncthreads();
    for (uint32_t p = threadIdx.x; p < P; p += blockDim.x)
        if (hist[p]) atomicAdd(&cnt1[p], hist[p]);
}

// ... and the entries into their level-1 bins (order inside a bin does not matter: the partitions are sorted at the end)
template <int ITERS>
static __global__ void __launch_bounds__(PBA_IX_TILE_THREADS)
k_seed_scatter(const uint8_t *seq, uint32_t len, uint32_t mask, ScanSeg sg, int bits, uint32_t *cursor, uint64_t *dst) {
    __shared__ uint32_t hist[1 << PBA_IX_LVL_BITS];
    __shared__ uint32_t base[1 << PBA_IX_LVL_BITS];
    const uint32_t P = 1u << bits;
    for (uint32_t p = threadIdx.x; p < P; p += blockDim.x) hist[p] = 0;
    __syncthreads();
    const uint32_t chunk0 = sg.lo >> 4;
    uint64_t be[ITERS];
    for (int it = 0; it < ITERS; ++it) {
        const uint32_t chunk = chunk0 + (blockIdx.x * ITERS + it) * PBA_IX_TILE_THREADS + threadIdx.x;
        be[it] = 0;
        if ((uint64_t)chunk * 16 >= sg.hi) continue;
        be[it] = chunk_bits(seq, chunk);
#pragma unroll
        for (uint32_t k = 0; k < 16; ++k) {
            const uint32_t pos = chunk * 16 + k;
            if (pos < sg.lo || pos >= sg.hi) continue;
            const uint32_t key = chunk_key(be[it], k, pos, len, mask);
            if (key) atomicAdd(&hist[ix_part(key, bits)], 1u);
        }
    }
    __syncthreads();
    for (uint32_t p = threadIdx.x; p < P; p += blockDim.x) {
        const uint32_t n = hist[p];
        base[p] = n ? atomicAdd(&cursor[p], n) : 0u;
        hist[p] = 0;
    }
    __syncthreads();
    for (int it = 0; it < ITERS; ++it) {
        const uint32_t chunk = chunk0 + (blockIdx.x * ITERS + it) * PBA_IX_TILE_THREADS + threadIdx.x;
        if ((uint64_t)chunk * 16 >= sg.hi) continue;
#pragma unroll
        for (uint32_t k = 0; k < 16; ++k) {
            const uint32_t pos = chunk * 16 + k;
            if (pos < sg.lo || pos >= sg.hi) continue;
            const uint32_t key = chunk_key(be[it], k, pos, len, mask);
            if (!key) continue;
            const uint32_t p = ix_part(key, bits);
            const uint32_t slot = base[p] + atomicAdd(&hist[p], 1u);
            const uint32_t ord = sg.ord0 + (sg.descending ? sg.hi - 1 - pos : pos - sg.lo);
            dst[slot] = (uint64_t)key << 32 | ord;
        }
    }
}

// ---- a further level: the entries, grouped by the top `done` bits of their hash (off_prev: the groups' offsets), are
// split by the next `bits` bits.  A workgroup takes one tile of PBA_IX_TILE_POS entries of one group (tile_pre: tiles of
// the groups before it).  Also level 1 of the exchange form: one group = the gathered list, all-ones entries = padding.
struct LvlSrc {
    const uint64_t *src;
    const uint32_t *off_prev, *tile_pre;
    uint32_t n_groups;
    int done, bits;
    uint32_t tile;           // entries of a tile: PBA_IX_TILE_POS, or 4 096 for small inputs (more workgroups; still >= 64 entries per bin at 6 bits)
};
__device__ __forceinline__ bool lvl_tile(const LvlSrc &L, uint32_t &group, uint32_t &lo, uint32_t &hi) {
    uint32_t a = 0, b = L.n_groups;                     // last group g with tile_pre[g] <= blockIdx.x
    if (blockIdx.x >= L.tile_pre[L.n_groups]) return false;
    while (b - a > 1) {
        const uint32_t mid = (a + b) >> 1;
        if (L.tile_pre[mid] <= blockIdx.x) a = mid; else b = mid;
    }
    group = a;
    lo = L.off_prev[a] + (blockIdx.x - L.tile_pre[a]) * L.tile;
    hi = min(L.off_prev[a + 1], lo + L.tile);
    return true;
}
// the entry's bin inside its group: the `bits` hash bits below the top `done`
__device__ __forceinline__ uint32_t lvl_bin(const LvlSrc &L, uint64_t e) {
    return L.bits ? ((((uint32_t)(e >> 32)) * 0x9E3779B1u) >> (32 - L.done - L.bits)) & ((1u << L.bits) - 1u) : 0u;
}
static __global__ void __launch_bounds__(256)
k_lvl_tiles(const uint32_t *off_prev, uint32_t n_groups, uint32_t tile, uint32_t *ntiles1) {   // ntiles1[-1] = 0: the scan's exclusive start
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g == 0) ntiles1[-1] = 0;
    if (g < n_groups) ntiles1[g] = (off_prev[g + 1] - off_prev[g] + tile - 1) / tile;
}
static __global__ void __launch_bounds__(PBA_IX_TILE_THREADS)
k_lvl_count(LvlSrc L, uint32_t *cnt1) {
    __shared__ uint32_t hist[1 << PBA_IX_LVL_BITS];
    uint32_t g, lo, hi;
    if (!lvl_tile(L, g, lo, hi)) return;
    const uint32_t P = 1u << L.bits;
    for (uint32_t p = threadIdx.x; p < P; p += blockDim.x) hist[p] = 0;
    __syncthreads();
    for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const uint64_t e = L.src[i];
        if (e != ~0ull) atomicAdd(&hist[lvl_bin(L, e)], 1u);
    }
    __syncthreads();
    for (uint32_t p = threadIdx.x; p < P; p += blockDim.x)
        if (hist[p]) atomicAdd(&cnt1[(g << L.bits) + p], hist[p]);
}
static __global__ void __launch_bounds__(PBA_IX_TILE_THREADS)
k_lvl_scatter(LvlSrc L, uint32_t *cursor, uint64_t *dst) {
    __shared__ uint32_t hist[1 << PBA_IX_LVL_BITS];
    __shared__ uint32_t base[1 << PBA_IX_LVL_BITS];
    uint32_t g, lo, hi;
    if (!lvl_tile(L, g, lo, hi)) return;
    const uint32_t P = 1u << L.bits;
    for (uint32_t p = threadIdx.x; p < P; p += blockDim.x) hist[p] = 0;
    __syncthreads();
    for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const uint64_t e = L.src[i];
        if (e != ~0ull) atomicAdd(&hist[lvl_bin(L, e)], 1u);
    }
    __syncthreads();
    for (uint32_t p = threadIdx.x; p < P; p += blockDim.x) {
        const uint32_t c = hist[p];
        base[p] = c ? atomicAdd(&cursor[(g << L.bits) + p], c) : 0u;
        hist[p] = 0;
    }
    __syncthreads();
    for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const uint64_t e = L.src[i];
        if (e == ~0ull) continue;
        const uint32_t p = lvl_bin(L, e);
        dst[base[p] + atomicAdd(&hist[p], 1u)] = e;
    }
}
// the largest partition (for the LDS the sort asks for) and how many outgrow the LDS sort
static __global__ void __launch_bounds__(256)
k_part_max(const uint32_t *part_off, uint32_t P, uint32_t *out3) {        // out3[0..1] zero on entry; out3[2] = entries in all
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    if (p == 0) out3[2] = part_off[P];
    const uint32_t n = part_off[p + 1] - part_off[p];
    if (n <= PBA_IX_LDS_SORT_CAP) atomicMax(&out3[0], n); else atomicAdd(&out3[1], 1u);
}

// ---- multi-GPU exchange form: a rank scans its slice of the visiting order into a flat entry list ...
static __global__ void __launch_bounds__(PBA_IX_TILE_THREADS)
k_seed_emit(const uint8_t *seq, uint32_t len, uint32_t mask, ScanSeg sg, uint64_t *out, unsigned long long cap,
            unsigned long long *counter) {
    __shared__ uint32_t wg_count, wg_base_lo, wg_base_hi;
    if (threadIdx.x == 0) wg_count = 0;
    __syncthreads();
    const uint32_t chunk0 = sg.lo >> 4;
    uint64_t be[PBA_IX_TILE_ITERS];
    uint32_t mine = 0;
    for (int it = 0; it < PBA_IX_TILE_ITERS; ++it) {
        const uint32_t chunk = chunk0 + (blockIdx.x * PBA_IX_TILE_ITERS + it) * PBA_IX_TILE_THREADS + threadIdx.x;
        be[it] = 0;
        if ((uint64_t)chunk * 16 >= sg.hi) continue;
        be[it] = chunk_bits(seq, chunk);
#pragma unroll
        for (uint32_t k = 0; k < 16; ++k) {
            const uint32_t pos = chunk * 16 + k;
            if (pos < sg.lo || pos >= sg.hi) continue;
            if (chunk_key(be[it], k, pos, len, mask)) ++mine;
        }
    }
    uint32_t slot = mine ? atomicAdd(&wg_count, mine) : 0u;      // LDS-staged reservation: one global atomic per workgroup
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long b = wg_count ? atomicAdd(counter, (unsigned long long)wg_count) : 0ull;
        wg_base_lo = (uint32_t)b; wg_base_hi = (uint32_t)(b >> 32);
    }
    __syncthreads();
    unsigned long long o = ((unsigned long long)wg_base_hi << 32 | wg_base_lo) + slot;
    for (int it = 0; it < PBA_IX_TILE_ITERS; ++it) {
        const uint32_t chunk = chunk0 + (blockIdx.x * PBA_IX_TILE_ITERS + it) * PBA_IX_TILE_THREADS + threadIdx.x;
        if ((uint64_t)chunk * 16 >= sg.hi) continue;
#pragma unroll
        for (uint32_t k = 0; k < 16; ++k) {
            const uint32_t pos = chunk * 16 + k;
            if (pos < sg.lo || pos >= sg.hi) continue;
            const uint32_t key = chunk_key(be[it], k, pos, len, mask);
            if (!key) continue;
            const uint32_t ord = sg.ord0 + (sg.descending ? sg.hi - 1 - pos : pos - sg.lo);
            if (o < cap) out[o] = (uint64_t)key << 32 | ord;
            ++o;
        }
    }
}

// In-place inclusive scan of a[0 .. n) in three launches: tiles of PBA_SCAN_TILE per workgroup, the tile sums by one
// workgroup, the carry-in added back.  (2^24 - 2^26 bucket counters, once per probe table.)
#define PBA_SCAN_TILE 2048
static __global__ void __launch_bounds__(256)
k_scan_tiles(uint32_t *a, uint64_t n, uint32_t *tile_sum) {
    __shared__ uint32_t wsum[4];
    const uint64_t base = (uint64_t)blockIdx.x * PBA_SCAN_TILE + (uint64_t)threadIdx.x * 8;
    uint32_t v[8], run = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { v[k] = base + k < n ? a[base + k] : 0u; run += v[k]; v[k] = run; }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t inc = run;                                  // inclusive scan of the per-thread sums across the wavefront
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint32_t carry = inc - run;
    for (int k = 0; k < w; ++k) carry += wsum[k];
#pragma unroll
    for (int k = 0; k < 8; ++k) if (base + k < n) a[base + k] = v[k] + carry;
    if (threadIdx.x == 255) tile_sum[blockIdx.x] = carry + run;
}
static __global__ void __launch_bounds__(1024)
k_scan_sums(uint32_t *tile_sum, uint32_t n_tiles) {      // one workgroup, exclusive scan in place
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t b0 = 0; b0 < n_tiles; b0 += 1024) {
        const uint32_t i = b0 + threadIdx.x;
        const uint32_t x = i < n_tiles ? tile_sum[i] : 0u;
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        uint32_t inc = x;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        uint32_t c = carry_s;
        for (int k = 0; k < w; ++k) c += wsum[k];
        if (i < n_tiles) tile_sum[i] = c + inc - x;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = c + inc;
        __syncthreads();
    }
}
// ... and for up to 2^13 values one launch of one workgroup (a level's 64 .. 4 096 bin counts: three launches of ~5 us each
// were a tenth of a 5 Mb index build).  shifted (nullable): shifted[0] = 0, shifted[i + 1] = a[i] after the scan, i + 1 < n -- the
// cursor copy the scatter kernels consume, written here instead of by a device-to-device copy.
#define PBA_SCAN_SMALL_MAX 8192             // (one workgroup: 4 096 values 7 us, 32 768 values 44 us -- the three launches take 15)
static __global__ void __launch_bounds__(1024)
k_scan_small(uint32_t *a, uint32_t n, uint32_t *shifted) {
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry_s;
    if (threadIdx.x == 0) { carry_s = 0; if (shifted) shifted[0] = 0; }
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (uint32_t b0 = 0; b0 < n; b0 += 1024 * 8) {
        const uint32_t base = b0 + threadIdx.x * 8;
        uint32_t v[8], run = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) { v[k] = base + k < n ? a[base + k] : 0u; run += v[k]; v[k] = run; }
        uint32_t inc = run;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        uint32_t carry = carry_s + inc - run;
        for (int k = 0; k < w; ++k) carry += wsum[k];
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (base + k < n) {
                a[base + k] = v[k] + carry;
                if (shifted && base + k + 1 < n) shifted[base + k + 1] = v[k] + carry;
            }
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + run;
        __syncthreads();
    }
}
static __global__ void __launch_bounds__(256)
k_scan_add(uint32_t *a, uint64_t n, const uint32_t *tile_pre) {
    const uint32_t c = tile_pre[blockIdx.x];
    const uint64_t base = (uint64_t)blockIdx.x * PBA_SCAN_TILE + (uint64_t)threadIdx.x * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) if (base + k < n) a[base + k] += c;
}

// ---------------------------------------------------------------------------------------------------------------
// Segment sort: the entries of a segment (a hash partition of the index, a target's candidate slice or a piece of it)
// are spread over buckets by an ORDER-PRESERVING map of their value -- the buckets in order are the segment in order --
// and every bucket (20-40 entries) is sorted in the registers of one wavefront: a bitonic network over 64 lanes, 21
// compare-exchange stages on lane shuffles, no LDS traffic, no workgroup barrier.  The LDS bitonic sort it replaces moved
// every entry through LDS ~100 times (27 MB of LDS traffic for a 16 384-entry piece: 2.8 s of the 9.9 s of a
// million-read all-vs-all, two thirds of a 500 Mb index build); here an entry is loaded once into a register, counted,
// written to its bucket, and sorted there.
//   mode 0: bucket = the top bits of (entry - smallest entry): candidate slices (query ids spread evenly over their
//           range); all-ones entries (empty slots) go to a bucket of their own behind the others, unsorted
//   mode 1: bucket = the top bits of the key's care bits gathered into a number (monotone in the key, and uniform: a hash
//           partition holds keys from all over the key space) -- index partitions
// Buckets of up to 256 entries are sorted (four registers per lane); a segment with a larger one (one query with
// thousands of candidates on a target: tandem repeats; a low-complexity partition) is reported for the caller's global
// bitonic pass, as are segments beyond PBA_SS_EPT entries per thread.
#define PBA_SS_EPT 16
#define PBA_SS_MAXBKT 1024
#ifndef PBA_SS_AVG
#define PBA_SS_AVG 40          // entries per bucket aimed at (upper end; tuning hook)
#endif
struct SegBkt {
    int mode;
    uint32_t mask, mv[5];     // mode 1: compress(key, mask) (Hacker's Delight 7-4)
    int care;
};
struct SegRef { uint32_t off, n; };
__device__ __forceinline__ uint64_t shfl_xor64(uint64_t x, int j) {
    return (uint64_t)(uint32_t)__shfl_xor((int)(uint32_t)x, j, PBA_WAVE) | (uint64_t)(uint32_t)__shfl_xor((int)(uint32_t)(x >> 32), j, PBA_WAVE) << 32;
}
__device__ __forceinline__ void wave_sort64(uint64_t &x, int lane) {          // 64 entries, one per lane, ascending by lane
#pragma unroll
    for (int k = 2; k <= PBA_WAVE; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            const uint64_t y = shfl_xor64(x, j);
            const bool keep_min = ((lane & j) == 0) == ((lane & k) == 0);
            x = keep_min ? (x < y ? x : y) : (x > y ? x : y);
        }
    }
}
__device__ __forceinline__ void wave_sort256(uint64_t (&x)[4], int lane) {   // entry r * 64 + lane in x[r]
#pragma unroll
    for (int k = 2; k <= 4 * PBA_WAVE; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= PBA_WAVE) {
                const int rj = j / PBA_WAVE;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (r & rj) continue;
                    const bool up = ((r * PBA_WAVE) & k) == 0;
                    const uint64_t a = x[r], b = x[r | rj];
                    if ((a > b) == up) { x[r] = b; x[r | rj] = a; }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const uint64_t y = shfl_xor64(x[r], j);
                    const bool keep_min = ((lane & j) == 0) == (((r * PBA_WAVE + lane) & k) == 0);
                    x[r] = keep_min ? (x[r] < y ? x[r] : y) : (x[r] > y ? x[r] : y);
                }
            }
        }
    }
}
// segs: SegRef per segment, or (segs == nullptr) seg_off[i] .. seg_off[i + 1].  src == dst is allowed (every entry is in a
// register before the first store).  oversize[0] counts, oversize[1 + k] names the segments left to the caller.
template <int T>
static __global__ void __launch_bounds__(T)
k_seg_sort(const uint64_t *src, uint64_t *dst, const uint32_t *seg_off, const SegRef *segs, SegBkt bk, uint32_t *oversize,
           uint32_t oversize_cap, uint32_t seg_base) {        // (seg_base: this launch's first segment -- what it reports is seg_base + its own index)
    __shared__ uint32_t hist[PBA_SS_MAXBKT + 2], cur[PBA_SS_MAXBKT + 2], wsum[T / PBA_WAVE];
    __shared__ unsigned long long s_min, s_max;
    const uint32_t lo = segs ? segs[blockIdx.x].off : seg_off[blockIdx.x];
    const uint32_t n = segs ? segs[blockIdx.x].n : seg_off[blockIdx.x + 1] - lo;
    if (n < 2) { if (n == 1 && src != dst && threadIdx.x == 0) dst[lo] = src[lo]; return; }
    if (n > T * PBA_SS_EPT) {                                        // beyond this launch: the caller's fallback
        if (src != dst) for (uint32_t i = threadIdx.x; i < n; i += T) dst[lo + i] = src[lo + i];
        if (threadIdx.x == 0) { const uint32_t k = atomicAdd(&oversize[0], 1u); if (k < oversize_cap) oversize[1 + k] = seg_base + blockIdx.x; }
        return;
    }
    const int lane = threadIdx.x & (PBA_WAVE - 1), wave = threadIdx.x / PBA_WAVE;
    // 20-40 entries per bucket: a bucket takes the 64 lanes of a wavefront whatever it holds, so fuller is cheaper, and
    // at 40 on average one bucket in 5 000 outgrows the one-register sort (measured: 8-16 per bucket left the sort as slow
    // as the LDS network it replaces)
    uint32_t nbkt = 1;
    while (nbkt < PBA_SS_MAXBKT && nbkt * PBA_SS_AVG < n) nbkt <<= 1;
    const int lg = 31 - __builtin_clz(nbkt);
    uint64_t e[PBA_SS_EPT];
    unsigned long long mn = ~0ull, mx = 0ull;
#pragma unroll
    for (int k = 0; k < PBA_SS_EPT; ++k) {
        const uint32_t i = (uint32_t)k * T + threadIdx.x;
        e[k] = i < n ? src[lo + i] : ~0ull;
        if (e[k] != ~0ull) { mn = e[k] < mn ? e[k] : mn; mx = e[k] > mx ? e[k] : mx; }
    }
    for (uint32_t b = threadIdx.x; b < nbkt + 2; b += T) hist[b] = 0;
    if (threadIdx.x == 0) { s_min = ~0ull; s_max = 0ull; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // (in place: every load has returned before any store)
    __syncthreads();
    int shift = 0;
    if (bk.mode == 0) {
#pragma unroll
        for (int d = 1; d < PBA_WAVE; d <<= 1) {
            const unsigned long long a = shfl_xor64(mn, d), b = shfl_xor64(mx, d);
            mn = a < mn ? a : mn; mx = b > mx ? b : mx;
        }
        if (lane == 0) { atomicMin(&s_min, mn); atomicMax(&s_max, mx); }
        __syncthreads();
        mn = s_min; mx = s_max;
        const unsigned long long range = mx >= mn ? mx - mn : 0ull;
        const int bits = range ? 64 - __builtin_clzll(range) : 0;
        shift = bits > lg ? min(bits - lg, 63) : 0;
    }
    auto bucket = [&](uint64_t v) -> uint32_t {
        if (v == ~0ull) return nbkt;                                 // empty slots: behind everything, never sorted
        if (bk.mode == 0) return (uint32_t)((v - mn) >> shift);
        uint32_t x = (uint32_t)(v >> 32) & bk.mask;
#pragma unroll
        for (int i = 0; i < 5; ++i) { const uint32_t t = x & bk.mv[i]; x = (x ^ t) | (t >> (1 << i)); }
        return lg == 0 ? 0u : (bk.care > lg ? x >> (bk.care - lg) : x);              // (lg == 0: one bucket; a shift by 32 is not a shift)
    };
#pragma unroll
    for (int k = 0; k < PBA_SS_EPT; ++k)
        if ((uint32_t)k * T + threadIdx.x < n) atomicAdd(&hist[bucket(e[k])], 1u);
    __syncthreads();
    // exclusive scan of hist[0 .. nbkt) (T >= nbkt: one bin per thread): cur[b] = first slot of bucket b; the empty slots'
    // bucket starts behind the last real one
    {
        const uint32_t c0 = threadIdx.x < nbkt ? hist[threadIdx.x] : 0u;
        uint32_t inc = c0;
#pragma unroll
        for (int d = 1; d < PBA_WAVE; d <<= 1) { const uint32_t t = __shfl_up(inc, d, PBA_WAVE); if (lane >= d) inc += t; }
        if (lane == PBA_WAVE - 1) wsum[wave] = inc;
        __syncthreads();
        uint32_t carry = 0;
        for (int w = 0; w < wave; ++w) carry += wsum[w];
        if (threadIdx.x < nbkt) cur[threadIdx.x] = carry + inc - c0;
        if (threadIdx.x == nbkt - 1) cur[nbkt] = carry + inc;
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < PBA_SS_EPT; ++k)
        if ((uint32_t)k * T + threadIdx.x < n) dst[lo + atomicAdd(&cur[bucket(e[k])], 1u)] = e[k];
    __threadfence_block();
    __syncthreads();                                                 // cur[b] is now the END of bucket b
    bool too_big = false;
    for (uint32_t b = wave; b < nbkt; b += T / PBA_WAVE) {
        const uint32_t c = hist[b], first = cur[b] - c;
        if (c < 2) continue;
        uint64_t *p = dst + lo + first;
        if (c <= PBA_WAVE) {
            uint64_t x = (uint32_t)lane < c ? p[lane] : ~0ull;
            wave_sort64(x, lane);
            if ((uint32_t)lane < c) p[lane] = x;
        } else if (c <= 4 * PBA_WAVE) {
            uint64_t x[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) x[r] = (uint32_t)(r * PBA_WAVE + lane) < c ? p[r * PBA_WAVE + lane] : ~0ull;
            wave_sort256(x, lane);
#pragma unroll
            for (int r = 0; r < 4; ++r) if ((uint32_t)(r * PBA_WAVE + lane) < c) p[r * PBA_WAVE + lane] = x[r];
        } else too_big = true;
    }
    if (__builtin_amdgcn_ballot_w64(too_big) && lane == 0) {
        const uint32_t k = atomicAdd(&oversize[0], 1u);
        if (k < oversize_cap) oversize[1 + k] = seg_base + blockIdx.x;
    }
}

// global-memory bitonic step for the segments k_seg_sort leaves to the caller (low-complexity targets, tandem repeats)
static __global__ void k_bitonic_step(uint64_t *buf, uint32_t N, uint32_t k, uint32_t j) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (N >> 1)) return;
    const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
    const uint32_t l = i | j;
    const uint64_t x = buf[i], y = buf[l];
    const bool up = (i & k) == 0;
    if ((x > y) == up) { buf[i] = y; buf[l] = x; }
}
static __global__ void k_fill_u64(uint64_t *buf, uint32_t from, uint32_t to, uint64_t v) {
    const uint32_t i = from + blockIdx.x * blockDim.x + threadIdx.x;
    if (i < to) buf[i] = v;
}

// batch find, two launches: counts, then positions in list order
static __global__ void k_find_count(IndexDev ix, const uint32_t *keys, uint32_t n, uint32_t *beg, uint32_t *cnt) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    uint32_t b = 0, c = 0;
    if (keys[q]) ix_find(ix, keys[q], b, c);
    beg[q] = b; cnt[q] = c;
}
static __global__ void k_find_fill(IndexDev ix, const uint32_t *beg, const uint64_t *hit_off, uint32_t n, int32_t *hit_pos,
                            uint64_t cap) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    const uint64_t o = hit_off[q], c = hit_off[q + 1] - o;
    for (uint64_t h = 0; h < c && o + h < cap; ++h)
        hit_pos[o + h] = ix_pos_of(ix, (uint32_t)ix.ent[beg[q] + h]);
}

#endif
