// dev_common.h -- device-side helpers shared by the gfx950 kernels of libpba.so.
#ifndef PBA_DEV_COMMON_H
#define PBA_DEV_COMMON_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#define PBA_WAVE 64
#define PBA_INF (1 << 29)

// A sequence set as the kernels see it: 2-bit packed bases in the reference byte layout
// (first base in bits 7:6 of a byte, /root/reference/src/dna_seq.h:147-159).
struct SeqSetDev {
    const uint8_t *packed;   // all sequences; >= 64 readable bytes after the last one
    const uint64_t *off;     // byte offset of sequence i's first packed byte
    const uint32_t *len;     // length in bases
    // The same bases as two bit planes (bit p of sequence i's stream = low / high bit of its base p, LSB first, every
    // sequence starting on a word): what the bit-vector kernels consume.  Built once per set (k_make_planes) so that
    // 32 bases of either plane are two dword loads and a funnel shift instead of a 60-instruction bit de-interleave.
    // The planes are stored side by side: plane[2w] = low word w, plane[2w+1] = high word w (one cache line for both).
    const uint32_t *plane;   // word pair 0; >= 32 zero pairs before the first and after the last sequence
    const uint64_t *poff;    // word (pair) offset of sequence i
};

// Reductions over the 64 lanes through ds_swizzle (the partner lane is in the instruction): __shfl_xor builds a vector
// of lane addresses per distance, which the compiler hoists out of the alignment loops and spills to scratch memory.
// Every lane must be active; the result is wave-uniform.
#define PBA_SWZ_XOR(v, d) __builtin_amdgcn_ds_swizzle((v), ((d) << 10) | 0x1F)
__device__ __forceinline__ int wave_min_i32(int v) {
    v = min(v, PBA_SWZ_XOR(v, 1)); v = min(v, PBA_SWZ_XOR(v, 2)); v = min(v, PBA_SWZ_XOR(v, 4));
    v = min(v, PBA_SWZ_XOR(v, 8)); v = min(v, PBA_SWZ_XOR(v, 16));
    return min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 32));
}
__device__ __forceinline__ int wave_sum_i32(int v) {
    v += PBA_SWZ_XOR(v, 1); v += PBA_SWZ_XOR(v, 2); v += PBA_SWZ_XOR(v, 4);
    v += PBA_SWZ_XOR(v, 8); v += PBA_SWZ_XOR(v, 16);
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 32);
}

__device__ __forceinline__ uint32_t ld_u32(const uint8_t *p) {
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
__device__ __forceinline__ uint64_t ld_u64(const uint8_t *p) {
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}

// The 16-base window at base `p` of a packed sequence as dna_seq::encode would return it
// (dna_seq.h:86-96: little-endian word whose byte k holds bases 4k..4k+3).  Bases at or
// beyond `len` read as code 3, which is what locator.cpp:62-63 sees past the contig (NUL).
__device__ __forceinline__ uint32_t window_key(const uint8_t *seq, uint32_t p, uint32_t len) {
    const uint64_t be = __builtin_bswap64(ld_u64(seq + (p >> 2)));
    uint32_t w = (uint32_t)((be << (2 * (p & 3))) >> 32);   // base p in bits 31:30
    const uint32_t valid = len - p;                        // caller guarantees p < len
    if (valid < 16) w |= 0xFFFFFFFFu >> (2 * valid);
    return __builtin_bswap32(w);
}

// element fetchers for the DP: k-th element of an accessor (dna_seq.h:211,221)
struct PackedFetch {
    const uint8_t *seq;   // first packed byte of the sequence
    int org;              // accessor origin (base index)
    int dir;              // +1 forward, -1 backward
    const uint32_t *pl;   // the sequence's first pair of plane words (low word, high word, low word of the next 32 bases, ...)
    __device__ __forceinline__ int operator()(int k) const {
        const int idx = org + dir * k;
        return (seq[idx >> 2] >> (6 - 2 * (idx & 3))) & 3;
    }
    // the same sequence from another origin / in another direction
    __device__ __forceinline__ PackedFetch at(int origin, int direction) const {
        return PackedFetch{seq, origin, direction, pl};
    }
};
// accessor on sequence `id` of a set (dna_seq.h:191: origin + direction)
__device__ __forceinline__ PackedFetch fetch_of(const SeqSetDev &S, uint32_t id, int org, int dir) {
    return PackedFetch{S.packed + S.off[id], org, dir, S.plane + 2 * S.poff[id]};
}
// The same for an id every lane of the wavefront agrees on: the two offsets are forced into scalar registers (the
// loads go through the vector path -- the compiler cannot prove the tables are not written by the kernel -- and their
// results would otherwise sit in four vector registers per accessor for as long as the read is worked on).
__device__ __forceinline__ uint64_t uniform_u64(uint64_t v) {
    return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v) |
           (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32;
}
__device__ __forceinline__ PackedFetch fetch_of_uniform(const SeqSetDev &S, uint32_t id, int org, int dir) {
    return PackedFetch{S.packed + uniform_u64(S.off[id]), org, dir, S.plane + 2 * uniform_u64(S.poff[id])};
}
struct ByteFetch {
    const uint8_t *org;   // accessor origin (pointer to element 0)
    int dir;
    __device__ __forceinline__ int operator()(int k) const { return org[dir * k]; }
};

// what one alignment produces (seq_aligner.h:73-81 plus the early-failure row)
struct AlnOut {
    int rc, cost, matlen_a, matlen_b, len_a, len_b, max_dst, fail_row;
    int diag;      // D(m, m), m = min(len_a, len_b): the end of the diagonal (locator.cpp:86 prints it); -1 unless the sweep got there
};

// parameter block of seq_aligner::align, seq_aligner.h:94-102.  The products are formed in
// FP64 exactly as the reference's `len * R` (int promoted to double, truncation toward zero).
__device__ __forceinline__ void aln_params(int la, int lb, double R, AlnOut &o) {
    if (lb >= la) {
        o.len_a = la;
        o.max_dst = 1 + (int)((double)la * R);
        o.len_b = min(lb, la + o.max_dst);
    } else {
        o.len_b = lb;
        o.max_dst = 1 + (int)((double)lb * R);
        o.len_a = min(la, lb + o.max_dst);
    }
    o.rc = -1;
    o.cost = o.matlen_a = o.matlen_b = o.fail_row = 0;
    o.diag = -1;
}

// band cells the reference sweep evaluates in rows 1..n (seq_aligner.h:158-161), closed form
__device__ __host__ inline long long band_cells(long long len_b, long long m, long long n) {
    if (n <= 0) return 0;
    long long t = len_b - m;
    long long k = t < 0 ? 0 : (t > n ? n : t);
    long long s_hi = k * (k + 1) / 2 + k * m + (n - k) * len_b;          // sum of min(len_b, i+m)
    long long k2 = n < m + 1 ? n : m + 1;
    long long s_lo = k2 + (n * (n + 1) / 2 - k2 * (k2 + 1) / 2) - (n - k2) * m;   // sum of max(1, i-m)
    return s_hi - s_lo + n;
}

#endif
