// align_common.h -- what every aligning kernel of libpba.so shares: the per-launch configuration, the wavefronts-per-
// workgroup choice, the kernel dispatch by NB and the work queue of the persistent wavefronts.
#ifndef PBA_ALIGN_COMMON_H
#define PBA_ALIGN_COMMON_H

#include "align_bitvec.h"
#include "align_rowsweep.h"
#include "dev_common.h"
#include "pba.h"

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
// waves per SIMD the register allocator leaves room for (2nd __launch_bounds__ argument); tuning hooks -DPBA_BV_OCC12=n
// (one or two blocks per lane) and -DPBA_BV_OCC34=n.  Measured on BASELINE configs[1] (NB = 2, profiles/r02_*): 5 -> 53.9 ms per
// step, 6 -> 51.3, 7 -> 49.7, 8 -> 48.7: the step loop of two blocks fits 64 registers, and two more resident wavefronts
// per SIMD fill the issue slots the ramp of every alignment leaves.
#ifndef PBA_BV_OCC12
#define PBA_BV_OCC12 8
#endif
#ifndef PBA_BV_OCC34
#define PBA_BV_OCC34 6
#endif
template <int NB> struct Wpb {
    static constexpr int v = NB ? 4 : 1;
    // waves per SIMD the register allocator must leave room for (2nd __launch_bounds__ argument)
    static constexpr int occ = NB == 0 ? 1 : (NB <= 2 ? PBA_BV_OCC12 : (NB <= 4 ? PBA_BV_OCC34 : 3));
};

// need_diag: the caller reports D(m,m), the end of the diagonal (pba_result::diag_cost, locator.cpp:86)
template <int NB>
__device__ __forceinline__ void align_dispatch(const PackedFetch &fa, int la, const PackedFetch &fb, int lb,
                                               const AlignCfg &cfg, void *lds, AlnOut &o, bool need_diag = false) {
    if constexpr (NB == 0)
        align_rowsweep(fa, la, fb, lb, cfg.R, cfg.maxn, cfg.maxm, (uint16_t *)lds, cfg.row_cap, o);
    else
        align_bitvec<NB>(fa, la, fb, lb, cfg.R, cfg.maxn, cfg.maxm, cfg.full_band != 0, (uint16_t *)lds, cfg.row_cap, o, need_diag);
}

__device__ __forceinline__ void store_result(pba_result *out, const AlnOut &o) {
    if ((threadIdx.x & (PBA_WAVE - 1)) == 0) {
        const bool ok = o.rc >= 0;
        out->rc = ok ? o.rc : (o.rc == PBA_RC_UNCERTIFIED ? PBA_RC_UNCERTIFIED : -1);
        out->cost = ok ? o.cost : 0;
        out->matlen_a = ok ? o.matlen_a : 0;
        out->matlen_b = ok ? o.matlen_b : 0;
        out->len_a = o.len_a; out->len_b = o.len_b; out->max_dst = o.max_dst;
        out->diag_cost = o.diag;
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

#endif
