// ubench_ops.hip -- cycles per wave64 instruction per SIMD for the individual VALU opcodes the bit-vector kernel
// uses (inline asm, 4 independent chains, 8 waves/SIMD, all CUs busy).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define DEF(NAME, ASM)                                                                   \
    __global__ void NAME(uint32_t *out, int iters, uint32_t seed) {                      \
        uint32_t a0 = threadIdx.x, a1 = threadIdx.x * 3, a2 = seed, a3 = seed + threadIdx.x; \
        uint32_t b = seed ^ 0x5555u, c = seed * 7u + threadIdx.x;                        \
        for (int it = 0; it < iters; ++it) {                                             \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) {                             \
                asm volatile(ASM : "+v"(a0) : "v"(b), "v"(c));                           \
                asm volatile(ASM : "+v"(a1) : "v"(b), "v"(c));                           \
                asm volatile(ASM : "+v"(a2) : "v"(b), "v"(c));                           \
                asm volatile(ASM : "+v"(a3) : "v"(b), "v"(c));                           \
            }                                                                            \
        }                                                                                \
        if ((a0 ^ a1 ^ a2 ^ a3) == 0x12345) out[0] = a0;                                 \
    }

DEF(k_xor, "v_xor_b32 %0, %0, %1")
DEF(k_add, "v_add_u32 %0, %0, %1")
DEF(k_lshl, "v_lshlrev_b32 %0, 1, %0")
DEF(k_bitop3, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x4c")
DEF(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
DEF(k_or3, "v_or3_b32 %0, %0, %1, %2")
DEF(k_lshl_or, "v_lshl_or_b32 %0, %0, 1, %1")
DEF(k_alignbit, "v_alignbit_b32 %0, %0, %1, 31")
DEF(k_bfe, "v_bfe_u32 %0, %0, 1, 31")
DEF(k_xor_e64, "v_xor_b32_e64 %0, %0, %1")
DEF(k_add3, "v_add3_u32 %0, %0, %1, %2")
DEF(k_xad, "v_xad_u32 %0, %0, %1, %2")
DEF(k_fma, "v_fma_f32 %0, %0, %1, %2")
DEF(k_mov_dpp, "v_mov_b32_dpp %0, %1 wave_ror:1 row_mask:0xf bank_mask:0xf")
DEF(k_mov_dpp_row, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")

// carry-chain forms: masks live in SGPR pairs
__global__ void k_addc(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = threadIdx.x * 3, a2 = seed, a3 = seed + threadIdx.x, b = seed ^ 0x5555u;
    uint64_t c0 = 1, c1 = 2, c2 = 3, c3 = 4;
    for (int it = 0; it < iters; ++it) {
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {
            asm volatile("v_addc_co_u32 %0, %1, %0, %2, %1" : "+v"(a0), "+s"(c0) : "v"(b));
            asm volatile("v_addc_co_u32 %0, %1, %0, %2, %1" : "+v"(a1), "+s"(c1) : "v"(b));
            asm volatile("v_addc_co_u32 %0, %1, %0, %2, %1" : "+v"(a2), "+s"(c2) : "v"(b));
            asm volatile("v_addc_co_u32 %0, %1, %0, %2, %1" : "+v"(a3), "+s"(c3) : "v"(b));
        }
    }
    if ((a0 ^ a1 ^ a2 ^ a3) == 0x12345 || (c0 ^ c1 ^ c2 ^ c3) == 77) out[0] = a0;
}
__global__ void k_cmp_cnd(uint32_t *out, int iters, uint32_t seed) {   // v_cmp -> sgpr mask, v_cndmask from it: 2 instrs
    uint32_t a0 = threadIdx.x, a1 = threadIdx.x * 3, a2 = seed, a3 = seed + threadIdx.x, b = seed ^ 0x5555u;
    uint64_t c0, c1, c2, c3;
    for (int it = 0; it < iters; ++it) {
        _Pragma("unroll") for (int r = 0; r < 8; ++r) {
            asm volatile("v_cmp_gt_i32 %1, %0, %2\n\tv_cndmask_b32 %0, %0, %2, %1" : "+v"(a0), "=&s"(c0) : "v"(b));
            asm volatile("v_cmp_gt_i32 %1, %0, %2\n\tv_cndmask_b32 %0, %0, %2, %1" : "+v"(a1), "=&s"(c1) : "v"(b));
            asm volatile("v_cmp_gt_i32 %1, %0, %2\n\tv_cndmask_b32 %0, %0, %2, %1" : "+v"(a2), "=&s"(c2) : "v"(b));
            asm volatile("v_cmp_gt_i32 %1, %0, %2\n\tv_cndmask_b32 %0, %0, %2, %1" : "+v"(a3), "=&s"(c3) : "v"(b));
        }
    }
    if ((a0 ^ a1 ^ a2 ^ a3) == 0x12345) out[0] = a0;
}
DEF(k_ashr, "v_ashrrev_i32 %0, 31, %1")
DEF(k_bfe_i, "v_bfe_i32 %0, %1, 3, 1")
DEF(k_and, "v_and_b32 %0, %0, %1")
DEF(k_or, "v_or_b32 %0, %0, %1")
DEF(k_sub, "v_sub_u32 %0, %0, %1")
DEF(k_not, "v_not_b32 %0, %0")
DEF(k_bfrev, "v_bfrev_b32 %0, %0")
DEF(k_bcnt, "v_bcnt_u32_b32 %0, %0, %1")
DEF(k_min, "v_min_i32 %0, %0, %1")
DEF(k_cndmask_vcc, "v_cndmask_b32 %0, %0, %1, vcc")
DEF(k_lshl_v, "v_lshlrev_b32 %0, %1, %0")
DEF(k_perm, "v_perm_b32 %0, %0, %1, %2")

template <class K>
int run(const char *name, K kern, uint32_t *d) {
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int iters = 2000, wps = 8, blocks = 256 * wps;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 10, 1u);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, iters, 1u);
    CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    const double winst = (double)iters * 64 * wps;
    printf("%-14s %.3f ms  %.2f cycles/instr/SIMD @2.4GHz\n", name, ms, ms * 1e-3 * 2.4e9 / winst);
    return 0;
}

int main() {
    uint32_t *d; CHK(hipMalloc(&d, 4096));
#define R(n) run(#n, n, d)
    R(k_xor); R(k_add); R(k_lshl); R(k_bitop3); R(k_and_or); R(k_or3); R(k_lshl_or); R(k_alignbit); R(k_bfe);
    R(k_addc); R(k_cmp_cnd); R(k_ashr); R(k_bfe_i); R(k_and); R(k_or); R(k_sub); R(k_not); R(k_bfrev); R(k_bcnt); R(k_min);
    R(k_cndmask_vcc); R(k_lshl_v); R(k_perm);
    R(k_xor_e64); R(k_add3); R(k_xad); R(k_fma); R(k_mov_dpp); R(k_mov_dpp_row);
    return 0;
}
