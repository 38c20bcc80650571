// ubench_valu.hip -- integer VALU issue rate on gfx950: wave64 ops per cycle per SIMD for the op mix of the
// bit-vector kernel (and, or, xor, add, shift, bitop3-able), at 1..8 waves per SIMD and ILP 1/4.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int ILP, int KIND>
__global__ void k(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a[ILP], b = seed ^ threadIdx.x, c = seed * 3 + blockIdx.x;
#pragma unroll
    for (int i = 0; i < ILP; ++i) a[i] = threadIdx.x * 2654435761u + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int i = 0; i < ILP; ++i) {
                if (KIND == 0) a[i] = (a[i] & b) + c;                 // 2 ops: and, add (dependent chain per i)
                else if (KIND == 1) a[i] = (a[i] << 1) | (a[i] >> 31) ;   // shifts + or (alignbit maybe)
                else if (KIND == 2) a[i] = (a[i] ^ b) | (a[i] & c);       // logic (bitop3)
                else { float f = __uint_as_float(a[i]); f = f * 1.0001f + 0.5f; a[i] = __float_as_uint(f); }  // fma
            }
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s ^= a[i];
    if (s == 0x12345) out[0] = s;
}

template <int ILP, int KIND>
int run(const char *name, uint32_t *d, int ops_per_inner) {
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int iters = 4000;
    for (int wps = 1; wps <= 8; wps *= 2) {                 // waves per SIMD: blocks of 256 threads = 1 wave/SIMD
        const int blocks = 256 * wps;
        hipLaunchKernelGGL((k<ILP, KIND>), dim3(blocks), dim3(256), 0, 0, d, 10, 1u);
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k<ILP, KIND>), dim3(blocks), dim3(256), 0, 0, d, iters, 1u);
        CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        const double winst = (double)iters * 16 * ILP * ops_per_inner * wps;   // wave-instructions per SIMD
        printf("%-10s ILP %d waves/SIMD %d: %.3f ms -> %.1f ns per 1000 wave-instr/SIMD -> %.2f cycles/instr @2.4GHz\n", name, ILP, wps,
               ms, ms * 1e6 / winst * 1000, ms * 1e-3 * 2.4e9 / winst);
    }
    return 0;
}

int main() {
    uint32_t *d; CHK(hipMalloc(&d, 4096));
    run<1, 0>("and+add", d, 2); run<4, 0>("and+add", d, 2);
    run<1, 1>("rot", d, 1);     run<4, 1>("rot", d, 1);
    run<1, 2>("logic", d, 1);   run<4, 2>("logic", d, 1);
    run<1, 3>("fma", d, 1);     run<4, 3>("fma", d, 1);
    return 0;
}
