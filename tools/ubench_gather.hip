// ubench_gather.hip -- how many scattered 64-byte lines per second the chip delivers, by table size and by the number of
// independent loads a lane keeps in flight (ILP): the ceiling the all-vs-all scan and its pre-sort stage live under.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_gather tools/ubench_gather.hip && tools/ubench_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int ILP>
__global__ void __launch_bounds__(256) k_gather(const uint4 *tab, uint32_t line_mask, uint32_t iters, uint32_t *sink) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0, x = gid * 2654435761u + 1u;
    for (uint32_t i = 0; i < iters; ++i) {
        uint4 v[ILP];
#pragma unroll
        for (int k = 0; k < ILP; ++k) { x = mix(x + 0x9e3779b9u); v[k] = tab[(size_t)(x & line_mask) * 4]; }   // one 16-byte read per 64-byte line
#pragma unroll
        for (int k = 0; k < ILP; ++k) acc += v[k].x ^ v[k].w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int ILP>
static double run(const uint4 *d_tab, uint32_t lines, uint32_t *d_sink, int blocks, uint32_t iters) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k_gather<ILP>, dim3(blocks), dim3(256), 0, 0, d_tab, lines - 1, iters / 8, d_sink);
    hipEventRecord(a);
    hipLaunchKernelGGL(k_gather<ILP>, dim3(blocks), dim3(256), 0, 0, d_tab, lines - 1, iters, d_sink);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return (double)blocks * 256.0 * iters * ILP / (ms * 1e-3) / 1e9;
}

int main() {
    uint32_t *d_sink; hipMalloc(&d_sink, 4);
    const int blocks = 256 * 8;                       // 8 workgroups of 4 wavefronts per CU: 8 waves per SIMD
    printf("table        ILP1   ILP2   ILP4   ILP8   (G scattered lines/s; 16-byte read per line)\n");
    for (uint64_t mb : {1ull, 4ull, 16ull, 64ull, 256ull, 1024ull, 4096ull}) {
        const uint64_t bytes = mb << 20;
        uint4 *d_tab; if (hipMalloc(&d_tab, bytes) != hipSuccess) break;
        hipMemset(d_tab, 1, bytes);
        const uint32_t lines = (uint32_t)(bytes / 64);
        const double r1 = run<1>(d_tab, lines, d_sink, blocks, 512), r2 = run<2>(d_tab, lines, d_sink, blocks, 256),
                     r4 = run<4>(d_tab, lines, d_sink, blocks, 128), r8 = run<8>(d_tab, lines, d_sink, blocks, 64);
        printf("%5llu MB  %6.1f %6.1f %6.1f %6.1f\n", (unsigned long long)mb, r1, r2, r4, r8);
        hipFree(d_tab);
    }
    return 0;
}
