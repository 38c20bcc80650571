// ubench_queue.hip -- the persistent-wavefront work-queue pattern in isolation (does every wave drain and leave?)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int VARIANT>
__device__ __forceinline__ uint32_t next_slot(uint32_t *queue) {
    if (VARIANT == 0) {
        uint32_t v = 0;
        if ((threadIdx.x & 63) == 0) v = atomicAdd(queue, 1u);
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
    } else {
        const uint32_t v = atomicAdd(queue, (threadIdx.x & 63) == 0 ? 1u : 0u);
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
    }
}

template <int VARIANT>
__global__ void k(uint32_t *queue, uint32_t n, const uint32_t *work, uint32_t *out) {
    for (;;) {
        const uint32_t slot = next_slot<VARIANT>(queue);
        if (slot >= n) break;
        uint32_t acc = threadIdx.x;
        const uint32_t iters = work[slot];                  // uniform, data dependent
        for (uint32_t i = 0; i < iters; ++i) {
            acc = acc * 1664525u + 1013904223u;
            if ((acc >> 28) == 15u && i > iters) break;     // never true: keeps a divergent exit in the loop
        }
        if ((threadIdx.x & 63) == 0) out[slot] = acc | 1u;
    }
}

template <int VARIANT> int run(uint32_t *q, uint32_t *w, uint32_t *o, uint32_t n) {
    CHK(hipMemset(q, 0, 4)); CHK(hipMemset(o, 0, 4 * n));
    hipLaunchKernelGGL(k<VARIANT>, dim3(64), dim3(256), 0, 0, q, n, w, o);
    CHK(hipDeviceSynchronize());
    uint32_t *h = new uint32_t[n]; CHK(hipMemcpy(h, o, 4 * n, hipMemcpyDeviceToHost));
    uint32_t done = 0; for (uint32_t i = 0; i < n; ++i) done += h[i] != 0;
    uint32_t hq; CHK(hipMemcpy(&hq, q, 4, hipMemcpyDeviceToHost));
    printf("variant %d: %u of %u items done, queue counter %u\n", VARIANT, done, n, hq);
    return 0;
}

int main(int argc, char **argv) {
    setvbuf(stdout, NULL, _IONBF, 0);
    const int which = argc > 1 ? atoi(argv[1]) : 1;
    const uint32_t n = 5000;
    uint32_t *q, *w, *o; CHK(hipMalloc(&q, 64)); CHK(hipMalloc(&w, 4 * n)); CHK(hipMalloc(&o, 4 * n));
    uint32_t *hw = new uint32_t[n]; for (uint32_t i = 0; i < n; ++i) hw[i] = (i % 7 == 0) ? 20000 : 10;
    CHK(hipMemcpy(w, hw, 4 * n, hipMemcpyHostToDevice));
    printf("running variant %d\n", which);
    if (which == 1) run<1>(q, w, o, n); else run<0>(q, w, o, n);
    return 0;
}
