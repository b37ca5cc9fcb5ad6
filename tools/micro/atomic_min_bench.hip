// Micro-benchmark: throughput of no-return 64-bit atomicMin into a framebuffer-sized array in HBM,
// for the access patterns a direct (bin-free) small-triangle rasterizer would produce.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void k_min(unsigned long long* vis, uint32_t n_pix, uint32_t n_ops, uint32_t mode, uint32_t width) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_ops) return;
    uint32_t p;
    if (mode == 0) p = i % n_pix;                                   // consecutive lanes -> consecutive pixels (row-major)
    else if (mode == 1) p = (uint32_t)(((uint64_t)i * 2654435761u) % n_pix);   // scattered
    else {                                                           // tile-major 32x32: consecutive lanes walk a row of a scene that maps to tiles
        uint32_t q = i % n_pix;
        uint32_t x = q % width, y = q / width;
        uint32_t tx = x >> 5, ty = y >> 5;
        p = ((ty * (width >> 5) + tx) << 10) + ((y & 31) << 5) + (x & 31);
    }
    unsigned long long v = ((unsigned long long)(0x3F000000u + (i & 0xFFFF)) << 32) | i;
    __hip_atomic_fetch_min(&vis[p], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

int main() {
    const uint32_t width = 3840, height = 2176;
    const uint32_t n_pix = width * height;
    unsigned long long* vis;
    CK(hipMalloc(&vis, (size_t)n_pix * 8));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (uint32_t mode = 0; mode < 3; ++mode) {
        for (uint32_t mult = 1; mult <= 2; ++mult) {
            uint32_t n_ops = n_pix * mult;
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipMemsetAsync(vis, 0xFF, (size_t)n_pix * 8, 0));
                CK(hipEventRecord(a, 0));
                hipLaunchKernelGGL(k_min, dim3((n_ops + 255) / 256), dim3(256), 0, 0, vis, n_pix, n_ops, mode, width);
                CK(hipEventRecord(b, 0));
                CK(hipEventSynchronize(b));
                float ms; CK(hipEventElapsedTime(&ms, a, b));
                if (ms < best) best = ms;
            }
            printf("mode %u ops %u: %.4f ms  (%.1f Gatomics/s)\n", mode, n_ops, best, n_ops / best * 1e-6);
        }
    }
    // memset cost alone
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(a, 0));
        CK(hipMemsetAsync(vis, 0xFF, (size_t)n_pix * 8, 0));
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    printf("memset %zu bytes: %.4f ms\n", (size_t)n_pix * 8, best);
    CK(hipFree(vis));
    return 0;
}
