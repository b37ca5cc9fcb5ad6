// VALU issue rate of gfx950, measured: independent v_fma_f32 (or v_pk_fma_f32) streams at 1..8 waves per SIMD.
// Prints SIMD cycles per instruction (2.4 GHz assumed).  Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int kMode> __global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* ticks) {
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    const float m = 0.999f, c = 0.001f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (kMode == 0) {
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        } else if (kMode == 1) {
            f2 mm = {m, m}, cc = {c, c};
            asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                         "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(mm), "v"(cc));
        } else if (kMode == 2) {      // a dependent chain
            asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                         "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                         : "+v"(a0) : "v"(m), "v"(c));
        } else {                      // transcendental
            asm volatile("v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1\n v_rsq_f32 %2, %2\n v_rsq_f32 %3, %3\n"
                         "v_rsq_f32 %4, %4\n v_rsq_f32 %5, %5\n v_rsq_f32 %6, %6\n v_rsq_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        }
      }
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) { ticks[blockIdx.x * 2] = c1 - c0; ticks[blockIdx.x * 2 + 1] = w1 - w0; }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}

int main() {
    float* out; CHECK(hipMalloc(&out, 256 * 4 * 16 * 64 * 4));
    unsigned long long* ticks; CHECK(hipMallocManaged(&ticks, 256 * 4 * 16 * 2 * 8));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 10000;
    const char* names[4] = {"v_fma_f32 x8 independent", "v_pk_fma_f32 x8 independent", "v_fma_f32 dependent chain", "v_rsq_f32 x8 independent"};
    for (int mode = 0; mode < 4; ++mode)
        for (int wps : {1, 2, 3, 4, 8}) {
            const int grid = 256 * wps;          // 256 threads = one wave per SIMD of a CU: wps waves on every SIMD when the workgroups spread evenly
            for (int rep = 0; rep < 2; ++rep) {
                CHECK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, out, iters, ticks);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, out, iters, ticks);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, out, iters, ticks);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, out, iters, ticks);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            }
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double cyc = ms * 1e-3 * 2.4e9, per_simd = (double)iters * 32 * wps;
            double sc = 0, sw = 0;
            for (int b = 0; b < grid; ++b) { sc += (double)ticks[b * 2]; sw += (double)ticks[b * 2 + 1]; }
            sc /= grid; sw /= grid;
            // per wave: shader-clock ticks (s_memtime) and 100 MHz ticks (s_memrealtime) over the loop
            printf("%-30s %d waves/SIMD: %.3f ms, %.2f cycles@2.4GHz per instruction per SIMD (%.2f at the measured clock); per wave %.0f clock64 ticks = %.2f per instruction, %.1f us -> clock64 runs at %.0f MHz\n",
                   names[mode], wps, ms, cyc / per_simd, ms * 1e-3 * (sc / (sw / 100.0)) * 1e6 / per_simd, sc, sc / (iters * 32.0), sw / 100.0, sc / (sw / 100.0));
        }
    return 0;
}
