// Issue cost of the VALU instruction classes of gfx950, measured: eight independent instructions of one opcode per
// loop trip, at 4 and 8 waves per SIMD.  Prints SIMD cycles per instruction at the clock the waves measured.
// (tools/microbench/valu_rate.hip found v_fma_f32 at 2.3 cycles and v_pk_fma_f32 / v_rsq_f32 at 4.2 / 8.2; this one asks
// which other opcodes run at the 2-cycle rate.)  Build: hipcc --offload-arch=gfx950 -O3 valu_ops.hip -o valu_ops
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#define REP8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define OUTS "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)

template <int kMode> __global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* ticks) {
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    unsigned long long q0 = threadIdx.x, q1 = q0 + 1, q2 = q0 + 2, q3 = q0 + 3;
    const float m = 0.999f, c = 0.001f;
    const unsigned long long qm = 12345;
    float sm = out[0];       // (a scalar operand: uniform load)
    sm = __builtin_amdgcn_readfirstlane(sm);
    const unsigned long long mask = __ballot(threadIdx.x & 1);
    const int smi = __builtin_amdgcn_readfirstlane(__float_as_int(sm));
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#define I3(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n"
        if (kMode == 0) asm volatile(REP8(I3) : OUTS : "v"(m), "v"(c));
#undef I3
#define I3(n) "v_mul_f32 %" #n ", %" #n ", %8\n"
        if (kMode == 1) asm volatile(REP8(I3) : OUTS : "v"(m), "v"(c));
#undef I3
#define I3(n) "v_add_f32 %" #n ", %" #n ", %8\n"
        if (kMode == 2) asm volatile(REP8(I3) : OUTS : "v"(m), "v"(c));
#undef I3
#define I3(n) "v_max_f32 %" #n ", %" #n ", %8\n"
        if (kMode == 3) asm volatile(REP8(I3) : OUTS : "v"(m), "v"(c));
#undef I3
#define I3(n) "v_mov_b32 %" #n ", %8\n"
        if (kMode == 4) asm volatile(REP8(I3) : OUTS : "v"(m), "v"(c));
#undef I3
#define I3(n) "v_add_u32 %" #n ", %" #n ", %8\n"
        if (kMode == 5) asm volatile(REP8(I3) : OUTS : "v"(m), "v"(c));
#undef I3
#define I3(n) "v_and_b32 %" #n ", %" #n ", %8\n"
        if (kMode == 6) asm volatile(REP8(I3) : OUTS : "v"(m), "v"(c));
#undef I3
#define I3(n) "v_lshlrev_b32 %" #n ", 1, %" #n "\n"
        if (kMode == 7) asm volatile(REP8(I3) : OUTS : "v"(m), "v"(c));
#undef I3
#define I3(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
        if (kMode == 8) asm volatile(REP8(I3) : OUTS : "v"(m), "v"(c) : "vcc");
#undef I3
#define I3(n) "v_cvt_f32_u32 %" #n ", %" #n "\n"
        if (kMode == 9) asm volatile(REP8(I3) : OUTS : "v"(m), "v"(c));
#undef I3
#define I3(n) "v_min_u32 %" #n ", %" #n ", %8\n"
        if (kMode == 10) asm volatile(REP8(I3) : OUTS : "v"(m), "v"(c));
#undef I3
        if (kMode == 11) asm volatile("v_lshl_add_u64 %0, %0, 1, %4\n v_lshl_add_u64 %1, %1, 1, %4\n v_lshl_add_u64 %2, %2, 1, %4\n v_lshl_add_u64 %3, %3, 1, %4\n"
                                      "v_lshl_add_u64 %0, %0, 1, %4\n v_lshl_add_u64 %1, %1, 1, %4\n v_lshl_add_u64 %2, %2, 1, %4\n v_lshl_add_u64 %3, %3, 1, %4\n"
                                      : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(qm));
#define I3(n) "v_cmp_lt_f32 vcc, %" #n ", %8\n"
        if (kMode == 12) asm volatile(REP8(I3) : OUTS : "v"(m), "v"(c) : "vcc");
#undef I3
#define I3(n) "v_mul_lo_u32 %" #n ", %" #n ", %8\n"
        if (kMode == 13) asm volatile(REP8(I3) : OUTS : "v"(m), "v"(c));
#undef I3
#define I3(n) "v_mad_i32_i24 %" #n ", %" #n ", %8, %9\n"
        if (kMode == 14) asm volatile(REP8(I3) : OUTS : "v"(m), "v"(c));
#undef I3
#define I3(n) "v_fmac_f32 %" #n ", %8, %9\n"
        if (kMode == 15) asm volatile(REP8(I3) : OUTS : "v"(m), "v"(c));
#undef I3
#define I3(n) "v_mul_f32_e64 %" #n ", %" #n ", %8 clamp\n"
        if (kMode == 16) asm volatile(REP8(I3) : OUTS : "v"(m), "v"(c));
#undef I3
#define I3(n) "v_fma_f32 %" #n ", %" #n ", s12, %9\n"
        if (kMode == 17) asm volatile(REP8(I3) : OUTS : "v"(m), "v"(c) : "s12");
#undef I3
#define I3(n) "v_sub_f32 %" #n ", s12, %" #n "\n"
        if (kMode == 18) asm volatile(REP8(I3) : OUTS : "v"(m), "v"(c) : "s12");
#undef I3
#define I3(n) "v_max3_f32 %" #n ", %" #n ", %8, %9\n"
        if (kMode == 19) asm volatile(REP8(I3) : OUTS : "v"(m), "v"(c));
#undef I3
#define I3(n) "v_cvt_f16_f32 %" #n ", %" #n "\n"
        if (kMode == 20) asm volatile(REP8(I3) : OUTS : "v"(m), "v"(c));
#undef I3
#define I3(n) "v_exp_f32 %" #n ", %" #n "\n"
        if (kMode == 21) asm volatile(REP8(I3) : OUTS : "v"(m), "v"(c));
#undef I3
#define I3(n) "v_cndmask_b32_e64 %" #n ", %" #n ", %8, s[10:11]\n"
        if (kMode == 23) asm volatile(REP8(I3) : OUTS : "v"(m), "v"(c) : "s10", "s11");
#undef I3
#define I3(n) "v_cndmask_b32_e64 %" #n ", %" #n ", %8, s[10:11]\n"
        if (kMode == 24) asm volatile("s_mov_b64 s[10:11], 0x5555\n" REP8(I3) : OUTS : "v"(m), "v"(c) : "s10", "s11");
#undef I3
#define I3(n) "v_readlane_b32 s10, %" #n ", 3\n"
        if (kMode == 25) asm volatile(REP8(I3) : OUTS : "v"(m), "v"(c) : "s10");
#undef I3
#define I3(n) "v_writelane_b32 %" #n ", s12, 3\n"
        if (kMode == 26) asm volatile(REP8(I3) : OUTS : "v"(m), "v"(c) : "s12");
#undef I3
#define I3(n) "v_readfirstlane_b32 s10, %" #n "\n"
        if (kMode == 27) asm volatile(REP8(I3) : OUTS : "v"(m), "v"(c) : "s10");
#undef I3
        if (kMode == 28) asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                                      "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n" : OUTS : "v"(m), "v"(c) : "vcc");
        if (kMode == 29) asm volatile("v_cmp_lt_f32 vcc, %1, %8\n v_cndmask_b32 %0, %0, %8, vcc\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                                      "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n" : OUTS : "v"(m), "v"(c) : "vcc");
        if (kMode == 30) asm volatile("v_cmp_lt_f32 s[10:11], %1, %8\n v_cndmask_b32_e64 %0, %0, %8, s[10:11]\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                                      "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n" : OUTS : "v"(m), "v"(c) : "s10", "s11");
#define I3(n) "v_min_f32_dpp %" #n ", %" #n ", %" #n " row_ror:1 row_mask:0xf bank_mask:0xf\n"
        if (kMode == 22) asm volatile("s_nop 1\n" REP8(I3) : OUTS : "v"(m), "v"(c));
#undef I3
      }
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) { ticks[blockIdx.x * 2] = c1 - c0; ticks[blockIdx.x * 2 + 1] = w1 - w0; }
    out[1 + blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(q0 + q1 + q2 + q3);
}

template <int kMode> void run(const char* name, float* out, unsigned long long* ticks, hipEvent_t e0, hipEvent_t e1) {
    const int iters = 10000;
    for (int wps : {4, 8}) {
        const int grid = 256 * wps;
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(k<kMode>, dim3(grid), dim3(256), 0, 0, out, iters, ticks);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        }
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double per_simd = (double)iters * 32 * wps;
        double sc = 0, sw = 0;
        for (int b = 0; b < grid; ++b) { sc += (double)ticks[b * 2]; sw += (double)ticks[b * 2 + 1]; }
        const double mhz = sc / (sw / 100.0);
        printf("%-28s %d waves/SIMD: %.3f ms, %.2f SIMD cycles per instruction at the measured clock (%.0f MHz)\n", name, wps, ms,
               ms * 1e-3 * mhz * 1e6 / per_simd, mhz);
    }
}

int main() {
    float* out; CHECK(hipMalloc(&out, (1 + 256 * 8 * 256) * 4)); CHECK(hipMemset(out, 0, 4));
    unsigned long long* ticks; CHECK(hipMallocManaged(&ticks, 256 * 8 * 2 * 8));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
#define RUN(n, s) run<n>(s, out, ticks, e0, e1)
    RUN(0, "v_fma_f32"); RUN(15, "v_fmac_f32"); RUN(17, "v_fma_f32 (sgpr operand)"); RUN(1, "v_mul_f32"); RUN(16, "v_mul_f32_e64 clamp");
    RUN(2, "v_add_f32"); RUN(18, "v_sub_f32 (sgpr operand)"); RUN(3, "v_max_f32"); RUN(19, "v_max3_f32"); RUN(22, "v_min_f32_dpp row_ror");
    RUN(4, "v_mov_b32"); RUN(5, "v_add_u32"); RUN(6, "v_and_b32"); RUN(7, "v_lshlrev_b32"); RUN(10, "v_min_u32"); RUN(8, "v_cndmask_b32");
    RUN(12, "v_cmp_lt_f32"); RUN(9, "v_cvt_f32_u32"); RUN(20, "v_cvt_f16_f32"); RUN(14, "v_mad_i32_i24"); RUN(13, "v_mul_lo_u32");
    RUN(11, "v_lshl_add_u64"); RUN(21, "v_exp_f32");
    RUN(23, "v_cndmask_b32_e64 s[10:11]"); RUN(24, "v_cndmask_b32_e64 sgpr pair"); RUN(25, "v_readlane_b32"); RUN(26, "v_writelane_b32"); RUN(27, "v_readfirstlane_b32");
    RUN(28, "1 cndmask vcc + 7 fma"); RUN(29, "cmp vcc, cndmask vcc, 6 fma"); RUN(30, "cmp sgpr, cndmask sgpr, 6 fma");
    return 0;
}
