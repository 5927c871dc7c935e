// VALU issue-rate microbenchmark for gfx950: N waves per SIMD run a loop of 32
// independent instructions of one kind; prints cycles per wave-instruction per SIMD
// (s_memtime based) for several waves/SIMD.  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)

template <int KIND>
__global__ void rate_kernel(unsigned *out, unsigned long long *cyc, int iters)
{
    unsigned a = threadIdx.x + 1, b = threadIdx.x * 3 + 7, c = 12345;
    float fa = a * 0.5f, fb = b * 0.25f, fc = 1.0f, fd = 2.0f;
    unsigned r0 = a, r1 = b, r2 = c, r3 = a + b, r4 = a ^ b, r5 = b - a, r6 = a * 3, r7 = b | 5;
    float f0 = fa, f1 = fb, f2 = fc, f3 = fd, f4 = fa + 1, f5 = fb + 1, f6 = fc + 1, f7 = fd + 1;
    float g0 = 1, g1 = 2, g2 = 3, g3 = 4, g4 = 5, g5 = 6, g6 = 7, g7 = 8;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {
            asm volatile(REP8("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n")
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a));
        } else if (KIND == 1) {
            asm volatile(REP8("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n")
                         : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(fa), "v"(fb));
        } else if (KIND == 2) {
            asm volatile(REP8("v_dot2_u32_u16 %0, %8, %9, %0\n v_dot2_u32_u16 %1, %8, %9, %1\n v_dot2_u32_u16 %2, %8, %9, %2\n v_dot2_u32_u16 %3, %8, %9, %3\n")
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
        } else if (KIND == 3) {
            asm volatile(REP8("v_perm_b32 %0, %0, %8, %9\n v_perm_b32 %1, %1, %8, %9\n v_perm_b32 %2, %2, %8, %9\n v_perm_b32 %3, %3, %8, %9\n")
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
        } else if (KIND == 4) {
            asm volatile(REP8("v_cvt_f32_i32 %0, %4\n v_cvt_f32_i32 %1, %5\n v_cvt_f32_i32 %2, %6\n v_cvt_f32_i32 %3, %7\n")
                         : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(r0), "v"(r1), "v"(r2), "v"(r3));
        } else if (KIND == 5) {
            asm volatile(REP8("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n")
                         : "+v"(*(double *)&f0), "+v"(*(double *)&f2), "+v"(*(double *)&f4), "+v"(*(double *)&f6)
                         : "v"(*(double *)&g0), "v"(*(double *)&g2));
        } else if (KIND == 6) {
            asm volatile(REP8("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n")
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a));
        } else if (KIND == 7) {
            asm volatile(REP8("v_pk_mad_u16 %0, %0, %8, %9\n v_pk_mad_u16 %1, %1, %8, %9\n v_pk_mad_u16 %2, %2, %8, %9\n v_pk_mad_u16 %3, %3, %8, %9\n")
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
        } else if (KIND == 8) {
            asm volatile(REP8("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 1\n v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 1\n v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 1\n v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 1\n")
                         : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));
        } else if (KIND == 9) {
            asm volatile(REP8("v_mad_i32_i24 %0, %0, %8, %9\n v_mad_i32_i24 %1, %1, %8, %9\n v_mad_i32_i24 %2, %2, %8, %9\n v_mad_i32_i24 %3, %3, %8, %9\n")
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
        } else if (KIND == 10) {
            asm volatile(REP8("v_lshrrev_b32 %0, 9, %0\n v_lshrrev_b32 %1, 9, %1\n v_lshrrev_b32 %2, 9, %2\n v_lshrrev_b32 %3, 9, %3\n")
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 +
        (unsigned)(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + g0 + g1 + g2 + g3 + g4 + g5 + g6 + g7);
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char *name, unsigned *out, unsigned long long *cyc)
{
    const int iters = 2000;
    for (int wps : {1, 2, 4, 8}) {
        // one block of 256*wps/... : use blocks of 256 threads (4 waves -> one per SIMD), wps blocks per CU
        int blocks = 256 * wps;
        hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks);
        hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
        double avg = 0;
        for (auto v : h) avg += v;
        avg /= blocks;
        // s_memtime ticks at 100 MHz on gfx9-family; report wall-based figure too
        double instr_per_wave = 32.0 * iters;
        printf("%-14s waves/SIMD %d: %.3f ms, %.2f ns per wave-instr per SIMD (%.2f cycles at 2.4 GHz)\n", name, wps, ms,
               ms * 1e6 / (instr_per_wave * wps), ms * 1e6 / (instr_per_wave * wps) * 2.4);
    }
}

int main()
{
    unsigned *out; unsigned long long *cyc;
    hipMalloc(&out, 256 * 8 * 256 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    run<0>("v_add_u32", out, cyc);
    run<1>("v_fma_f32", out, cyc);
    run<5>("v_pk_fma_f32", out, cyc);
    run<2>("v_dot2_u32_u16", out, cyc);
    run<3>("v_perm_b32", out, cyc);
    run<4>("v_cvt_f32_i32", out, cyc);
    run<10>("v_lshrrev_b32", out, cyc);
    run<9>("v_mad_i32_i24", out, cyc);
    run<7>("v_pk_mad_u16", out, cyc);
    run<6>("v_mul_lo_u32", out, cyc);
    run<8>("add_dpp+nop1", out, cyc);
    return 0;
}
