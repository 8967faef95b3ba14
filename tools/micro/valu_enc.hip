// Microbenchmark: issue cost of VALU encodings on gfx950 (wave64), inline asm so the encoding is fixed.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = 0.999f, c = 0.5f, d = 1.5f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {0.999f, 0.998f}, p3 = {0.5f, 0.25f};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) { REP16(asm volatile("v_fmac_f32_e32 %0, %4, %5\n v_fmac_f32_e32 %1, %4, %5\n v_fmac_f32_e32 %2, %4, %5\n v_fmac_f32_e32 %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (MODE == 1) { REP16(asm volatile("v_fma_f32 %0, %4, %5, %6\n v_fma_f32 %1, %4, %5, %6\n v_fma_f32 %2, %4, %5, %6\n v_fma_f32 %3, %4, %5, %6" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c), "v"(d));) }
        if (MODE == 2) { REP16(asm volatile("v_mul_f32_e32 %0, %4, %0\n v_mul_f32_e32 %1, %4, %1\n v_mul_f32_e32 %2, %4, %2\n v_mul_f32_e32 %3, %4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
        if (MODE == 3) { REP16(asm volatile("v_add_f32_e32 %0, %4, %0\n v_add_f32_e32 %1, %4, %1\n v_add_f32_e32 %2, %4, %2\n v_add_f32_e32 %3, %4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));) }
        if (MODE == 4) { REP16(asm volatile("v_add_f32_e64 %0, %4, |%0|\n v_add_f32_e64 %1, %4, |%1|\n v_add_f32_e64 %2, %4, |%2|\n v_add_f32_e64 %3, %4, |%3|" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));) }
        if (MODE == 5) { REP16(asm volatile("v_cmp_gt_f32_e32 vcc, %4, %0\n v_cndmask_b32_e32 %0, %0, %5, vcc\n v_cmp_gt_f32_e32 vcc, %4, %1\n v_cndmask_b32_e32 %1, %1, %5, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");) }
        if (MODE == 6) { REP16(asm volatile("v_cmp_gt_f32_e64 s[20:21], %4, %0\n v_cndmask_b32_e64 %0, %0, %5, s[20:21]\n v_cmp_gt_f32_e64 s[22:23], %4, %1\n v_cndmask_b32_e64 %1, %1, %5, s[22:23]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "s20", "s21", "s22", "s23");) }
        if (MODE == 7) { REP16(asm volatile("v_exp_f32_e32 %0, %0\n v_exp_f32_e32 %1, %1\n v_exp_f32_e32 %2, %2\n v_exp_f32_e32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (MODE == 9) { REP16(asm volatile("v_pk_fma_f32 %0, %2, %3, %0\n v_pk_fma_f32 %1, %2, %3, %1\n v_pk_fma_f32 %0, %2, %3, %0\n v_pk_fma_f32 %1, %2, %3, %1" : "+v"(p0), "+v"(p1) : "v"(p2), "v"(p3));) }
        if (MODE == 10) { REP16(asm volatile("v_pk_mul_f32 %0, %2, %0\n v_pk_mul_f32 %1, %2, %1\n v_pk_mul_f32 %0, %2, %0\n v_pk_mul_f32 %1, %2, %1" : "+v"(p0), "+v"(p1) : "v"(p2), "v"(p3));) }
        if (MODE == 11) { REP16(asm volatile("v_pk_add_f32 %0, %2, %0\n v_pk_add_f32 %1, %2, %1\n v_pk_add_f32 %0, %2, %0\n v_pk_add_f32 %1, %2, %1" : "+v"(p0), "+v"(p1) : "v"(p2), "v"(p3));) }
        if (MODE == 12) { REP16(asm volatile("v_fmac_f32_e32 %0, s20, %4\n v_fmac_f32_e32 %1, s21, %4\n v_fmac_f32_e32 %2, s22, %4\n v_fmac_f32_e32 %3, s23, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c) : "s20", "s21", "s22", "s23");) }
        if (MODE == 13) { REP16(asm volatile("v_fmac_f32_e32 %0, 0x3e99999a, %4\n v_fmac_f32_e32 %1, 0x3e99999a, %4\n v_fmac_f32_e32 %2, 0x3e99999a, %4\n v_fmac_f32_e32 %3, 0x3e99999a, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));) }
        if (MODE == 14) { REP16(asm volatile("v_mul_f32_e32 %0, s20, %0\n v_mul_f32_e32 %1, s21, %1\n v_mul_f32_e32 %2, s22, %2\n v_mul_f32_e32 %3, s23, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "s20", "s21", "s22", "s23");) }
        if (MODE == 15) { REP16(asm volatile("v_cmp_le_f32_e64 s[24:25], %4, %0\n v_cmp_le_f32_e64 s[26:27], %4, %1\n v_cmp_le_f32_e64 s[24:25], %4, %2\n v_cmp_le_f32_e64 s[26:27], %4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "s24", "s25", "s26", "s27");) }
        if (MODE == 16) { REP16(asm volatile("v_cmp_le_f32_e64 s[24:25], s20, %0\n v_cmp_le_f32_e64 s[26:27], s21, %1\n v_cmp_le_f32_e64 s[24:25], s22, %2\n v_cmp_le_f32_e64 s[26:27], s23, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");) }
        if (MODE == 17) { REP16(asm volatile("v_cmp_le_f32_e32 vcc, 0x3b808081, %0\n v_cmp_le_f32_e32 vcc, 0x3b808081, %1\n v_cmp_le_f32_e32 vcc, 0x3b808081, %2\n v_cmp_le_f32_e32 vcc, 0x3b808081, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "vcc");) }
        if (MODE == 18) { REP16(asm volatile("v_cndmask_b32_e64 %0, %0, %4, s[24:25]\n v_cndmask_b32_e64 %1, %1, %4, s[24:25]\n v_cndmask_b32_e64 %2, %2, %4, s[24:25]\n v_cndmask_b32_e64 %3, %3, %4, s[24:25]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "s24", "s25");) }
        if (MODE == 19) { REP16(asm volatile("v_add_f32_e32 %0, s20, %0\n v_add_f32_e32 %1, s21, %1\n v_add_f32_e32 %2, s22, %2\n v_add_f32_e32 %3, s23, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "s20", "s21", "s22", "s23");) }
        if (MODE == 20) { REP16(asm volatile("v_fma_f32 %0, %4, %5, %0\n v_mov_b32_dpp %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_fma_f32 %2, %4, %5, %2\n v_mov_b32_dpp %3, %2 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (MODE == 8) { REP16(asm volatile("v_mov_b32_e32 %0, %4\n v_mov_b32_e32 %1, %4\n v_mov_b32_e32 %2, %4\n v_mov_b32_e32 %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + p0.x + p0.y + p1.x + p1.y;
}
template <int MODE> float run(float* out, int grid, int iters) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    return ms;
}
int main() {
    float* out; (void)hipMalloc(&out, 256 * 4096 * sizeof(float));
    const int iters = 2000, wps = 4, grid = 256 * wps;
    const char* names[] = {"v_fmac_f32_e32 (VOP2)", "v_fma_f32 (VOP3, 3 vgpr)", "v_mul_f32_e32", "v_add_f32_e32", "v_add_f32_e64 |abs|", "v_cmp_e32+v_cndmask_e32 (vcc)", "v_cmp_e64+v_cndmask_e64 (sgpr)", "v_exp_f32", "v_mov_b32", "v_pk_fma_f32 (vgpr pairs)", "v_pk_mul_f32", "v_pk_add_f32", "v_fmac_f32_e32 (sgpr src0)", "v_fmac_f32_e32 (literal src0)", "v_mul_f32_e32 (sgpr src0)", "v_cmp_le_f32_e64 -> sgpr pair (vgpr srcs)", "v_cmp_le_f32_e64 -> sgpr pair (sgpr src0)", "v_cmp_le_f32_e32 vcc (literal src0)", "v_cndmask_b32_e64 (sgpr-pair mask)", "v_add_f32_e32 (sgpr src0)", "v_fma_f32 + v_mov_b32_dpp row_shr:1"};
    float ms[21] = {run<0>(out, grid, iters), run<1>(out, grid, iters), run<2>(out, grid, iters), run<3>(out, grid, iters), run<4>(out, grid, iters),
                   run<5>(out, grid, iters), run<6>(out, grid, iters), run<7>(out, grid, iters), run<8>(out, grid, iters), run<9>(out, grid, iters), run<10>(out, grid, iters), run<11>(out, grid, iters), run<12>(out, grid, iters), run<13>(out, grid, iters), run<14>(out, grid, iters), run<15>(out, grid, iters), run<16>(out, grid, iters), run<17>(out, grid, iters), run<18>(out, grid, iters), run<19>(out, grid, iters), run<20>(out, grid, iters)};
    for (int m = 0; m < 21; ++m) {
        const double instr = (double)iters * 64;   // 64 instructions per iteration per wave in every mode
        printf("%-34s %.3f ms -> %.2f cycles per wave-instruction per SIMD (2.4 GHz, %d waves/SIMD)\n", names[m], ms[m], ms[m] * 1e-3 * 2.4e9 / (instr * wps), wps);
    }
    return 0;
}
