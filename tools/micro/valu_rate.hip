// Microbenchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 on gfx950 (wave64), N waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float s) {
    float a[16];
    f2 p[8];
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;
    for (int i = 0; i < 8; ++i) { p[i].x = a[2 * i]; p[i].y = a[2 * i + 1]; }
    const f2 s2 = {s, s * 1.0001f};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) a[i] = __builtin_fmaf(a[i], s, 0.5f);
        } else if (MODE == 1) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) p[i] = __builtin_elementwise_fma(p[i], s2, (f2){0.5f, 0.25f});
        } else if (MODE == 2) {   // packed multiply
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) p[i] = p[i] * s2;
        } else if (MODE == 3) {   // packed add
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) p[i] = p[i] + s2;
        } else if (MODE == 4) {   // packed fma with a broadcast (op_sel) operand taken from a scalar VGPR
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) p[i] = __builtin_elementwise_fma(p[i], (f2){a[i], a[i]}, p[(i + 1) & 7]);
        } else {   // packed fma, all three operands distinct VGPR pairs
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) p[i] = __builtin_elementwise_fma(p[i], p[(i + 3) & 7], p[(i + 1) & 7]);
        }
    }
    float acc = 0.f;
    if (MODE == 0) for (int i = 0; i < 16; ++i) acc += a[i];
    else for (int i = 0; i < 8; ++i) acc += p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main() {
    float* out; hipMalloc(&out, 256 * 4096 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int mode = 0; mode < 6; ++mode)
        for (int blocks_per_cu : {4}) {
            const int grid = 256 * blocks_per_cu;   // 256 CUs, 4 waves per block -> blocks_per_cu waves per SIMD
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                switch (mode) {
                    case 0: hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, out, iters, 0.999f); break;
                    case 1: hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, out, iters, 0.999f); break;
                    case 2: hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, out, iters, 0.999f); break;
                    case 3: hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, out, iters, 0.999f); break;
                    case 4: hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, 0, out, iters, 0.999f); break;
                    default: hipLaunchKernelGGL(k<5>, dim3(grid), dim3(256), 0, 0, out, iters, 0.999f); break;
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep == 1) {
                    const double fma_per_lane = (double)iters * 128;             // 128 FMAs per lane per iteration in both modes
                    const double instr_per_wave = mode == 0 ? fma_per_lane : fma_per_lane / 2;
                    const double waves_per_simd = blocks_per_cu;
                    const double cycles = ms * 1e-3 * 2.4e9;
                    printf("%s waves/SIMD=%d  %.3f ms  -> %.2f cycles per wave-instruction per SIMD (at 2.4 GHz), %.1f TFLOP/s\n",
                           (const char*[]){"v_fma_f32          ", "v_pk_fma (consts)  ", "v_pk_mul_f32       ", "v_pk_add_f32       ", "v_pk_fma bcast src ", "v_pk_fma 3 vgpr prs"}[mode], blocks_per_cu, ms, cycles / (instr_per_wave * waves_per_simd),
                           2.0 * fma_per_lane * 64 * 4 * grid / (ms * 1e-3) / 1e12);
                }
            }
        }
    return 0;
}
