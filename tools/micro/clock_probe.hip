// Sustained shader clock under a VALU-saturating kernel: every wave reads the shader-clock counter
// (clock64 = s_memtime) and the constant-rate counter (wall_clock64 = s_memrealtime, hipDeviceAttributeWallClockRate kHz)
// at its first and last instruction; clock = d(s_memtime) / d(s_memrealtime) x wall rate.  Cross-checks printed next to it:
// (1) the instruction count of the kernel divided by its HIP-event duration, in cycles of that measured clock; (2) what the
// driver reports as the current sclk while the kernel runs is left to `rocm-smi --showclocks` in tools/clock_probe.sh.
//   hipcc -O3 --offload-arch=gfx950 -o clock_probe clock_probe.hip && ./clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define REP16(x) x x x x x x x x x x x x x x x x
// MODE 0: v_fma_f32 (VOP3), four independent accumulators;  1: v_exp_f32;  2: v_cmp + v_cndmask;  3: v_pk_fma_f32;
// 4: idle-ish (s_sleep): the clock of a chip that is NOT under vector load
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, long long* probe, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = 0.999f, c = 0.5f, d = 1.5f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {0.999f, 0.998f}, p3 = {0.5f, 0.25f};
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) { REP16(asm volatile("v_fma_f32 %0, %4, %5, %6\n v_fma_f32 %1, %4, %5, %6\n v_fma_f32 %2, %4, %5, %6\n v_fma_f32 %3, %4, %5, %6" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c), "v"(d));) }
        if (MODE == 1) { REP16(asm volatile("v_exp_f32_e32 %0, %0\n v_exp_f32_e32 %1, %1\n v_exp_f32_e32 %2, %2\n v_exp_f32_e32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (MODE == 2) { REP16(asm volatile("v_cmp_gt_f32_e32 vcc, %4, %0\n v_cndmask_b32_e32 %0, %0, %5, vcc\n v_cmp_gt_f32_e32 vcc, %4, %1\n v_cndmask_b32_e32 %1, %1, %5, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");) }
        if (MODE == 3) { REP16(asm volatile("v_pk_fma_f32 %0, %2, %3, %0\n v_pk_fma_f32 %1, %2, %3, %1\n v_pk_fma_f32 %0, %2, %3, %0\n v_pk_fma_f32 %1, %2, %3, %1" : "+v"(p0), "+v"(p1) : "v"(p2), "v"(p3));) }
        if (MODE == 4) { REP16(asm volatile("s_sleep 8\n s_sleep 8\n s_sleep 8\n s_sleep 8");) }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    if ((threadIdx.x & 63) == 0) {
        const int wv = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        probe[2 * wv] = c1 - c0;
        probe[2 * wv + 1] = w1 - w0;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + p0.x + p0.y + p1.x + p1.y;
}
template <int MODE> void run(const char* name, float* out, long long* probe, int wps, int iters, double wall_khz, int per_iter) {
    const int grid = 256 * wps, waves = grid * 4;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {   // the last repetition is reported: the chip has been under this load for two kernels by then
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, probe, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<long long> h(2 * waves);
    (void)hipMemcpy(h.data(), probe, sizeof(long long) * 2 * waves, hipMemcpyDeviceToHost);
    double sc = 0, sw = 0; double lo = 1e30, hi = 0;
    for (int i = 0; i < waves; ++i) {
        sc += (double)h[2 * i]; sw += (double)h[2 * i + 1];
        const double mhz = (double)h[2 * i] / (double)h[2 * i + 1] * wall_khz * 1e-3;
        lo = mhz < lo ? mhz : lo; hi = mhz > hi ? mhz : hi;
    }
    const double mhz = sc / sw * wall_khz * 1e-3;
    const double instr = (double)iters * per_iter;   // wave-instructions per wave
    // cycles per wave-instruction per SIMD: the SIMD's busy time (event-timed kernel) x measured clock / (instructions of its wps waves)
    printf("{\"kernel\": \"%s\", \"waves_per_simd\": %d, \"kernel_ms\": %.4f, \"s_memtime_per_s_memrealtime_MHz\": %.1f, \"min_wave_MHz\": %.1f, "
           "\"max_wave_MHz\": %.1f, \"wave_instr\": %.0f, \"cycles_per_wave_instr_at_measured_clock\": %.3f, \"cycles_per_wave_instr_at_2400MHz\": %.3f}\n",
           name, wps, ms, mhz, lo, hi, instr, per_iter ? ms * 1e-3 * mhz * 1e6 / (instr * wps) : 0.0, per_iter ? ms * 1e-3 * 2.4e9 / (instr * wps) : 0.0);
}
int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;   // ~6 ms per launch: long enough for the power management to settle
    int khz = 0;
    (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
    int sclk_khz = 0;
    (void)hipDeviceGetAttribute(&sclk_khz, hipDeviceAttributeClockRate, 0);
    printf("{\"wall_clock_rate_kHz\": %d, \"device_attribute_clock_rate_kHz\": %d}\n", khz, sclk_khz);
    float* out; (void)hipMalloc(&out, 256 * 4096 * 8 * sizeof(float));
    long long* probe; (void)hipMalloc(&probe, sizeof(long long) * 2 * 256 * 8 * 4 * 4);
    for (int wps : {1, 2, 4, 8}) run<0>("v_fma_f32", out, probe, wps, iters, khz, 64);
    run<1>("v_exp_f32", out, probe, 4, iters / 2, khz, 64);
    run<2>("v_cmp+v_cndmask", out, probe, 4, iters, khz, 64);
    run<3>("v_pk_fma_f32", out, probe, 4, iters, khz, 64);
    run<4>("s_sleep (idle SIMDs)", out, probe, 1, iters / 40, khz, 0);
    return 0;
}
