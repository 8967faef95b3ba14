// Microbenchmark: the memory side of the SSIM forward kernel alone -- every block reads a 42x42-pixel tile (+halo) of two
// channel-last images (12-byte pixels) and writes 32x32 pixels x 3 planes x 12 bytes, no arithmetic.  Variants of the access shape.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct F3 { float x, y, z; };
constexpr int H = 1080, W = 1920, LT = 32, HALO = 5, LR = 42;
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
// MODE 0: as the kernel (thread = column x 7 rows, 12-byte loads; 12-byte stores per pixel and plane)
// MODE 1: loads only   MODE 2: stores only   MODE 3: as 0 but tiles in plain row-major block order (no XCD swizzle)
// MODE 4: loads only, 64x16 tiles (74 x 26 staged)
template <int MODE>
__global__ __launch_bounds__(256) void k(const float* gt, const float* render, float* maps, float* sink) {
    const int ntx = (W + LT - 1) / LT, nty = (H + LT - 1) / LT, nt = ntx * nty, per = (nt + 7) >> 3;
    int t;
    if (MODE == 3) t = blockIdx.x; else { t = (blockIdx.x & 7) * per + (blockIdx.x >> 3); if ((int)(blockIdx.x >> 3) >= per) return; }
    if (t >= nt) return;
    const int ty = t / ntx, x0 = (t - ty * ntx) * LT, y0 = ty * LT, tid = threadIdx.x;
    const unsigned row_bytes = 12u * W;
    float acc = 0.f;
    if (MODE != 2 && tid < LR * 6) {
        const int rg = tid / LR, col = tid - rg * LR, cx = clampi(x0 - HALO + col, 0, W - 1);
        F3 g[7], r[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const unsigned cy = clampi(y0 - HALO + rg + 6 * i, 0, H - 1), o = cy * row_bytes + 12u * cx;
            g[i] = *(const F3*)((const char*)gt + o); r[i] = *(const F3*)((const char*)render + o);
        }
#pragma unroll
        for (int i = 0; i < 7; ++i) acc += g[i].x + g[i].y + g[i].z + r[i].x + r[i].y + r[i].z;
    }
    if (MODE != 1) {
        const int q = tid / LT, col = tid - q * LT, gx = x0 + col;
        for (int ch = 0; ch < 3; ++ch)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int gy = y0 + 4 * q + j;
                if (gy < H && gx < W) *(F3*)((char*)(maps + (size_t)ch * H * W * 3) + gy * row_bytes + 12u * gx) = F3{acc, 1.f, 2.f};
            }
    } else if (acc == 123.456f) sink[tid] = acc;
}
template <int MODE> float run(const std::vector<float*>& gts, const std::vector<float*>& rs, float* maps, float* sink, int grid) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, gts[i % gts.size()], rs[i % rs.size()], maps, sink);
    (void)hipEventRecord(e0);
    const int reps = 200;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, gts[i % gts.size()], rs[i % rs.size()], maps, sink);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / reps * 1e3f;
}
int main() {
    const size_t img = (size_t)H * W * 3 * sizeof(float);
    std::vector<float*> gts(24), rs(24);
    for (auto& p : gts) { (void)hipMalloc(&p, img); (void)hipMemset(p, 0, img); }
    for (auto& p : rs) { (void)hipMalloc(&p, img); (void)hipMemset(p, 0, img); }
    float *maps, *sink; (void)hipMalloc(&maps, 3 * img); (void)hipMalloc(&sink, 4096);
    const int nt = ((W + LT - 1) / LT) * ((H + LT - 1) / LT), grid = 8 * ((nt + 7) / 8);
    printf("loads + stores (kernel's shape, XCD order): %.1f us\n", run<0>(gts, rs, maps, sink, grid));
    printf("loads only:                                 %.1f us\n", run<1>(gts, rs, maps, sink, grid));
    printf("stores only:                                %.1f us\n", run<2>(gts, rs, maps, sink, grid));
    printf("loads + stores, row-major block order:      %.1f us\n", run<3>(gts, rs, maps, sink, nt));
    std::vector<float*> one(1, gts[0]), oner(1, rs[0]);
    printf("loads + stores, one resident image pair:    %.1f us\n", run<0>(one, oner, maps, sink, grid));
    printf("loads only, one resident image pair:        %.1f us\n", run<1>(one, oner, maps, sink, grid));
    return 0;
}
