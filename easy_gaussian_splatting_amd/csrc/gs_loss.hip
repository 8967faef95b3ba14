// gs_loss.hip -- fused L1 + (1 - SSIM) loss for gfx950 ("next" row):
// the producer of v_render_colors for the blend backward inside the train step
// (/root/reference/model/gaussian.py:415-453; torchmetrics SSIM: 11x11 Gaussian window, sigma 1.5,
// K1 = 0.01, K2 = 0.03, data_range 1, mean over the un-padded interior, where the reflect padding
// never reaches -- every contributing window lies inside the image).
//
//   l1_ssim_fwd_kernel : one 32x32 tile per block.  Stages the tile + 5-pixel halo of both images
//                        (mask-composited, all 3 channels, coalesced channel-last rows) in LDS,
//                        runs the separable 11-tap window for the five moment maps, evaluates
//                        SSIM and its three partial derivatives (wrt mu_x, E[x^2], E[xy]) per
//                        pixel, and writes per-block partial sums (deterministic reduction).
//   l1_ssim_bwd_kernel : d loss / d render = window (*) derivative maps (+ L1 sign term), same
//                        tiling; the mask composite's (1 - mask) factor is applied here.
//   loss_reduce_kernel : single block, fixed order sum of the per-block partials.
// Pure HBM-streaming + LDS stencil work; no atomics.
#include "gs_common.h"

namespace gs {

constexpr int kLT = 32;               // tile edge (outputs)
constexpr int kHalo = 5;
constexpr int kLR = kLT + 2 * kHalo;  // 42 staged rows / cols
constexpr int kLRP = kLR + 1;         // padded row stride
constexpr float kC1 = 0.01f * 0.01f, kC2 = 0.03f * 0.03f;

__device__ __constant__ float kWin[11] = {1.0283800845e-03f, 7.5987581352e-03f, 3.6000772128e-02f,
                                          1.0936068951e-01f, 2.1300553771e-01f, 2.6601172486e-01f,
                                          2.1300553771e-01f, 1.0936068951e-01f, 3.6000772128e-02f,
                                          7.5987581352e-03f, 1.0283800845e-03f};

struct LossArgs {
    int H, W;
    float lambda_ssim;
    const float *render, *gt, *mask;   // [H,W,3], [H,W,3], [H,W] or null
    float* maps;                       // [3 (dmu, dxx, dxy)][3 ch][H][W]
    float* partial;                    // [nblocks][2] (l1 sum, ssim sum)
    const float* gout;                 // device scalar: d loss_total
    float* v_render;                   // [H,W,3]
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__global__ __launch_bounds__(256) void l1_ssim_fwd_kernel(const LossArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* sx = lds;                         // [3][kLR][kLRP]
    float* sy = sx + 3 * kLR * kLRP;         // [3][kLR][kLRP]
    float* hp = sy + 3 * kLR * kLRP;         // [5][kLR][kLT+1] horizontal pass of one channel
    __shared__ float red[2][4];
    const int x0 = blockIdx.x * kLT, y0 = blockIdx.y * kLT;
    const int tid = threadIdx.x;
    // ---- stage tile + halo (coordinates clamped; clamped values only feed discarded outputs)
    float l1 = 0.f;
    for (int e = tid; e < kLR * kLR * 3; e += 256) {
        const int row = e / (kLR * 3), rem = e - row * (kLR * 3);
        const int col = rem / 3, ch = rem - col * 3;
        const int gy = y0 - kHalo + row, gx = x0 - kHalo + col;
        const int cy = clampi(gy, 0, a.H - 1), cx = clampi(gx, 0, a.W - 1);
        const size_t o = ((size_t)cy * a.W + cx) * 3 + ch;
        const float g = a.gt[o];
        float r = a.render[o];
        if (a.mask) { const float m = a.mask[(size_t)cy * a.W + cx]; r = m * g + (1.f - m) * r; }
        sx[(ch * kLR + row) * kLRP + col] = r;
        sy[(ch * kLR + row) * kLRP + col] = g;
        const bool own = row >= kHalo && row < kHalo + kLT && col >= kHalo && col < kHalo + kLT && gy < a.H && gx < a.W;
        if (own) l1 += fabsf(r - g);
    }
    __syncthreads();
    float ssim_sum = 0.f;
    const size_t plane = (size_t)a.H * a.W;
    for (int ch = 0; ch < 3; ++ch) {
        const float* X = sx + ch * kLR * kLRP;
        const float* Y = sy + ch * kLR * kLRP;
        // horizontal 11-tap pass: kLR rows x kLT cols x 5 maps
        for (int e = tid; e < kLR * kLT; e += 256) {
            const int row = e / kLT, col = e - row * kLT;
            float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f, m4 = 0.f;
#pragma unroll
            for (int k = 0; k < 11; ++k) {
                const float xv = X[row * kLRP + col + k], yv = Y[row * kLRP + col + k], w = kWin[k];
                m0 = fmaf(w, xv, m0); m1 = fmaf(w, yv, m1); m2 = fmaf(w * xv, xv, m2);
                m3 = fmaf(w * yv, yv, m3); m4 = fmaf(w * xv, yv, m4);
            }
            float* h = hp + row * (kLT + 1) + col;
            h[0] = m0; h[kLR * (kLT + 1)] = m1; h[2 * kLR * (kLT + 1)] = m2; h[3 * kLR * (kLT + 1)] = m3; h[4 * kLR * (kLT + 1)] = m4;
        }
        __syncthreads();
        // vertical pass + SSIM: 4 outputs per thread
        for (int e = tid; e < kLT * kLT; e += 256) {
            const int row = e / kLT, col = e - row * kLT;
            const int gy = y0 + row, gx = x0 + col;
            float mu_x = 0.f, mu_y = 0.f, exx = 0.f, eyy = 0.f, exy = 0.f;
#pragma unroll
            for (int k = 0; k < 11; ++k) {
                const float w = kWin[k];
                const float* h = hp + (row + k) * (kLT + 1) + col;
                mu_x = fmaf(w, h[0], mu_x); mu_y = fmaf(w, h[kLR * (kLT + 1)], mu_y);
                exx = fmaf(w, h[2 * kLR * (kLT + 1)], exx); eyy = fmaf(w, h[3 * kLR * (kLT + 1)], eyy);
                exy = fmaf(w, h[4 * kLR * (kLT + 1)], exy);
            }
            const bool interior = gy >= kHalo && gy < a.H - kHalo && gx >= kHalo && gx < a.W - kHalo;
            if (gy < a.H && gx < a.W) {
                float dmu = 0.f, dxx = 0.f, dxy = 0.f;
                if (interior) {
                    const float sxx = exx - mu_x * mu_x, syy = eyy - mu_y * mu_y, sxy = exy - mu_x * mu_y;
                    const float n1 = 2.f * mu_x * mu_y + kC1, n2 = 2.f * sxy + kC2;
                    const float d1 = mu_x * mu_x + mu_y * mu_y + kC1, d2 = sxx + syy + kC2;
                    const float inv = 1.f / (d1 * d2);
                    const float s = n1 * n2 * inv;
                    ssim_sum += s;
                    dxx = -s / d2;
                    dxy = 2.f * n1 * inv;
                    dmu = 2.f * mu_y * (n2 - n1) * inv - 2.f * mu_x * s / d1 + 2.f * mu_x * s / d2;
                }
                const size_t o = (size_t)gy * a.W + gx;
                a.maps[(0 * 3 + ch) * plane + o] = dmu;
                a.maps[(1 * 3 + ch) * plane + o] = dxx;
                a.maps[(2 * 3 + ch) * plane + o] = dxy;
            }
        }
        __syncthreads();
    }
    l1 = wave_reduce_add(l1);
    ssim_sum = wave_reduce_add(ssim_sum);
    if (lane_id() == 0) { red[0][tid >> 6] = l1; red[1][tid >> 6] = ssim_sum; }
    __syncthreads();
    if (tid == 0) {
        const int bidx = blockIdx.y * gridDim.x + blockIdx.x;
        a.partial[2 * bidx] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        a.partial[2 * bidx + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}

// out[0] = l1, out[1] = 1 - ssim, out[2] = (1-lambda) l1 + lambda (1 - ssim)
__global__ __launch_bounds__(256) void loss_reduce_kernel(int nblocks, const float* __restrict__ partial, int H, int W,
                                                          float lambda_ssim, float* __restrict__ out) {
    __shared__ double red[2][4];
    double s0 = 0.0, s1 = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) { s0 += partial[2 * i]; s1 += partial[2 * i + 1]; }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { s0 += __shfl_xor(s0, d, 64); s1 += __shfl_xor(s1, d, 64); }
    if (lane_id() == 0) { red[0][threadIdx.x >> 6] = s0; red[1][threadIdx.x >> 6] = s1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double l1 = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) / ((double)H * W * 3.0);
        const double cnt = (double)(H - 2 * kHalo) * (double)(W - 2 * kHalo) * 3.0;
        const double ssim = (red[1][0] + red[1][1] + red[1][2] + red[1][3]) / cnt;
        out[0] = (float)l1; out[1] = (float)(1.0 - ssim);
        out[2] = (float)((1.0 - lambda_ssim) * l1 + lambda_ssim * (1.0 - ssim));
    }
}

__global__ __launch_bounds__(256) void l1_ssim_bwd_kernel(const LossArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* sm = lds;                       // [3 maps][kLR][kLRP] of one channel
    float* hp = sm + 3 * kLR * kLRP;       // [3][kLR][kLT+1]
    const int x0 = blockIdx.x * kLT, y0 = blockIdx.y * kLT;
    const int tid = threadIdx.x;
    const size_t plane = (size_t)a.H * a.W;
    const float g = a.gout[0];
    const float cnt = (float)(a.H - 2 * kHalo) * (float)(a.W - 2 * kHalo) * 3.f;
    const float k_ssim = -g * a.lambda_ssim / cnt;                     // d(1 - mean ssim)
    const float k_l1 = g * (1.f - a.lambda_ssim) / ((float)a.H * (float)a.W * 3.f);
    for (int ch = 0; ch < 3; ++ch) {
        for (int e = tid; e < 3 * kLR * kLR; e += 256) {
            const int mi = e / (kLR * kLR), rem = e - mi * (kLR * kLR);
            const int row = rem / kLR, col = rem - row * kLR;
            const int gy = y0 - kHalo + row, gx = x0 - kHalo + col;
            float v = 0.f;   // derivative maps are zero outside the image (and outside the interior)
            if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) v = a.maps[(mi * 3 + ch) * plane + (size_t)gy * a.W + gx];
            sm[(mi * kLR + row) * kLRP + col] = v;
        }
        __syncthreads();
        for (int e = tid; e < kLR * kLT; e += 256) {
            const int row = e / kLT, col = e - row * kLT;
            float m0 = 0.f, m1 = 0.f, m2 = 0.f;
#pragma unroll
            for (int k = 0; k < 11; ++k) {
                const float w = kWin[k];
                m0 = fmaf(w, sm[(0 * kLR + row) * kLRP + col + k], m0);
                m1 = fmaf(w, sm[(1 * kLR + row) * kLRP + col + k], m1);
                m2 = fmaf(w, sm[(2 * kLR + row) * kLRP + col + k], m2);
            }
            float* h = hp + row * (kLT + 1) + col;
            h[0] = m0; h[kLR * (kLT + 1)] = m1; h[2 * kLR * (kLT + 1)] = m2;
        }
        __syncthreads();
        for (int e = tid; e < kLT * kLT; e += 256) {
            const int row = e / kLT, col = e - row * kLT;
            const int gy = y0 + row, gx = x0 + col;
            if (gy >= a.H || gx >= a.W) continue;
            float c0 = 0.f, c1 = 0.f, c2 = 0.f;
#pragma unroll
            for (int k = 0; k < 11; ++k) {
                const float w = kWin[k];
                const float* h = hp + (row + k) * (kLT + 1) + col;
                c0 = fmaf(w, h[0], c0); c1 = fmaf(w, h[kLR * (kLT + 1)], c1); c2 = fmaf(w, h[2 * kLR * (kLT + 1)], c2);
            }
            const size_t o = ((size_t)gy * a.W + gx) * 3 + ch;
            const float gtv = a.gt[o];
            float r = a.render[o], keep = 1.f;
            if (a.mask) { const float m = a.mask[(size_t)gy * a.W + gx]; r = m * gtv + (1.f - m) * r; keep = 1.f - m; }
            const float d = r - gtv;
            const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
            a.v_render[o] = keep * (k_ssim * (c0 + 2.f * r * c1 + gtv * c2) + k_l1 * sgn);
        }
        __syncthreads();
    }
}

}  // namespace gs

using namespace gs;

extern "C" size_t gs_loss_workspace_floats(int height, int width) {
    const size_t nb = (size_t)((width + kLT - 1) / kLT) * ((height + kLT - 1) / kLT);
    return 9 * (size_t)height * width + 2 * nb;
}

extern "C" int gs_l1_ssim_fwd(void* stream, int height, int width, float lambda_ssim, const float* render,
                              const float* gt, const float* mask, float* workspace, float* out3) {
    GS_REQUIRE(height > 2 * kHalo && width > 2 * kHalo, "image must be larger than the 11x11 window");
    GS_REQUIRE(render && gt && workspace && out3, "null pointer");
    LossArgs a;
    a.H = height; a.W = width; a.lambda_ssim = lambda_ssim; a.render = render; a.gt = gt; a.mask = mask;
    a.maps = workspace; a.partial = workspace + 9 * (size_t)height * width; a.gout = nullptr; a.v_render = nullptr;
    dim3 grid((width + kLT - 1) / kLT, (height + kLT - 1) / kLT);
    const size_t lds = sizeof(float) * (6 * kLR * kLRP + 5 * kLR * (kLT + 1));
    hipStream_t st = (hipStream_t)stream;
    if (lds > 64 * 1024)
        GS_HIP_CHECK(hipFuncSetAttribute((const void*)l1_ssim_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(l1_ssim_fwd_kernel, grid, dim3(256), lds, st, a);
    GS_LAUNCH_CHECK("l1_ssim_fwd_kernel");
    hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(256), 0, st, (int)(grid.x * grid.y), a.partial, height, width,
                       lambda_ssim, out3);
    GS_LAUNCH_CHECK("loss_reduce_kernel");
    return GS_OK;
}

extern "C" int gs_l1_ssim_bwd(void* stream, int height, int width, float lambda_ssim, const float* render,
                              const float* gt, const float* mask, const float* workspace, const float* v_total,
                              float* v_render) {
    GS_REQUIRE(height > 2 * kHalo && width > 2 * kHalo, "image must be larger than the 11x11 window");
    GS_REQUIRE(render && gt && workspace && v_total && v_render, "null pointer");
    LossArgs a;
    a.H = height; a.W = width; a.lambda_ssim = lambda_ssim; a.render = render; a.gt = gt; a.mask = mask;
    a.maps = const_cast<float*>(workspace); a.partial = nullptr; a.gout = v_total; a.v_render = v_render;
    dim3 grid((width + kLT - 1) / kLT, (height + kLT - 1) / kLT);
    const size_t lds = sizeof(float) * (3 * kLR * kLRP + 3 * kLR * (kLT + 1));
    hipLaunchKernelGGL(l1_ssim_bwd_kernel, grid, dim3(256), lds, (hipStream_t)stream, a);
    GS_LAUNCH_CHECK("l1_ssim_bwd_kernel");
    return GS_OK;
}
