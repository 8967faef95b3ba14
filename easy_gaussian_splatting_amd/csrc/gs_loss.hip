// gs_loss.hip -- fused L1 + (1 - SSIM) loss for gfx950 ("next" row):
// the producer of v_render_colors for the blend backward inside the train step
// (/root/reference/model/gaussian.py:415-453; torchmetrics SSIM: 11x11 Gaussian window, sigma 1.5,
// K1 = 0.01, K2 = 0.03, data_range 1, mean over the un-padded interior, where the reflect padding
// never reaches -- every contributing window lies inside the image).
//
//   l1_ssim_fwd_kernel : one 32x32 tile per block.  Stages the tile + 5-pixel halo of both images
//                        (mask-composited, all 3 channels, coalesced channel-last rows; all global
//                        loads issued before the first LDS store) in LDS, runs the separable
//                        11-tap window for the five moment maps with register sliding windows
//                        (row x 8 columns, then column x 4 rows per thread), evaluates
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
constexpr int kLRP = kLR + 1;         // padded row stride of the staged tiles
constexpr int kHP = kLT + 1;          // row stride of the horizontal-pass buffers
static_assert(kLT * (kLT / 4) == 256 && kLR * (kLT / 8) <= 256 && kLR * 6 <= 256, "thread mapping of the separable passes");
static_assert(kLR % 2 == 0 && kLR * 3 <= 128, "staging: two 128-thread halves, one staged row each");
constexpr float kC1 = 0.01f * 0.01f, kC2 = 0.03f * 0.03f;

__device__ __constant__ float kWin[11] = {1.0283800845e-03f, 7.5987581352e-03f, 3.6000772128e-02f,
                                          1.0936068951e-01f, 2.1300553771e-01f, 2.6601172486e-01f,
                                          2.1300553771e-01f, 1.0936068951e-01f, 3.6000772128e-02f,
                                          7.5987581352e-03f, 1.0283800845e-03f};

#ifndef GS_SSIM_IEEE_DIV
#define GS_SSIM_IEEE_DIV 0
#endif
#ifndef GS_SSIM_HPASS_6
#define GS_SSIM_HPASS_6 1
#endif
struct LossArgs {
    int H, W;
    int clamp_input;                   // render is the un-clamped image: clamp to [0,1] on load, mask the gradient
    float lambda_ssim;
    const float *render, *gt, *mask;   // [H,W,3], [H,W,3], [H,W] or null
    float* maps;                       // [3 ch][H][W][3 (dmu, dxx, dxy)]: a tile row of one channel is one contiguous run
    float* partial;                    // [nblocks][2] (l1 sum, ssim sum)
    const float* gout;                 // device scalar: d loss_total
    float* v_render;                   // [H,W,3]
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__global__ __launch_bounds__(256) void l1_ssim_fwd_kernel(const LossArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* sx = lds;                         // [3][kLR][kLRP]
    float* sy = sx + 3 * kLR * kLRP;         // [3][kLR][kLRP]
    float* hp = sy + 3 * kLR * kLRP;         // [5][kLR][kHP] horizontal pass of one channel
    __shared__ float red[2][4];
    const int x0 = blockIdx.x * kLT, y0 = blockIdx.y * kLT;
    const int tid = threadIdx.x;
    // ---- stage tile + halo (coordinates clamped; clamped values only feed discarded outputs).
    // All global loads of a thread are issued before its first LDS store: hipcc keeps a
    // load -> LDS-store loop in program order (the store may alias the next load), which would
    // cost one HBM round trip per element.
    float l1 = 0.f;
    {
        // Row-wise mapping: a staged row is one contiguous run of 42 x 3 floats in the channel-last
        // images; thread j < 126 of each 128-thread half owns element j (column j/3, channel j%3,
        // computed once) of every second row -- no per-element index arithmetic.
        constexpr int kRowsPer = kLR / 2;   // 21 rows per half
        const int half = tid >> 7, j = tid & 127;
        const bool lane_on = j < kLR * 3;
        const int col = j / 3, ch = j - col * 3;
        const int gx = x0 - kHalo + col, cx = clampi(gx, 0, a.W - 1);
        const bool col_own = col >= kHalo && col < kHalo + kLT && gx < a.W;
        float rv[kRowsPer], gv[kRowsPer], mv[kRowsPer];
#pragma unroll
        for (int i = 0; i < kRowsPer; ++i) {
            rv[i] = gv[i] = mv[i] = 0.f;
            if (lane_on) {
                const int cy = clampi(y0 - kHalo + half + 2 * i, 0, a.H - 1);
                const size_t o = ((size_t)cy * a.W + cx) * 3 + ch;
                gv[i] = a.gt[o]; rv[i] = a.render[o];
                if (a.mask) mv[i] = a.mask[(size_t)cy * a.W + cx];
            }
        }
        float* dx = sx + (ch * kLR + half) * kLRP + col;
        float* dy = sy + (ch * kLR + half) * kLRP + col;
#pragma unroll
        for (int i = 0; i < kRowsPer; ++i) {
            if (lane_on) {
                const int row = half + 2 * i, gy = y0 - kHalo + row;
                const float g = gv[i];
                float r = rv[i];
                if (a.clamp_input) r = fminf(fmaxf(r, 0.f), 1.f);   // torch.clamp(render, 0, 1) of the model, folded in
                if (a.mask) r = mv[i] * g + (1.f - mv[i]) * r;
                dx[2 * i * kLRP] = r;
                dy[2 * i * kLRP] = g;
                if (col_own && row >= kHalo && row < kHalo + kLT && gy < a.H) l1 += fabsf(r - g);
            }
        }
    }
    __syncthreads();
    float ssim_sum = 0.f;
    const size_t plane = (size_t)a.H * a.W;
    for (int ch = 0; ch < 3; ++ch) {
        const float* X = sx + ch * kLR * kLRP;
        const float* Y = sy + ch * kLR * kLRP;
        // horizontal 11-tap pass with a register sliding window: one thread = one staged row x 8
        // output columns (18 + 18 LDS reads feed 8 x 5 outputs); lanes run down the rows, so the
        // odd row strides keep the reads and the writes conflict-free
#if GS_SSIM_HPASS_6
        // 42 rows x 6 column groups of 6 outputs (the last group starts at column 26 and recomputes two): 252 of the 256
        // threads work, 330 instead of 440 FMAs on the longest path (42 x 4 groups of 8 left a third of the block idle)
        if (tid < kLR * 6) {
            constexpr int kOut = 6, kWinN = kOut + 10;
            const int g = tid / kLR, row = tid - g * kLR, c0 = min(kOut * g, kLT - kOut);
#else
        if (tid < kLR * (kLT / 8)) {
            constexpr int kOut = 8, kWinN = kOut + 10;
            const int g = tid / kLR, row = tid - g * kLR, c0 = 8 * g;
#endif
            float xv[kWinN], yv[kWinN], xx[kWinN], yy[kWinN], xy[kWinN];
#pragma unroll
            for (int i = 0; i < kWinN; ++i) {
                xv[i] = X[row * kLRP + c0 + i]; yv[i] = Y[row * kLRP + c0 + i];
                xx[i] = xv[i] * xv[i]; yy[i] = yv[i] * yv[i]; xy[i] = xv[i] * yv[i];
            }
            float* h = hp + row * kHP + c0;
#pragma unroll
            for (int j = 0; j < kOut; ++j) {
                float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f, m4 = 0.f;
#pragma unroll
                for (int k = 0; k < 11; ++k) {
                    const float w = kWin[k];
                    m0 = fmaf(w, xv[j + k], m0); m1 = fmaf(w, yv[j + k], m1); m2 = fmaf(w, xx[j + k], m2);
                    m3 = fmaf(w, yy[j + k], m3); m4 = fmaf(w, xy[j + k], m4);
                }
                h[j] = m0; h[kLR * kHP + j] = m1; h[2 * kLR * kHP + j] = m2; h[3 * kLR * kHP + j] = m3; h[4 * kLR * kHP + j] = m4;
            }
        }
        __syncthreads();
        // vertical pass + SSIM: one thread = one column x 4 output rows (14 LDS reads per map)
        {
            const int q = tid / kLT, col = tid - q * kLT, r0 = 4 * q;
            float mom[5][4];
#pragma unroll
            for (int mi = 0; mi < 5; ++mi) {
                float hv[14];
#pragma unroll
                for (int i = 0; i < 14; ++i) hv[i] = hp[mi * kLR * kHP + (r0 + i) * kHP + col];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float acc = 0.f;
#pragma unroll
                    for (int k = 0; k < 11; ++k) acc = fmaf(kWin[k], hv[j + k], acc);
                    mom[mi][j] = acc;
                }
            }
            const int gx = x0 + col;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int gy = y0 + r0 + j;
                const float mu_x = mom[0][j], mu_y = mom[1][j], exx = mom[2][j], eyy = mom[3][j], exy = mom[4][j];
                const bool interior = gy >= kHalo && gy < a.H - kHalo && gx >= kHalo && gx < a.W - kHalo;
                if (gy < a.H && gx < a.W) {
                    float dmu = 0.f, dxx = 0.f, dxy = 0.f;
                    if (interior) {
                        const float sxx = exx - mu_x * mu_x, syy = eyy - mu_y * mu_y, sxy = exy - mu_x * mu_y;
                        const float n1 = 2.f * mu_x * mu_y + kC1, n2 = 2.f * sxy + kC2;
                        const float d1 = mu_x * mu_x + mu_y * mu_y + kC1, d2 = sxx + syy + kC2;
#if GS_SSIM_IEEE_DIV
                        const float inv = 1.f / (d1 * d2);
                        const float s = n1 * n2 * inv;
                        ssim_sum += s;
                        dxx = -s / d2;
                        dxy = 2.f * n1 * inv;
                        dmu = 2.f * mu_y * (n2 - n1) * inv - 2.f * mu_x * s / d1 + 2.f * mu_x * s / d2;
#else
                        // two hardware reciprocals (1 ulp) instead of four IEEE divisions (~10 instructions each on gfx950):
                        // the kernel is VALU-bound, and d1, d2 >= C1, C2 > 0 are far from any range the refinement steps guard
                        const float i1 = __builtin_amdgcn_rcpf(d1), i2 = __builtin_amdgcn_rcpf(d2), inv = i1 * i2;
                        const float s = n1 * n2 * inv;
                        ssim_sum += s;
                        dxx = -s * i2;
                        dxy = 2.f * n1 * inv;
                        dmu = 2.f * mu_y * (n2 - n1) * inv + 2.f * mu_x * s * (i2 - i1);
#endif
                    }
                    const size_t o = (size_t)gy * a.W + gx;
                    float* mp = a.maps + ((size_t)ch * plane + o) * 3;   // [ch][H][W][dmu, dxx, dxy]
                    mp[0] = dmu; mp[1] = dxx; mp[2] = dxy;
                }
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
    float* hp = sm + 3 * kLR * kLRP;       // [3][kLR][kHP]
    const int x0 = blockIdx.x * kLT, y0 = blockIdx.y * kLT;
    const int tid = threadIdx.x;
    const size_t plane = (size_t)a.H * a.W;
    const float g = a.gout[0];
    const float cnt = (float)(a.H - 2 * kHalo) * (float)(a.W - 2 * kHalo) * 3.f;
    const float k_ssim = -g * a.lambda_ssim / cnt;                     // d(1 - mean ssim)
    const float k_l1 = g * (1.f - a.lambda_ssim) / ((float)a.H * (float)a.W * 3.f);
    // Row-wise mapping as in the forward kernel: a tile row of one channel's derivative maps is one contiguous run of
    // 42 x 3 floats.  The NEXT channel's maps are requested while this channel's two passes run (a block is a chain of
    // three load -> stage -> filter rounds otherwise: 68 us for 22 us worth of HBM traffic).
    constexpr int kRowsPer = kLR / 2;
    const int half = tid >> 7, jj = tid & 127;
    const int scol = jj / 3, smi = jj - scol * 3;
    const int sgx = x0 - kHalo + scol;
    const bool lane_on = jj < kLR * 3 && sgx >= 0 && sgx < a.W;
    float v[kRowsPer];
    auto fetch = [&](int ch) {
#pragma unroll
        for (int i = 0; i < kRowsPer; ++i) {
            const int gy = y0 - kHalo + half + 2 * i;
            v[i] = 0.f;   // derivative maps are zero outside the image (and outside the interior)
            if (lane_on && gy >= 0 && gy < a.H) v[i] = a.maps[((size_t)ch * plane + (size_t)gy * a.W + sgx) * 3 + smi];
        }
    };
    fetch(0);
    for (int ch = 0; ch < 3; ++ch) {
        if (jj < kLR * 3) {
            float* d = sm + (smi * kLR + half) * kLRP + scol;
#pragma unroll
            for (int i = 0; i < kRowsPer; ++i) d[2 * i * kLRP] = v[i];
        }
#ifndef GS_LOSS_BWD_PREFETCH
#define GS_LOSS_BWD_PREFETCH 1
#endif
        if (GS_LOSS_BWD_PREFETCH && ch < 2) fetch(ch + 1);
        __syncthreads();
#if GS_SSIM_HPASS_6
        if (tid < kLR * 6) {   // horizontal pass, sliding window: one row x 6 columns per thread (see the forward kernel)
            constexpr int kOut = 6, kWinN = kOut + 10;
            const int g = tid / kLR, row = tid - g * kLR, c0 = min(kOut * g, kLT - kOut);
#else
        if (tid < kLR * (kLT / 8)) {   // horizontal pass, sliding window: one row x 8 columns per thread
            constexpr int kOut = 8, kWinN = kOut + 10;
            const int g = tid / kLR, row = tid - g * kLR, c0 = 8 * g;
#endif
#pragma unroll
            for (int mi = 0; mi < 3; ++mi) {
                float w[kWinN];
#pragma unroll
                for (int i = 0; i < kWinN; ++i) w[i] = sm[(mi * kLR + row) * kLRP + c0 + i];
#pragma unroll
                for (int j = 0; j < kOut; ++j) {
                    float acc = 0.f;
#pragma unroll
                    for (int k = 0; k < 11; ++k) acc = fmaf(kWin[k], w[j + k], acc);
                    hp[mi * kLR * kHP + row * kHP + c0 + j] = acc;
                }
            }
        }
        __syncthreads();
        {   // vertical pass: one column x 4 rows per thread
            const int q = tid / kLT, col = tid - q * kLT, r0 = 4 * q;
            float c[3][4];
#pragma unroll
            for (int mi = 0; mi < 3; ++mi) {
                float hv[14];
#pragma unroll
                for (int i = 0; i < 14; ++i) hv[i] = hp[mi * kLR * kHP + (r0 + i) * kHP + col];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float acc = 0.f;
#pragma unroll
                    for (int k = 0; k < 11; ++k) acc = fmaf(kWin[k], hv[j + k], acc);
                    c[mi][j] = acc;
                }
            }
            const int gx = x0 + col;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int gy = y0 + r0 + j;
                if (gy >= a.H || gx >= a.W) continue;
                const size_t o = ((size_t)gy * a.W + gx) * 3 + ch;
                const float gtv = a.gt[o];
                float r = a.render[o], keep = 1.f;
                if (a.clamp_input) { keep = (r >= 0.f && r <= 1.f) ? 1.f : 0.f; r = fminf(fmaxf(r, 0.f), 1.f); }   // clamp's backward
                if (a.mask) { const float m = a.mask[(size_t)gy * a.W + gx]; r = m * gtv + (1.f - m) * r; keep *= 1.f - m; }
                const float d = r - gtv;
                const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
                a.v_render[o] = keep * (k_ssim * (c[0][j] + 2.f * r * c[1][j] + gtv * c[2][j]) + k_l1 * sgn);
            }
        }
        if (!GS_LOSS_BWD_PREFETCH && ch < 2) fetch(ch + 1);
        __syncthreads();
    }
}

// torch.clamp(x, 0, 1) of /root/reference/model/gaussian.py:368 and its backward
// (gradient passes where 0 <= x <= 1, the aten convention) as single passes over the image.
__global__ __launch_bounds__(256) void clamp01_kernel(int64_t n4, int64_t n, const float* __restrict__ x,
                                                      const float* __restrict__ v_out, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) {
        const float4 a = reinterpret_cast<const float4*>(x)[i];
        float4 r;
        if (v_out) {
            const float4 g = reinterpret_cast<const float4*>(v_out)[i];
            r.x = (a.x >= 0.f && a.x <= 1.f) ? g.x : 0.f; r.y = (a.y >= 0.f && a.y <= 1.f) ? g.y : 0.f;
            r.z = (a.z >= 0.f && a.z <= 1.f) ? g.z : 0.f; r.w = (a.w >= 0.f && a.w <= 1.f) ? g.w : 0.f;
        } else {
            r.x = fminf(fmaxf(a.x, 0.f), 1.f); r.y = fminf(fmaxf(a.y, 0.f), 1.f);
            r.z = fminf(fmaxf(a.z, 0.f), 1.f); r.w = fminf(fmaxf(a.w, 0.f), 1.f);
        }
        reinterpret_cast<float4*>(out)[i] = r;
    }
    if (i == 0)
        for (int64_t k = n4 * 4; k < n; ++k) {
            const float a = x[k];
            out[k] = v_out ? ((a >= 0.f && a <= 1.f) ? v_out[k] : 0.f) : fminf(fmaxf(a, 0.f), 1.f);
        }
}

}  // namespace gs

using namespace gs;

extern "C" int gs_clamp01(void* stream, int64_t n, const float* x, const float* v_out, float* out) {
    GS_REQUIRE(n >= 0, "n >= 0");
    if (n == 0) return GS_OK;
    GS_REQUIRE(x && out, "null pointer");
    GS_REQUIRE((((uintptr_t)x | (uintptr_t)out | (uintptr_t)v_out) & 15) == 0, "buffers must be 16-byte aligned");
    const int64_t n4 = n >> 2, blocks = (n4 + 255) / 256;
    hipLaunchKernelGGL(clamp01_kernel, dim3((unsigned)(blocks > 0 ? blocks : 1)), dim3(256), 0, (hipStream_t)stream, n4, n, x, v_out, out);
    GS_LAUNCH_CHECK("clamp01_kernel");
    return GS_OK;
}

extern "C" size_t gs_loss_workspace_floats(int height, int width) {
    const size_t nb = (size_t)((width + kLT - 1) / kLT) * ((height + kLT - 1) / kLT);
    return 9 * (size_t)height * width + 2 * nb;
}

extern "C" int gs_l1_ssim_fwd(void* stream, int height, int width, float lambda_ssim, const float* render,
                              const float* gt, const float* mask, int clamp_input, float* workspace, float* out3) {
    GS_REQUIRE(height > 2 * kHalo && width > 2 * kHalo, "image must be larger than the 11x11 window");
    GS_REQUIRE(render && gt && workspace && out3, "null pointer");
    LossArgs a;
    a.H = height; a.W = width; a.lambda_ssim = lambda_ssim; a.render = render; a.gt = gt; a.mask = mask;
    a.clamp_input = clamp_input != 0;
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
                              const float* gt, const float* mask, int clamp_input, const float* workspace,
                              const float* v_total, float* v_render) {
    GS_REQUIRE(height > 2 * kHalo && width > 2 * kHalo, "image must be larger than the 11x11 window");
    GS_REQUIRE(render && gt && workspace && v_total && v_render, "null pointer");
    LossArgs a;
    a.H = height; a.W = width; a.lambda_ssim = lambda_ssim; a.render = render; a.gt = gt; a.mask = mask;
    a.clamp_input = clamp_input != 0;
    a.maps = const_cast<float*>(workspace); a.partial = nullptr; a.gout = v_total; a.v_render = v_render;
    dim3 grid((width + kLT - 1) / kLT, (height + kLT - 1) / kLT);
    const size_t lds = sizeof(float) * (3 * kLR * kLRP + 3 * kLR * (kLT + 1));
    hipLaunchKernelGGL(l1_ssim_bwd_kernel, grid, dim3(256), lds, (hipStream_t)stream, a);
    GS_LAUNCH_CHECK("l1_ssim_bwd_kernel");
    return GS_OK;
}
