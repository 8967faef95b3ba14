// gs_loss.hip -- fused L1 + (1 - SSIM) loss for gfx950 ("next" row):
// the producer of v_render_colors for the blend backward inside the train step
// (/root/reference/model/gaussian.py:415-453; torchmetrics SSIM: 11x11 Gaussian window, sigma 1.5,
// K1 = 0.01, K2 = 0.03, data_range 1, mean over the un-padded interior, where the reflect padding
// never reaches -- every contributing window lies inside the image).
//
//   l1_ssim_fwd_kernel : one block per 32x32 tile, XCD-aware block order (loss_tile).  Requests the tile + 5-pixel
//                        halo of both images once (whole 12-byte pixels, coalesced), then per channel: stages it
//                        (mask-composited) in LDS, runs the separable 11-tap window for FOUR moment maps (x, y,
//                        x^2 + y^2, xy) with register sliding windows (row x 6 columns, then column x 4 rows
//                        per thread), evaluates SSIM and its three partial derivatives (wrt mu_x, E[x^2],
//                        E[xy]) per pixel (one 12-byte store), and writes per-block partial sums
//                        (deterministic reduction).
//   l1_ssim_bwd_kernel : one block per (tile, channel): d loss / d render = window (*) derivative maps (+ L1 sign
//                        term); the mask composite's (1 - mask) factor is applied here.
//   loss_reduce_kernel : single block, fixed order sum of the per-block partials.
// Round 5 (68 + 68 us -> 44 + 31 at 1080p inside the train step): a channel at a time in 37 / 22 KB of
// LDS instead of 71 / 38 KB, block-uniform bases with 32-bit byte offsets instead of 64-bit index arithmetic per element, the
// window taps as literal operands, whole-pixel global accesses and four moment maps instead of five in the forward.
// Pure HBM-streaming + LDS stencil work; no atomics.
#include "gs_common.h"

namespace gs {

constexpr int kLT = 32;               // tile edge (outputs)
constexpr int kHalo = 5;
constexpr int kLR = kLT + 2 * kHalo;  // 42 staged rows / cols
constexpr int kLRP = kLR + 1;         // padded row stride of the staged tiles
constexpr int kHP = kLT + 1;          // row stride of the horizontal-pass buffers
static_assert(kLT * (kLT / 4) == 256 && kLR * 6 <= 256 && kLR % 6 == 0, "thread mapping of the staging and the separable passes");
static_assert(kLR % 2 == 0 && kLR * 3 <= 128, "backward staging: two 128-thread halves, one staged row each");
constexpr int64_t kLossMaxPixels = (int64_t)1 << 28;   // 12 bytes per pixel and plane under 2^32
constexpr int kLossMaxWidth = 1 << 20;                 // a row's 12 W bytes under 2^24; the row INDEX is held under 2^24 by the entry points (24-bit multiplies)
constexpr float kC1 = 0.01f * 0.01f, kC2 = 0.03f * 0.03f;

// The window as compile-time constants: every tap is a LITERAL operand of its FMA.  From __constant__ memory the taps sat in
// scalar registers, and a VALU instruction with a scalar-register source issues every 4.4 cycles on gfx950 against 3.0 with
// vector-register or literal sources (tools/micro/valu_enc.hip) -- two thirds of these kernels' instructions.
#define GS_WIN_TAPS {1.0283800845e-03f, 7.5987581352e-03f, 3.6000772128e-02f, 1.0936068951e-01f, 2.1300553771e-01f, \
                     2.6601172486e-01f, 2.1300553771e-01f, 1.0936068951e-01f, 3.6000772128e-02f, 7.5987581352e-03f, 1.0283800845e-03f}

struct LossArgs {
    int H, W;
    int clamp_input;                   // render is the un-clamped image: clamp to [0,1] on load, mask the gradient
    float lambda_ssim;
    const float *render, *gt, *mask;   // [H,W,3], [H,W,3], [H,W] or null
    const float* const* slots;         // optional: {gt, mask} read from device memory at kernel start instead (gs_step_inputs)
    float* maps;                       // [3 ch][H][W][3 (dmu, dxx, dxy)]: a tile row of one channel is one contiguous run
    float* partial;                    // [nblocks][2] (l1 sum, ssim sum)
    const float* gout;                 // device scalar: d loss_total
    float* v_render;                   // [H,W,3]
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
// base + 32-bit BYTE offset: with a block-uniform base this is one global_load with a scalar base and a vector offset
__device__ __forceinline__ float ld_off(const float* base, unsigned byte_off) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ void st_off(float* base, unsigned byte_off, float v) {
    *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + byte_off) = v;
}

struct F3 { float x, y, z; };   // one pixel of a channel-last image / one pixel's three derivative maps: a 12-byte access
__device__ __forceinline__ F3 ld3_off(const float* base, unsigned byte_off) {
    return *reinterpret_cast<const F3*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ void st3_off(float* base, unsigned byte_off, float x, float y, float z) {
    *reinterpret_cast<F3*>(reinterpret_cast<char*>(base) + byte_off) = F3{x, y, z};
}

// Block -> tile.  Consecutive block ids go round-robin over the eight XCDs, each with its own L2: XCD x takes the contiguous
// row-major run of tiles [x * per, (x + 1) * per) -- the halos neighbouring tiles share are met in that L2.
__device__ __forceinline__ bool loss_tile(const LossArgs& a, int& x0, int& y0) {
    const int ntx = (a.W + kLT - 1) / kLT, nt = ntx * ((a.H + kLT - 1) / kLT), per = (nt + 7) >> 3;
    const int id = blockIdx.x, t = (id & 7) * per + (id >> 3);
    if ((id >> 3) >= per || t >= nt) return false;
    const int ty = t / ntx;
    x0 = (t - ty * ntx) * kLT; y0 = ty * kLT;
    return true;
}

// Block -> (tile, channel) for the backward kernel: the same XCD runs, the three channels of a tile one after the other on the
// same XCD (the channel-last image lines the three blocks share and the lines their strided stores fill are met in that L2).
__device__ __forceinline__ bool loss_tile_channel(const LossArgs& a, int& x0, int& y0, int& ch) {
    const int ntx = (a.W + kLT - 1) / kLT, nt = ntx * ((a.H + kLT - 1) / kLT), per = (nt + 7) >> 3;
    const int id = blockIdx.x, k = id >> 3, local = k / 3, t = (id & 7) * per + local;
    ch = k - 3 * local;
    if (t >= nt) return false;
    const int ty = t / ntx;
    x0 = (t - ty * ntx) * kLT; y0 = ty * kLT;
    return true;
}

// the separable window's horizontal pass over one staged row: thread = row x 6 output columns, register sliding window
// (42 rows x 6 column groups; the last group starts at column 26 and recomputes two: 252 of the 256 threads work)
constexpr int kHOut = 6, kHWin = kHOut + 10;

// One block per tile.  The tile + halo of both images is requested ONCE, a whole pixel (12 bytes, three channels) per lane
// and row: 7 + 7 (+ 7 mask) fully coalesced loads per thread; the three channels then take turns in the same 37 KB of LDS
// (four blocks per CU, four waves per SIMD at 124 registers), and a pixel's three derivatives leave as one 12-byte store.
// Measured at 1080p with a mask on inputs that are cold in every cache, as in the train step (tools/loss_time.py; the
// forward entry including its one-block reduction): round 4's three-channel block with 71 KB of LDS 68 us; a block per
// (tile, channel) with 4-byte accesses at a 12-byte stride 58-64; this form with five moment maps 51, with four 44-48.
// Round 5's timing builds (no stores / synthetic pixels instead of loads / both; since removed from the sources): 52 -> 47 / 42 /
// 39 us without a mask, and the memory side alone (tools/micro/tile_stream.hip: the same loads and stores, no arithmetic)
// 23 us: the kernel is bound by its ~900 VALU instructions per tile and channel, with the memory side two thirds hidden.
template <bool MASK>
__global__ __launch_bounds__(256) void l1_ssim_fwd_kernel(const LossArgs a) {
    constexpr float kWin[11] = GS_WIN_TAPS;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* sx = lds;                         // [kLR][kLRP] render (composited), one channel
    float* sy = sx + kLR * kLRP;             // [kLR][kLRP] ground truth
    float* hp = sy + kLR * kLRP;             // [4][kLR][kHP] horizontal sums (a buffer of their own: parked in registers across a
                                             // barrier and written over the staged tile they cost this kernel its fourth wave)
    __shared__ float red[2][4];
    int x0, y0;
    if (!loss_tile(a, x0, y0)) {   // (a block past the end of its XCD's run: the reduction sums every block's pair)
        if (threadIdx.x == 0) { a.partial[2 * blockIdx.x] = 0.f; a.partial[2 * blockIdx.x + 1] = 0.f; }
        return;
    }
    const int tid = threadIdx.x;
    const unsigned row_bytes = 12u * (unsigned)a.W;
    const float* gt_img = a.slots ? a.slots[0] : a.gt;        // (block-uniform scalar loads)
    const float* mask_img = a.slots ? a.slots[1] : a.mask;
    // ---- staging role: thread = one staged column x 7 rows (coordinates clamped; clamped values only feed discarded
    // outputs).  Addresses: a block-uniform base (scalar registers) + a 32-bit byte offset per lane (the entry points bound
    // H * W), a 24-bit multiply-add per row.
    constexpr int kPer = kLR / 6;   // 7 rows per thread
    float rv[kPer][3], gv[kPer][3], mv[MASK ? kPer : 1];
    if (tid < kLR * 6) {
        const int rg = tid / kLR, col = tid - rg * kLR;
        const int cx = clampi(x0 - kHalo + col, 0, a.W - 1), ytop = y0 - kHalo + rg;
#pragma unroll
        for (int i = 0; i < kPer; ++i) {
            const unsigned cy = (unsigned)clampi(ytop + 6 * i, 0, a.H - 1);
            const unsigned o = __umul24(cy, row_bytes) + 12u * (unsigned)cx;
            const F3 g3 = ld3_off(gt_img, o), r3 = ld3_off(a.render, o);
            gv[i][0] = g3.x; gv[i][1] = g3.y; gv[i][2] = g3.z;
            rv[i][0] = r3.x; rv[i][1] = r3.y; rv[i][2] = r3.z;
            if (MASK) mv[i] = ld_off(mask_img, __umul24(cy, 4u * (unsigned)a.W) + 4u * (unsigned)cx);
        }
    }
    float l1 = 0.f, ssim_sum = 0.f;
#pragma unroll 1
    for (int ch = 0; ch < 3; ++ch) {
        // (the thread's roles are re-derived from an opaque copy of its index every round: hoisted out of the loop, the
        //  addresses and predicates of all four phases sit in some 40 registers and cost the kernel a wave per SIMD)
        int t = tid;
        asm volatile("" : "+v"(t));
        const bool st_on = t < kLR * 6;
        const int rg = t / kLR, col = t - rg * kLR;
        if (st_on) {
            const int gx = x0 - kHalo + col;
            const bool col_own = col >= kHalo && col < kHalo + kLT && gx < a.W;
            float* dx = sx + rg * kLRP + col;
            float* dy = sy + rg * kLRP + col;
#pragma unroll
            for (int i = 0; i < kPer; ++i) {
                const int row = rg + 6 * i, gy = y0 - kHalo + row;
                const float g = gv[i][0];
                float r = rv[i][0];
                if (a.clamp_input) r = fminf(fmaxf(r, 0.f), 1.f);   // torch.clamp(render, 0, 1) of the model, folded in
                if (MASK) r = mv[i] * g + (1.f - mv[i]) * r;
                dx[6 * i * kLRP] = r;
                dy[6 * i * kLRP] = g;
                if (col_own && row >= kHalo && row < kHalo + kLT && gy < a.H) l1 += fabsf(r - g);
                gv[i][0] = gv[i][1]; gv[i][1] = gv[i][2];   // the next channel moves up (a rolled loop cannot index registers)
                rv[i][0] = rv[i][1]; rv[i][1] = rv[i][2];
            }
        }
        __syncthreads();
        float* maps_c = a.maps + (size_t)ch * a.H * a.W * 3;
        {
            // horizontal pass: 16 + 16 LDS reads feed 6 x 4 outputs; lanes run down the rows, the odd row strides keep reads and
            // writes conflict-free.  FOUR moment maps, not five: SSIM and its derivatives see E[xx] and E[yy] only through
            // their sum (d2 = sxx + syy + C2), so x^2 + y^2 goes through the window as one map.
            const bool h_on = st_on;
            const int row = col, c0 = min(kHOut * rg, kLT - kHOut);
            if (h_on) {
                float* h = hp + row * kHP + c0;
                float xv[kHWin], yv[kHWin];
#pragma unroll
                for (int i = 0; i < kHWin; ++i) { xv[i] = sx[row * kLRP + c0 + i]; yv[i] = sy[row * kLRP + c0 + i]; }
#pragma unroll
                for (int j = 0; j < kHOut; ++j) {
                    float m0 = kWin[0] * xv[j], m1 = kWin[0] * yv[j], m2 = kWin[0] * fmaf(xv[j], xv[j], yv[j] * yv[j]),
                          m3 = kWin[0] * (xv[j] * yv[j]);   // (first tap as a product: no zero to materialise per sum)
#pragma unroll
                    for (int kk = 1; kk < 11; ++kk) {
                        const float w = kWin[kk], x = xv[j + kk], y = yv[j + kk];
                        m0 = fmaf(w, x, m0); m1 = fmaf(w, y, m1); m2 = fmaf(w, fmaf(x, x, y * y), m2); m3 = fmaf(w, x * y, m3);
                    }
                    h[j] = m0; h[kLR * kHP + j] = m1; h[2 * kLR * kHP + j] = m2; h[3 * kLR * kHP + j] = m3;
                }
            }
        }
        __syncthreads();
        // vertical pass + SSIM: one thread = one column x 4 output rows (14 LDS reads per map)
        {
            const int q = t / kLT, vcol = t - q * kLT, r0 = 4 * q;
            float mom[4][4];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                float hv[14];
#pragma unroll
                for (int i = 0; i < 14; ++i) hv[i] = hp[mi * kLR * kHP + (r0 + i) * kHP + vcol];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float acc = kWin[0] * hv[j];
#pragma unroll
                    for (int kk = 1; kk < 11; ++kk) acc = fmaf(kWin[kk], hv[j + kk], acc);
                    mom[mi][j] = acc;
                }
            }
            const int gx = x0 + vcol;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int gy = y0 + r0 + j;
                const float mu_x = mom[0][j], mu_y = mom[1][j], ess = mom[2][j], exy = mom[3][j];   // ess = E[xx] + E[yy]
                const bool interior = gy >= kHalo && gy < a.H - kHalo && gx >= kHalo && gx < a.W - kHalo;
                if (gy < a.H && gx < a.W) {
                    float dmu = 0.f, dxx = 0.f, dxy = 0.f;
                    if (interior) {
                        const float mm = mu_x * mu_x + mu_y * mu_y, sxy = exy - mu_x * mu_y;
                        const float n1 = 2.f * mu_x * mu_y + kC1, n2 = 2.f * sxy + kC2;
                        const float d1 = mm + kC1, d2 = (ess - mm) + kC2;
                        // two hardware reciprocals (1 ulp) instead of four IEEE divisions (~10 instructions each on gfx950):
                        // d1, d2 >= C1, C2 > 0 are far from any range the refinement steps guard
                        const float i1 = __builtin_amdgcn_rcpf(d1), i2 = __builtin_amdgcn_rcpf(d2), inv = i1 * i2;
                        const float sv = n1 * n2 * inv;
                        ssim_sum += sv;
                        dxx = -sv * i2;
                        dxy = 2.f * n1 * inv;
                        dmu = 2.f * mu_y * (n2 - n1) * inv + 2.f * mu_x * sv * (i2 - i1);
                    }
                    // [ch][H][W][dmu, dxx, dxy]: one 12-byte store per pixel, a tile row one contiguous run
                    st3_off(maps_c, __umul24((unsigned)gy, row_bytes) + 12u * (unsigned)gx, dmu, dxx, dxy);
                }
            }
        }
        if (ch < 2) __syncthreads();   // the next channel is staged over the buffers the vertical pass has just read
    }
    l1 = wave_reduce_add(l1);
    ssim_sum = wave_reduce_add(ssim_sum);
    if (lane_id() == 0) { red[0][tid >> 6] = l1; red[1][tid >> 6] = ssim_sum; }
    __syncthreads();
    if (tid == 0) {
        a.partial[2 * blockIdx.x] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        a.partial[2 * blockIdx.x + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
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

// One block per (tile, channel): the derivative maps are channel-planar, a channel's tile rows contiguous runs.  (A block per
// tile with a channel loop, the images as whole pixels and the three answers as one 12-byte store, needs 163 registers for the
// prefetched maps, the twelve pixels and the answers: three blocks per CU, 37 us against 34.)
__global__ __launch_bounds__(256) void l1_ssim_bwd_kernel(const LossArgs a) {
    constexpr float kWin[11] = GS_WIN_TAPS;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* sm = lds;                       // [3 maps][kLR][kLRP] of this block's channel
    float* hp = lds;                       // [3][kLR][kHP], written OVER the staged maps once every thread holds its windows' sums
                                           // in registers: 22 KB per block instead of 38, seven blocks per CU instead of four (37 -> 33 us)
    int x0, y0, ch;
    if (!loss_tile_channel(a, x0, y0, ch)) return;
    const int tid = threadIdx.x;
    const size_t plane = (size_t)a.H * a.W;
    const float g = a.gout[0];
    const float cnt = (float)(a.H - 2 * kHalo) * (float)(a.W - 2 * kHalo) * 3.f;
    const float k_ssim = -g * a.lambda_ssim / cnt;                     // d(1 - mean ssim)
    const float k_l1 = g * (1.f - a.lambda_ssim) / ((float)a.H * (float)a.W * 3.f);
    // this thread's four output pixels (one column x 4 rows): their images are requested first, the answers are due last
    const unsigned row_bytes = 12u * (unsigned)a.W;
    const int q = tid / kLT, ocol = tid - q * kLT, r0 = 4 * q, ogx = x0 + ocol;
    const bool col_in = ogx < a.W;
    const unsigned o0 = __umul24((unsigned)min(y0 + r0, a.H - 1), row_bytes) + 12u * (unsigned)min(ogx, a.W - 1);
    const float* mask_img = a.slots ? a.slots[1] : a.mask;   // (block-uniform scalar loads)
    const float* gt_c = (a.slots ? a.slots[0] : a.gt) + ch;
    const float* render_c = a.render + ch;
    float gtv[4], rr[4], mk[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned o = o0 + (y0 + r0 + j < a.H ? j * row_bytes : 0u);   // (rows below the image re-read a valid one; never stored)
        gtv[j] = ld_off(gt_c, o); rr[j] = ld_off(render_c, o);
        if (mask_img) mk[j] = ld_off(mask_img, o / 3u);
    }
    {
        // a tile row of one channel's derivative maps is one contiguous run of 42 x 3 floats: lane j < 126 of each 128-thread
        // half owns element j (column j / 3, map j % 3) of every second row.  The half, hence the row, is uniform over a
        // wave: the row's address and its in-image test live in scalar registers, the lane adds one constant offset.
        constexpr int kRowsPer = kLR / 2;
        const int half = __builtin_amdgcn_readfirstlane(tid >> 7), jj = tid & 127;
        const int scol = jj / 3, smi = jj - scol * 3;
        const int sgx = x0 - kHalo + scol;
        const float* maps_c = a.maps + (size_t)ch * plane * 3;
        if (jj < kLR * 3) {
            const bool col_ok = sgx >= 0 && sgx < a.W;
            const unsigned lane_off = 4u * (unsigned)(clampi(sgx, 0, a.W - 1) * 3 + smi);
            float v[kRowsPer];
            const int gy0 = y0 - kHalo + half;
            const char* rowp = reinterpret_cast<const char*>(maps_c) + (int64_t)gy0 * row_bytes;   // scalar, stepped by two rows
#pragma unroll
            for (int i = 0; i < kRowsPer; ++i) {
                const int gy = gy0 + 2 * i;
                v[i] = 0.f;   // derivative maps are zero outside the image (and outside the interior)
                if (gy >= 0 && gy < a.H) v[i] = ld_off(reinterpret_cast<const float*>(rowp), lane_off);
                rowp += 2 * row_bytes;
            }
            float* d = sm + (smi * kLR + half) * kLRP + scol;
#pragma unroll
            for (int i = 0; i < kRowsPer; ++i) d[2 * i * kLRP] = col_ok ? v[i] : 0.f;
        }
    }
    __syncthreads();
    {   // horizontal pass, sliding window: one row x 6 columns per thread (see the forward kernel)
        const bool h_on = tid < kLR * 6;
        const int hg = tid / kLR, row = tid - hg * kLR, c0 = min(kHOut * hg, kLT - kHOut);
        float ho[3][kHOut];
        if (h_on) {
#pragma unroll
            for (int mi = 0; mi < 3; ++mi) {
                float w[kHWin];
#pragma unroll
                for (int i = 0; i < kHWin; ++i) w[i] = sm[(mi * kLR + row) * kLRP + c0 + i];
#pragma unroll
                for (int j = 0; j < kHOut; ++j) {
                    float acc = kWin[0] * w[j];
#pragma unroll
                    for (int k = 1; k < 11; ++k) acc = fmaf(kWin[k], w[j + k], acc);
                    ho[mi][j] = acc;
                }
            }
        }
        __syncthreads();
        if (h_on) {
#pragma unroll
            for (int mi = 0; mi < 3; ++mi)
#pragma unroll
                for (int j = 0; j < kHOut; ++j) hp[mi * kLR * kHP + row * kHP + c0 + j] = ho[mi][j];
        }
    }
    __syncthreads();
    {   // vertical pass: one column x 4 rows per thread
        float c[3][4];
#pragma unroll
        for (int mi = 0; mi < 3; ++mi) {
            float hv[14];
#pragma unroll
            for (int i = 0; i < 14; ++i) hv[i] = hp[mi * kLR * kHP + (r0 + i) * kHP + ocol];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float acc = kWin[0] * hv[j];
#pragma unroll
                for (int k = 1; k < 11; ++k) acc = fmaf(kWin[k], hv[j + k], acc);
                c[mi][j] = acc;
            }
        }
        float* out_c = a.v_render + ch;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (y0 + r0 + j >= a.H || !col_in) continue;
            float r = rr[j], keep = 1.f;
            if (a.clamp_input) { keep = (r >= 0.f && r <= 1.f) ? 1.f : 0.f; r = fminf(fmaxf(r, 0.f), 1.f); }   // clamp's backward
            if (mask_img) { r = mk[j] * gtv[j] + (1.f - mk[j]) * r; keep *= 1.f - mk[j]; }
            const float d = r - gtv[j];
            const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
            st_off(out_c, o0 + j * row_bytes, keep * (k_ssim * (c[0][j] + 2.f * r * c[1][j] + gtv[j] * c[2][j]) + k_l1 * sgn));
        }
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

static int loss_tile_count(int height, int width) { return ((width + kLT - 1) / kLT) * ((height + kLT - 1) / kLT); }
static dim3 loss_grid(int nt) { return dim3((unsigned)(8 * ((nt + 7) / 8))); }   // see loss_tile()

extern "C" size_t gs_loss_workspace_floats(int height, int width) {
    const size_t nb = (size_t)((width + kLT - 1) / kLT) * ((height + kLT - 1) / kLT);
    return 9 * (size_t)height * width + 2 * (nb + 8);   // derivative maps + (l1, ssim) partial sums per forward block
}

static int loss_fwd_launch(void* stream, int height, int width, float lambda_ssim, const float* render, const float* gt,
                           const float* mask, const float* const* slots, bool has_mask, int clamp_input, float* workspace, float* out3) {
    GS_REQUIRE(height > 2 * kHalo && width > 2 * kHalo, "image must be larger than the 11x11 window");
    GS_REQUIRE((int64_t)height * width <= kLossMaxPixels && width <= kLossMaxWidth && height <= (1 << 24),
               "image too large for the loss kernels' 32-bit byte offsets (H * W <= 2^28, W <= 2^20, H <= 2^24: row indices go through 24-bit multiplies)");
    GS_REQUIRE(render && (gt || slots) && workspace && out3, "null pointer");
    LossArgs a;
    a.H = height; a.W = width; a.lambda_ssim = lambda_ssim; a.render = render; a.gt = gt; a.mask = mask; a.slots = slots;
    a.clamp_input = clamp_input != 0;
    a.maps = workspace; a.partial = workspace + 9 * (size_t)height * width; a.gout = nullptr; a.v_render = nullptr;
    const int blocks = (int)loss_grid(loss_tile_count(height, width)).x;
    const size_t lds = sizeof(float) * (2 * kLR * kLRP + 4 * kLR * kHP);
    hipStream_t st = (hipStream_t)stream;
    if (has_mask) hipLaunchKernelGGL(l1_ssim_fwd_kernel<true>, dim3((unsigned)blocks), dim3(256), lds, st, a);
    else hipLaunchKernelGGL(l1_ssim_fwd_kernel<false>, dim3((unsigned)blocks), dim3(256), lds, st, a);
    GS_LAUNCH_CHECK("l1_ssim_fwd_kernel");
    hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(256), 0, st, blocks, a.partial, height, width, lambda_ssim, out3);
    GS_LAUNCH_CHECK("loss_reduce_kernel");
    return GS_OK;
}

static int loss_bwd_launch(void* stream, int height, int width, float lambda_ssim, const float* render, const float* gt,
                           const float* mask, const float* const* slots, int clamp_input, const float* workspace,
                           const float* v_total, float* v_render) {
    GS_REQUIRE(height > 2 * kHalo && width > 2 * kHalo, "image must be larger than the 11x11 window");
    GS_REQUIRE((int64_t)height * width <= kLossMaxPixels && width <= kLossMaxWidth && height <= (1 << 24),
               "image too large for the loss kernels' 32-bit byte offsets (H * W <= 2^28, W <= 2^20, H <= 2^24: row indices go through 24-bit multiplies)");
    GS_REQUIRE(render && (gt || slots) && workspace && v_total && v_render, "null pointer");
    LossArgs a;
    a.H = height; a.W = width; a.lambda_ssim = lambda_ssim; a.render = render; a.gt = gt; a.mask = mask; a.slots = slots;
    a.clamp_input = clamp_input != 0;
    a.maps = const_cast<float*>(workspace); a.partial = nullptr; a.gout = v_total; a.v_render = v_render;
    const size_t lds = sizeof(float) * (3 * kLR * kLRP);
    hipLaunchKernelGGL(l1_ssim_bwd_kernel, dim3(3 * loss_grid(loss_tile_count(height, width)).x), dim3(256), lds, (hipStream_t)stream, a);
    GS_LAUNCH_CHECK("l1_ssim_bwd_kernel");
    return GS_OK;
}

extern "C" int gs_l1_ssim_fwd(void* stream, int height, int width, float lambda_ssim, const float* render,
                              const float* gt, const float* mask, int clamp_input, float* workspace, float* out3) {
    return loss_fwd_launch(stream, height, width, lambda_ssim, render, gt, mask, nullptr, mask != nullptr, clamp_input, workspace, out3);
}

extern "C" int gs_l1_ssim_bwd(void* stream, int height, int width, float lambda_ssim, const float* render,
                              const float* gt, const float* mask, int clamp_input, const float* workspace,
                              const float* v_total, float* v_render) {
    return loss_bwd_launch(stream, height, width, lambda_ssim, render, gt, mask, nullptr, clamp_input, workspace, v_total, v_render);
}

// The same two entries with the ground truth and the mask read THROUGH device memory: slots_dev[0] = gt, slots_dev[1] = mask
// (NULL: none), written by gs_step_inputs in front of every replay of a captured step -- the graph's kernel arguments are
// frozen at capture, the image a step trains on is not.  has_mask selects the instantiation (known at capture).
extern "C" int gs_l1_ssim_fwd_slots(void* stream, int height, int width, float lambda_ssim, const float* render,
                                    const float* const* slots_dev, int has_mask, int clamp_input, float* workspace, float* out3) {
    GS_REQUIRE(slots_dev != nullptr, "slots_dev is required");
    return loss_fwd_launch(stream, height, width, lambda_ssim, render, nullptr, nullptr, slots_dev, has_mask != 0, clamp_input, workspace, out3);
}

extern "C" int gs_l1_ssim_bwd_slots(void* stream, int height, int width, float lambda_ssim, const float* render,
                                    const float* const* slots_dev, int clamp_input, const float* workspace,
                                    const float* v_total, float* v_render) {
    GS_REQUIRE(slots_dev != nullptr, "slots_dev is required");
    return loss_bwd_launch(stream, height, width, lambda_ssim, render, nullptr, nullptr, slots_dev, clamp_input, workspace, v_total, v_render);
}
