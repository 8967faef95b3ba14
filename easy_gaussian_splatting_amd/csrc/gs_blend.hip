// gs_blend.hip -- alpha-blend forward and backward for gfx950.
//
// Forward: ONE wavefront per 16x16 tile, four pixels per lane (same column, rows y, y+4, y+8,
// y+12).  The tile's depth-sorted list is walked in buckets of 64 entries: each lane gathers one
// packed 48-byte record into LDS, then all lanes read the records back at a wave-uniform address
// (LDS broadcast).  Four pixels per lane amortise every broadcast read over 4x the VALU work and
// share the column terms of the conic form; a single wave needs no workgroup barrier and leaves
// the tile with one ballot when all 256 pixels are saturated.
//
// Backward: ONE wavefront per bucket, GAUSSIAN-parallel.  Lane l owns entry l of the bucket and
// keeps its eleven gradient sums in registers while the tile's 256 pixels stream through the
// wave as a systolic pipeline: at step t lane l treats pixel t-l, receives that pixel's running
// (transmittance T, P = prefix colour . v_colour) from lane l-1 through one DPP wave_shr:1 each
// and hands it on.  Lane 0 is fed from the per-bucket checkpoint the forward wrote.  There are no
// cross-lane reductions and no atomics: each lane finally stores its 48-byte gradient row, and
// gs_project_bwd sums the (contiguous) rows of every Gaussian.  Buckets are independent work
// units of identical size, which also removes the heavy-tailed per-tile load imbalance of the
// pixel-parallel backward (SURVEY.md section 7 "hard parts").
//
// Semantics: SURVEY.md Appendix A.4 / A.5 (gsplat 1.0.0 rasterize_to_pixels fwd/bwd).
#include "gs_common.h"
#include "gs_math.h"

namespace gs {

struct BlendFwdArgs {
    int C, W, H, tw, tiles;
    const float4* rec;
    const float* bg;
    const int32_t *isect_offsets, *bucket_offsets, *flatten_ids;
    float *out_colors, *out_alphas;
    int32_t *last_ids, *tile_used, *bucket_tile;
    float4* ckpt;
};

__device__ __forceinline__ float fast_exp(float x) { return __expf(x); }

template <bool CKPT>
__global__ __launch_bounds__(64) void blend_fwd_kernel(const BlendFwdArgs a) {
    __shared__ float4 srec[GS_BUCKET * 3];
    const int t = blockIdx.x;
    const int cam = t / a.tiles, tt = t - cam * a.tiles;
    const int tyi = tt / a.tw, txi = tt - tyi * a.tw;
    const int lane = threadIdx.x;
    const int px = txi * GS_TILE + (lane & 15);
    const int py0 = tyi * GS_TILE + (lane >> 4);
    const float fx = (float)px + 0.5f, fy0 = (float)py0 + 0.5f;
    const int lo = a.isect_offsets[t], hi = a.isect_offsets[t + 1];
    const int nb = (hi - lo + GS_BUCKET - 1) / GS_BUCKET;
    const int bucket0 = a.bucket_offsets[t];

    float T[4], cr[4], cg[4], cb[4];
    int last[4];
    bool done[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        T[k] = 1.f; cr[k] = cg[k] = cb[k] = 0.f; last[k] = -1;
        done[k] = !(px < a.W && (py0 + 4 * k) < a.H);
    }
    if (CKPT)
        for (int b = lane; b < nb; b += 64) a.bucket_tile[bucket0 + b] = t;

    int used = 0;
    for (int b = 0; b < nb; ++b) {
        if (__all(done[0] && done[1] && done[2] && done[3])) break;
        used = b + 1;
        if (CKPT) {
            float4* ck = a.ckpt + (size_t)(bucket0 + b) * 256;
#pragma unroll
            for (int k = 0; k < 4; ++k) ck[k * 64 + lane] = make_float4(T[k], cr[k], cg[k], cb[k]);
        }
        const int first = lo + b * GS_BUCKET;
        const int m = min(GS_BUCKET, hi - first);
        if (lane < m) {
            const float4* r = a.rec + 3 * (size_t)a.flatten_ids[first + lane];
            srec[lane * 3] = r[0]; srec[lane * 3 + 1] = r[1]; srec[lane * 3 + 2] = r[2];
        }
        __syncthreads();
        for (int j = 0; j < m; ++j) {
            const float4 q0 = srec[j * 3], q1 = srec[j * 3 + 1], q2 = srec[j * 3 + 2];
            const float dx = q0.x - fx;
            const float hA = 0.5f * q0.z * dx * dx, Bdx = q0.w * dx, hC = 0.5f * q1.x;
            const float dy0 = q0.y - fy0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (done[k]) continue;
                const float dy = dy0 - 4.f * k;
                const float sigma = hA + dy * (Bdx + hC * dy);
                if (sigma < 0.f) continue;
                const float alpha = fminf(kAlphaMax, q1.y * fast_exp(-sigma));
                if (alpha < kAlphaMin) continue;
                const float Tn = T[k] * (1.f - alpha);
                if (Tn <= kTMin) { done[k] = true; continue; }
                const float w = alpha * T[k];
                cr[k] += q1.z * w; cg[k] += q1.w * w; cb[k] += q2.x * w;
                T[k] = Tn;
                last[k] = first + j;
            }
        }
        __syncthreads();
    }
    if (lane == 0) a.tile_used[t] = used;
    float bgr = 0.f, bgg = 0.f, bgb = 0.f;
    if (a.bg) { bgr = a.bg[3 * cam]; bgg = a.bg[3 * cam + 1]; bgb = a.bg[3 * cam + 2]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int py = py0 + 4 * k;
        if (px < a.W && py < a.H) {
            const size_t o = ((size_t)cam * a.H + py) * a.W + px;
            a.out_colors[3 * o] = cr[k] + T[k] * bgr;
            a.out_colors[3 * o + 1] = cg[k] + T[k] * bgg;
            a.out_colors[3 * o + 2] = cb[k] + T[k] * bgb;
            a.out_alphas[o] = 1.f - T[k];
            a.last_ids[o] = last[k];
        }
    }
}

// ------------------------------------------------------------------------------------------------
struct BlendBwdArgs {
    int C, W, H, tw, tiles;
    int64_t n_buckets;
    const float4* rec;
    const int32_t *isect_offsets, *bucket_offsets, *flatten_ids, *slots, *bucket_tile, *tile_used,
        *last_ids;
    const float4* ckpt;
    const float *out_colors, *out_alphas, *v_colors, *v_alphas;
    float4* rows;
};

constexpr int kBwdWaves = 4;

__device__ __forceinline__ float dpp_wave_shr1(float v) {
    // lane l receives lane l-1's value; lane 0 keeps its own (overwritten by the caller)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float readlane_f(float v, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

__global__ __launch_bounds__(kBwdWaves * 64) void blend_bwd_kernel(const BlendBwdArgs a) {
    // per wave: 256 pixels x 2 float4 = 8 KB
    __shared__ float4 spix_all[kBwdWaves][256 * 2];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t B = (int64_t)blockIdx.x * kBwdWaves + wave;
    if (B >= a.n_buckets) return;  // wave-uniform; no workgroup barriers below
    float4* spix = spix_all[wave];

    const int t = a.bucket_tile[B];
    const int b = (int)(B - a.bucket_offsets[t]);
    const int lo = a.isect_offsets[t] + b * GS_BUCKET, hi = a.isect_offsets[t + 1];
    const int m = min(GS_BUCKET, hi - lo);
    const bool has = lane < m;
    const int idx = lo + lane;
    const int slot = has ? a.slots[idx] : 0;
    if (b >= a.tile_used[t]) {  // the forward never reached this bucket: all-zero rows
        if (has) {
            float4* r = a.rows + 3 * (size_t)slot;
            r[0] = r[1] = r[2] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        return;
    }
    const int cam = t / a.tiles, tt = t - cam * a.tiles;
    const int tyi = tt / a.tw, txi = tt - tyi * a.tw;

    // stage the tile's per-pixel constants: (v_r, v_g, v_b, E) and (px, py, last, -)
    float4 ck[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int p = k * 64 + lane;
        const int px = txi * GS_TILE + (lane & 15), py = tyi * GS_TILE + (lane >> 4) + 4 * k;
        float4 d0 = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 d1 = make_float4((float)px + 0.5f, (float)py + 0.5f, __int_as_float(-1), 0.f);
        if (px < a.W && py < a.H) {
            const size_t o = ((size_t)cam * a.H + py) * a.W + px;
            const float vr = a.v_colors[3 * o], vg = a.v_colors[3 * o + 1], vb = a.v_colors[3 * o + 2];
            const float Tf = 1.f - a.out_alphas[o];
            const float va = a.v_alphas ? a.v_alphas[o] : 0.f;
            const float E = Tf * va - (a.out_colors[3 * o] * vr + a.out_colors[3 * o + 1] * vg + a.out_colors[3 * o + 2] * vb);
            d0 = make_float4(vr, vg, vb, E);
            d1.z = __int_as_float(a.last_ids[o]);
        }
        spix[p * 2] = d0; spix[p * 2 + 1] = d1;
        ck[k] = a.ckpt[(size_t)B * 256 + p];
    }
    __builtin_amdgcn_wave_barrier();

    float mx = 0.f, my = 0.f, A = 0.f, Bc = 0.f, Cc = 0.f, op = 0.f, colr = 0.f, colg = 0.f, colb = 0.f;
    if (has) {
        const float4* r = a.rec + 3 * (size_t)a.flatten_ids[idx];
        const float4 q0 = r[0], q1 = r[1], q2 = r[2];
        mx = q0.x; my = q0.y; A = q0.z; Bc = q0.w; Cc = q1.x; op = q1.y; colr = q1.z; colg = q1.w; colb = q2.x;
    }
    float s_mx = 0.f, s_my = 0.f, s_ax = 0.f, s_ay = 0.f, s_A = 0.f, s_B = 0.f, s_C = 0.f, s_op = 0.f,
          s_r = 0.f, s_g = 0.f, s_b = 0.f;
    float T_out = 0.f, P_out = 0.f;

    auto step = [&](const int tstep, const bool inject, const float4 ckv, const int src_lane) {
        float T_in = dpp_wave_shr1(T_out), P_in = dpp_wave_shr1(P_out);
        const int p = tstep - lane;
        const bool act = has && p >= 0 && p < 256;
        const int pc = min(max(p, 0), 255);
        const float4 d0 = spix[pc * 2], d1 = spix[pc * 2 + 1];
        if (inject) {
            const float cT = readlane_f(ckv.x, src_lane), cR = readlane_f(ckv.y, src_lane),
                        cG = readlane_f(ckv.z, src_lane), cB = readlane_f(ckv.w, src_lane);
            if (lane == 0) { T_in = cT; P_in = cR * d0.x + cG * d0.y + cB * d0.z; }
        }
        T_out = T_in; P_out = P_in;
        if (act && idx <= __float_as_int(d1.z)) {
            const float dx = mx - d1.x, dy = my - d1.y;
            const float sigma = 0.5f * (A * dx * dx + Cc * dy * dy) + Bc * dx * dy;
            if (sigma >= 0.f) {
                const float vis = fast_exp(-sigma);
                const float ov = op * vis;
                const float alpha = fminf(kAlphaMax, ov);
                if (alpha >= kAlphaMin) {
                    const float fac = alpha * T_in;
                    const float cv = colr * d0.x + colg * d0.y + colb * d0.z;
                    const float Pn = P_in + fac * cv;
                    const float ra = 1.f / (1.f - alpha);
                    const float v_alpha = T_in * cv + ra * (d0.w + Pn);
                    s_r += fac * d0.x; s_g += fac * d0.y; s_b += fac * d0.z;
                    if (ov <= kAlphaMax) {
                        const float v_sigma = -ov * v_alpha;
                        s_A += 0.5f * v_sigma * dx * dx; s_B += v_sigma * dx * dy; s_C += 0.5f * v_sigma * dy * dy;
                        const float gx = v_sigma * (A * dx + Bc * dy), gy = v_sigma * (Bc * dx + Cc * dy);
                        s_mx += gx; s_my += gy; s_ax += fabsf(gx); s_ay += fabsf(gy);
                        s_op += vis * v_alpha;
                    }
                    T_out = T_in * (1.f - alpha);
                    P_out = Pn;
                }
            }
        }
    };
#pragma unroll
    for (int seg = 0; seg < 4; ++seg)
        for (int tl = 0; tl < 64; ++tl) step(seg * 64 + tl, true, ck[seg], tl);
    for (int tl = 0; tl < 63; ++tl) step(256 + tl, false, ck[0], 0);

    if (has) {
        float4* r = a.rows + 3 * (size_t)slot;
        r[0] = make_float4(s_mx, s_my, s_ax, s_ay);
        r[1] = make_float4(s_A, s_B, s_C, s_op);
        r[2] = make_float4(s_r, s_g, s_b, 0.f);
    }
}

}  // namespace gs

using namespace gs;

extern "C" int gs_blend_fwd(void* stream, int C, int width, int height, const float* rec,
                            const float* backgrounds, const int32_t* isect_offsets,
                            const int32_t* bucket_offsets, const int32_t* flatten_ids,
                            float* render_colors, float* render_alphas, int32_t* last_ids,
                            int32_t* tile_used, float* ckpt, int32_t* bucket_tile) {
    GS_REQUIRE(C >= 1 && width > 0 && height > 0, "C>=1 and positive image size");
    GS_REQUIRE(isect_offsets && bucket_offsets && render_colors && render_alphas && last_ids && tile_used, "null pointer");
    GS_REQUIRE((ckpt == nullptr) == (bucket_tile == nullptr), "ckpt and bucket_tile go together");
    BlendFwdArgs a;
    a.C = C; a.W = width; a.H = height;
    a.tw = (width + GS_TILE - 1) / GS_TILE;
    a.tiles = a.tw * ((height + GS_TILE - 1) / GS_TILE);
    a.rec = reinterpret_cast<const float4*>(rec); a.bg = backgrounds; a.isect_offsets = isect_offsets;
    a.bucket_offsets = bucket_offsets; a.flatten_ids = flatten_ids; a.out_colors = render_colors;
    a.out_alphas = render_alphas; a.last_ids = last_ids; a.tile_used = tile_used;
    a.bucket_tile = bucket_tile; a.ckpt = reinterpret_cast<float4*>(ckpt);
    const unsigned grid = (unsigned)(C * a.tiles);
    hipStream_t st = (hipStream_t)stream;
    if (ckpt) hipLaunchKernelGGL(blend_fwd_kernel<true>, dim3(grid), dim3(64), 0, st, a);
    else hipLaunchKernelGGL(blend_fwd_kernel<false>, dim3(grid), dim3(64), 0, st, a);
    GS_LAUNCH_CHECK("blend_fwd_kernel");
    return GS_OK;
}

extern "C" int gs_blend_bwd(void* stream, int C, int width, int height, const float* rec,
                            const int32_t* isect_offsets, const int32_t* bucket_offsets,
                            const int32_t* flatten_ids, const int32_t* slots, int64_t n_buckets,
                            const int32_t* bucket_tile, const int32_t* tile_used, const float* ckpt,
                            const float* render_colors, const float* render_alphas,
                            const int32_t* last_ids, const float* v_render_colors,
                            const float* v_render_alphas, float* rows) {
    GS_REQUIRE(C >= 1 && width > 0 && height > 0 && n_buckets >= 0, "C>=1, positive image size, n_buckets>=0");
    if (n_buckets == 0) return GS_OK;
    GS_REQUIRE(rec && isect_offsets && bucket_offsets && flatten_ids && slots && bucket_tile && tile_used && ckpt, "null list pointer");
    GS_REQUIRE(render_colors && render_alphas && last_ids && v_render_colors && rows, "null image pointer");
    BlendBwdArgs a;
    a.C = C; a.W = width; a.H = height;
    a.tw = (width + GS_TILE - 1) / GS_TILE;
    a.tiles = a.tw * ((height + GS_TILE - 1) / GS_TILE);
    a.n_buckets = n_buckets; a.rec = reinterpret_cast<const float4*>(rec);
    a.isect_offsets = isect_offsets; a.bucket_offsets = bucket_offsets; a.flatten_ids = flatten_ids;
    a.slots = slots; a.bucket_tile = bucket_tile; a.tile_used = tile_used; a.last_ids = last_ids;
    a.ckpt = reinterpret_cast<const float4*>(ckpt); a.out_colors = render_colors;
    a.out_alphas = render_alphas; a.v_colors = v_render_colors; a.v_alphas = v_render_alphas;
    a.rows = reinterpret_cast<float4*>(rows);
    const unsigned grid = (unsigned)((n_buckets + kBwdWaves - 1) / kBwdWaves);
    hipLaunchKernelGGL(blend_bwd_kernel, dim3(grid), dim3(kBwdWaves * 64), 0, (hipStream_t)stream, a);
    GS_LAUNCH_CHECK("blend_bwd_kernel");
    return GS_OK;
}
