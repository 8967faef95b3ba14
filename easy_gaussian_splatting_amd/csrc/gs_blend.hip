// gs_blend.hip -- alpha-blend forward and backward for gfx950.
//
// Forward: ONE wavefront per 16x16 tile, four pixels per lane -- lane (lx, ly) of an 8x8 grid
// owns the pixel at that position in each of the tile's four 8x8 QUADRANTS.  The tile's
// depth-sorted list is walked in buckets of 64 entries: each lane gathers one packed 48-byte
// record, tests its opacity-aware extent against the four quadrants and the wave ballots the
// results into four 64-bit scalar masks.  The inner loop then runs on scalar control flow only:
// it visits just the entries that can touch the tile at all (s_ff1 over the OR of the masks),
// reads the record back at a wave-uniform LDS address (broadcast) and evaluates only the
// quadrants whose bit is set.  The per-pixel arithmetic is straight-line predicated code
// (v_cndmask, no divergent branches).  A single wave needs no workgroup barrier and leaves the
// tile with one ballot once all 256 pixels are saturated.
//
// Backward: ONE wavefront per bucket, GAUSSIAN-parallel.  Lane l owns entry l of the bucket and
// keeps its eleven gradient sums in registers while the tile's 256 pixels stream through the
// wave as a systolic pipeline: at step t lane l treats pixel t-l, receives that pixel's running
// (transmittance T, P = prefix colour . v_colour) from lane l-1 through one DPP wave_shr:1 each
// and hands it on.  Lane 0 is fed from the per-bucket checkpoint the forward wrote (T < 0 marks
// a pixel that is already saturated or outside the image).  There are no cross-lane reductions
// and no atomics: each lane finally stores its 48-byte gradient row, and gs_project_bwd sums the
// (contiguous) rows of every Gaussian.  Buckets are independent work units of identical size,
// which also removes the heavy-tailed per-tile load imbalance of a pixel-parallel backward
// (SURVEY.md section 7 "hard parts").
//
// Forward and backward evaluate alpha, w = alpha*T and T' = T - w with the same instruction
// sequence, so the backward re-derives the forward's contributor set exactly (no last_ids).
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
    int32_t *tile_used, *bucket_tile;
    float4* ckpt;
};

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// One (pixel, Gaussian) pair of the forward; straight-line, predicated.
__device__ __forceinline__ void blend_pair(const float sigma, const float op, const float r,
                                           const float g, const float b, float& T, float& cr,
                                           float& cg, float& cb, bool& done) {
    const float alpha = fminf(kAlphaMax, op * fast_exp2(-sigma));
    const bool ok = !done && sigma >= 0.f && alpha >= kAlphaMin;
    const float w = alpha * T;
    const float Tn = fmaf(-alpha, T, T);   // explicit: fwd and bwd must round identically
    const bool stop = ok && Tn <= kTMin;
    const bool contrib = ok && !stop;
    const float wm = contrib ? w : 0.f;
    cr = fmaf(r, wm, cr); cg = fmaf(g, wm, cg); cb = fmaf(b, wm, cb);
    T = contrib ? Tn : T;
    done = done || stop;
}

template <bool CKPT>
__global__ __launch_bounds__(64) void blend_fwd_kernel(const BlendFwdArgs a) {
    __shared__ float4 srec[GS_BUCKET * 3];
    const int t = blockIdx.x;
    const int cam = t / a.tiles, tt = t - cam * a.tiles;
    const int tyi = tt / a.tw, txi = tt - tyi * a.tw;
    const int lane = threadIdx.x;
    const int x0 = txi * GS_TILE, y0 = tyi * GS_TILE;
    const int px0 = x0 + (lane & 7), py0 = y0 + (lane >> 3);   // quadrant 0 pixel; +8 for the others
    const float fx0 = (float)px0 + 0.5f, fy0 = (float)py0 + 0.5f, fx1 = fx0 + 8.f, fy1 = fy0 + 8.f;
    const int lo = a.isect_offsets[t], hi = a.isect_offsets[t + 1];
    const int nb = (hi - lo + GS_BUCKET - 1) / GS_BUCKET;
    const int bucket0 = a.bucket_offsets[t];

    float T[4], cr[4], cg[4], cb[4];
    bool done[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        T[k] = 1.f; cr[k] = cg[k] = cb[k] = 0.f;
        done[k] = !((px0 + 8 * (k & 1)) < a.W && (py0 + 8 * (k >> 1)) < a.H);
    }
    if (CKPT)
        for (int b = lane; b < nb; b += 64) a.bucket_tile[bucket0 + b] = t;
    // pixel-centre bounds of the four quadrants
    const float qxlo[2] = {(float)x0 + 0.5f, (float)x0 + 8.5f}, qxhi[2] = {(float)x0 + 7.5f, (float)x0 + 15.5f};
    const float qylo[2] = {(float)y0 + 0.5f, (float)y0 + 8.5f}, qyhi[2] = {(float)y0 + 7.5f, (float)y0 + 15.5f};

    int used = 0;
    for (int b = 0; b < nb; ++b) {
        if (__all(done[0] && done[1] && done[2] && done[3])) break;
        used = b + 1;
        if (CKPT) {
            float4* ck = a.ckpt + (size_t)(bucket0 + b) * 256;
#pragma unroll
            for (int k = 0; k < 4; ++k) ck[k * 64 + lane] = make_float4(done[k] ? -1.f : T[k], cr[k], cg[k], cb[k]);
        }
        const int first = lo + b * GS_BUCKET;
        const int m = min(GS_BUCKET, hi - first);
        bool hx[2] = {false, false}, hy[2] = {false, false};
        if (lane < m) {
            const float4* r = a.rec + 3 * (size_t)a.flatten_ids[first + lane];
            const float4 q0 = r[0], q1 = r[1], q2 = r[2];
            srec[lane * 3] = q0; srec[lane * 3 + 1] = q1; srec[lane * 3 + 2] = q2;
            const float ex = q2.y, ey = q2.z;   // negative when alpha can never reach 1/255
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                hx[h] = ex >= 0.f && q0.x + ex >= qxlo[h] && q0.x - ex <= qxhi[h];
                hy[h] = ey >= 0.f && q0.y + ey >= qylo[h] && q0.y - ey <= qyhi[h];
            }
        }
        const unsigned long long m0 = __ballot(hx[0] && hy[0]), m1 = __ballot(hx[1] && hy[0]),
                                 m2 = __ballot(hx[0] && hy[1]), m3 = __ballot(hx[1] && hy[1]);
        __syncthreads();
        for (unsigned long long rem = m0 | m1 | m2 | m3; rem; rem &= rem - 1) {
            const int j = __builtin_ctzll(rem);
            const float4 q0 = srec[j * 3], q1 = srec[j * 3 + 1], q2 = srec[j * 3 + 2];
            const float dxa = q0.x - fx0, dxb = q0.x - fx1, dya = q0.y - fy0, dyb = q0.y - fy1;
            const float sxa = q0.z * dxa * dxa, sxb = q0.z * dxb * dxb;   // hA dx^2
            const float bxa = q0.w * dxa, bxb = q0.w * dxb;               // B dx
            const float op = q1.y, r = q1.z, g = q1.w, bl = q2.x;
            if ((m0 >> j) & 1) blend_pair(fmaf(dya, fmaf(q1.x, dya, bxa), sxa), op, r, g, bl, T[0], cr[0], cg[0], cb[0], done[0]);
            if ((m1 >> j) & 1) blend_pair(fmaf(dya, fmaf(q1.x, dya, bxb), sxb), op, r, g, bl, T[1], cr[1], cg[1], cb[1], done[1]);
            if ((m2 >> j) & 1) blend_pair(fmaf(dyb, fmaf(q1.x, dyb, bxa), sxa), op, r, g, bl, T[2], cr[2], cg[2], cb[2], done[2]);
            if ((m3 >> j) & 1) blend_pair(fmaf(dyb, fmaf(q1.x, dyb, bxb), sxb), op, r, g, bl, T[3], cr[3], cg[3], cb[3], done[3]);
        }
        __syncthreads();
    }
    if (lane == 0) a.tile_used[t] = used;
    float bgr = 0.f, bgg = 0.f, bgb = 0.f;
    if (a.bg) { bgr = a.bg[3 * cam]; bgg = a.bg[3 * cam + 1]; bgb = a.bg[3 * cam + 2]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int px = px0 + 8 * (k & 1), py = py0 + 8 * (k >> 1);
        if (px < a.W && py < a.H) {
            const size_t o = ((size_t)cam * a.H + py) * a.W + px;
            a.out_colors[3 * o] = cr[k] + T[k] * bgr;
            a.out_colors[3 * o + 1] = cg[k] + T[k] * bgg;
            a.out_colors[3 * o + 2] = cb[k] + T[k] * bgb;
            a.out_alphas[o] = 1.f - T[k];
        }
    }
}

// ------------------------------------------------------------------------------------------------
struct BlendBwdArgs {
    int C, W, H, tw, tiles;
    int64_t n_buckets;
    const float4* rec;
    const int32_t *isect_offsets, *bucket_offsets, *flatten_ids, *slots, *bucket_tile, *tile_used;
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
    // per wave: 256 pixels x 2 float4 = 8 KB : (v_r, v_g, v_b, E) and (px, py, -, -)
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

    float4 ck[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int p = k * 64 + lane;
        const int px = txi * GS_TILE + (lane & 7) + 8 * (k & 1), py = tyi * GS_TILE + (lane >> 3) + 8 * (k >> 1);
        float4 d0 = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 d1 = make_float4((float)px + 0.5f, (float)py + 0.5f, 0.f, 0.f);
        if (px < a.W && py < a.H) {
            const size_t o = ((size_t)cam * a.H + py) * a.W + px;
            const float vr = a.v_colors[3 * o], vg = a.v_colors[3 * o + 1], vb = a.v_colors[3 * o + 2];
            const float Tf = 1.f - a.out_alphas[o];
            const float va = a.v_alphas ? a.v_alphas[o] : 0.f;
            // E = T_final * v_alpha - render_colour . v_colour  (background terms cancel)
            const float E = Tf * va - (a.out_colors[3 * o] * vr + a.out_colors[3 * o + 1] * vg + a.out_colors[3 * o + 2] * vb);
            d0 = make_float4(vr, vg, vb, E);
        }
        spix[p * 2] = d0; spix[p * 2 + 1] = d1;
        ck[k] = a.ckpt[(size_t)B * 256 + p];
    }
    __builtin_amdgcn_wave_barrier();

    float mx = 0.f, my = 0.f, hA = 0.f, Bc = 0.f, hC = 0.f, op = 0.f, colr = 0.f, colg = 0.f, colb = 0.f;
    if (has) {
        const float4* r = a.rec + 3 * (size_t)a.flatten_ids[idx];
        const float4 q0 = r[0], q1 = r[1], q2 = r[2];
        mx = q0.x; my = q0.y; hA = q0.z; Bc = q0.w; hC = q1.x; op = q1.y; colr = q1.z; colg = q1.w; colb = q2.x;
    }
    // true conic entries for the mean gradient (the record stores them scaled by log2(e))
    const float At = 2.f * kLn2 * hA, Bt = kLn2 * Bc, Ct = 2.f * kLn2 * hC;
    float s_mx = 0.f, s_my = 0.f, s_ax = 0.f, s_ay = 0.f, s_A = 0.f, s_B = 0.f, s_C = 0.f, s_vs = 0.f,
          s_r = 0.f, s_g = 0.f, s_b = 0.f;
    float T_out = -1.f, P_out = 0.f;

    auto step = [&](const int tstep, const bool inject, const float4 ckv, const int src_lane) {
        float T_in = dpp_wave_shr1(T_out), P_in = dpp_wave_shr1(P_out);
        const int p = tstep - lane;
        const bool act = has && (unsigned)p < 256u;
        const int pc = min(max(p, 0), 255);
        const float4 d0 = spix[pc * 2], d1 = spix[pc * 2 + 1];
        if (inject) {
            const float cT = readlane_f(ckv.x, src_lane), cR = readlane_f(ckv.y, src_lane),
                        cG = readlane_f(ckv.z, src_lane), cB = readlane_f(ckv.w, src_lane);
            const float P0 = cR * d0.x + cG * d0.y + cB * d0.z;
            T_in = lane == 0 ? cT : T_in;
            P_in = lane == 0 ? P0 : P_in;
        }
        const float dx = mx - d1.x, dy = my - d1.y;
        const float sigma = fmaf(dy, fmaf(hC, dy, Bc * dx), hA * dx * dx);   // same op sequence as the forward
        const float vis = fast_exp2(-sigma);
        const float ov = op * vis;
        const float alpha = fminf(kAlphaMax, ov);
        const bool ok = act && T_in > 0.f && sigma >= 0.f && alpha >= kAlphaMin;
        const float w = alpha * T_in;
        const float Tn = fmaf(-alpha, T_in, T_in);   // identical to the forward's update
        const bool stop = ok && Tn <= kTMin;
        const bool contrib = ok && !stop;
        const float cv = colr * d0.x + colg * d0.y + colb * d0.z;
        const float Pn = fmaf(w, cv, P_in);
        const float ra = fast_rcp(1.f - alpha);
        const float v_alpha = fmaf(T_in, cv, ra * (d0.w + Pn));
        const float wm = contrib ? w : 0.f;
        s_r = fmaf(wm, d0.x, s_r); s_g = fmaf(wm, d0.y, s_g); s_b = fmaf(wm, d0.z, s_b);
        const float vs = (contrib && ov <= kAlphaMax) ? -ov * v_alpha : 0.f;   // d loss / d sigma
        const float hx = vs * dx, hy = vs * dy;
        s_A = fmaf(hx, dx, s_A); s_B = fmaf(hx, dy, s_B); s_C = fmaf(hy, dy, s_C);
        const float gx = fmaf(At, hx, Bt * hy), gy = fmaf(Bt, hx, Ct * hy);
        s_mx += gx; s_my += gy; s_ax += fabsf(gx); s_ay += fabsf(gy);
        s_vs += vs;
        T_out = contrib ? Tn : (stop ? -1.f : T_in);
        P_out = contrib ? Pn : P_in;
    };
#pragma unroll
    for (int seg = 0; seg < 4; ++seg)
        for (int tl = 0; tl < 64; ++tl) step(seg * 64 + tl, true, ck[seg], tl);
    for (int tl = 0; tl < 63; ++tl) step(256 + tl, false, ck[0], 0);

    if (has) {
        float4* r = a.rows + 3 * (size_t)slot;
        // v_opacity = sum vis * v_alpha = -sum(vs) / opacity   (vs = -opacity*vis*v_alpha)
        const float v_op = op > 0.f ? -s_vs / op : 0.f;
        r[0] = make_float4(s_mx, s_my, s_ax, s_ay);
        r[1] = make_float4(0.5f * s_A, s_B, 0.5f * s_C, v_op);
        r[2] = make_float4(s_r, s_g, s_b, 0.f);
    }
}

}  // namespace gs

using namespace gs;

extern "C" int gs_blend_fwd(void* stream, int C, int width, int height, const float* rec,
                            const float* backgrounds, const int32_t* isect_offsets,
                            const int32_t* bucket_offsets, const int32_t* flatten_ids,
                            float* render_colors, float* render_alphas, int32_t* tile_used,
                            float* ckpt, int32_t* bucket_tile) {
    GS_REQUIRE(C >= 1 && width > 0 && height > 0, "C>=1 and positive image size");
    GS_REQUIRE(isect_offsets && bucket_offsets && render_colors && render_alphas && tile_used, "null pointer");
    GS_REQUIRE((ckpt == nullptr) == (bucket_tile == nullptr), "ckpt and bucket_tile go together");
    BlendFwdArgs a;
    a.C = C; a.W = width; a.H = height;
    a.tw = (width + GS_TILE - 1) / GS_TILE;
    a.tiles = a.tw * ((height + GS_TILE - 1) / GS_TILE);
    a.rec = reinterpret_cast<const float4*>(rec); a.bg = backgrounds; a.isect_offsets = isect_offsets;
    a.bucket_offsets = bucket_offsets; a.flatten_ids = flatten_ids; a.out_colors = render_colors;
    a.out_alphas = render_alphas; a.tile_used = tile_used;
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
                            const float* v_render_colors, const float* v_render_alphas, float* rows) {
    GS_REQUIRE(C >= 1 && width > 0 && height > 0 && n_buckets >= 0, "C>=1, positive image size, n_buckets>=0");
    if (n_buckets == 0) return GS_OK;
    GS_REQUIRE(rec && isect_offsets && bucket_offsets && flatten_ids && slots && bucket_tile && tile_used && ckpt, "null list pointer");
    GS_REQUIRE(render_colors && render_alphas && v_render_colors && rows, "null image pointer");
    BlendBwdArgs a;
    a.C = C; a.W = width; a.H = height;
    a.tw = (width + GS_TILE - 1) / GS_TILE;
    a.tiles = a.tw * ((height + GS_TILE - 1) / GS_TILE);
    a.n_buckets = n_buckets; a.rec = reinterpret_cast<const float4*>(rec);
    a.isect_offsets = isect_offsets; a.bucket_offsets = bucket_offsets; a.flatten_ids = flatten_ids;
    a.slots = slots; a.bucket_tile = bucket_tile; a.tile_used = tile_used;
    a.ckpt = reinterpret_cast<const float4*>(ckpt); a.out_colors = render_colors;
    a.out_alphas = render_alphas; a.v_colors = v_render_colors; a.v_alphas = v_render_alphas;
    a.rows = reinterpret_cast<float4*>(rows);
    const unsigned grid = (unsigned)((n_buckets + kBwdWaves - 1) / kBwdWaves);
    hipLaunchKernelGGL(blend_bwd_kernel, dim3(grid), dim3(kBwdWaves * 64), 0, (hipStream_t)stream, a);
    GS_LAUNCH_CHECK("blend_bwd_kernel");
    return GS_OK;
}
