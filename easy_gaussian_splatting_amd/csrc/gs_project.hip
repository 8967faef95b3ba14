// gs_project.hip -- per-Gaussian stages for gfx950:
//   project_fwd_kernel : P-fwd + SH-fwd fused (one pass over means/quats/scales/shs); when training with SH colours it
//                        also leaves d colour / d view direction (sh_jac), so that the backward reads no coefficient
//   project_bwd_kernel : gradient-row reduction (compact rows, a row per lane: row_sum_wave) + SH-bwd + P-bwd fused, no atomics;
//                        <DEG, true>: Adam and update_statistics applied in the same pass (gs_project_bwd_adam)
// The [N,K,3] SH block (192 B per Gaussian at SH3) is moved through LDS with coalesced 16-byte accesses and read per
// thread at an odd row stride (3K+1 dwords) so the per-thread walk over its own row is bank-conflict free.  The
// geometry half of the forward is VALU-bound (the fp64 chain), the rest HBM streams and gathers.
#include "gs_common.h"
#include "gs_math.h"

namespace gs {

#define GS_PROJ_THREADS 256
constexpr int kProjThreads = GS_PROJ_THREADS;

struct ProjFwdArgs {
    int C, K, colors_per_camera, W, H, tw, th, tight, activations;
    int64_t N;
    float eps2d, near_p, far_p, radius_clip;
    const float *means, *quats, *scales, *opacities, *colors_in, *sh_rest, *viewmats, *Ks;
    int32_t* radii;
    float *means2d, *depths, *conics, *colors_out;
    float4* rec;
    uint4* bbox;
    int32_t* tiles_per_gauss;
    uint2* rect_ref;   // optional [C*N]: the 3-sigma tile rectangle itself (x0 | x1 << 16, y0 | y1 << 16), whatever `tight` does to bbox
    float4* sh_jac;    // optional, 9 floats per (camera, Gaussian): d(pre-clamp colour)/d(view direction) of the visible Gaussians, for a backward without the coefficients
};

// activations != 0: `scales` / `opacities` hold the reference model's parameters (log-scales, logit
// opacities, /root/reference/model/gaussian.py:98-103) and exp / sigmoid are applied here, so that the
// model's four activation kernels (two forward, two backward) disappear.
__device__ __forceinline__ float act_scale(float v, int activations) { return activations ? expf(v) : v; }
__device__ __forceinline__ float act_opacity(float v, int activations) { return activations ? 1.0f / (1.0f + expf(-v)) : v; }

// dynamic LDS carve (dwords): [0,32) camera | [32, 32+256) visibility | SH tile 256*(3K+1)
__device__ __forceinline__ void load_camera(const float* viewmats, const float* Ks, int c, int W,
                                            int H, float* lds_cam, Camera& cam) {
    if (threadIdx.x == 0) {
        Camera tmp;
        make_camera(viewmats + 16 * c, Ks + 9 * c, W, H, tmp);
        const float* src = reinterpret_cast<const float*>(&tmp);
#pragma unroll
        for (int i = 0; i < (int)(sizeof(Camera) / 4); ++i) lds_cam[i] = src[i];
    }
    __syncthreads();
    float* dst = reinterpret_cast<float*>(&cam);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(Camera) / 4); ++i) dst[i] = lds_cam[i];
}

// The same in two steps, for a kernel whose first phase does not need the camera: thread 0 leaves it in LDS (+ block barrier),
// every thread reads it when it gets there -- the ~30 camera registers are then not live across that phase (project_bwd_kernel's
// row sum is its register peak).
__device__ __forceinline__ void stage_camera(const float* viewmats, const float* Ks, int c, int W, int H, float* lds_cam) {
    if (threadIdx.x == 0) {
        Camera tmp;
        make_camera(viewmats + 16 * c, Ks + 9 * c, W, H, tmp);
        const float* src = reinterpret_cast<const float*>(&tmp);
#pragma unroll
        for (int i = 0; i < (int)(sizeof(Camera) / 4); ++i) lds_cam[i] = src[i];
    }
    __syncthreads();
}
__device__ __forceinline__ void read_camera(const float* lds_cam, Camera& cam) {
    // (member by member: through a float* alias of the struct one instantiation kept the whole Camera as a 100-byte stack object)
    static_assert(sizeof(Camera) == 21 * sizeof(float), "Camera layout");
#pragma unroll
    for (int i = 0; i < 9; ++i) cam.R[i] = lds_cam[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) { cam.t[i] = lds_cam[9 + i]; cam.pos[i] = lds_cam[12 + i]; }
    cam.fx = lds_cam[15]; cam.fy = lds_cam[16]; cam.cx = lds_cam[17]; cam.cy = lds_cam[18];
    cam.half_w = lds_cam[19]; cam.half_h = lds_cam[20];
}

// Cooperative global -> LDS copy of the first `ka3` floats of every visible Gaussian's SH row.
// KC > 0 fixes K at compile time (K = 16 is the reference's SH3 layout) so the per-element
// row/offset divisions become multiply-shifts; KC = 0 keeps K a run-time value.
template <int KC>
__device__ __forceinline__ void stage_sh_rows(const float* __restrict__ shs, int64_t n0, int rows,
                                              int Krt, int ka3, const int* vis, float* tile) {
    const int K = KC > 0 ? KC : Krt;
    const int row_f = 3 * K, stride = row_f + 1;
    const float* src = shs + n0 * row_f;
    if (ka3 == row_f && (row_f & 3) == 0) {
        // All of a thread's (<= 12) 16-byte loads are issued before the first LDS write: a single
        // loop "load, then store to LDS" makes hipcc wait for each load before the next is issued
        // (the LDS store may alias the visibility flags it reads), i.e. 12 serial HBM round trips.
        const int per_row = row_f >> 2, total4 = rows * per_row;
        const float4* src4 = reinterpret_cast<const float4*>(src);
        constexpr int kMaxQ = 12;   // K <= 16 -> at most 48 floats = 12 quads per Gaussian
        float4 v[kMaxQ];
        bool ld[kMaxQ];
#pragma unroll
        for (int q = 0; q < kMaxQ; ++q) {
            const int e = threadIdx.x + q * kProjThreads;
            ld[q] = e < total4 && vis[e / per_row];
            if (ld[q]) v[q] = src4[e];
        }
#pragma unroll
        for (int q = 0; q < kMaxQ; ++q) {
            if (ld[q]) {
                const int e = threadIdx.x + q * kProjThreads;
                const int g = e / per_row, o = e - g * per_row;
                float* d = tile + g * stride + 4 * o;
                d[0] = v[q].x; d[1] = v[q].y; d[2] = v[q].z; d[3] = v[q].w;
            }
        }
    } else {
        for (int e = threadIdx.x; e < rows * ka3; e += blockDim.x) {
            const int g = e / ka3, o = e - g * ka3;
            if (vis[g]) tile[g * stride + o] = src[(int64_t)g * row_f + o];
        }
    }
}

// Split parameter layout of the reference model (sh_0[N,1,3] and sh_rest[N,K-1,3] are separate
// nn.Parameters, /root/reference/model/gaussian.py:49-50, concatenated on every forward at :105-107):
// staging straight from the two tensors removes that cat (and the split of its gradient).
template <int KC>
__device__ __forceinline__ void stage_sh_rows_split(const float* __restrict__ sh0, const float* __restrict__ shr,
                                                    int64_t n0, int rows, int Krt, int ka3, const int* vis, float* tile) {
    const int K = KC > 0 ? KC : Krt;
    const int row_f = 3 * K, stride = row_f + 1, rest_f = row_f - 3, kr = ka3 - 3;
    for (int e = threadIdx.x; e < rows * 3; e += blockDim.x) {
        const int g = e / 3;
        if (vis[g]) tile[g * stride + (e - 3 * g)] = sh0[n0 * 3 + e];
    }
    if (kr <= 0) return;
    const float* src = shr + n0 * rest_f;
    const int total = rows * rest_f;
    if (kr == rest_f) {  // whole rows needed: one contiguous, 16-byte aligned stream for the block
        const float4* src4 = reinterpret_cast<const float4*>(src);
        constexpr int kMaxQ = 12;   // <= 45 floats per Gaussian -> at most 12 quads per thread
        const int total4 = total >> 2;
        float4 v[kMaxQ];
        bool ld[kMaxQ];
#pragma unroll
        for (int q = 0; q < kMaxQ; ++q) {   // issue every load first (see stage_sh_rows)
            const int e4 = threadIdx.x + q * kProjThreads;
            ld[q] = e4 < total4;   // culled rows are loaded too: 12 % more bytes, but a dense stream (-8 % on the kernel)
            if (ld[q]) v[q] = nt_load4(src4 + e4);
        }
#pragma unroll
        for (int q = 0; q < kMaxQ; ++q) {
            if (ld[q]) {
                const int e = (threadIdx.x + q * kProjThreads) << 2;
                const float vv[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int g = (e + i) / rest_f, o = (e + i) - g * rest_f;
                    if (vis[g]) tile[g * stride + 3 + o] = vv[i];
                }
            }
        }
        for (int e = (total & ~3) + threadIdx.x; e < total; e += blockDim.x) {
            const int g = e / rest_f, o = e - g * rest_f;
            if (vis[g]) tile[g * stride + 3 + o] = src[e];
        }
    } else {
        for (int e = threadIdx.x; e < rows * kr; e += blockDim.x) {
            const int g = e / kr, o = e - g * kr;
            if (vis[g]) tile[g * stride + 3 + o] = src[(int64_t)g * rest_f + o];
        }
    }
}

// The same staging in two halves, for the one-launch forward (STAGE 0) with the reference's full SH3 rows in the split
// layout: `sh_rows_issue` requests a block's whole slice (every thread its 12 quads of sh_rest and <= 3 floats of sh_0) at
// the top of the kernel, `sh_rows_commit` moves it into the LDS tile once the visibility flags exist -- the 192 bytes
// per Gaussian then arrive while the fp64 projection chain runs instead of after it (the block used to stream them in
// a phase of its own, between two barriers, with nothing else to do).
struct ShPrefetch {
    float4 v[12];
    float s0[3];
};

__device__ __forceinline__ void sh_rows_issue(const float* __restrict__ sh0, const float* __restrict__ shr, int64_t n0,
                                              int rows, ShPrefetch& pf) {
    constexpr int rest_f = 45;
    const float4* src4 = reinterpret_cast<const float4*>(shr + n0 * rest_f);
    const int total4 = (rows * rest_f) >> 2;
#pragma unroll
    for (int q = 0; q < 12; ++q) {
        const int e4 = threadIdx.x + q * kProjThreads;
        pf.v[q] = e4 < total4 ? nt_load4(src4 + e4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int e = threadIdx.x + q * kProjThreads;
        pf.s0[q] = e < rows * 3 ? sh0[n0 * 3 + e] : 0.f;
    }
}

__device__ __forceinline__ void sh_rows_commit(const ShPrefetch& pf, const float* __restrict__ shr, int64_t n0, int rows,
                                               float* tile) {
    // Every row is written, visible or not (the loads were unconditional; a test per element was an LDS read of the flag, a
    // compare and an exec-mask round trip in front of each of the 51 LDS writes).  Element e of the sh_rest stream lies in row
    // g = e / 45 at tile[49 g + 3 + (e - 45 g)] = tile[e + 4 g + 3]: one division per quad, the row stepped where a quad
    // straddles two rows.
    constexpr int rest_f = 45, stride = 49;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int e = threadIdx.x + q * kProjThreads;
        if (e < rows * 3) { const int g = e / 3; tile[g * stride + (e - 3 * g)] = pf.s0[q]; }
    }
    const int total = rows * rest_f, total4 = total >> 2;
#pragma unroll
    for (int q = 0; q < 12; ++q) {
        const int e4 = threadIdx.x + q * kProjThreads;
        if (e4 < total4) {
            const float vv[4] = {pf.v[q].x, pf.v[q].y, pf.v[q].z, pf.v[q].w};
            const int e = e4 << 2, g = e / rest_f, o = e - g * rest_f;
            float* d = tile + e + 4 * g + 3;
#pragma unroll
            for (int i = 0; i < 4; ++i) d[i + (o + i >= rest_f ? 4 : 0)] = vv[i];
        }
    }
    const float* src = shr + n0 * rest_f;
    for (int e = (total & ~3) + threadIdx.x; e < total; e += blockDim.x) {   // (the last block's 0-3 trailing floats)
        const int g = e / rest_f;
        tile[e + 4 * g + 3] = src[e];
    }
}

// STAGE 0: geometry + colour in one launch.  STAGE 1: geometry only (radii, means2d, depths,
// conics, footprint, record quads 0-1).  STAGE 2: colour only (SH evaluation -> colors_out and
// record quad 2) for the Gaussians stage 1 found visible.  The split lets the host enqueue the
// colour pass BEHIND the tile-count kernels: the one host read-back of the path (I, to size the
// lists) then overlaps ~0.1 ms of SH streaming instead of leaving the GPU idle.
template <int DEG, int STAGE>
__global__ __launch_bounds__(kProjThreads) void project_fwd_kernel(const ProjFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* lds_cam = smem;
    int* vis_s = reinterpret_cast<int*>(smem + 32);
    float* tile = smem + 32 + kProjThreads;

    const int c = blockIdx.y;
    const int64_t n0 = (int64_t)blockIdx.x * kProjThreads;
    const int64_t n = n0 + threadIdx.x;
    const int64_t f = (int64_t)c * a.N + n;
    const bool in_range = n < a.N;
    Camera cam;
    load_camera(a.viewmats, a.Ks, c, a.W, a.H, lds_cam, cam);

    // one-launch forward, full SH3 rows in the split layout (what the reference model trains with): the block's SH slice is
    // requested now and consumed after the projection chain (sh_rows_issue)
    const bool prefetch_sh = STAGE == 0 && DEG == 3 && a.sh_rest != nullptr && a.K == 16 && (((uintptr_t)a.sh_rest & 15) == 0);
    // (vector memory loads return in issue order: the geometry inputs are requested FIRST, so that the chain below waits
    //  for them only and runs while the SH slice is still in flight)
    float mean[3] = {0.f, 0.f, 0.f};
    if (in_range) { mean[0] = a.means[3 * n]; mean[1] = a.means[3 * n + 1]; mean[2] = a.means[3 * n + 2]; }
    float4 q4 = make_float4(1.f, 0.f, 0.f, 0.f);
    float sc_raw[3] = {0.f, 0.f, 0.f}, op_raw = 0.f;
    if (STAGE != 2 && in_range) {
        q4 = reinterpret_cast<const float4*>(a.quats)[n];
        sc_raw[0] = a.scales[3 * n]; sc_raw[1] = a.scales[3 * n + 1]; sc_raw[2] = a.scales[3 * n + 2];
        op_raw = a.opacities[n];
    }
    ShPrefetch pf;
    if (STAGE == 0 && DEG == 3 && prefetch_sh)
        sh_rows_issue(a.colors_in, a.sh_rest, n0, (int)min((int64_t)kProjThreads, a.N - n0), pf);

    bool vis = false;
    if (STAGE != 2) {
        Splat2D s;
        s.radius = 0; s.mx = s.my = s.depth = s.A = s.B = s.C = s.cxx = s.cyy = 0.f; s.x0 = s.x1 = s.y0 = s.y1 = 0;
        if (in_range) {
            const float quat[4] = {q4.x, q4.y, q4.z, q4.w};
            const float scale[3] = {act_scale(sc_raw[0], a.activations), act_scale(sc_raw[1], a.activations),
                                    act_scale(sc_raw[2], a.activations)};
            s = project_gaussian(mean, quat, scale, cam, a.W, a.H, a.eps2d, a.near_p, a.far_p, a.radius_clip, GS_TILE, a.tw, a.th);
        }
        vis = s.radius > 0;
        if (!vis && in_range && a.rect_ref) a.rect_ref[f] = make_uint2(0u, 0u);
        int x0 = 0, x1 = 0, y0 = 0, y1 = 0;
        float op = 0.f, ex = -1.f, ey = -1.f;
        if (vis) {
            x0 = s.x0; x1 = s.x1; y0 = s.y0; y1 = s.y1;   // from the unrounded centre (gs_math.h: preal)
            if (a.rect_ref) a.rect_ref[f] = make_uint2((uint32_t)x0 | ((uint32_t)x1 << 16), (uint32_t)y0 | ((uint32_t)y1 << 16));
            op = act_opacity(op_raw, a.activations);
            alpha_extent(op, s.cxx, s.cyy, ex, ey);
            // tight mode: keep only the tiles of the 3-sigma rectangle that hold a pixel centre where
            // alpha can reach 1/255 (the blend skips every other pixel anyway: identical image)
            if (a.tight) tile_rect_tight(s.mx, s.my, ex, ey, a.W, a.H, GS_TILE, x0, x1, y0, y1);
        }
        // tile footprint: rectangle + (for rectangles of <= 32 tiles) a row-major bit per tile
        const int rw = x1 - x0, rect_tiles = rw * (y1 - y0);
        uint32_t tmask = rect_tiles >= 32 ? 0xffffffffu : ((1u << rect_tiles) - 1u);
        if (a.tight && rect_tiles > 0 && rect_tiles <= 32) {
            // exact test per tile: some pixel centre of the tile must allow alpha >= 1/255
            const float tau = __log2f(255.f * op) + 0.02f, lim = tau + 1e-4f * fabsf(tau);
            const float qa = 0.5f * kLog2e * s.A, qb = kLog2e * s.B, qc = 0.5f * kLog2e * s.C;
            tmask = 0u;
            const float inv_rw = 1.0f / (float)rw;
            for (int i = 0; i < rect_tiles; ++i) {
                const int row = div_by_width(i, inv_rw), ty = y0 + row, tx = x0 + (i - row * rw);
                const float dx0 = (float)(tx * GS_TILE) + 0.5f - s.mx, dy0 = (float)(ty * GS_TILE) + 0.5f - s.my;
                if (quad_min_on_rect(qa, qb, qc, dx0, dx0 + (float)(GS_TILE - 1), dy0, dy0 + (float)(GS_TILE - 1)) <= lim) tmask |= 1u << i;
            }
        }
        const int cnt = rect_tiles <= 32 ? __popc(tmask) : rect_tiles;
        if (in_range) {
            a.radii[f] = s.radius;
            reinterpret_cast<float2*>(a.means2d)[f] = make_float2(s.mx, s.my);
            a.depths[f] = s.depth;
            a.conics[3 * f] = s.A; a.conics[3 * f + 1] = s.B; a.conics[3 * f + 2] = s.C;
            a.bbox[f] = make_uint4((uint32_t)x0 | ((uint32_t)x1 << 16), (uint32_t)y0 | ((uint32_t)y1 << 16), tmask, (uint32_t)cnt);
            a.tiles_per_gauss[f] = cnt;
            if (vis) {
                // blend record: conic pre-scaled so the kernels evaluate exp2(-(hA dx^2 + B dx dy + hC dy^2))
                float4* r = a.rec + 3 * f;
                r[0] = make_float4(s.mx, s.my, 0.5f * kLog2e * s.A, kLog2e * s.B);
                r[1] = make_float4(0.5f * kLog2e * s.C, op, ex, ey);
            }
        }
    } else {
        vis = in_range && a.radii[f] > 0;
    }
    if (STAGE == 1) return;

    float rgb[3] = {0.5f, 0.5f, 0.5f};
    if (DEG >= 0) {
        vis_s[threadIdx.x] = vis ? 1 : 0;
        const int any_vis = __syncthreads_or(vis ? 1 : 0);
        if (any_vis) {
            const int rows = (int)min((int64_t)kProjThreads, a.N - n0);
            constexpr int ka3 = 3 * (DEG + 1) * (DEG + 1);
            if (STAGE == 0 && DEG == 3 && prefetch_sh) {
                sh_rows_commit(pf, a.sh_rest, n0, rows, tile);
            } else if (a.sh_rest) {
                if (a.K == 16) stage_sh_rows_split<16>(a.colors_in, a.sh_rest, n0, rows, 16, ka3, vis_s, tile);
                else stage_sh_rows_split<0>(a.colors_in, a.sh_rest, n0, rows, a.K, ka3, vis_s, tile);
            } else {
                if (a.K == 16) stage_sh_rows<16>(a.colors_in, n0, rows, 16, ka3, vis_s, tile);
                else stage_sh_rows<0>(a.colors_in, n0, rows, a.K, ka3, vis_s, tile);
            }
            __syncthreads();
            if (vis) {
                float ux, uy, uz;
                view_dir(mean, cam, ux, uy, uz);
                sh_to_rgb(DEG < 0 ? 0 : DEG, tile + threadIdx.x * (3 * a.K + 1), ux, uy, uz, rgb);
                if (DEG >= 1 && a.sh_jac) {
                    float G[12];
                    sh_dir_jacobian(DEG, tile + threadIdx.x * (3 * a.K + 1), ux, uy, uz, G);
                    // nine floats per Gaussian: eight as two aligned quads [C*N][8], the ninth in a plane of its own behind them
                    float4* jp = a.sh_jac + 2 * f;
                    jp[0] = make_float4(G[0], G[1], G[2], G[4]); jp[1] = make_float4(G[5], G[6], G[8], G[9]);
                    reinterpret_cast<float*>(a.sh_jac + 2 * (int64_t)a.C * a.N)[f] = G[10];
                }
            }
        }
    } else if (in_range) {
        const float* src = a.colors_in + 3 * (a.colors_per_camera ? f : n);
        rgb[0] = src[0]; rgb[1] = src[1]; rgb[2] = src[2];
    }
    if (in_range) {
        a.colors_out[3 * f] = rgb[0]; a.colors_out[3 * f + 1] = rgb[1]; a.colors_out[3 * f + 2] = rgb[2];
        if (vis) a.rec[3 * f + 2] = make_float4(rgb[0], rgb[1], rgb[2], 0.f);
    }
}

// ------------------------------------------------------------------------------------------------
struct ProjBwdArgs {
    int C, cam, K, colors_per_camera, W, H, accumulate;
    int64_t N;
    float eps2d, near_p, far_p;
    const float *means, *quats, *scales, *colors_in, *sh_rest, *viewmats, *Ks, *colors_post;
    const int32_t *radii, *tiles_per_gauss, *cum_tiles;
    const float4* rows;      // [I*4][3]: one row per (intersection slot, tile quadrant)
    const int32_t* row_base; // [I/16+1]: gradient rows in front of every 16th slot (gs_blend_fwd's scan of the quadrant masks)
    const uint8_t* qmask;    // [I] by slot: which of the four quadrant rows exist (the readers' share of the scan: rows_before)
    const float4* sh_jac;    // optional, from gs_project_fwd ([C*N][8] + [C*N]): the SH rows are then not read at all
    float *v_means, *v_quats, *v_scales, *v_opacities, *v_colors, *v_sh_rest, *v_means2d_abs, *v_means2d,
        *v_conics, *v_colors_post, *v_colors_pre;
    const float* opacities;   // raw (logit) opacities, read only when activations != 0
    int activations;
    const int64_t* guard;     // step guard (gs_guard_set) or nullptr
    // Fused Adam (gs_project_bwd_adam): when `adam` is set no gradient is written; every parameter element is updated
    // in place right where its gradient is formed.  Tensor order of the reference's param_names:
    // 0 means, 1 log_scales, 2 quats, 3 sh_0, 4 sh_rest, 5 logit_opacities.
    int adam;
    // (three bases + six offsets instead of eighteen pointers: the fused kernel is SGPR-bound -- 106 live -- and every pointer it
    //  keeps across the row sum is two of them; adam_p / adam_m / adam_v below)
    float *ad_pbase, *ad_mbase, *ad_vbase;
    int64_t ad_off[6];
    const float* ad_hyper;    // {1/sqrt(1-beta2^t), lr_k/(1-beta1^t) x 6}  (gs_adam_hyper)
    float ad_b1, ad_b2, ad_eps;
    int64_t* ad_applied;
    // ... and update_statistics (/root/reference/model/gaussian.py:188-197) for the single camera, from the radius and
    // the absgrad this thread holds anyway (optional: NULL = not fused)
    float *st_max_radii, *st_grad_norm, *st_counts;
    float st_max_hw;
    // View-parallel step (distributed.ViewParallelStep): the row sums come from gs_row_sums (formed early, for the colour
    // exchange), and this view's two additive statistics are WRITTEN (not accumulated) into the all-reduce bucket:
    // |absgrad|_2 * max_hw and the visibility flag, 0 for culled Gaussians (st_max_hw as above)
    const float* row_sums;            // optional [C*N][12]
    float *st_gn_out, *st_cnt_out;    // optional [N] each (single camera)
    const int64_t* rblk;              // depth rounds (gs_rounds_set phase 3): the rows are two ranges, split at slot rblk[GS_ROUND_BASE]
};

__device__ __forceinline__ float* adam_p(const ProjBwdArgs& a, int t) { return a.ad_pbase + a.ad_off[t]; }
__device__ __forceinline__ float* adam_m(const ProjBwdArgs& a, int t) { return a.ad_mbase + a.ad_off[t]; }
__device__ __forceinline__ float* adam_v(const ProjBwdArgs& a, int t) { return a.ad_vbase + a.ad_off[t]; }
__device__ __forceinline__ float sh_adam_isbc2(const ProjBwdArgs& a) { return a.ad_hyper[0]; }
__device__ __forceinline__ float sh_adam_ss(const ProjBwdArgs& a, int t) { return a.ad_hyper[1 + t]; }

struct RowSum {
    float v[12];
};
constexpr int kRow4 = GS_ROW_FLOATS / 4;   // float4 per gradient row (include/gs_raster.h)

// Dense, coalesced write-out of one block's SH-gradient tile (LDS rows of 3K floats, row stride
// 3K+1) to v_shs[n0 : n0+rows]: unsplit [N,K,3] or split v_sh_0[N,1,3] + v_sh_rest[N,K-1,3].
template <int KC>   // KC > 0: K at compile time (divisions by 45 / 48 / 12 as multiply-shifts), see stage_sh_rows
__device__ __forceinline__ void write_sh_tile_k(const float* tile, int rows, int Krt, int64_t n0, float* v_colors,
                                                float* v_sh_rest, bool accumulate) {
    const int K = KC > 0 ? KC : Krt;
    const int row_f = 3 * K, stride = row_f + 1;
        if (v_sh_rest) {  // split layout: v_sh_0[N,1,3] and v_sh_rest[N,K-1,3]
            const int rest_f = row_f - 3, total = rows * rest_f;
            float* d0 = v_colors + n0 * 3;
            for (int e = threadIdx.x; e < rows * 3; e += blockDim.x) {
                const int g = e / 3;
                float v = tile[g * stride + (e - 3 * g)];
                if (accumulate) v += d0[e];
                d0[e] = v;
            }
            float* dr = v_sh_rest + n0 * rest_f;
            float4* dr4 = reinterpret_cast<float4*>(dr);
            for (int e4 = threadIdx.x; e4 < (total >> 2); e4 += blockDim.x) {
                float vv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int e = (e4 << 2) + i, g = e / rest_f;
                    vv[i] = tile[g * stride + 3 + (e - g * rest_f)];
                }
                float4 v = make_float4(vv[0], vv[1], vv[2], vv[3]);
                if (accumulate) { const float4 o = dr4[e4]; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
                nt_store4(v, dr4 + e4);
            }
            for (int e = (total & ~3) + threadIdx.x; e < total; e += blockDim.x) {
                const int g = e / rest_f;
                float v = tile[g * stride + 3 + (e - g * rest_f)];
                if (accumulate) v += dr[e];
                dr[e] = v;
            }
        } else {
        float* dst = v_colors + n0 * row_f;
        if ((row_f & 3) == 0) {
            const int per_row = row_f >> 2;
            float4* dst4 = reinterpret_cast<float4*>(dst);
            for (int e = threadIdx.x; e < rows * per_row; e += blockDim.x) {
                const int g = e / per_row, q = e - g * per_row;
                const float* sp = tile + g * stride + 4 * q;
                float4 v = make_float4(sp[0], sp[1], sp[2], sp[3]);
                if (accumulate) { const float4 o = dst4[e]; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
                dst4[e] = v;
            }
        } else {
            for (int e = threadIdx.x; e < rows * row_f; e += blockDim.x) {
                const int g = e / row_f, o = e - g * row_f;
                float v = tile[g * stride + o];
                if (accumulate) v += dst[e];
                dst[e] = v;
            }
        }
        }
}

__device__ __forceinline__ void write_sh_tile(const float* tile, int rows, int K, int64_t n0, float* v_colors,
                                              float* v_sh_rest, bool accumulate) {
    if (K == 16) write_sh_tile_k<16>(tile, rows, 16, n0, v_colors, v_sh_rest, accumulate);
    else write_sh_tile_k<0>(tile, rows, K, n0, v_colors, v_sh_rest, accumulate);
}

#define GS_ADAM_BATCH 3
// The same walk over one block's SH-gradient tile, but as an in-place Adam update of sh_0 / sh_rest (split layout)
// and their moments: the 48 SH gradients per Gaussian (81 % of all gradient bytes at SH3) are never written to HBM
// nor read back by a separate optimizer pass.  (The block staged its own SH rows into LDS before the barrier in
// front of this call, and no other block touches them, so updating in place is race-free.)
// KC > 0 fixes K at compile time (K = 16: the reference's SH3 layout): the per-element Gaussian / offset divisions by 45
// become multiply-shifts (four run-time integer divisions per 16 bytes otherwise).
// (templated on the argument block: project_bwd_kernel reads the bias corrections from the device array gs_adam_hyper wrote,
//  sh_grad_views_kernel<.., ADAM> carries them as kernel arguments; both expose ad_p / ad_m / ad_v [3], [4] and the betas)
struct ProjBwdArgs;
__device__ __forceinline__ float* adam_p(const ProjBwdArgs& a, int t);
__device__ __forceinline__ float* adam_m(const ProjBwdArgs& a, int t);
__device__ __forceinline__ float* adam_v(const ProjBwdArgs& a, int t);
__device__ __forceinline__ float sh_adam_isbc2(const ProjBwdArgs& a);
__device__ __forceinline__ float sh_adam_ss(const ProjBwdArgs& a, int t);
template <int KC, class A>
__device__ __forceinline__ void adam_sh_tile(const float* tile, int rows, int Krt, int64_t n0, const A& a) {
    const int K = KC > 0 ? KC : Krt;
    const int row_f = 3 * K, stride = row_f + 1;
    const float isbc2 = sh_adam_isbc2(a), ss0 = sh_adam_ss(a, 3), ssr = sh_adam_ss(a, 4);
    {
        float *p0 = adam_p(a, 3) + n0 * 3, *m0 = adam_m(a, 3) + n0 * 3, *v0 = adam_v(a, 3) + n0 * 3;
        for (int e = threadIdx.x; e < rows * 3; e += blockDim.x) {
            const int g = e / 3;
            float p = p0[e], m = m0[e], v = v0[e];
            adam1(p, tile[g * stride + (e - 3 * g)], m, v, a.ad_b1, a.ad_b2, a.ad_eps, isbc2, ss0);
            p0[e] = p; m0[e] = m; v0[e] = v;
        }
    }
    if (K > 1) {
        const int rest_f = row_f - 3, total = rows * rest_f;
        float *pr = adam_p(a, 4) + n0 * rest_f, *mr = adam_m(a, 4) + n0 * rest_f, *vr = adam_v(a, 4) + n0 * rest_f;
        // (n0 is a multiple of the block size and rest_f * kProjThreads of 4: the block's slice starts 16-byte aligned
        //  whenever the tensor does)
        const bool aligned = ((((uintptr_t)pr | (uintptr_t)mr | (uintptr_t)vr) & 15) == 0);
        const int vec_end = aligned ? (total & ~3) : 0;
        float4 *pr4 = reinterpret_cast<float4*>(pr), *mr4 = reinterpret_cast<float4*>(mr), *vr4 = reinterpret_cast<float4*>(vr);
        // kAdamBatch iterations' worth of parameters and moments are requested before the first update is stored: the
        // in-place stores may alias the next loads as far as the compiler can tell, and a thread would otherwise pay one
        // HBM round trip per 16 bytes x 3
        constexpr int kAdamBatch = GS_ADAM_BATCH;
        const int n4 = vec_end >> 2;
        for (int e0 = threadIdx.x; e0 < n4; e0 += kAdamBatch * (int)blockDim.x) {
            float4 p[kAdamBatch], m[kAdamBatch], v[kAdamBatch];
#pragma unroll
            for (int u = 0; u < kAdamBatch; ++u) {
                const int e4 = e0 + u * (int)blockDim.x;
                if (e4 < n4) { p[u] = pr4[e4]; m[u] = nt_load4(mr4 + e4); v[u] = nt_load4(vr4 + e4); }
            }
#pragma unroll
            for (int u = 0; u < kAdamBatch; ++u) {
                const int e4 = e0 + u * (int)blockDim.x;
                if (e4 < n4) {
                    float gg[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int e = (e4 << 2) + i, g = e / rest_f;
                        gg[i] = tile[g * stride + 3 + (e - g * rest_f)];
                    }
                    adam1(p[u].x, gg[0], m[u].x, v[u].x, a.ad_b1, a.ad_b2, a.ad_eps, isbc2, ssr);
                    adam1(p[u].y, gg[1], m[u].y, v[u].y, a.ad_b1, a.ad_b2, a.ad_eps, isbc2, ssr);
                    adam1(p[u].z, gg[2], m[u].z, v[u].z, a.ad_b1, a.ad_b2, a.ad_eps, isbc2, ssr);
                    adam1(p[u].w, gg[3], m[u].w, v[u].w, a.ad_b1, a.ad_b2, a.ad_eps, isbc2, ssr);
                    pr4[e4] = p[u]; nt_store4(m[u], mr4 + e4); nt_store4(v[u], vr4 + e4);
                }
            }
        }
        for (int e = vec_end + threadIdx.x; e < total; e += blockDim.x) {
            const int g = e / rest_f;
            float p = pr[e], m = mr[e], v = vr[e];
            adam1(p, tile[g * stride + 3 + (e - g * rest_f)], m, v, a.ad_b1, a.ad_b2, a.ad_eps, isbc2, ssr);
            pr[e] = p; mr[e] = m; vr[e] = v;
        }
    }
}

// In-place Adam of one block's slice of the four geometry tensors (groups 0 means[N,3], 1 log_scales[N,3], 2 quats[N,4],
// 5 logit_opacities[N]) from gradients staged in LDS in element order: tile + {0, 3, 6, 10} * kProjThreads.
__device__ __forceinline__ void adam_geo_tile(const float* tile, int rows, int64_t n0, const ProjBwdArgs& a) {
    constexpr int kSeg = 4;
    const int group[kSeg] = {0, 1, 2, 5}, width[kSeg] = {3, 3, 4, 1}, goff[kSeg] = {0, 3 * kProjThreads, 6 * kProjThreads, 10 * kProjThreads};
    const float isbc2 = a.ad_hyper[0];
    float4 p[kSeg], m[kSeg], v[kSeg];
    bool has[kSeg];
    const int e4 = threadIdx.x;   // (a block's slice of a tensor holds at most kProjThreads 16-byte groups: 4 floats per Gaussian)
#pragma unroll
    for (int sgi = 0; sgi < kSeg; ++sgi) {   // every load first
        const int t = group[sgi], total = rows * width[sgi];
        float *pp = adam_p(a, t) + n0 * width[sgi], *mm = adam_m(a, t) + n0 * width[sgi], *vv = adam_v(a, t) + n0 * width[sgi];
        const bool aligned = ((((uintptr_t)pp | (uintptr_t)mm | (uintptr_t)vv) & 15) == 0);
        has[sgi] = aligned && e4 < (total >> 2);
        if (has[sgi]) {
            p[sgi] = reinterpret_cast<const float4*>(pp)[e4];
            m[sgi] = nt_load4(reinterpret_cast<const float4*>(mm) + e4);
            v[sgi] = nt_load4(reinterpret_cast<const float4*>(vv) + e4);
        }
    }
#pragma unroll
    for (int sgi = 0; sgi < kSeg; ++sgi) {
        const int t = group[sgi], total = rows * width[sgi];
        float *pp = adam_p(a, t) + n0 * width[sgi], *mm = adam_m(a, t) + n0 * width[sgi], *vv = adam_v(a, t) + n0 * width[sgi];
        const float ss = a.ad_hyper[1 + t];
        const float* g = tile + goff[sgi];
        if (has[sgi]) {
            const float4 g4 = *reinterpret_cast<const float4*>(g + 4 * e4);
            adam1(p[sgi].x, g4.x, m[sgi].x, v[sgi].x, a.ad_b1, a.ad_b2, a.ad_eps, isbc2, ss);
            adam1(p[sgi].y, g4.y, m[sgi].y, v[sgi].y, a.ad_b1, a.ad_b2, a.ad_eps, isbc2, ss);
            adam1(p[sgi].z, g4.z, m[sgi].z, v[sgi].z, a.ad_b1, a.ad_b2, a.ad_eps, isbc2, ss);
            adam1(p[sgi].w, g4.w, m[sgi].w, v[sgi].w, a.ad_b1, a.ad_b2, a.ad_eps, isbc2, ss);
            reinterpret_cast<float4*>(pp)[e4] = p[sgi];
            nt_store4(m[sgi], reinterpret_cast<float4*>(mm) + e4);
            nt_store4(v[sgi], reinterpret_cast<float4*>(vv) + e4);
        }
        // what the 16-byte groups do not cover: the whole slice when a tensor is not 16-byte aligned, else its last 0-3 floats
        const bool aligned = ((((uintptr_t)pp | (uintptr_t)mm | (uintptr_t)vv) & 15) == 0);
        for (int e = (aligned ? (total & ~3) : 0) + (int)threadIdx.x; e < total; e += blockDim.x) {
            float ps = pp[e], ms = mm[e], vs = vv[e];
            adam1(ps, g[e], ms, vs, a.ad_b1, a.ad_b2, a.ad_eps, isbc2, ss);
            pp[e] = ps; mm[e] = ms; vv[e] = vs;
        }
    }
}


// ---- gradient rows of the wave's Gaussians -------------------------------------------------------------------------
// blend_bwd leaves one 48-byte row per (intersection, quadrant) some pixel took, COMPACT and in slot order (round 6:
// row_base = gs_blend_fwd's scan of the quadrant masks): the rows of a Gaussian are contiguous, the Gaussians of a wave follow
// one another with nothing in between -- the wave's rows are ONE contiguous range [R0, R0 + T), read 64 rows at a time (an
// ITEM), a row per lane, 3 KB of fully used cache lines per item and the next item in flight while this one is added.
// (Rounds 3-5 kept a row per LISTED (intersection, quadrant): 192 bytes per list entry, found through a mask byte per entry;
// a lane gathered up to four predicated rows, an owner lookup through LDS told it whose they were, and on realistic
// footprints -- where 1-3 % of the listed entries are walked -- the pass read masks of rows that did not exist.)
// Each lane leaves its row in LDS; every Gaussian adds the rows of its own range [lo, hi) of the item in row order.  An item
// that lies inside ONE Gaussian's range is not staged at all: the lanes keep partial sums in registers across such items and
// one DPP reduction closes the run (a splat over the whole image holds 10^5 rows).  Fixed order -> reproducible sums, and the
// same bits from project_bwd_kernel and row_sums_kernel.
constexpr int kRowWaveFloats = 64 * 12;   // LDS of one wave: an item's 64 rows

// Depth rounds (a.rblk: gs_rounds_set phase 3): the wave's rows are TWO such ranges -- the slots of the front round's Gaussians lie
// in front of slot GS_ROUND_BASE, those of the back round's behind it, and the Gaussians of one round follow one another inside
// their range.  The walk below runs over the two ranges laid end to end: row j of the wave is row R0 + j of the front range for
// j < Ta and row R0b + (j - Ta) of the back range behind (an item may straddle the seam: every lane forms its own address).
template <class A>
__device__ __forceinline__ void row_sum_wave(const A& a, int nr, int r0, int base, RowSum& s, float* wl) {
    const int lane = lane_id();
#pragma unroll
    for (int i = 0; i < 12; ++i) s.v[i] = 0.f;
    const bool two = a.rblk != nullptr;   // (kernel argument: uniform)
    const bool back = two && (int64_t)base >= a.rblk[GS_ROUND_BASE];
    const int incl = wave_incl_scan_add(back ? 0 : nr);
    int o = incl - nr;
    const int Ta = __builtin_amdgcn_readlane(incl, 63);
    int T = Ta;
    int64_t R0b = 0;
    if (two) {
        const int inb = wave_incl_scan_add(back ? nr : 0);
        if (back) o = Ta + inb - nr;
        T += __builtin_amdgcn_readlane(inb, 63);
        const unsigned long long bb = __ballot(back && nr > 0);
        if (bb) R0b = __shfl(r0, __builtin_ctzll(bb), 64);
    }
    if (T == 0) return;   // wave-uniform
    const unsigned long long fb = __ballot(!back && nr > 0);
    const int64_t R0 = fb ? __shfl(r0, __builtin_ctzll(fb), 64) : 0;
    float4* item = reinterpret_cast<float4*>(wl);
    const int n_items = (T + 63) >> 6;
    // (the third quad of a row holds three live floats: loaded as 12 bytes)
    struct Row { float4 a, b; float cx, cy, cz; };
    auto fetch = [&](Row& d, int it) {
        const int j = 64 * it + lane;
        d.a = d.b = make_float4(0.f, 0.f, 0.f, 0.f); d.cx = d.cy = d.cz = 0.f;
        if (j < T) {
            const float4* rp = a.rows + kRow4 * (j < Ta ? R0 + j : R0b + (j - Ta));
            d.a = rp[0]; d.b = rp[1];
            const float* c = reinterpret_cast<const float*>(rp + 2);
            d.cx = c[0]; d.cy = c[1]; d.cz = c[2];
        }
    };
    auto add = [](RowSum& p, const Row& d) {
        p.v[0] += d.a.x; p.v[1] += d.a.y; p.v[2] += d.a.z; p.v[3] += d.a.w;
        p.v[4] += d.b.x; p.v[5] += d.b.y; p.v[6] += d.b.z; p.v[7] += d.b.w;
        p.v[8] += d.cx; p.v[9] += d.cy; p.v[10] += d.cz;
    };
    // the wave-wide sums of `p` go to lane `owner`
    auto close = [&](RowSum& p, int owner) {
#pragma unroll
        for (int i = 0; i < 11; ++i) {
            const float t = wave_reduce_add_dpp(p.v[i]);
            if (lane == owner) s.v[i] += t;
        }
    };
    Row cur, nxt;
    fetch(cur, 0);
    int it = 0;
    while (it < n_items) {
        if (it + 1 < n_items) fetch(nxt, it + 1);
        const int jb = 64 * it;
        const int lo = max(o, jb) - jb, hi = min(o + nr, jb + 64) - jb;   // this Gaussian's rows inside the item
        const unsigned long long whole = __ballot(nr > 0 && lo == 0 && hi == 64);
        if (whole) {   // wave-uniform: the item lies inside ONE Gaussian's range -- and so do the items up to `last`
            const int owner = __builtin_ctzll(whole);
            const int last = (__shfl(o + nr, owner, 64) >> 6) - 1;   // the last item that is all this Gaussian's
            RowSum p;
#pragma unroll
            for (int i = 0; i < 12; ++i) p.v[i] = 0.f;
            add(p, cur);
            int k = it + 1;
            if (k <= last) { add(p, nxt); ++k; }
            // four items in flight at a time: a wave that holds a splat over the whole image walks 10^4 .. 10^5 rows on its own, and
            // one item per round trip made it the kernel's tail (round 6: 0.29 ms on the heavy-tailed 1 M scene, 0.21 before)
            for (; k + 3 <= last; k += 4) {
                Row b0, b1, b2, b3;
                fetch(b0, k); fetch(b1, k + 1); fetch(b2, k + 2); fetch(b3, k + 3);
                add(p, b0); add(p, b1); add(p, b2); add(p, b3);
            }
            for (; k <= last; ++k) { Row b0; fetch(b0, k); add(p, b0); }
            close(p, owner);
            it = last + 1;
            if (it < n_items) fetch(cur, it);   // (when the run was one item long this is `nxt` again: rare, and cached)
            continue;
        }
        // segments of 16 rows and more: one masked wave reduction each (an owner that adds its rows one by one holds the other
        // 63 lanes for as many LDS round trips); the shorter ones: every owner walks its own rows in LDS
        unsigned long long longs = __ballot(nr > 0 && hi - lo >= 16);
        for (; longs; longs &= longs - 1ull) {
            const int owner = __builtin_ctzll(longs);
            const int slo = __shfl(lo, owner, 64), shi = __shfl(hi, owner, 64);
            const bool in = lane >= slo && lane < shi;
            RowSum p;
            p.v[0] = in ? cur.a.x : 0.f; p.v[1] = in ? cur.a.y : 0.f; p.v[2] = in ? cur.a.z : 0.f; p.v[3] = in ? cur.a.w : 0.f;
            p.v[4] = in ? cur.b.x : 0.f; p.v[5] = in ? cur.b.y : 0.f; p.v[6] = in ? cur.b.z : 0.f; p.v[7] = in ? cur.b.w : 0.f;
            p.v[8] = in ? cur.cx : 0.f; p.v[9] = in ? cur.cy : 0.f; p.v[10] = in ? cur.cz : 0.f;
            close(p, owner);
        }
        if (__any(nr > 0 && hi > lo && hi - lo < 16)) {   // wave-uniform
            item[3 * lane] = cur.a; item[3 * lane + 1] = cur.b; item[3 * lane + 2] = make_float4(cur.cx, cur.cy, cur.cz, 0.f);
            __builtin_amdgcn_wave_barrier();   // (LDS operations of one wave complete in issue order)
            if (hi - lo < 16) {
                for (int r = lo; r < hi; ++r) {
                    const float4 ix = item[3 * r], iy = item[3 * r + 1], iz = item[3 * r + 2];
                    s.v[0] += ix.x; s.v[1] += ix.y; s.v[2] += ix.z; s.v[3] += ix.w;
                    s.v[4] += iy.x; s.v[5] += iy.y; s.v[6] += iy.z; s.v[7] += iy.w;
                    s.v[8] += iz.x; s.v[9] += iz.y; s.v[10] += iz.z;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        cur = nxt;
        ++it;
    }
}

// SUMS: the row sums come from gs_row_sums (a.row_sums) -- an instantiation of its own, so that the row-walking form keeps the
// registers it had (a run-time branch cost it its third wave per SIMD and put the fused form into scratch).
template <int DEG, bool ADAM, bool SUMS = false>
__global__ __launch_bounds__(kProjThreads) void project_bwd_kernel(const ProjBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if (guard_tripped(a.guard)) return;
    if (ADAM && a.ad_applied != nullptr && blockIdx.x == 0 && threadIdx.x == 0) a.ad_applied[0] += 1;
    float* lds_cam = smem;
    int* vis_s = reinterpret_cast<int*>(smem + 32);
    float* tile = smem + 32 + kProjThreads;

    const int c = a.cam;
    const int64_t n0 = (int64_t)blockIdx.x * kProjThreads;
    const int64_t n = n0 + threadIdx.x;
    const int64_t f = (int64_t)c * a.N + n;
    const bool in_range = n < a.N;
    stage_camera(a.viewmats, a.Ks, c, a.W, a.H, lds_cam);   // (read after the row sum: read_camera)

    const bool vis = in_range && a.radii[f] > 0;
    const int cnt = vis ? a.tiles_per_gauss[f] : 0;
    const int base = vis ? a.cum_tiles[f] : 0;
    int r0 = 0, nr = 0;   // this Gaussian's gradient rows: [r0, r0 + nr)
    if (!SUMS && cnt > 0) { r0 = rows_before(a.row_base, a.qmask, base); nr = rows_before(a.row_base, a.qmask, base + cnt) - r0; }

    // ---- 1. sum this Gaussian's gradient rows (contiguous and compact, written once each by blend_bwd) -- or take the sums gs_row_sums
    //         left (a.row_sums: the view-parallel step forms them early, for the colour-gradient exchange)
    RowSum s;
    if (SUMS) {
        const float4* rs = reinterpret_cast<const float4*>(a.row_sums) + 3 * f;
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f), y = x, z = x;
        if (vis) { x = rs[0]; y = rs[1]; z = rs[2]; }
        s.v[0] = x.x; s.v[1] = x.y; s.v[2] = x.z; s.v[3] = x.w; s.v[4] = y.x; s.v[5] = y.y; s.v[6] = y.z; s.v[7] = y.w;
        s.v[8] = z.x; s.v[9] = z.y; s.v[10] = z.z; s.v[11] = 0.f;
    } else {
        row_sum_wave(a, nr, r0, base, s, tile + (threadIdx.x >> 6) * kRowWaveFloats);
    }

    // ---- 2. colour path
    Camera cam;
    read_camera(lds_cam, cam);
    float v_mean[3] = {0.f, 0.f, 0.f}, v_quat[4] = {0.f, 0.f, 0.f, 0.f}, v_scale[3] = {0.f, 0.f, 0.f};
    float mean[3] = {0.f, 0.f, 0.f};
    // (the fused form reads its parameters through the flat buffer it updates: six pointers fewer to keep in SGPRs)
    const float* p_means = ADAM ? adam_p(a, 0) : a.means;
    const float* p_scales = ADAM ? adam_p(a, 1) : a.scales;
    const float* p_quats = ADAM ? adam_p(a, 2) : a.quats;
    const float* p_opac = ADAM ? adam_p(a, 5) : a.opacities;
    if (in_range) { mean[0] = p_means[3 * n]; mean[1] = p_means[3 * n + 1]; mean[2] = p_means[3 * n + 2]; }
    const float v_rgb[3] = {s.v[8], s.v[9], s.v[10]};
    if (DEG >= 0) {
        const int row_f = 3 * a.K, stride = row_f + 1;
        constexpr int ka3 = 3 * (DEG + 1) * (DEG + 1);
        // With the forward's direction Jacobian (a.sh_jac) the backward needs no SH coefficient: v_sh = Y (x) v_pre, and the
        // direction term of v_mean is J^T v_pre -- the block neither streams its 48 floats per Gaussian into LDS nor waits
        // for them (0.04 ms of the fused step kernel; the forward pays 36 B per visible Gaussian and ~200 FMAs).
        const bool use_jac = a.sh_jac != nullptr;
        float G[12] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float rgb[3] = {0.f, 0.f, 0.f};
        if (vis) {   // (requested before the barrier)
            rgb[0] = a.colors_post[3 * f]; rgb[1] = a.colors_post[3 * f + 1]; rgb[2] = a.colors_post[3 * f + 2];
            if (use_jac && DEG >= 1) {
                const float4 j0 = a.sh_jac[2 * f], j1 = a.sh_jac[2 * f + 1];
                G[10] = reinterpret_cast<const float*>(a.sh_jac + 2 * (int64_t)a.C * a.N)[f];
                G[0] = j0.x; G[1] = j0.y; G[2] = j0.z; G[4] = j0.w; G[5] = j1.x; G[6] = j1.y; G[8] = j1.z; G[9] = j1.w;
            }
        }
        vis_s[threadIdx.x] = vis ? 1 : 0;
        __syncthreads();   // (every wave is done with its row-sum scratch in `tile`)
        const int rows = (int)min((int64_t)kProjThreads, a.N - n0);
        if (!use_jac) {
            if (a.sh_rest) {
                if (a.K == 16) stage_sh_rows_split<16>(a.colors_in, a.sh_rest, n0, rows, 16, ka3, vis_s, tile);
                else stage_sh_rows_split<0>(a.colors_in, a.sh_rest, n0, rows, a.K, ka3, vis_s, tile);
            } else {
                if (a.K == 16) stage_sh_rows<16>(a.colors_in, n0, rows, 16, ka3, vis_s, tile);
                else stage_sh_rows<0>(a.colors_in, n0, rows, a.K, ka3, vis_s, tile);
            }
            __syncthreads();
        }
        float* my = tile + threadIdx.x * stride;
        if (vis) {
            float ux, uy, uz;
            const float dn = view_dir(mean, cam, ux, uy, uz);
            if (use_jac) sh_vjp_jac(DEG < 0 ? 0 : DEG, G, rgb, v_rgb, ux, uy, uz, dn, my, v_mean);
            else sh_vjp(DEG < 0 ? 0 : DEG, my, rgb, v_rgb, ux, uy, uz, dn, my, v_mean, false);
            for (int o = ka3; o < row_f; ++o) my[o] = 0.f;
            if (a.v_colors_pre) {  // gradient w.r.t. the pre-clamp colour: what gs_sh_grad_views consumes
                float* d = a.v_colors_pre + 3 * f;
                d[0] = rgb[0] > 0.f ? v_rgb[0] : 0.f; d[1] = rgb[1] > 0.f ? v_rgb[1] : 0.f; d[2] = rgb[2] > 0.f ? v_rgb[2] : 0.f;
            }
        } else if (in_range) {
            for (int o = 0; o < row_f; ++o) my[o] = 0.f;
            if (a.v_colors_pre) { float* d = a.v_colors_pre + 3 * f; d[0] = 0.f; d[1] = 0.f; d[2] = 0.f; }
        }
        if (ADAM) {
            __syncthreads();
            if (a.K == 16) adam_sh_tile<16>(tile, rows, 16, n0, a);
            else adam_sh_tile<0>(tile, rows, a.K, n0, a);
        } else if (a.v_colors) {  // NULL: the caller rebuilds the SH gradients from v_colors_pre (gs_sh_grad_views)
            __syncthreads();
            write_sh_tile(tile, rows, a.K, n0, a.v_colors, a.v_sh_rest, a.accumulate);
        }
    } else if (in_range) {
        if (a.colors_per_camera) {
            float* d = a.v_colors + 3 * f;
            d[0] = v_rgb[0]; d[1] = v_rgb[1]; d[2] = v_rgb[2];
        } else {
            float* d = a.v_colors + 3 * n;
            if (a.accumulate) { d[0] += v_rgb[0]; d[1] += v_rgb[1]; d[2] += v_rgb[2]; }
            else { d[0] = v_rgb[0]; d[1] = v_rgb[1]; d[2] = v_rgb[2]; }
        }
    }

    // ---- 3. projection VJP
    float sc_fac[3] = {1.f, 1.f, 1.f}, op_fac = 1.f;   // d activated / d raw (identity without activations)
    if (vis) {
        const float4 q4 = reinterpret_cast<const float4*>(p_quats)[n];
        const float quat[4] = {q4.x, q4.y, q4.z, q4.w};
        const float scale[3] = {act_scale(p_scales[3 * n], a.activations), act_scale(p_scales[3 * n + 1], a.activations),
                                act_scale(p_scales[3 * n + 2], a.activations)};
        if (a.activations) { sc_fac[0] = scale[0]; sc_fac[1] = scale[1]; sc_fac[2] = scale[2]; }
        ProjChainT<preal> p;
        if (project_chain<preal>(mean, quat, scale, cam, a.eps2d, a.near_p, a.far_p, p))
            project_vjp<preal>(scale, cam, p, s.v[0], s.v[1], s.v[4], s.v[5], s.v[6], 0.f, v_mean, v_quat, v_scale);
    }
    float geo_op = 0.f;
    if (in_range) {
        if (a.activations && vis) { const float o = act_opacity(p_opac[n], 1); op_fac = o * (1.f - o); }
        v_scale[0] *= sc_fac[0]; v_scale[1] *= sc_fac[1]; v_scale[2] *= sc_fac[2];
        const float v_op = s.v[7] * op_fac;
        float* vm = a.v_means + 3 * n; float* vq = a.v_quats + 4 * n; float* vs = a.v_scales + 3 * n;
        if (ADAM) {
            geo_op = v_op;   // (applied block-wide below: adam_geo_tile)
            if (a.st_max_radii != nullptr && vis) {   // same arithmetic as update_statistics_kernel
                a.st_max_radii[n] = fmaxf(a.st_max_radii[n], (float)a.radii[f] * (1.f / a.st_max_hw));
                a.st_grad_norm[n] += sqrtf(s.v[2] * s.v[2] + s.v[3] * s.v[3]) * a.st_max_hw;
                a.st_counts[n] += 1.f;
            }
        } else if (a.accumulate) {
            vm[0] += v_mean[0]; vm[1] += v_mean[1]; vm[2] += v_mean[2];
            vq[0] += v_quat[0]; vq[1] += v_quat[1]; vq[2] += v_quat[2]; vq[3] += v_quat[3];
            vs[0] += v_scale[0]; vs[1] += v_scale[1]; vs[2] += v_scale[2];
            a.v_opacities[n] += v_op;
        } else {
            vm[0] = v_mean[0]; vm[1] = v_mean[1]; vm[2] = v_mean[2];
            vq[0] = v_quat[0]; vq[1] = v_quat[1]; vq[2] = v_quat[2]; vq[3] = v_quat[3];
            vs[0] = v_scale[0]; vs[1] = v_scale[1]; vs[2] = v_scale[2];
            a.v_opacities[n] = v_op;
        }
        reinterpret_cast<float2*>(a.v_means2d_abs)[f] = make_float2(s.v[2], s.v[3]);
        if (SUMS && a.st_gn_out) {   // (only with row_sums: the view-parallel step; the row-walking form keeps HEAD's registers)
            a.st_gn_out[n] = vis ? sqrtf(s.v[2] * s.v[2] + s.v[3] * s.v[3]) * a.st_max_hw : 0.f;   // (update_statistics_kernel's arithmetic)
            a.st_cnt_out[n] = vis ? 1.f : 0.f;
        }
        if (a.v_means2d) reinterpret_cast<float2*>(a.v_means2d)[f] = make_float2(s.v[0], s.v[1]);
        if (a.v_conics) { a.v_conics[3 * f] = s.v[4]; a.v_conics[3 * f + 1] = s.v[5]; a.v_conics[3 * f + 2] = s.v[6]; }
        if (a.v_colors_post) { a.v_colors_post[3 * f] = v_rgb[0]; a.v_colors_post[3 * f + 1] = v_rgb[1]; a.v_colors_post[3 * f + 2] = v_rgb[2]; }
    }
    if (DEG >= 0 && ADAM) {
        // Geometry Adam, block-wide: the 11 gradients of every Gaussian of the block go through LDS into element order, and
        // means / log-scales / quaternions / logit-opacities (+ their moments) are updated with 16-byte accesses, every load
        // of a thread issued before its first store.  One gradient per thread and launch-order scalar accesses (33 dependent
        // load -> store round trips per thread: the in-place stores may alias the next loads as far as the compiler can
        // tell) took 0.08 ms of the step for 0.3 GB; same arithmetic per element, bit-identical update.
        __syncthreads();   // (adam_sh_tile is done with the tile; every thread has read its own parameters)
        float* gm = tile;
        float* gs = tile + 3 * kProjThreads;
        float* gq = tile + 6 * kProjThreads;
        float* go = tile + 10 * kProjThreads;
        const int tid = threadIdx.x;
        gm[3 * tid] = v_mean[0]; gm[3 * tid + 1] = v_mean[1]; gm[3 * tid + 2] = v_mean[2];
        gs[3 * tid] = v_scale[0]; gs[3 * tid + 1] = v_scale[1]; gs[3 * tid + 2] = v_scale[2];
        gq[4 * tid] = v_quat[0]; gq[4 * tid + 1] = v_quat[1]; gq[4 * tid + 2] = v_quat[2]; gq[4 * tid + 3] = v_quat[3];
        go[tid] = geo_op;
        __syncthreads();
        adam_geo_tile(tile, (int)min((int64_t)kProjThreads, a.N - n0), n0, a);
    }
}

static size_t proj_lds_bytes(int K, int degree);
// backward: the tile region also holds the four waves' row-sum scratch (row_sum_slots), with or without SH
static size_t proj_bwd_lds_bytes(int K, int degree) {
    const size_t need = sizeof(float) * (32 + kProjThreads + (size_t)(kProjThreads / 64) * kRowWaveFloats);
    const size_t base = proj_lds_bytes(K, degree);
    return base > need ? base : need;
}

static size_t proj_lds_bytes(int K, int degree) {
    // (the SH tile also stages the 11 geometry gradients per Gaussian of the fused Adam: at least 11 floats per thread)
    const size_t tile = (size_t)kProjThreads * (size_t)((3 * K + 1) > 11 ? (3 * K + 1) : 11);
    return sizeof(float) * (32 + kProjThreads + (degree >= 0 ? tile : 0));
}


// ------------------------------------------------------------------------------------------------
// SH-parameter gradients of R views from the per-view pre-clamp colour gradients:
//   v_sh[n][k][:] = sum_r Y_k(dir(mean_n, camera_r)) * v_colors_pre[r][n][:]      (views in order r = 0..R-1)
// The view-parallel step exchanges v_colors_pre (3 floats per Gaussian and view) instead of the 48
// SH gradients per Gaussian; every rank then rebuilds the identical dense SH gradient with this
// kernel (distributed.py).  Same write-out as project_bwd_kernel.
struct ShGradArgs {
    int R, K;
    int64_t N;
    const float *means, *viewmats, *v_colors_pre;
    int64_t view_stride;   // floats between view r and view r+1 of v_colors_pre (3N for a dense [R,N,3]; the payload stride of
    int64_t cam_stride;    // gs_sh_adam_views) and of viewmats (16 for a dense [R,4,4])
    float *v_colors, *v_sh_rest;
    // gs_sh_adam_views: no gradient is written -- sh_0 / sh_rest and their moments are updated in place (ShAdam), gradients
    // scaled by grad_scale first (1/world), and max_radii[n] = max(max_radii[n], max_r radii_norm[r][n])
    float *ad_p[6], *ad_m[6], *ad_v[6];                       // (only [3] = sh_0 and [4] = sh_rest are set: param_names order)
    float ad_b1, ad_b2, ad_eps, ad_isbc2, ad_ss0, ad_ssr;     // betas, eps, 1/sqrt(1-beta2^t), lr/(1-beta1^t) of the two groups
    float grad_scale;
    const float* radii_norm;   // optional: view r's normalised radii at radii_norm + r * view_stride
    float* max_radii;
    const int64_t* guard;
};

__device__ __forceinline__ float* adam_p(const ShGradArgs& a, int t) { return a.ad_p[t]; }
__device__ __forceinline__ float* adam_m(const ShGradArgs& a, int t) { return a.ad_m[t]; }
__device__ __forceinline__ float* adam_v(const ShGradArgs& a, int t) { return a.ad_v[t]; }
__device__ __forceinline__ float sh_adam_isbc2(const ShGradArgs& a) { return a.ad_isbc2; }
__device__ __forceinline__ float sh_adam_ss(const ShGradArgs& a, int t) { return t == 3 ? a.ad_ss0 : a.ad_ssr; }

constexpr int kMaxViews = 64;

template <int DEG, bool ADAM>
__global__ __launch_bounds__(kProjThreads) void sh_grad_views_kernel(const ShGradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if (ADAM && guard_tripped(a.guard)) return;
    float* cam_pos = smem;                       // [kMaxViews * 3]
    float* tile = smem + kMaxViews * 3;          // [kProjThreads * (3K + 1)]
    const int64_t n0 = (int64_t)blockIdx.x * kProjThreads, n = n0 + threadIdx.x;
    if ((int)threadIdx.x < a.R) {
        Camera tmp;
        const float kid[9] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f};
        make_camera(a.viewmats + a.cam_stride * threadIdx.x, kid, 1, 1, tmp);
        cam_pos[3 * threadIdx.x] = tmp.pos[0]; cam_pos[3 * threadIdx.x + 1] = tmp.pos[1]; cam_pos[3 * threadIdx.x + 2] = tmp.pos[2];
    }
    __syncthreads();
    constexpr int Ka = (DEG + 1) * (DEG + 1);
    const int row_f = 3 * a.K, stride = row_f + 1;
    float acc[3 * Ka];
#pragma unroll
    for (int i = 0; i < 3 * Ka; ++i) acc[i] = 0.f;
    if (n < a.N) {
        const float mean[3] = {a.means[3 * n], a.means[3 * n + 1], a.means[3 * n + 2]};
        float rad = 0.f;
        for (int r = 0; r < a.R; ++r) {
            const float* v = a.v_colors_pre + (int64_t)r * a.view_stride + 3 * n;
            const float vr = v[0], vg = v[1], vb = v[2];
            if (ADAM && a.radii_norm) rad = fmaxf(rad, a.radii_norm[(int64_t)r * a.view_stride + n]);
            if (vr == 0.f && vg == 0.f && vb == 0.f) continue;   // not visible in view r (x + 0 is exact)
            Camera cam;
            cam.pos[0] = cam_pos[3 * r]; cam.pos[1] = cam_pos[3 * r + 1]; cam.pos[2] = cam_pos[3 * r + 2];
            float ux, uy, uz, Y[16];
            view_dir(mean, cam, ux, uy, uz);
            sh_basis(DEG, ux, uy, uz, Y);
#pragma unroll
            for (int k = 0; k < Ka; ++k) {
                acc[3 * k] += Y[k] * vr; acc[3 * k + 1] += Y[k] * vg; acc[3 * k + 2] += Y[k] * vb;
            }
        }
        if (ADAM && a.max_radii) a.max_radii[n] = fmaxf(a.max_radii[n], rad);
        float* my = tile + threadIdx.x * stride;
#pragma unroll
        for (int i = 0; i < 3 * Ka; ++i) my[i] = ADAM ? acc[i] * a.grad_scale : acc[i];   // (gs_adam_step scales the gradient it reads the same way)
        for (int o = 3 * Ka; o < row_f; ++o) my[o] = 0.f;
    }
    __syncthreads();
    const int rows = (int)min((int64_t)kProjThreads, a.N - n0);
    if (ADAM) {
        if (a.K == 16) adam_sh_tile<16>(tile, rows, 16, n0, a);
        else adam_sh_tile<0>(tile, rows, a.K, n0, a);
    } else {
        write_sh_tile(tile, rows, a.K, n0, a.v_colors, a.v_sh_rest, false);
    }
}

// ------------------------------------------------------------------------------------------------
// Row sums of every Gaussian as a kernel of its own (view-parallel step): what project_bwd_kernel's phase 1 computes, with
// the same function -> the same sums bit for bit.  Leaves row_sums[C*N][12] for gs_project_bwd(row_sums = ...) and, from
// them, everything another rank needs of this view BEFORE the long projection backward runs: the clamp-masked pre-clamp
// colour gradient v_colors_pre[f] (rounds 2-4: a second pass over the rows, gs_colors_pre_grad), optionally the
// normalised radii (radius / max_hw, 0 for culled Gaussians: the MAX statistic of update_statistics) and a copy of the view
// matrix -- the three pieces of the all-gather payload of distributed.ViewParallelStep.
struct RowSumsArgs {
    int64_t total;   // C * N
    const int32_t *radii, *tiles_per_gauss, *cum_tiles;
    const float* colors_post;
    const float4* rows;
    const int32_t* row_base;
    const uint8_t* qmask;
    float4* row_sums;
    float *v_colors_pre, *radii_norm, *cam_out;
    const float* viewmats;
    float max_hw;
    const int64_t* guard;
    const int64_t* rblk;   // depth rounds (see ProjBwdArgs)
};

__global__ __launch_bounds__(256) void row_sums_kernel(const RowSumsArgs a) {
    __shared__ __attribute__((aligned(16))) float wl_all[4][kRowWaveFloats];
    if (guard_tripped(a.guard)) return;
    const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool in_range = f < a.total;
    const int radius = in_range ? a.radii[f] : 0;
    const bool vis = radius > 0;
    const int cnt = vis ? a.tiles_per_gauss[f] : 0, base = vis ? a.cum_tiles[f] : 0;
    int r0 = 0, nr = 0;
    if (cnt > 0) { r0 = rows_before(a.row_base, a.qmask, base); nr = rows_before(a.row_base, a.qmask, base + cnt) - r0; }
    float rgb[3] = {0.f, 0.f, 0.f};
    if (vis) { rgb[0] = a.colors_post[3 * f]; rgb[1] = a.colors_post[3 * f + 1]; rgb[2] = a.colors_post[3 * f + 2]; }
    if (a.cam_out && blockIdx.x == 0 && threadIdx.x < 16) a.cam_out[threadIdx.x] = a.viewmats[threadIdx.x];
    RowSum s;
    row_sum_wave(a, nr, r0, base, s, wl_all[threadIdx.x >> 6]);
    if (!in_range) return;
    if (vis) {   // (culled Gaussians: gs_project_bwd does not read their sums)
        float4* rs = a.row_sums + 3 * f;
        rs[0] = make_float4(s.v[0], s.v[1], s.v[2], s.v[3]);
        rs[1] = make_float4(s.v[4], s.v[5], s.v[6], s.v[7]);
        rs[2] = make_float4(s.v[8], s.v[9], s.v[10], 0.f);
    }
    float* d = a.v_colors_pre + 3 * f;   // (project_bwd_kernel's v_colors_pre: masked where the clamp max(c + 0.5, 0) is active)
    d[0] = (vis && rgb[0] > 0.f) ? s.v[8] : 0.f; d[1] = (vis && rgb[1] > 0.f) ? s.v[9] : 0.f; d[2] = (vis && rgb[2] > 0.f) ? s.v[10] : 0.f;
    if (a.radii_norm) a.radii_norm[f] = vis ? (float)radius * (1.f / a.max_hw) : 0.f;   // (update_statistics_kernel's arithmetic)
}

}  // namespace gs

using namespace gs;

static int sh_views_launch(hipStream_t st, int sh_degree, bool adam, const ShGradArgs& a) {
    dim3 grid((unsigned)((a.N + kProjThreads - 1) / kProjThreads));
    const size_t lds = sizeof(float) * (kMaxViews * 3 + (size_t)kProjThreads * (3 * a.K + 1));
#define GS_SHV(D)                                                                                           \
    if (adam) hipLaunchKernelGGL((sh_grad_views_kernel<D, true>), grid, dim3(kProjThreads), lds, st, a);    \
    else hipLaunchKernelGGL((sh_grad_views_kernel<D, false>), grid, dim3(kProjThreads), lds, st, a)
    switch (sh_degree) {
        case 0: GS_SHV(0); break;
        case 1: GS_SHV(1); break;
        case 2: GS_SHV(2); break;
        default: GS_SHV(3); break;
    }
#undef GS_SHV
    GS_LAUNCH_CHECK("sh_grad_views_kernel");
    return GS_OK;
}

extern "C" int gs_sh_grad_views(void* stream, int R, int64_t N, int K, int sh_degree, const float* means,
                                const float* viewmats, const float* v_colors_pre, float* v_colors,
                                float* v_sh_rest) {
    GS_REQUIRE(R >= 1 && R <= kMaxViews, "1..64 views");
    GS_REQUIRE(N >= 0 && sh_degree >= 0 && sh_degree <= 3, "N >= 0 and sh_degree in 0..3");
    GS_REQUIRE(K >= (sh_degree + 1) * (sh_degree + 1) && K <= 16, "K must hold (sh_degree+1)^2 coefficients and be <= 16");
    if (N == 0) return GS_OK;
    GS_REQUIRE(means && viewmats && v_colors_pre && v_colors, "null pointer");
    ShGradArgs a = {};
    a.R = R; a.K = K; a.N = N; a.means = means; a.viewmats = viewmats; a.v_colors_pre = v_colors_pre;
    a.view_stride = 3 * N; a.cam_stride = 16;
    a.v_colors = v_colors; a.v_sh_rest = v_sh_rest;
    return sh_views_launch((hipStream_t)stream, sh_degree, false, a);
}

extern "C" int gs_sh_adam_views(void* stream, int R, int64_t N, int K, int sh_degree, const float* means, const float* payload,
                                int64_t payload_stride, float* sh_0, float* sh_0_exp_avg, float* sh_0_exp_avg_sq, float* sh_rest,
                                float* sh_rest_exp_avg, float* sh_rest_exp_avg_sq, float lr_sh_0, float lr_sh_rest, float beta1,
                                float beta2, float eps, int64_t step, float grad_scale, float* max_radii) {
    GS_REQUIRE(R >= 1 && R <= kMaxViews, "1..64 views");
    GS_REQUIRE(N >= 0 && sh_degree >= 0 && sh_degree <= 3, "N >= 0 and sh_degree in 0..3");
    GS_REQUIRE(K >= (sh_degree + 1) * (sh_degree + 1) && K <= 16, "K must hold (sh_degree+1)^2 coefficients and be <= 16");
    GS_REQUIRE(step >= 1, "step counts from 1");
    GS_REQUIRE(payload_stride >= 4 * N + 16, "payload_stride must hold [3N colour gradients | N radii | 16 camera floats]");
    if (N == 0) return GS_OK;
    GS_REQUIRE(means && payload && sh_0 && sh_0_exp_avg && sh_0_exp_avg_sq, "null pointer");
    GS_REQUIRE(K == 1 || (sh_rest && sh_rest_exp_avg && sh_rest_exp_avg_sq), "K > 1 needs sh_rest and its moments");
    ShGradArgs a = {};
    a.R = R; a.K = K; a.N = N; a.means = means;
    a.v_colors_pre = payload; a.radii_norm = max_radii ? payload + 3 * N : nullptr; a.viewmats = payload + 4 * N;
    a.view_stride = payload_stride; a.cam_stride = payload_stride;
    a.max_radii = max_radii; a.grad_scale = grad_scale;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    a.ad_p[3] = sh_0; a.ad_m[3] = sh_0_exp_avg; a.ad_v[3] = sh_0_exp_avg_sq;
    a.ad_p[4] = sh_rest; a.ad_m[4] = sh_rest_exp_avg; a.ad_v[4] = sh_rest_exp_avg_sq;
    a.ad_isbc2 = (float)(1.0 / sqrt(bc2));   // (adam_launch's host arithmetic, gs_adam.hip)
    a.ad_ss0 = (float)((double)lr_sh_0 / bc1); a.ad_ssr = (float)((double)lr_sh_rest / bc1);
    a.ad_b1 = beta1; a.ad_b2 = beta2; a.ad_eps = eps;
    a.guard = current_guard().info;
    return sh_views_launch((hipStream_t)stream, sh_degree, true, a);
}

extern "C" int gs_row_sums(void* stream, int C, int64_t N, const int32_t* radii, const float* colors_post,
                           const int32_t* tiles_per_gauss, const int32_t* cum_tiles, const float* rows, const int32_t* row_base, const uint8_t* qmask,
                           float* row_sums, float* v_colors_pre, float* radii_norm, float max_hw, const float* viewmats,
                           float* cam_out) {
    GS_REQUIRE(C >= 1 && N >= 0, "C>=1, N>=0");
    if (N == 0) return GS_OK;
    GS_REQUIRE(radii && colors_post && tiles_per_gauss && cum_tiles && rows && row_base && qmask && row_sums && v_colors_pre, "null pointer");
    GS_REQUIRE(((uintptr_t)row_sums & 15) == 0, "row_sums must be 16-byte aligned");
    GS_REQUIRE(!radii_norm || max_hw > 0.f, "radii_norm needs a positive image extent");
    GS_REQUIRE(!cam_out || (viewmats && C == 1), "cam_out: the single camera's view matrix");
    RowSumsArgs a;
    a.total = (int64_t)C * N; a.radii = radii; a.tiles_per_gauss = tiles_per_gauss; a.cum_tiles = cum_tiles;
    a.colors_post = colors_post; a.rows = reinterpret_cast<const float4*>(rows); a.row_base = row_base; a.qmask = qmask;
    a.row_sums = reinterpret_cast<float4*>(row_sums); a.v_colors_pre = v_colors_pre; a.radii_norm = radii_norm;
    a.cam_out = cam_out; a.viewmats = viewmats; a.max_hw = max_hw;
    a.guard = current_guard().info;
    a.rblk = current_rounds().phase == 3 ? current_rounds().blk : nullptr;   // (depth rounds: the rows are two ranges)
    hipLaunchKernelGGL(row_sums_kernel, dim3((unsigned)((a.total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    GS_LAUNCH_CHECK("row_sums_kernel");
    return GS_OK;
}

extern "C" int gs_project_fwd(void* stream, int C, int64_t N, int K, int sh_degree, const float* means,
                              const float* quats, const float* scales, const float* opacities,
                              const float* colors_in, const float* sh_rest, int colors_per_camera,
                              const float* viewmats, const float* Ks, int width, int height,
                              float eps2d, float near_plane, float far_plane, float radius_clip,
                              int tile_culling, int stage, int activations, int32_t* radii,
                              float* means2d, float* depths, float* conics, float* colors_out, float* rec,
                              uint32_t* bbox, int32_t* tiles_per_gauss, uint32_t* rect_ref, float* sh_jac) {
    GS_REQUIRE(C >= 1 && N >= 0 && width > 0 && height > 0, "C>=1, N>=0, positive image size");
    GS_REQUIRE(sh_degree <= 3, "sh_degree must be <= 3");
    GS_REQUIRE(sh_degree < 0 || (K >= (sh_degree + 1) * (sh_degree + 1) && K <= 16), "K must hold (sh_degree+1)^2 coefficients and be <= 16");
    GS_REQUIRE((width + GS_TILE - 1) / GS_TILE < 65536 && (height + GS_TILE - 1) / GS_TILE < 65536, "tile grid must fit 16 bits per axis");
    if (N == 0) return GS_OK;
    GS_REQUIRE(means && quats && scales && opacities && colors_in && viewmats && Ks, "null input pointer");
    GS_REQUIRE(radii && means2d && depths && conics && colors_out && rec && bbox && tiles_per_gauss, "null output pointer");
    ProjFwdArgs a;
    a.C = C; a.N = N; a.K = K; a.colors_per_camera = colors_per_camera; a.W = width; a.H = height;
    a.tw = (width + GS_TILE - 1) / GS_TILE; a.th = (height + GS_TILE - 1) / GS_TILE;
    a.eps2d = eps2d; a.near_p = near_plane; a.far_p = far_plane; a.radius_clip = radius_clip;
    a.tight = tile_culling != 0; a.activations = activations != 0;
    a.means = means; a.quats = quats; a.scales = scales; a.opacities = opacities; a.colors_in = colors_in;
    a.sh_rest = sh_degree >= 0 ? sh_rest : nullptr;
    a.viewmats = viewmats; a.Ks = Ks; a.radii = radii; a.means2d = means2d; a.depths = depths;
    a.conics = conics; a.colors_out = colors_out; a.rec = reinterpret_cast<float4*>(rec);
    a.bbox = reinterpret_cast<uint4*>(bbox); a.tiles_per_gauss = tiles_per_gauss;
    a.rect_ref = reinterpret_cast<uint2*>(rect_ref);
    a.sh_jac = sh_degree >= 1 ? reinterpret_cast<float4*>(sh_jac) : nullptr;
    dim3 grid((unsigned)((N + kProjThreads - 1) / kProjThreads), (unsigned)C);
    const size_t lds = proj_lds_bytes(K, sh_degree);
    hipStream_t st = (hipStream_t)stream;
    GS_REQUIRE(stage >= 0 && stage <= 2, "stage: 0 = geometry+colour, 1 = geometry, 2 = colour");
#define GS_PF(D, S) hipLaunchKernelGGL((project_fwd_kernel<D, S>), grid, dim3(kProjThreads), (S) == 1 ? proj_lds_bytes(K, -1) : lds, st, a)
#define GS_PF_DEG(S)                                                      \
    switch (sh_degree) {                                                  \
        case 0: GS_PF(0, S); break;                                       \
        case 1: GS_PF(1, S); break;                                       \
        case 2: GS_PF(2, S); break;                                       \
        case 3: GS_PF(3, S); break;                                       \
        default: GS_PF(-1, S); break;                                     \
    }
    if (stage == 0) { GS_PF_DEG(0) } else if (stage == 1) { GS_PF_DEG(1) } else { GS_PF_DEG(2) }
#undef GS_PF_DEG
#undef GS_PF
    GS_LAUNCH_CHECK("project_fwd_kernel");
    return GS_OK;
}

extern "C" int gs_project_bwd(void* stream, int C, int64_t N, int K, int sh_degree, const float* means,
                              const float* quats, const float* scales, const float* colors_in,
                              const float* sh_rest, int colors_per_camera, const float* viewmats,
                              const float* Ks, int width,
                              int height, float eps2d, float near_plane, float far_plane,
                              const int32_t* radii, const float* colors_post,
                              const int32_t* tiles_per_gauss, const int32_t* cum_tiles,
                              const float* rows, const int32_t* row_base, const uint8_t* qmask, float* v_means, float* v_quats, float* v_scales,
                              float* v_opacities, float* v_colors, float* v_sh_rest, float* v_means2d_abs,
                              float* v_means2d, float* v_conics, float* v_colors_post, float* v_colors_pre,
                              const float* opacities, int activations, const float* sh_jac, const float* row_sums,
                              float* stat_grad_norm, float* stat_count) {
    GS_REQUIRE(C >= 1 && N >= 0 && width > 0 && height > 0, "C>=1, N>=0, positive image size");
    GS_REQUIRE((stat_grad_norm == nullptr) == (stat_count == nullptr) && (!stat_count || (C == 1 && row_sums)),
               "stat_grad_norm / stat_count: both or neither, single camera, together with row_sums");
    GS_REQUIRE(sh_degree <= 3, "sh_degree must be <= 3");
    GS_REQUIRE(sh_degree < 0 || (K >= (sh_degree + 1) * (sh_degree + 1) && K <= 16), "K must hold (sh_degree+1)^2 coefficients and be <= 16");
    if (N == 0) return GS_OK;
    GS_REQUIRE(means && quats && scales && colors_in && viewmats && Ks && radii && colors_post && tiles_per_gauss && cum_tiles, "null input pointer");
    GS_REQUIRE(row_sums || (rows && row_base && qmask), "the gradient rows (rows + row_base + qmask) or their sums (row_sums, from gs_row_sums)");
    GS_REQUIRE(v_means && v_quats && v_scales && v_opacities && v_means2d_abs, "null output pointer");
    GS_REQUIRE(v_colors || sh_degree >= 0, "v_colors may be NULL only with SH colours (gradients rebuilt by gs_sh_grad_views)");
    ProjBwdArgs a;
    a.C = C; a.N = N; a.K = K; a.colors_per_camera = colors_per_camera; a.W = width; a.H = height;
    a.eps2d = eps2d; a.near_p = near_plane; a.far_p = far_plane;
    a.means = means; a.quats = quats; a.scales = scales; a.colors_in = colors_in; a.viewmats = viewmats;
    a.sh_rest = sh_degree >= 0 ? sh_rest : nullptr; a.v_sh_rest = a.sh_rest ? v_sh_rest : nullptr;
    a.Ks = Ks; a.colors_post = colors_post; a.radii = radii; a.tiles_per_gauss = tiles_per_gauss;
    a.cum_tiles = cum_tiles; a.rows = reinterpret_cast<const float4*>(rows); a.row_base = row_base; a.qmask = qmask;
    a.v_means = v_means; a.v_quats = v_quats; a.v_scales = v_scales; a.v_opacities = v_opacities;
    a.v_colors = v_colors; a.v_means2d_abs = v_means2d_abs; a.v_means2d = v_means2d;
    a.v_conics = v_conics; a.v_colors_post = v_colors_post; a.v_colors_pre = sh_degree >= 0 ? v_colors_pre : nullptr;
    GS_REQUIRE(!activations || opacities, "activations need the raw opacities");
    a.opacities = opacities; a.activations = activations != 0;
    a.sh_jac = sh_degree >= 0 ? reinterpret_cast<const float4*>(sh_jac) : nullptr;
    a.guard = current_guard().info;
    a.rblk = current_rounds().phase == 3 ? current_rounds().blk : nullptr;   // (depth rounds: the rows are two ranges)
    a.adam = 0; a.ad_hyper = nullptr; a.ad_applied = nullptr; a.ad_b1 = a.ad_b2 = a.ad_eps = 0.f;
    a.st_max_radii = a.st_grad_norm = a.st_counts = nullptr; a.st_max_hw = (float)(width > height ? width : height);
    a.row_sums = row_sums; a.st_gn_out = stat_grad_norm; a.st_cnt_out = stat_count;
    a.ad_pbase = a.ad_mbase = a.ad_vbase = nullptr;
    for (int t = 0; t < 6; ++t) a.ad_off[t] = 0;
    dim3 grid((unsigned)((N + kProjThreads - 1) / kProjThreads));
    const size_t lds = proj_bwd_lds_bytes(K, sh_degree);
    hipStream_t st = (hipStream_t)stream;
    // One launch per camera: launch c accumulates onto launch c-1 (same thread owns Gaussian n in
    // every launch, stream order serialises them) -> deterministic sum over cameras, no atomics.
    for (int c = 0; c < C; ++c) {
        a.cam = c; a.accumulate = c > 0;
#define GS_PB(D)                                                                                                        \
        if (row_sums) hipLaunchKernelGGL((project_bwd_kernel<D, false, true>), grid, dim3(kProjThreads), lds, st, a);      \
        else hipLaunchKernelGGL((project_bwd_kernel<D, false, false>), grid, dim3(kProjThreads), lds, st, a)
        switch (sh_degree) {
            case 0: GS_PB(0); break;
            case 1: GS_PB(1); break;
            case 2: GS_PB(2); break;
            case 3: GS_PB(3); break;
            default: GS_PB(-1); break;
        }
#undef GS_PB
        GS_LAUNCH_CHECK("project_bwd_kernel");
    }
    return GS_OK;
}


// Row reduction + SH-bwd + P-bwd + Adam in ONE pass (train-step graph, single camera): gs_project_bwd with the
// reference model's raw parameters (log-scales, logit opacities, split SH), but instead of writing the 59 gradients
// per Gaussian and reading them back in gs_adam_step, every parameter and both of its moments are updated in place
// where the gradient is formed.  params / exp_avg / exp_avg_sq: the flat buffers of gs_adam_step, the six tensors
// of param_names at offsets_host[6] floats.  hyper_dev: gs_adam_hyper.  Only v_means2d_abs (the `.absgrad`
// side channel) is still written; with max_radii / grad_norm_accum / counts given, update_statistics is applied too.
extern "C" int gs_project_bwd_adam(void* stream, int64_t N, int K, int sh_degree, float* params, float* exp_avg, float* exp_avg_sq,
                                   const int64_t* offsets_host, const float* viewmats, const float* Ks, int width, int height,
                                   float eps2d, float near_plane, float far_plane, const int32_t* radii, const float* colors_post,
                                   const int32_t* tiles_per_gauss, const int32_t* cum_tiles, const float* rows, const int32_t* row_base, const uint8_t* qmask,
                                   float* v_means2d_abs, float beta1, float beta2, float eps, const float* hyper_dev,
                                   int64_t* applied_dev, float* max_radii, float* grad_norm_accum, float* counts,
                                   const float* sh_jac) {
    GS_REQUIRE(N >= 0 && width > 0 && height > 0, "N>=0, positive image size");
    GS_REQUIRE(sh_degree >= 0 && sh_degree <= 3 && K >= (sh_degree + 1) * (sh_degree + 1) && K <= 16, "SH colours: 0 <= degree <= 3, (degree+1)^2 <= K <= 16");
    if (N == 0) return GS_OK;
    GS_REQUIRE(params && exp_avg && exp_avg_sq && offsets_host && viewmats && Ks && radii && colors_post && tiles_per_gauss &&
               cum_tiles && rows && row_base && qmask && v_means2d_abs && hyper_dev, "null pointer");
    ProjBwdArgs a;
    a.C = 1; a.N = N; a.K = K; a.colors_per_camera = 0; a.W = width; a.H = height;
    a.eps2d = eps2d; a.near_p = near_plane; a.far_p = far_plane;
    a.means = params + offsets_host[0]; a.scales = params + offsets_host[1]; a.quats = params + offsets_host[2];
    a.colors_in = params + offsets_host[3]; a.sh_rest = K > 1 ? params + offsets_host[4] : nullptr;
    a.opacities = params + offsets_host[5]; a.activations = 1;
    GS_REQUIRE(((uintptr_t)a.quats & 15) == 0, "the quaternion tensor must be 16-byte aligned");
    a.viewmats = viewmats; a.Ks = Ks; a.colors_post = colors_post; a.radii = radii; a.tiles_per_gauss = tiles_per_gauss;
    a.cum_tiles = cum_tiles; a.rows = reinterpret_cast<const float4*>(rows); a.row_base = row_base; a.qmask = qmask;
    a.v_means = a.v_quats = a.v_scales = a.v_opacities = a.v_colors = a.v_sh_rest = nullptr;
    a.v_means2d_abs = v_means2d_abs; a.v_means2d = a.v_conics = a.v_colors_post = a.v_colors_pre = nullptr;
    a.sh_jac = reinterpret_cast<const float4*>(sh_jac);
    a.guard = current_guard().info;
    a.rblk = current_rounds().phase == 3 ? current_rounds().blk : nullptr;   // (depth rounds: the rows are two ranges)
    a.adam = 1; a.ad_hyper = hyper_dev; a.ad_applied = applied_dev; a.ad_b1 = beta1; a.ad_b2 = beta2; a.ad_eps = eps;
    GS_REQUIRE((max_radii == nullptr) == (grad_norm_accum == nullptr) && (max_radii == nullptr) == (counts == nullptr),
               "the three statistics buffers come together or not at all");
    a.st_max_radii = max_radii; a.st_grad_norm = grad_norm_accum; a.st_counts = counts;
    a.st_max_hw = (float)(width > height ? width : height);
    a.row_sums = nullptr; a.st_gn_out = a.st_cnt_out = nullptr;
    a.ad_pbase = params; a.ad_mbase = exp_avg; a.ad_vbase = exp_avg_sq;
    for (int t = 0; t < 6; ++t) a.ad_off[t] = offsets_host[t];
    a.cam = 0; a.accumulate = 0;
    dim3 grid((unsigned)((N + kProjThreads - 1) / kProjThreads));
    const size_t lds = proj_bwd_lds_bytes(K, sh_degree);
    hipStream_t st = (hipStream_t)stream;
    // (no row_sums form of the fused kernel: its instantiation kept a dead 36-byte stack object, and a kernel that declares
    //  scratch makes the runtime provision it -- gs_blend.hip, GS_FWD_TRAIN_WAVES_PER_EU)
#define GS_PB(D) hipLaunchKernelGGL((project_bwd_kernel<D, true, false>), grid, dim3(kProjThreads), lds, st, a)
    switch (sh_degree) {
        case 0: GS_PB(0); break;
        case 1: GS_PB(1); break;
        case 2: GS_PB(2); break;
        default: GS_PB(3); break;
    }
#undef GS_PB
    GS_LAUNCH_CHECK("project_bwd_kernel (fused Adam)");
    return GS_OK;
}
