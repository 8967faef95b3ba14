// gs_refine.hip -- device-side densify / prune ("next" row): the clone / split / prune
// decisions of /root/reference/model/gaussian.py:259-349 and the parameter + Adam-moment surgery of :199-257 as
// three launches around one prefix scan, with ONE host read (the new N, which sizes the new buffers):
//
//   refine_flags_kernel    per Gaussian: densification decision (average view-space gradient vs threshold; split if
//                          its largest scale is above the scale threshold, clone otherwise) and the prune decisions
//                          for the old Gaussian, for its split children and for its clone; tb_info counters
//   (host: inclusive scan of the three 0/1 flag rows -- any scan will do, the binding uses torch.cumsum -- and one
//    read of the three totals + five counters)
//   refine_map_kernel      scatters, per surviving output row, its source Gaussian and what it is (survivor,
//                          split child j, clone): the reference's order [old | split children, copy-major | clones]
//   refine_gather_kernel   one thread per OUTPUT element of the six parameters: gathers value and both moments from
//                          the old flat buffers into the new ones (coalesced stores; new Gaussians get zero moments;
//                          split children get mean + R(q) (s * noise) and log(s / (0.8 S)))
//
// The reference spells this as ~20 boolean-index / cat copies of every tensor and ~8 host syncs.
#include "gs_common.h"
#include "gs_math.h"

namespace gs {

struct RefineThresholds {
    float densify_grad, densify_scale, prune_radii, prune_scale, min_opacity;
    int num_splits;
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

__global__ __launch_bounds__(256) void refine_flags_kernel(int64_t n, RefineThresholds th, const float* __restrict__ grad_norm_accum,
                                                           const float* __restrict__ counts, const float* __restrict__ max_radii,
                                                           const float* __restrict__ log_scales, const float* __restrict__ logit_opac,
                                                           int32_t* __restrict__ flags /* [3][n] */, int64_t* __restrict__ counters /* [5] */) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int split = 0, clone = 0, c_low = 0, c_low_r = 0, c_low_r_s = 0;
    if (i < n) {
        float avg = grad_norm_accum[i] / (counts[i] + 1e-8f);
        if (avg != avg) avg = 0.f;
        const bool high = avg >= th.densify_grad;
        const float smax = fmaxf(fmaxf(expf(log_scales[3 * i]), expf(log_scales[3 * i + 1])), expf(log_scales[3 * i + 2]));
        const bool big = smax >= th.densify_scale;
        split = big && high; clone = !big && high;
        const bool low = sigmoidf_(logit_opac[i]) < th.min_opacity;
        const bool big_r = max_radii[i] > th.prune_radii;
        const bool big_s = smax > th.prune_scale;
        // children of a split: same opacity, max_radii 0, scales / (0.8 S); the clone: same opacity and scales, max_radii 0
        const bool child_big_s = expf(logf(smax / (0.8f * (float)th.num_splits))) > th.prune_scale;
        flags[i] = !(low || big_r || big_s || split);
        flags[n + i] = split && !(low || child_big_s);
        flags[2 * n + i] = clone && !(low || big_s);
        // the reference counts over [old | new] after the concatenation
        const int copies_split = split ? th.num_splits : 0;
        c_low = (low ? 1 : 0) * (1 + copies_split + clone);
        c_low_r = c_low + ((!low && big_r) ? 1 : 0);
        c_low_r_s = c_low_r + ((!low && !big_r && big_s) ? 1 : 0) + ((!low && child_big_s) ? copies_split : 0) + ((!low && big_s && clone) ? 1 : 0);
    }
    // block totals -> five device counters
    __shared__ int sm[5][4];
    int v[5] = {split, clone, c_low, c_low_r, c_low_r_s};
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int w = wave_reduce_add(v[k]);
        if (lane_id() == 0) sm[k][threadIdx.x >> 6] = w;
    }
    __syncthreads();
    if (threadIdx.x < 5) {
        const int tot = sm[threadIdx.x][0] + sm[threadIdx.x][1] + sm[threadIdx.x][2] + sm[threadIdx.x][3];
        if (tot) atomicAdd(reinterpret_cast<unsigned long long*>(counters + threadIdx.x), (unsigned long long)tot);
    }
}

// incl: inclusive scans of the three flag rows [3][n]; tot = {survivors, surviving children per copy, surviving clones}
__global__ __launch_bounds__(256) void refine_map_kernel(int64_t n, int num_splits, const int32_t* __restrict__ flags,
                                                         const int32_t* __restrict__ incl, int64_t tot_old, int64_t tot_child,
                                                         int32_t* __restrict__ src, int8_t* __restrict__ tag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (flags[i]) { const int64_t o = incl[i] - 1; src[o] = (int32_t)i; tag[o] = 0; }
    if (flags[n + i]) {
        const int64_t e = incl[n + i] - 1;
        for (int j = 0; j < num_splits; ++j) { const int64_t o = tot_old + (int64_t)j * tot_child + e; src[o] = (int32_t)i; tag[o] = (int8_t)(1 + j); }
    }
    if (flags[2 * n + i]) { const int64_t o = tot_old + (int64_t)num_splits * tot_child + incl[2 * n + i] - 1; src[o] = (int32_t)i; tag[o] = 127; }
}

struct RefineGatherArgs {
    int64_t n_old, n_new;
    int num_splits;
    const float *old_p, *old_m, *old_v;     // old flat buffers
    float *new_p, *new_m, *new_v;           // new flat buffers
    int64_t off_old[6], off_new[6];         // first float of each parameter tensor in the flat buffers
    int width[6];                           // floats per Gaussian: 3, 3, 4, 3, 3(K-1), 1
    const int32_t* src;
    const int8_t* tag;
    const float* noise;                     // [num_splits][n_old][3] standard normal samples
};

// tensor order of the reference's param_names: 0 means, 1 log_scales, 2 quats, 3 sh_0, 4 sh_rest, 5 logit_opacities
__global__ __launch_bounds__(256) void refine_gather_kernel(const RefineGatherArgs a) {
    const int t = blockIdx.y;
    const int w = a.width[t];
    const int64_t total = a.n_new * w;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = idx / w;
        const int col = (int)(idx - row * w);
        const int64_t r = a.src[row];
        const int tg = a.tag[row];
        const int64_t so = a.off_old[t] + r * w + col;
        float p = a.old_p[so], m = 0.f, v = 0.f;
        if (tg == 0) {
            m = a.old_m[so]; v = a.old_v[so];
        } else if (tg != 127) {   // split child number tg-1
            if (t == 0) {
                // mean + R(q / |q|) (exp(log s) * noise)   (/root/reference/model/gaussian.py:166-173)
                const float* q = a.old_p + a.off_old[2] + 4 * r;
                const float* ls = a.old_p + a.off_old[1] + 3 * r;
                const float* nz = a.noise + ((int64_t)(tg - 1) * a.n_old + r) * 3;
                float qw = q[0], qx = q[1], qy = q[2], qz = q[3];
                const float inv = 1.f / fmaxf(sqrtf(qw * qw + qx * qx + qy * qy + qz * qz), 1e-12f);
                qw *= inv; qx *= inv; qy *= inv; qz *= inv;
                const float s0 = expf(ls[0]) * nz[0], s1 = expf(ls[1]) * nz[1], s2 = expf(ls[2]) * nz[2];
                float r0, r1, r2;
                if (col == 0) { r0 = 1.f - 2.f * (qy * qy + qz * qz); r1 = 2.f * (qx * qy - qw * qz); r2 = 2.f * (qx * qz + qw * qy); }
                else if (col == 1) { r0 = 2.f * (qx * qy + qw * qz); r1 = 1.f - 2.f * (qx * qx + qz * qz); r2 = 2.f * (qy * qz - qw * qx); }
                else { r0 = 2.f * (qx * qz - qw * qy); r1 = 2.f * (qy * qz + qw * qx); r2 = 1.f - 2.f * (qx * qx + qy * qy); }
                p += r0 * s0 + r1 * s1 + r2 * s2;
            } else if (t == 1) {
                p = logf(expf(p) / (0.8f * (float)a.num_splits));
            }
        }
        const int64_t d = a.off_new[t] + idx;
        a.new_p[d] = p; a.new_m[d] = m; a.new_v[d] = v;
    }
}

}  // namespace gs

using namespace gs;

// ------------------------------------------------------------------------------------------------
// Inclusive prefix scan of each row of an int32 [rows][n] array (the three flag rows): per-block sums, one block per
// row over the block sums, per-block rescan + base.  (torch.cumsum takes 2.25 ms for 3 x 1 M int32 here -- it was two
// thirds of the whole refinement; this takes ~15 us.)
namespace gs {
constexpr int kScanThreads = 1024, kScanPer = 16, kScanChunk = kScanThreads * kScanPer;

__global__ __launch_bounds__(kScanThreads) void scan_partial_kernel(int64_t n, const int32_t* __restrict__ in, int32_t* __restrict__ block_sums) {
    __shared__ int32_t red[16];
    const int32_t* row = in + (size_t)blockIdx.y * n;
    const int64_t first = (int64_t)blockIdx.x * kScanChunk + (int64_t)threadIdx.x * kScanPer;
    int32_t s = 0;
#pragma unroll
    for (int k = 0; k < kScanPer; ++k) s += first + k < n ? row[first + k] : 0;
    s = wave_reduce_add(s);
    if (lane_id() == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int32_t t = 0;
        for (int i = 0; i < kScanThreads / 64; ++i) t += red[i];
        block_sums[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(kScanThreads) void scan_blocksums_kernel(int nb, int32_t* __restrict__ block_sums) {
    __shared__ int32_t scratch[17];
    int32_t* row = block_sums + (size_t)blockIdx.x * nb;
    int32_t carry = 0;
    for (int base = 0; base < nb; base += kScanThreads) {
        const int i = base + threadIdx.x;
        const int32_t v = i < nb ? row[i] : 0;
        int32_t total;
        const int32_t ex = block_excl_scan_add(v, scratch, &total);
        if (i < nb) row[i] = carry + ex;
        carry += total;
    }
}

__global__ __launch_bounds__(kScanThreads) void scan_final_kernel(int64_t n, const int32_t* __restrict__ in, const int32_t* __restrict__ block_sums,
                                                                  int32_t* __restrict__ out) {
    __shared__ int32_t scratch[17];
    const int32_t* row = in + (size_t)blockIdx.y * n;
    int32_t* orow = out + (size_t)blockIdx.y * n;
    const int64_t first = (int64_t)blockIdx.x * kScanChunk + (int64_t)threadIdx.x * kScanPer;
    int32_t v[kScanPer], s = 0;
#pragma unroll
    for (int k = 0; k < kScanPer; ++k) { v[k] = first + k < n ? row[first + k] : 0; s += v[k]; }
    int32_t total;
    int32_t run = block_sums[(size_t)blockIdx.y * gridDim.x + blockIdx.x] + block_excl_scan_add(s, scratch, &total);
#pragma unroll
    for (int k = 0; k < kScanPer; ++k) {
        run += v[k];
        if (first + k < n) orow[first + k] = run;
    }
}
}  // namespace gs

extern "C" size_t gs_scan_rows_workspace_ints(int rows, int64_t n) {
    return (size_t)rows * (size_t)((n + gs::kScanChunk - 1) / gs::kScanChunk) + 1;
}

extern "C" int gs_scan_rows_i32(void* stream, int rows, int64_t n, const int32_t* in, int32_t* out, int32_t* workspace) {
    GS_REQUIRE(rows >= 1 && n >= 0, "rows >= 1, n >= 0");
    if (n == 0) return GS_OK;
    GS_REQUIRE(in && out && workspace, "null pointer");
    const int64_t nb = (n + gs::kScanChunk - 1) / gs::kScanChunk;
    GS_REQUIRE(nb < (1ll << 31), "row too long");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gs::scan_partial_kernel, dim3((unsigned)nb, rows), dim3(gs::kScanThreads), 0, st, n, in, workspace);
    GS_LAUNCH_CHECK("scan_partial_kernel");
    hipLaunchKernelGGL(gs::scan_blocksums_kernel, dim3(rows), dim3(gs::kScanThreads), 0, st, (int)nb, workspace);
    GS_LAUNCH_CHECK("scan_blocksums_kernel");
    hipLaunchKernelGGL(gs::scan_final_kernel, dim3((unsigned)nb, rows), dim3(gs::kScanThreads), 0, st, n, in, (const int32_t*)workspace, out);
    GS_LAUNCH_CHECK("scan_final_kernel");
    return GS_OK;
}

extern "C" int gs_refine_flags(void* stream, int64_t n, int num_splits, float densify_grad_thresh, float densify_scale_thresh,
                               float prune_radii_ratio_thresh, float prune_scale_thresh, float min_opacity,
                               const float* grad_norm_accum, const float* counts, const float* max_radii,
                               const float* log_scales, const float* logit_opacities, int32_t* flags, int64_t* counters) {
    GS_REQUIRE(n >= 0 && num_splits >= 1 && num_splits <= 100, "n >= 0, 1 <= num_splits <= 100");
    GS_REQUIRE(counters != nullptr, "null counters");
    hipStream_t st = (hipStream_t)stream;
    GS_HIP_CHECK(hipMemsetAsync(counters, 0, 5 * sizeof(int64_t), st));
    if (n == 0) return GS_OK;
    GS_REQUIRE(grad_norm_accum && counts && max_radii && log_scales && logit_opacities && flags, "null pointer");
    RefineThresholds th;
    th.densify_grad = densify_grad_thresh; th.densify_scale = densify_scale_thresh; th.prune_radii = prune_radii_ratio_thresh;
    th.prune_scale = prune_scale_thresh; th.min_opacity = min_opacity; th.num_splits = num_splits;
    hipLaunchKernelGGL(refine_flags_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, th, grad_norm_accum, counts, max_radii,
                       log_scales, logit_opacities, flags, counters);
    GS_LAUNCH_CHECK("refine_flags_kernel");
    return GS_OK;
}

extern "C" int gs_refine_apply(void* stream, int64_t n_old, int num_splits, int K, const int32_t* flags, const int32_t* flags_incl,
                               int64_t tot_old, int64_t tot_child, int64_t tot_clone, const float* noise,
                               const float* old_params, const float* old_exp_avg, const float* old_exp_avg_sq,
                               const int64_t* old_offsets_host, float* new_params, float* new_exp_avg, float* new_exp_avg_sq,
                               const int64_t* new_offsets_host, int32_t* src_scratch, int8_t* tag_scratch) {
    GS_REQUIRE(n_old >= 0 && num_splits >= 1 && K >= 1 && K <= 16, "n_old >= 0, num_splits >= 1, 1 <= K <= 16");
    GS_REQUIRE(tot_old >= 0 && tot_child >= 0 && tot_clone >= 0, "negative totals");
    const int64_t n_new = tot_old + (int64_t)num_splits * tot_child + tot_clone;
    GS_REQUIRE(n_new < (1ll << 31) && n_old < (1ll << 31), "Gaussian counts must fit int32");
    if (n_new == 0 || n_old == 0) return GS_OK;
    GS_REQUIRE(flags && flags_incl && old_params && old_exp_avg && old_exp_avg_sq && new_params && new_exp_avg && new_exp_avg_sq &&
               old_offsets_host && new_offsets_host && src_scratch && tag_scratch, "null pointer");
    GS_REQUIRE(tot_child == 0 || noise != nullptr, "split children need noise[num_splits][n_old][3]");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(refine_map_kernel, dim3((unsigned)((n_old + 255) / 256)), dim3(256), 0, st, n_old, num_splits, flags, flags_incl,
                       tot_old, tot_child, src_scratch, tag_scratch);
    GS_LAUNCH_CHECK("refine_map_kernel");
    RefineGatherArgs a;
    a.n_old = n_old; a.n_new = n_new; a.num_splits = num_splits;
    a.old_p = old_params; a.old_m = old_exp_avg; a.old_v = old_exp_avg_sq;
    a.new_p = new_params; a.new_m = new_exp_avg; a.new_v = new_exp_avg_sq;
    const int widths[6] = {3, 3, 4, 3, 3 * (K - 1), 1};
    for (int t = 0; t < 6; ++t) { a.off_old[t] = old_offsets_host[t]; a.off_new[t] = new_offsets_host[t]; a.width[t] = widths[t]; }
    a.src = src_scratch; a.tag = tag_scratch; a.noise = noise;
    const int64_t biggest = n_new * (int64_t)(K > 1 ? 3 * (K - 1) : 4);
    const int64_t want = (biggest + 255) / 256;
    const unsigned gx = (unsigned)(want < 65536 ? (want > 0 ? want : 1) : 65536);
    hipLaunchKernelGGL(refine_gather_kernel, dim3(gx, 6), dim3(256), 0, st, a);
    GS_LAUNCH_CHECK("refine_gather_kernel");
    return GS_OK;
}
