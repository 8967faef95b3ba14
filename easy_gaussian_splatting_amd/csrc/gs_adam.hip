// gs_adam.hip -- fused multi-group Adam step for gfx950 ("next" row):
// the reference drives ONE torch.optim.Adam with six named parameter groups
// (/root/reference/model/gaussian.py:389-412, default betas/eps, no weight decay, no amsgrad).
// Parameters and both moments live in three flat fp32 buffers (segments padded to 16-byte quads);
// the gradients stay where autograd put them (one tensor per segment, no accumulate-into-bucket
// pass, no zero-fill pass).  The whole step is a single HBM-streaming kernel: 16 B/lane loads of
// p, g, m, v and stores of p, m, v -- 28 B per element.  Same arithmetic as torch's
// `_single_tensor_adam`:  denom = sqrt(v)/sqrt(1-beta2^t) + eps;  p -= (lr/(1-beta1^t)) * m/denom.
// A segment whose gradient pointer is NULL is skipped entirely (torch skips `p.grad is None`);
// the view-parallel step uses that to update the SH segments and the geometry segments in two
// launches of the same step (distributed.py).  `grad_scale` folds the 1/world of a mean over ranks in.
#include "gs_common.h"

namespace gs {

constexpr int kMaxSeg = 8;
struct AdamArgs {
    int64_t n4;                   // number of float4 quads in the flat buffers
    float4 *p, *m, *v;
    int nseg;
    int64_t seg_begin4[kMaxSeg];  // first quad of each segment
    int64_t seg_end4[kMaxSeg];    // exclusive end, in quads
    int64_t seg_len[kMaxSeg];     // un-padded element count (length of the gradient tensor)
    const float* g[kMaxSeg];      // gradient tensor of each segment (may be null)
    float step_size[kMaxSeg];     // lr / (1 - beta1^t)
    float beta1, beta2, eps, inv_sqrt_bc2;
    float grad_scale;             // gradients are multiplied by this first (1/world for a mean over ranks)
    const float* hyper;           // optional device array {1/sqrt(1-beta2^t), step_size[0..nseg-1]} (gs_adam_hyper): replaces
                                  // inv_sqrt_bc2 / step_size above, so a captured launch can be replayed with a new step count
    const int64_t* guard;         // step guard (gs_guard_set) or nullptr
    int64_t* applied;             // optional device counter of steps that were really applied (not skipped by the guard)
    // Only the segments WITH a gradient are walked (a partial step -- the view-parallel step's geometry half is 11 of 59 floats
    // per Gaussian -- does not pay for the quads it skips): act_end4[k] = quads of the active segments up to and including k.
    int64_t act_end4[kMaxSeg];
    int64_t n_act4;
    // ... and two additive statistics ride along (optional; the view-parallel step's all-reduced |absgrad| and visibility count,
    // /root/reference/model/gaussian.py:188-197): stat_dst[j][0..stat_n) += stat_src[j][0..stat_n), one launch instead of three
    const float* stat_src[2];
    float* stat_dst[2];
    int64_t stat_n;
};

__global__ __launch_bounds__(256) void adam_step_kernel(const AdamArgs a) {
    if (guard_tripped(a.guard)) return;
    if (a.applied != nullptr && blockIdx.x == 0 && threadIdx.x == 0) a.applied[0] += 1;
    const float isbc2 = a.hyper ? a.hyper[0] : a.inv_sqrt_bc2;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < a.n_act4; j += (int64_t)gridDim.x * blockDim.x) {
        int s = 0;
#pragma unroll
        for (int k = 0; k < kMaxSeg - 1; ++k) s += (k < a.nseg - 1 && j >= a.act_end4[k]) ? 1 : 0;
        const float* gp = a.g[s];
        if (gp == nullptr) continue;   // (cannot happen: inactive segments hold no quads of j's range)
        const int64_t q = j - (s > 0 ? a.act_end4[s - 1] : 0);   // quad inside the segment
        const int64_t i = a.seg_begin4[s] + q;                   // quad inside the flat buffers
        const int64_t e = q << 2;                                // element index inside the segment
        const int64_t len = a.seg_len[s];
        float4 g;
        if (e + 4 <= len) {
            g = nt_load4(reinterpret_cast<const float4*>(gp + e));   // read once, then dead
        } else {  // the segment's padded tail
            g.x = e < len ? gp[e] : 0.f; g.y = e + 1 < len ? gp[e + 1] : 0.f;
            g.z = e + 2 < len ? gp[e + 2] : 0.f; g.w = 0.f;
        }
        g.x *= a.grad_scale; g.y *= a.grad_scale; g.z *= a.grad_scale; g.w *= a.grad_scale;
        const float ss = a.hyper ? a.hyper[1 + s] : a.step_size[s];
        float4 p = a.p[i], m = nt_load4(a.m + i), v = nt_load4(a.v + i);
        adam1(p.x, g.x, m.x, v.x, a.beta1, a.beta2, a.eps, isbc2, ss);
        adam1(p.y, g.y, m.y, v.y, a.beta1, a.beta2, a.eps, isbc2, ss);
        adam1(p.z, g.z, m.z, v.z, a.beta1, a.beta2, a.eps, isbc2, ss);
        adam1(p.w, g.w, m.w, v.w, a.beta1, a.beta2, a.eps, isbc2, ss);
        a.p[i] = p;   // re-read by the next forward
        nt_store4(m, a.m + i); nt_store4(v, a.v + i);   // streamed once per step
    }
    if (a.stat_n > 0) {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.stat_n; i += (int64_t)gridDim.x * blockDim.x) {
            a.stat_dst[0][i] += a.stat_src[0][i];
            a.stat_dst[1][i] += a.stat_src[1][i];
        }
    }
}

}  // namespace gs

using namespace gs;

static int adam_launch(void* stream, int64_t n, float* params, float* exp_avg, float* exp_avg_sq,
                       int n_segments, const int64_t* seg_ends_host, const int64_t* seg_lens_host,
                       const float* const* seg_grads_host, const float* seg_lrs_host, float beta1,
                       float beta2, float eps, int64_t step, float grad_scale, const float* hyper_dev, int64_t* applied_dev,
                       int64_t stat_n = 0, const float* stat_src0 = nullptr, const float* stat_src1 = nullptr,
                       float* stat_dst0 = nullptr, float* stat_dst1 = nullptr) {
    GS_REQUIRE(n >= 0 && (n & 3) == 0, "flat length must be a multiple of 4 (pad the buffers)");
    GS_REQUIRE(n_segments >= 1 && n_segments <= kMaxSeg, "1..8 segments");
    GS_REQUIRE(step >= 1, "step counts from 1");
    if (n == 0) return GS_OK;
    GS_REQUIRE(params && exp_avg && exp_avg_sq && seg_ends_host && seg_lens_host && seg_grads_host && seg_lrs_host, "null pointer");
    AdamArgs a;
    a.hyper = hyper_dev; a.applied = applied_dev; a.guard = current_guard().info;
    a.n4 = n >> 2;
    a.p = reinterpret_cast<float4*>(params);
    a.m = reinterpret_cast<float4*>(exp_avg); a.v = reinterpret_cast<float4*>(exp_avg_sq);
    a.nseg = n_segments;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    for (int k = 0; k < kMaxSeg; ++k) { a.seg_begin4[k] = a.n4; a.seg_end4[k] = a.n4; a.seg_len[k] = 0; a.g[k] = nullptr; a.step_size[k] = 0.f; }
    int64_t begin = 0;
    for (int k = 0; k < n_segments; ++k) {
        GS_REQUIRE((seg_ends_host[k] & 3) == 0 && seg_ends_host[k] >= begin && seg_ends_host[k] <= n, "segment ends must be ascending multiples of 4 within n");
        GS_REQUIRE(seg_lens_host[k] >= 0 && seg_lens_host[k] <= seg_ends_host[k] - begin, "segment length exceeds its padded extent");
        GS_REQUIRE(((uintptr_t)seg_grads_host[k] & 15) == 0, "gradient tensors must be 16-byte aligned");
        a.seg_begin4[k] = begin >> 2;
        a.seg_end4[k] = seg_ends_host[k] >> 2;
        a.seg_len[k] = seg_lens_host[k];
        a.g[k] = seg_grads_host[k];
        a.step_size[k] = (float)((double)seg_lrs_host[k] / bc1);
        begin = seg_ends_host[k];
    }
    a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    a.grad_scale = grad_scale;
    int64_t act = 0;
    for (int k = 0; k < kMaxSeg; ++k) {
        if (k < n_segments && a.g[k] != nullptr) act += a.seg_end4[k] - a.seg_begin4[k];
        a.act_end4[k] = act;
    }
    a.n_act4 = act;
    GS_REQUIRE(stat_n >= 0 && (stat_n == 0 || (stat_src0 && stat_src1 && stat_dst0 && stat_dst1)), "statistics: four pointers or none");
    a.stat_n = stat_n; a.stat_src[0] = stat_src0; a.stat_src[1] = stat_src1; a.stat_dst[0] = stat_dst0; a.stat_dst[1] = stat_dst1;
    const int64_t work = a.n_act4 > stat_n ? a.n_act4 : stat_n;
    if (work == 0) return GS_OK;
    const int64_t want = (work + 255) / 256;
    const unsigned grid = (unsigned)(want < 256 * 16 ? want : 256 * 16);
    hipLaunchKernelGGL(adam_step_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    GS_LAUNCH_CHECK("adam_step_kernel");
    return GS_OK;
}

extern "C" int gs_adam_step(void* stream, int64_t n, float* params, float* exp_avg, float* exp_avg_sq,
                            int n_segments, const int64_t* seg_ends_host, const int64_t* seg_lens_host,
                            const float* const* seg_grads_host, const float* seg_lrs_host, float beta1,
                            float beta2, float eps, int64_t step, float grad_scale) {
    return adam_launch(stream, n, params, exp_avg, exp_avg_sq, n_segments, seg_ends_host, seg_lens_host, seg_grads_host,
                       seg_lrs_host, beta1, beta2, eps, step, grad_scale, nullptr, nullptr);
}

extern "C" int gs_adam_step_stats(void* stream, int64_t n, float* params, float* exp_avg, float* exp_avg_sq,
                                  int n_segments, const int64_t* seg_ends_host, const int64_t* seg_lens_host,
                                  const float* const* seg_grads_host, const float* seg_lrs_host, float beta1,
                                  float beta2, float eps, int64_t step, float grad_scale, int64_t stat_n,
                                  const float* stat_src0, const float* stat_src1, float* stat_dst0, float* stat_dst1) {
    return adam_launch(stream, n, params, exp_avg, exp_avg_sq, n_segments, seg_ends_host, seg_lens_host, seg_grads_host,
                       seg_lrs_host, beta1, beta2, eps, step, grad_scale, nullptr, nullptr, stat_n, stat_src0, stat_src1, stat_dst0, stat_dst1);
}

extern "C" int gs_adam_step_dev(void* stream, int64_t n, float* params, float* exp_avg, float* exp_avg_sq,
                                int n_segments, const int64_t* seg_ends_host, const int64_t* seg_lens_host,
                                const float* const* seg_grads_host, float beta1, float beta2, float eps, float grad_scale,
                                const float* hyper_dev, int64_t* applied_dev) {
    GS_REQUIRE(hyper_dev != nullptr, "hyper_dev (gs_adam_hyper) is required");
    const float zero_lrs[kMaxSeg] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    return adam_launch(stream, n, params, exp_avg, exp_avg_sq, n_segments, seg_ends_host, seg_lens_host, seg_grads_host,
                       zero_lrs, beta1, beta2, eps, 1, grad_scale, hyper_dev, applied_dev);
}

namespace gs {
struct HyperArgs { float v[1 + kMaxSeg]; int n; };
__global__ void adam_hyper_kernel(const HyperArgs h, float* __restrict__ out) {
    if (threadIdx.x < (unsigned)h.n) out[threadIdx.x] = h.v[threadIdx.x];
}
}  // namespace gs

extern "C" int gs_adam_hyper(void* stream, int n_segments, const float* seg_lrs_host, float beta1, float beta2,
                             int64_t step, float* hyper_dev) {
    GS_REQUIRE(n_segments >= 1 && n_segments <= kMaxSeg && seg_lrs_host && hyper_dev, "1..8 segments, non-null pointers");
    GS_REQUIRE(step >= 1, "step counts from 1");
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    HyperArgs h;
    h.n = 1 + n_segments;
    h.v[0] = (float)(1.0 / sqrt(bc2));
    for (int k = 0; k < kMaxSeg; ++k) h.v[1 + k] = k < n_segments ? (float)((double)seg_lrs_host[k] / bc1) : 0.f;
    // values travel as kernel arguments (copied at launch): no host buffer that a later call could overwrite
    hipLaunchKernelGGL(gs::adam_hyper_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, h, hyper_dev);
    GS_LAUNCH_CHECK("adam_hyper_kernel");
    return GS_OK;
}

// Everything that changes from one captured step to the next, in ONE launch in front of the replay (gs_step_inputs): Adam's
// bias corrections and learning rates as gs_adam_hyper writes them, the camera (two device-to-device copies of 16 + 9 floats)
// and the POINTERS to the step's ground-truth image and mask, which the loss entries read through (gs_l1_ssim_fwd_slots):
// a step on another view used to cost three copy launches and 24 H W bytes of traffic for an image nobody changes.
namespace gs {
struct StepInputArgs {
    HyperArgs h;
    float* hyper;
    const float *vm_src, *k_src;
    float *vm_dst, *k_dst;
    const float *gt, *mask;
    const float** slots;
};
__global__ void step_inputs_kernel(const StepInputArgs a) {
    const unsigned t = threadIdx.x;
    if (t < (unsigned)a.h.n) a.hyper[t] = a.h.v[t];
    if (a.vm_src && t < 16u) a.vm_dst[t] = a.vm_src[t];
    if (a.k_src && t < 9u) a.k_dst[t] = a.k_src[t];
    if (a.slots && t == 0u) { a.slots[0] = a.gt; a.slots[1] = a.mask; }
}
}  // namespace gs

extern "C" int gs_step_inputs(void* stream, int n_segments, const float* seg_lrs_host, float beta1, float beta2, int64_t step,
                              float* hyper_dev, const float* viewmat_src, float* viewmat_dst, const float* K_src, float* K_dst,
                              const float* gt, const float* mask, const float** slots_dev) {
    GS_REQUIRE(n_segments >= 1 && n_segments <= kMaxSeg && seg_lrs_host && hyper_dev, "1..8 segments, non-null pointers");
    GS_REQUIRE(step >= 1, "step counts from 1");
    GS_REQUIRE((viewmat_src == nullptr || viewmat_dst != nullptr) && (K_src == nullptr || K_dst != nullptr), "a source needs its destination");
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    gs::StepInputArgs a;
    a.h.n = 1 + n_segments;
    a.h.v[0] = (float)(1.0 / sqrt(bc2));
    for (int k = 0; k < kMaxSeg; ++k) a.h.v[1 + k] = k < n_segments ? (float)((double)seg_lrs_host[k] / bc1) : 0.f;
    a.hyper = hyper_dev;
    a.vm_src = viewmat_src; a.vm_dst = viewmat_dst; a.k_src = K_src; a.k_dst = K_dst;
    a.gt = gt; a.mask = mask; a.slots = slots_dev;
    // values travel as kernel arguments (copied at launch): no host buffer that a later call could overwrite
    hipLaunchKernelGGL(gs::step_inputs_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a);
    GS_LAUNCH_CHECK("step_inputs_kernel");
    return GS_OK;
}

// ------------------------------------------------------------------------------------------------
// update_statistics (/root/reference/model/gaussian.py:188-197) as ONE launch: the consumer of the
// `.absgrad` / `radii` side channels.  For visible Gaussians (radius > 0):
//   max_radii = max(max_radii, radius / max_hw);  grad_norm_accum += |absgrad|_2 * max_hw;  counts += 1
// The reference spells this as ~10 boolean-index kernels.  Single camera (the reference's C = 1).
namespace gs {
__global__ __launch_bounds__(256) void update_statistics_kernel(int64_t n, float max_hw, const int32_t* __restrict__ radii,
                                                                const float2* __restrict__ absgrad, float* __restrict__ max_radii,
                                                                float* __restrict__ grad_norm_accum, float* __restrict__ counts,
                                                                const int64_t* __restrict__ guard) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || guard_tripped(guard)) return;
    const int r = radii[i];
    if (r > 0) {
        const float2 g = absgrad[i];
        // (radius * (1 / max_hw), not radius / max_hw: the reference's `radii / max_hw` is torch's division of a CUDA tensor by a
        //  Python scalar, which multiplies by the reciprocal -- one arithmetic on every path that forms this statistic)
        max_radii[i] = fmaxf(max_radii[i], (float)r * (1.f / max_hw));
        grad_norm_accum[i] += sqrtf(g.x * g.x + g.y * g.y) * max_hw;
        counts[i] += 1.f;
    }
}
}  // namespace gs

extern "C" int gs_update_statistics(void* stream, int64_t n, float max_hw, const int32_t* radii, const float* absgrad,
                                    float* max_radii, float* grad_norm_accum, float* counts) {
    GS_REQUIRE(n >= 0 && max_hw > 0.f, "n >= 0 and positive image extent");
    if (n == 0) return GS_OK;
    GS_REQUIRE(radii && absgrad && max_radii && grad_norm_accum && counts, "null pointer");
    hipLaunchKernelGGL(gs::update_statistics_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, max_hw,
                       radii, reinterpret_cast<const float2*>(absgrad), max_radii, grad_norm_accum, counts, gs::current_guard().info);
    GS_LAUNCH_CHECK("update_statistics_kernel");
    return GS_OK;
}

// ------------------------------------------------------------------------------------------------
// View-parallel step (distributed.ViewParallelStep): everything one rank contributes to the SUM
// all-reduce in ONE pass -- the four geometry gradients and this view's two additive statistics of
// /root/reference/model/gaussian.py:188-197 (|absgrad|_2 * max_hw and the visibility count), packed
// into a flat buffer whose six segments start at multiples of 4 floats: [means 3N | log_scales 3N |
// quats 4N | logit_opacities N | grad_norm N | count N].  Replaces ~12 elementwise launches.
namespace gs {
__global__ __launch_bounds__(256) void pack_view_step_kernel(int64_t n, float max_hw, const float* __restrict__ v_means,
                                                             const float* __restrict__ v_scales, const float* __restrict__ v_quats,
                                                             const float* __restrict__ v_opac, const int32_t* __restrict__ radii,
                                                             const float2* __restrict__ absgrad, float* __restrict__ flat,
                                                             int64_t o_scales, int64_t o_quats, int64_t o_opac, int64_t o_gn, int64_t o_cnt) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (v_means) {   // (NULL: the projection backward wrote its gradients into `flat` itself -- only the statistics are left)
        flat[3 * i] = v_means[3 * i]; flat[3 * i + 1] = v_means[3 * i + 1]; flat[3 * i + 2] = v_means[3 * i + 2];
        flat[o_scales + 3 * i] = v_scales[3 * i]; flat[o_scales + 3 * i + 1] = v_scales[3 * i + 1]; flat[o_scales + 3 * i + 2] = v_scales[3 * i + 2];
        reinterpret_cast<float4*>(flat + o_quats)[i] = reinterpret_cast<const float4*>(v_quats)[i];
        flat[o_opac + i] = v_opac[i];
    }
    const bool vis = radii[i] > 0;
    const float2 g = absgrad[i];
    flat[o_gn + i] = vis ? sqrtf(g.x * g.x + g.y * g.y) * max_hw : 0.f;
    flat[o_cnt + i] = vis ? 1.f : 0.f;
}
}  // namespace gs

extern "C" int gs_pack_view_step(void* stream, int64_t n, float max_hw, const float* v_means, const float* v_scales,
                                 const float* v_quats, const float* v_opacities, const int32_t* radii, const float* absgrad,
                                 float* flat) {
    GS_REQUIRE(n >= 0 && max_hw > 0.f, "n >= 0 and positive image extent");
    if (n == 0) return GS_OK;
    GS_REQUIRE(radii && absgrad && flat, "null pointer");
    GS_REQUIRE((v_means && v_scales && v_quats && v_opacities) || (!v_means && !v_scales && !v_quats && !v_opacities),
               "the four gradients: all (packed here) or none (already in place in flat)");
    GS_REQUIRE((((uintptr_t)v_quats | (uintptr_t)flat) & 15) == 0, "v_quats and flat must be 16-byte aligned");
    auto pad4 = [](int64_t x) { return (x + 3) / 4 * 4; };
    const int64_t o_scales = pad4(3 * n), o_quats = o_scales + pad4(3 * n), o_opac = o_quats + 4 * n, o_gn = o_opac + pad4(n),
                  o_cnt = o_gn + pad4(n);
    hipLaunchKernelGGL(gs::pack_view_step_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, max_hw,
                       v_means, v_scales, v_quats, v_opacities, radii, reinterpret_cast<const float2*>(absgrad), flat, o_scales, o_quats,
                       o_opac, o_gn, o_cnt);
    GS_LAUNCH_CHECK("pack_view_step_kernel");
    return GS_OK;
}
