// gs_adam.hip -- fused multi-group Adam step for gfx950 ("next" row):
// the reference drives ONE torch.optim.Adam with six named parameter groups
// (/root/reference/model/gaussian.py:389-412, default betas/eps, no weight decay, no amsgrad).
// With all parameters, gradients and both moments living in four flat fp32 buffers (the gradient
// one is the RCCL bucket of distributed.py) the whole step is a single HBM-streaming kernel:
// 16 B/lane loads of p, g, m, v; stores of p, m, v and -- optionally -- the zeroed gradient,
// 28-32 B per element instead of several launches per group.  Same arithmetic as torch's
// `_single_tensor_adam`:  denom = sqrt(v)/sqrt(1-beta2^t) + eps;  p -= (lr/(1-beta1^t)) * m/denom.
#include "gs_common.h"

namespace gs {

constexpr int kMaxSeg = 8;
struct AdamArgs {
    int64_t n4;                 // number of float4 quads
    float4 *p, *g, *m, *v;
    int nseg;
    int64_t seg_end4[kMaxSeg];  // exclusive end of each group, in quads
    float step_size[kMaxSeg];   // lr / (1 - beta1^t)
    float beta1, beta2, eps, inv_sqrt_bc2;
    int zero_grad;
};

__device__ __forceinline__ float adam1(float& p, float g, float& m, float& v, float b1, float b2, float eps,
                                       float isbc2, float ss) {
    m = fmaf(b1, m, (1.f - b1) * g);
    v = fmaf(b2, v, (1.f - b2) * g * g);
    const float denom = sqrtf(v) * isbc2 + eps;
    p = p - ss * (m / denom);
    return p;
}

__global__ __launch_bounds__(256) void adam_step_kernel(const AdamArgs a) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n4; i += (int64_t)gridDim.x * blockDim.x) {
        int s = 0;
#pragma unroll
        for (int k = 0; k < kMaxSeg - 1; ++k) s += (k < a.nseg - 1 && i >= a.seg_end4[k]) ? 1 : 0;
        const float ss = a.step_size[s];
        float4 p = a.p[i], m = a.m[i], v = a.v[i];
        const float4 g = a.g[i];
        adam1(p.x, g.x, m.x, v.x, a.beta1, a.beta2, a.eps, a.inv_sqrt_bc2, ss);
        adam1(p.y, g.y, m.y, v.y, a.beta1, a.beta2, a.eps, a.inv_sqrt_bc2, ss);
        adam1(p.z, g.z, m.z, v.z, a.beta1, a.beta2, a.eps, a.inv_sqrt_bc2, ss);
        adam1(p.w, g.w, m.w, v.w, a.beta1, a.beta2, a.eps, a.inv_sqrt_bc2, ss);
        a.p[i] = p; a.m[i] = m; a.v[i] = v;
        if (a.zero_grad) a.g[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

}  // namespace gs

using namespace gs;

extern "C" int gs_adam_step(void* stream, int64_t n, float* params, float* grads, float* exp_avg,
                            float* exp_avg_sq, int n_groups, const int64_t* group_ends_host,
                            const float* group_lrs_host, float beta1, float beta2, float eps, int64_t step,
                            int zero_grad) {
    GS_REQUIRE(n >= 0 && (n & 3) == 0, "flat length must be a multiple of 4 (pad the buffers)");
    GS_REQUIRE(n_groups >= 1 && n_groups <= kMaxSeg, "1..8 parameter groups");
    GS_REQUIRE(step >= 1, "step counts from 1");
    if (n == 0) return GS_OK;
    GS_REQUIRE(params && grads && exp_avg && exp_avg_sq && group_ends_host && group_lrs_host, "null pointer");
    AdamArgs a;
    a.n4 = n >> 2;
    a.p = reinterpret_cast<float4*>(params); a.g = reinterpret_cast<float4*>(grads);
    a.m = reinterpret_cast<float4*>(exp_avg); a.v = reinterpret_cast<float4*>(exp_avg_sq);
    a.nseg = n_groups;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    for (int k = 0; k < kMaxSeg; ++k) { a.seg_end4[k] = a.n4; a.step_size[k] = 0.f; }
    for (int k = 0; k < n_groups; ++k) {
        GS_REQUIRE((group_ends_host[k] & 3) == 0, "group boundaries must be multiples of 4 elements");
        a.seg_end4[k] = group_ends_host[k] >> 2;
        a.step_size[k] = (float)((double)group_lrs_host[k] / bc1);
    }
    a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    a.zero_grad = zero_grad;
    const int64_t want = (a.n4 + 255) / 256;
    const unsigned grid = (unsigned)(want < 256 * 16 ? want : 256 * 16);
    hipLaunchKernelGGL(adam_step_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    GS_LAUNCH_CHECK("adam_step_kernel");
    return GS_OK;
}
