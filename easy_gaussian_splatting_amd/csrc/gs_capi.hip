// gs_capi.hip -- library identity and the thread-local error channel of the C ABI
// (include/gs_raster.h).  The stage entry points live next to their kernels.
#include <stdarg.h>

#include "gs_common.h"

namespace gs {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
static thread_local Guard g_guard = {nullptr, 0, 0, 0};
Guard current_guard() { return g_guard; }
static thread_local int64_t* g_info_mirror = nullptr;
int64_t* current_info_mirror() { return g_info_mirror; }
static thread_local int64_t* g_walk_mirror = nullptr;
int64_t* current_walk_mirror() { return g_walk_mirror; }
static thread_local Rounds g_rounds = {nullptr, nullptr, nullptr, nullptr, 0};
Rounds current_rounds() { return g_rounds; }
}  // namespace gs

extern "C" int gs_rounds_set(int64_t* rounds_dev, uint8_t* tile_live, float* tile_state, int32_t* tile_rec, int phase) {
    if (phase < 0 || phase > 4 || (phase != 0 && !rounds_dev) || ((phase == 1 || phase == 2 || phase == 4) && (!tile_live || !tile_state))) {
        gs::set_error("invalid argument: phase 0 (off), 1 (front round), 2 (back round), 3 (behind both) or 4 (front round alone); a round needs its buffers");
        return GS_ERR_ARG;
    }
    if ((((uintptr_t)tile_state) & 15) != 0 || (((uintptr_t)tile_rec) & 15) != 0) {
        gs::set_error("invalid argument: tile_state / tile_rec 16-byte aligned");
        return GS_ERR_ARG;
    }
    gs::g_rounds.blk = phase ? rounds_dev : nullptr;
    gs::g_rounds.live = phase ? tile_live : nullptr;
    gs::g_rounds.state = phase ? reinterpret_cast<float4*>(tile_state) : nullptr;
    gs::g_rounds.tile_rec = phase ? tile_rec : nullptr;
    gs::g_rounds.phase = phase;
    return GS_OK;
}

extern "C" int gs_info_mirror_set(int64_t* info_host_mapped) {
    gs::g_info_mirror = info_host_mapped;
    return GS_OK;
}

extern "C" int gs_walk_mirror_set(int64_t* walk_host_mapped) {
    gs::g_walk_mirror = walk_host_mapped;
    return GS_OK;
}

extern "C" int gs_guard_set(const int64_t* info_dev, int64_t cap_isects, int64_t cap_tile) {
    if (info_dev != nullptr && (cap_isects <= 0 || cap_tile <= 0)) {
        gs::set_error("invalid argument: a guard needs positive capacities");
        return GS_ERR_ARG;
    }
    gs::g_guard.info = info_dev;
    gs::g_guard.cap_isects = info_dev ? cap_isects : 0;
    gs::g_guard.cap_tile = info_dev ? cap_tile : 0;
    gs::g_guard.per_call = 0;
    return GS_OK;
}

extern "C" int gs_guard_set_call(const int64_t* info_dev, int64_t cap_isects, int64_t cap_tile) {
    if (int rc = gs_guard_set(info_dev, cap_isects, cap_tile)) return rc;
    gs::g_guard.per_call = info_dev != nullptr;
    return GS_OK;
}

namespace gs {
__global__ void step_status_kernel(const int64_t* __restrict__ info, const int64_t* __restrict__ applied,
                                   volatile int64_t* __restrict__ status, const float* __restrict__ loss3,
                                   float* __restrict__ loss_ring, int ring_len, const int32_t* __restrict__ walk,
                                   const int64_t* __restrict__ rblk) {
    if (threadIdx.x < 4) status[threadIdx.x] = info[threadIdx.x];
    if (threadIdx.x == 7) status[7] = rblk ? rblk[GS_ROUND_LIVE] : -1;   // depth rounds: tiles the front round left live (-1: one round)
    if (threadIdx.x == 4) status[4] = applied ? applied[0] : 0;
    // what the walk of this step needed (a step the guard skipped before its blend leaves the cleared words of the step before)
    if (threadIdx.x == 5) status[5] = walk ? walk[GS_WALK_STORAGE] : 0;
    if (threadIdx.x == 6) status[6] = walk ? walk[GS_WALK_ROWS] : 0;
    // per-step loss log: slot (applied - 1) mod ring_len, written only by steps the guard did not skip, so a replayed
    // step overwrites nothing but its own slot
    if (loss_ring != nullptr && applied != nullptr && info[3] == 0 && threadIdx.x < 3 && applied[0] > 0)
        loss_ring[3 * ((applied[0] - 1) % ring_len) + threadIdx.x] = loss3[threadIdx.x];
    __threadfence_system();
}
}  // namespace gs

extern "C" int gs_step_status(void* stream, const int64_t* info_dev, const int64_t* applied_dev, int64_t* status,
                              const float* loss3_dev, float* loss_ring_dev, int ring_len, const int32_t* walk_state) {
    if (!info_dev || !status || (loss_ring_dev && (!loss3_dev || ring_len <= 0))) {
        gs::set_error("invalid argument: null pointer / empty ring");
        return GS_ERR_ARG;
    }
    const gs::Rounds R = gs::current_rounds();
    hipLaunchKernelGGL(gs::step_status_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, info_dev, applied_dev, status, loss3_dev,
                       loss_ring_dev, ring_len, walk_state, (const int64_t*)(R.phase == 3 ? R.blk : nullptr));
    GS_LAUNCH_CHECK("step_status_kernel");
    return GS_OK;
}

// ---- the step guard across ranks (view-parallel captured step: one view per rank, two collectives per step).  A rank whose
// lists or walk outgrew its capacities must not be the only one that skips the step: its flag travels inside both
// collectives (gs_guard_flag_out writes it where the record / the bucket carry it), and every rank ORs what arrives into
// its own guard (gs_guard_merge) in front of the kernels that apply the step -- all replicas skip, or none.
namespace gs {
__global__ void guard_flag_out_kernel(const int64_t* __restrict__ info, float* __restrict__ dst0, float* __restrict__ dst1) {
    const float f = info[3] != 0 ? 1.f : 0.f;
    if (dst0) dst0[0] = f;
    if (dst1) dst1[0] = f;
}
__global__ void guard_merge_kernel(int64_t* __restrict__ info, const float* __restrict__ src, int n, int64_t stride) {
    bool any = false;
    for (int i = threadIdx.x; i < n; i += 64) any |= src[(int64_t)i * stride] != 0.f;
    if (__ballot(any) != 0ull && threadIdx.x == 0) info[3] |= GS_FLAG_PEER;
}
__global__ void step_applied_kernel(const int64_t* __restrict__ info, int64_t* __restrict__ applied) {
    if (info[3] == 0) applied[0] += 1;
}
}  // namespace gs

extern "C" int gs_guard_flag_out(void* stream, const int64_t* info_dev, float* dst0, float* dst1) {
    if (!info_dev || (!dst0 && !dst1)) { gs::set_error("invalid argument: null pointer"); return GS_ERR_ARG; }
    hipLaunchKernelGGL(gs::guard_flag_out_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, info_dev, dst0, dst1);
    GS_LAUNCH_CHECK("guard_flag_out_kernel");
    return GS_OK;
}

extern "C" int gs_guard_merge(void* stream, int64_t* info_dev, const float* flags, int n, int64_t stride) {
    if (!info_dev || !flags || n < 1) { gs::set_error("invalid argument: null pointer / n < 1"); return GS_ERR_ARG; }
    hipLaunchKernelGGL(gs::guard_merge_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, info_dev, flags, n, stride);
    GS_LAUNCH_CHECK("guard_merge_kernel");
    return GS_OK;
}

extern "C" int gs_step_applied(void* stream, const int64_t* info_dev, int64_t* applied_dev) {
    if (!info_dev || !applied_dev) { gs::set_error("invalid argument: null pointer"); return GS_ERR_ARG; }
    hipLaunchKernelGGL(gs::step_applied_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, info_dev, applied_dev);
    GS_LAUNCH_CHECK("step_applied_kernel");
    return GS_OK;
}

extern "C" int gs_version(void) { return 300; }
// The preprocessor flags beyond the Makefile's own this library was built with ("" = the product build): a -DGS_BWD_CHECK,
// -DGS_BWD_ACC64, -DGS_EXACT_MATH or -DGS_CLOCK_PROBE diagnostic variant names itself, and the Python binding refuses to load
// one unless it is asked to (easy_gaussian_splatting_amd/_native.py).
#ifndef GS_BUILD_FLAGS
#define GS_BUILD_FLAGS ""
#endif
extern "C" const char* gs_build_flags(void) { return GS_BUILD_FLAGS; }
extern "C" const char* gs_last_error(void) { return gs::g_err; }
extern "C" const char* gs_arch(void) { return "gfx950"; }
