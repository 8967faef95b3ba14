// gs_common.h -- internal helpers shared by the gfx950 translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/gs_raster.h"

namespace gs {

void set_error(const char* fmt, ...);

#define GS_HIP_CHECK(expr)                                                                   \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            gs::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,   \
                          __LINE__);                                                         \
            return GS_ERR_HIP;                                                               \
        }                                                                                    \
    } while (0)

#define GS_LAUNCH_CHECK(name)                                                                \
    do {                                                                                     \
        hipError_t _e = hipGetLastError();                                                   \
        if (_e != hipSuccess) {                                                              \
            gs::set_error("launch of %s failed: %s", name, hipGetErrorString(_e));           \
            return GS_ERR_HIP;                                                               \
        }                                                                                    \
    } while (0)

#define GS_REQUIRE(cond, msg)                                                                \
    do {                                                                                     \
        if (!(cond)) {                                                                       \
            gs::set_error("invalid argument: %s (%s)", msg, #cond);                          \
            return GS_ERR_ARG;                                                               \
        }                                                                                    \
    } while (0)

constexpr int kWave = 64;

// Step guard (gs_guard_set, include/gs_raster.h): a device int64[4] info block {I, n_buckets, max tile, flags} plus
// the capacities the caller sized its list buffers for.  While set (per host thread), gs_bin_count raises
// flags when a capacity is exceeded and every kernel that walks or fills the lists -- and the kernels that
// apply the step (statistics, Adam) -- returns at once when flags != 0.
struct Guard {
    const int64_t* info;   // device pointer, nullptr = unguarded
    int64_t cap_isects;    // capacity of the intersection-indexed buffers
    int64_t cap_tile;      // longest tile list the sort classes launched can take
    int per_call;          // gs_guard_set_call: the flags belong to ONE call -- its first flag writer overwrites info[3]
};
Guard current_guard();
// Host mirror of the info block (gs_info_mirror_set): page-locked, device-visible host memory the tile scan writes the
// eight info words into directly -- the eager seam learns the list sizes without a device-to-host copy on the stream.
int64_t* current_info_mirror();
// Host mirror of a training forward's walk record (gs_walk_mirror_set): int64[4] {work units, storage units, rows, flags}.
int64_t* current_walk_mirror();
__device__ __forceinline__ bool guard_tripped(const int64_t* info) { return info != nullptr && info[3] != 0; }

// Depth rounds (gs_rounds_set, include/gs_raster.h): the list stages, the blend forward and the row gather of the backward run
// the frame's Gaussians in TWO rounds -- the front slab by depth first, the rest only into tiles the front slab has not
// finished.  `phase` 0: off; 1: front round; 2: back round; 3: behind both (the backward: rows of the two rounds are two ranges);
// 4: the front round alone (a tile it leaves live voids the step: GS_FLAG_BACK) -- the list stages see it as 1.
struct Rounds {
    int64_t* blk;        // device int64[GS_ROUND_WORDS]
    uint8_t* live;       // [tiles]  1: the front round left the tile with live pixels
    float4* state;       // [tiles][4][64]  pixel states of the live tiles, the front round's lanes' own
    int32_t* tile_rec;   // [tiles][8]  training: the quadrant sublists' lengths and part-filled work units
    int phase;
};
Rounds current_rounds();
// the back round has nothing to do: the front round left no live tile (its kernels return at once)
__device__ __forceinline__ bool round_idle(const int64_t* blk, int phase) { return phase == 2 && blk[GS_ROUND_LIVE] == 0; }

// Gradient rows in front of slot s (training: gs_blend_fwd's row-base scan).  The scan leaves ONE base per 16 slots -- a base per
// slot was 4 bytes written per LISTED intersection for the 2 % a saturated scene walks (65 us at 57 M) --; the reader adds the
// popcounts of the slot's predecessors inside its group: one aligned 16-byte load of their 4-bit quadrant masks.  *slot_mask (if
// asked for) = the slot's own mask.  s may be the slot one past the end (its group exists; bytes at or beyond s are not looked at).
__device__ __forceinline__ int rows_before(const int32_t* __restrict__ row_base16, const uint8_t* __restrict__ qmask, int s, int* slot_mask = nullptr) {
    const int g = s >> 4, k = s & 15;
    const uint4 q = reinterpret_cast<const uint4*>(qmask)[g];
    const uint32_t w[4] = {q.x, q.y, q.z, q.w};
    int n = row_base16[g];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int b = min(max(k - 4 * j, 0), 4);                        // bytes of word j in front of the slot
        const uint32_t m = b >= 4 ? 0xffffffffu : ((1u << (8 * b)) - 1u);
        n += __popc(w[j] & m & 0x0f0f0f0fu);
    }
    if (slot_mask) *slot_mask = (int)((w[k >> 2] >> (8 * (k & 3))) & 0xfu);
    return n;
}

// Workspace layout of the binning stage (all offsets in bytes, 256-B aligned).
struct BinLayout {
    int groups;          // Gaussian groups per camera
    int64_t per_group;   // Gaussians per group
    size_t hist_off;     // u32 [C*groups][tiles]   per-group tile histogram -> per-group tile base
    size_t tile_cnt_off; // u32 [C*tiles]
    size_t grp_tot_off;  // u32 [C*groups]          intersections emitted by each group
    size_t grp_base_off; // u32 [C*groups]          exclusive scan of the above
    size_t items_off;    // i32 work lists of the sort classes above 1024 keys (gs_binning.hip: class_items_kernel)
    size_t total;
};
BinLayout bin_layout(int C, int64_t N, int tiles);

// i / w for the small non-negative integers of a tile footprint (i < 2^20, w <= 2^12), w's reciprocal formed once per
// footprint: exact ((i + 0.5) / w is at least 0.5 / w away from an integer, far beyond fp32 rounding), and 3 instructions
// instead of the ~35 of a run-time integer division -- which was 20 % of the projection kernel's instruction count.
__device__ __forceinline__ int div_by_width(int i, float inv_w) { return (int)(((float)i + 0.5f) * inv_w); }

// ---- wavefront helpers (wave = 64 lanes on gfx950) ----
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

template <typename T>
__device__ __forceinline__ T wave_incl_scan_add(T v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        T o = __shfl_up(v, d, 64);
        if (lane_id() >= d) v += o;
    }
    return v;
}

template <typename T>
__device__ __forceinline__ T wave_reduce_add(T v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// Sum over the 64 lanes with DPP only (no LDS crossbar): two quad permutes and two mirrors leave every lane of a 16-lane row with
// the row's sum, row_bcast:15 / row_bcast:31 carry the rows along; lane 63 holds the total, handed to every lane as a scalar.
// Six full-rate v_add_f32_dpp per value instead of six ds_bpermute round trips (wave_reduce_add): for wave-wide sums on a
// latency-bound path (gs_rowsum_body.inc).  Fixed order -> reproducible.
__device__ __forceinline__ float wave_reduce_add_dpp(float v) {
#define GS_DPP_ADD(ctrl, row_mask) \
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, row_mask, 0xf, false))
    GS_DPP_ADD(0xB1, 0xf);    // quad_perm [1,0,3,2]
    GS_DPP_ADD(0x4E, 0xf);    // quad_perm [2,3,0,1]
    GS_DPP_ADD(0x141, 0xf);   // row_half_mirror
    GS_DPP_ADD(0x140, 0xf);   // row_mirror
    GS_DPP_ADD(0x142, 0xa);   // row_bcast:15 into rows 1 and 3
    GS_DPP_ADD(0x143, 0xc);   // row_bcast:31 into rows 2 and 3
#undef GS_DPP_ADD
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// Block-wide exclusive scan for blockDim.x <= 1024 (16 waves).  `scratch` >= 17 entries of T.
template <typename T>
__device__ __forceinline__ T block_excl_scan_add(T v, T* scratch, T* total) {
    const int lane = lane_id(), wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    T incl = wave_incl_scan_add(v);
    if (lane == 63) scratch[wave] = incl;
    __syncthreads();
    if (wave == 0) {
        T w = lane < nw ? scratch[lane] : T(0);
        T wi = wave_incl_scan_add(w);
        if (lane < nw) scratch[lane] = wi - w;
        if (lane == nw - 1) scratch[16] = wi;
    }
    __syncthreads();
    T res = incl - v + scratch[wave];
    *total = scratch[16];
    __syncthreads();
    return res;
}

// One Adam update, torch's `_single_tensor_adam` arithmetic (gs_adam.hip; also applied in place by the fused
// projection-backward + Adam kernel of gs_project.hip):  isbc2 = 1/sqrt(1-beta2^t),  ss = lr/(1-beta1^t).
__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, float b1, float b2, float eps,
                                      float isbc2, float ss) {
    m = fmaf(b1, m, (1.f - b1) * g);
    v = fmaf(b2, v, (1.f - b2) * g * g);
    const float denom = sqrtf(v) * isbc2 + eps;
    p = p - ss * (m / denom);
}

// streaming (non-temporal) 16-byte accesses for data that is touched once and then dead
#ifdef __HIPCC__
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load4(const float4* p) {
    const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void nt_store4(float4 x, float4* p) {
    f4v v; v.x = x.x; v.y = x.y; v.z = x.z; v.w = x.w;
    __builtin_nontemporal_store(v, reinterpret_cast<f4v*>(p));
}

#endif

}  // namespace gs
