// gs_rounds.hip -- depth rounds: the split of a frame's Gaussians at a depth quantile and the footprints of each round
// (include/gs_raster.h: "Depth rounds").  The list stages, the blend and the row gather read the round they work for from the
// per-thread context gs_rounds_set leaves (gs_common.h: Rounds); this file holds the two passes that exist only with rounds.
//
// Why rounds: the list stages price every LISTED intersection (count, emit, sort: 52 bytes each through the two-level binning),
// the blend reads the few per cent of them in front of each tile's saturation depth.  Nothing tells which those are before
// the blend has run -- except depth: a tile's walk is a prefix of its depth order, so the nearest slab of the frame holds the
// whole walk of every tile it saturates, and the lists of the rest are needed only where it does not.
#include <algorithm>

#include "gs_common.h"

namespace gs {

constexpr int kDepthBins = 4096;      // float bits >> 19: sign 0, 8 exponent bits, 4 mantissa bits -- 16 bins per octave
constexpr int kDepthShift = 19;
constexpr int kRoundThreads = 1024;

__global__ __launch_bounds__(kRoundThreads) void round_hist_kernel(int64_t N, const float* __restrict__ depths,
                                                                   const int32_t* __restrict__ tiles_per_gauss,
                                                                   uint32_t* __restrict__ hist) {
    __shared__ uint32_t h[kDepthBins];
    for (int i = threadIdx.x; i < kDepthBins; i += kRoundThreads) h[i] = 0u;
    __syncthreads();
    for (int64_t n = (int64_t)blockIdx.x * kRoundThreads + threadIdx.x; n < N; n += (int64_t)gridDim.x * kRoundThreads) {
        const int cnt = tiles_per_gauss[n];
        if (cnt > 0) atomicAdd(&h[min(__float_as_uint(depths[n]) >> kDepthShift, (uint32_t)(kDepthBins - 1))], (uint32_t)cnt);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kDepthBins; i += kRoundThreads)
        if (h[i]) atomicAdd(&hist[i], h[i]);
}

// one block: the first bin edge behind which at least `fraction` of the weight lies; leaves the histogram zero
__global__ __launch_bounds__(kRoundThreads) void round_select_kernel(uint32_t* __restrict__ hist, float fraction, int64_t* __restrict__ blk) {
    __shared__ unsigned long long scratch[17];
    __shared__ int s_bin;
    constexpr int kPer = kDepthBins / kRoundThreads;
    uint32_t v[kPer];
    unsigned long long mine = 0;
#pragma unroll
    for (int k = 0; k < kPer; ++k) { v[k] = hist[kPer * threadIdx.x + k]; hist[kPer * threadIdx.x + k] = 0u; mine += v[k]; }
    if (threadIdx.x == 0) s_bin = kDepthBins;
    unsigned long long total;
    unsigned long long run = block_excl_scan_add(mine, scratch, &total);
    // (fp64: the weights sum to the listed intersections, up to 2^31)
    const unsigned long long target = (unsigned long long)ceil((double)fraction * (double)total);
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        run += v[k];
        if (v[k] != 0u && run >= target) atomicMin(&s_bin, kPer * (int)threadIdx.x + k);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const bool all = !(fraction < 1.f) || total == 0ull || s_bin >= kDepthBins - 1;
        blk[GS_ROUND_SPLIT] = all ? (int64_t)0x7f800000 : ((int64_t)(s_bin + 1) << kDepthShift);
        blk[GS_ROUND_BASE] = 0; blk[GS_ROUND_LIVE] = 0; blk[GS_ROUND_FRONT_N] = 0; blk[GS_ROUND_LISTED_ALL] = (int64_t)total;
    }
}

// w (<= 32) bits of the live-tile bitmap's row y from column x on
__device__ __forceinline__ uint32_t live_row_bits(const unsigned long long* __restrict__ bits, int W64, int x, int y, int w) {
    const int j = x >> 6, sh = x & 63;
    unsigned long long v = bits[y * W64 + j] >> sh;
    if (sh + w > 64 && j + 1 < W64) v |= bits[y * W64 + j + 1] << (64 - sh);   // (sh > 32 here: the shift is in range)
    return (uint32_t)v & (w >= 32 ? 0xffffffffu : ((1u << w) - 1u));
}

// Footprints of <= 32 tiles carry a bit per tile: dead tiles leave the mask, the rectangle stays (the bits are relative to it).
// A ROW of the rectangle at a time -- w bits of the bitmap shifted into place: a bit at a time (an LDS read, a division and a
// test per tile) made this the back round's footprint pass: 48 of its 70 us at 2 M Gaussians.
__device__ __forceinline__ uint4 window_small(uint4 fp, const unsigned long long* __restrict__ bits, int W64) {
    const int x0 = fp.x & 0xffff, x1 = fp.x >> 16, y0 = fp.y & 0xffff, y1 = fp.y >> 16;
    const int w = x1 - x0;
    uint32_t lv = 0u;
    for (int y = y0, sh = 0; y < y1; ++y, sh += w) lv |= live_row_bits(bits, W64, x0, y, w) << sh;   // (sh + w <= 32)
    const uint32_t m = fp.z & lv;
    return m ? make_uint4(fp.x, fp.y, m, (uint32_t)__popc(m)) : make_uint4(0u, 0u, 0u, 0u);
}

// Larger footprints shrink to the bounding rectangle of their live tiles.  The WAVE finds it: lane l looks at tile rows y0 + l,
// y0 + l + 64, ... of the footprint and four min / max reductions close it -- a thread walking its own footprint's rows held its
// 63 neighbours for up to th x W64 LDS reads (the first form: 70 us per frame at 2 M Gaussians with 38 % of the tiles live).
__device__ __forceinline__ void live_bounds_wave(int x0, int x1, int y0, int y1, const unsigned long long* __restrict__ bits, int W64,
                                                 int& nx0, int& nx1, int& ny0, int& ny1) {
    nx0 = 1 << 20; nx1 = -1; ny0 = 1 << 20; ny1 = -1;
    const int j0 = x0 >> 6, j1 = (x1 - 1) >> 6;
    for (int y = y0 + lane_id(); y < y1; y += 64) {
        for (int j = j0; j <= j1; ++j) {
            const int lo = max(x0 - 64 * j, 0), hi = min(x1 - 64 * j, 64);
            const unsigned long long mask = (hi >= 64 ? ~0ull : ((1ull << hi) - 1ull)) & ~((1ull << lo) - 1ull);
            const unsigned long long word = bits[y * W64 + j] & mask;
            if (word) {
                ny0 = min(ny0, y); ny1 = max(ny1, y);
                nx0 = min(nx0, 64 * j + (int)__builtin_ctzll(word));
                nx1 = max(nx1, 64 * j + 63 - (int)__builtin_clzll(word));
            }
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        nx0 = min(nx0, __shfl_xor(nx0, d, 64)); ny0 = min(ny0, __shfl_xor(ny0, d, 64));
        nx1 = max(nx1, __shfl_xor(nx1, d, 64)); ny1 = max(ny1, __shfl_xor(ny1, d, 64));
    }
}

// the footprint of a bounding rectangle of live tiles [x0, x1] x [y0, y1] (inclusive): a mask when it holds <= 32 tiles
__device__ __forceinline__ uint4 box_footprint(int x0, int x1, int y0, int y1, const unsigned long long* __restrict__ bits, int W64) {
    if (y1 < 0) return make_uint4(0u, 0u, 0u, 0u);
    ++x1; ++y1;
    const int w = x1 - x0, rect = w * (y1 - y0);
    uint32_t m = 0xffffffffu, cnt = (uint32_t)rect;
    if (rect <= 32) {
        m = 0u;
        for (int y = y0, sh = 0; y < y1; ++y, sh += w) m |= live_row_bits(bits, W64, x0, y, w) << sh;
        cnt = (uint32_t)__popc(m);
    }
    return make_uint4((uint32_t)x0 | ((uint32_t)x1 << 16), (uint32_t)y0 | ((uint32_t)y1 << 16), m, cnt);
}

__global__ __launch_bounds__(kRoundThreads) void round_footprints_kernel(int64_t N, int tw, int th, int W64,
                                                                         const uint4* __restrict__ bbox, const float* __restrict__ depths,
                                                                         const uint8_t* __restrict__ live, int64_t* __restrict__ blk, int phase,
                                                                         uint4* __restrict__ out, int32_t* __restrict__ tpg) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long bits[];   // [th][W64]
    if (round_idle(blk, phase)) return;   // (nothing of the back round is read: every count behind the split is already 0)
    const uint32_t split = (uint32_t)blk[GS_ROUND_SPLIT];
    if (phase == 2) {
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        for (int p = wave; p < th * W64; p += kRoundThreads / 64) {
            const int y = p / W64, x = 64 * (p - y * W64) + lane;
            const unsigned long long word = __ballot(x < tw && live[y * tw + x] != 0);
            if (lane == 0) bits[p] = word;
        }
        __syncthreads();
    }
    uint32_t front_n = 0;
    const int64_t stride = (int64_t)gridDim.x * kRoundThreads;
    for (int64_t base = (int64_t)blockIdx.x * kRoundThreads; base < N; base += stride) {   // (block-uniform trip count: the waves cooperate)
        const int64_t n = base + threadIdx.x;
        const bool in = n < N;
        const uint4 fp = in ? bbox[n] : make_uint4(0u, 0u, 0u, 0u);
        const bool behind = in && fp.w != 0u && __float_as_uint(depths[n]) >= split;
        uint4 o = make_uint4(0u, 0u, 0u, 0u);
        if (phase == 1) {
            if (!behind) o = fp;
            if (in) { out[n] = o; tpg[n] = (int32_t)o.w; }
            front_n += o.w != 0u ? 1u : 0u;
        } else {
            const int x0 = fp.x & 0xffff, x1 = fp.x >> 16, y0 = fp.y & 0xffff, y1 = fp.y >> 16;
            const bool large = behind && (x1 - x0) * (y1 - y0) > 32;
            if (behind && !large) o = window_small(fp, bits, W64);
            int bx0 = 0, bx1 = -1, by0 = 0, by1 = -1;
            for (unsigned long long todo = __ballot(large); todo; todo &= todo - 1ull) {
                const int src = (int)__builtin_ctzll(todo);
                int a0, a1, c0, c1;
                live_bounds_wave(__shfl(x0, src, 64), __shfl(x1, src, 64), __shfl(y0, src, 64), __shfl(y1, src, 64), bits, W64, a0, a1, c0, c1);
                if (lane_id() == src) { bx0 = a0; bx1 = a1; by0 = c0; by1 = c1; }
            }
            if (large) o = box_footprint(bx0, bx1, by0, by1, bits, W64);
            if (behind) tpg[n] = (int32_t)o.w;
            if (in) out[n] = o;
        }
    }
    if (phase == 1) {   // (diagnostic count: one global atomic per block -- one per wave, 32 k on one address at 2 M Gaussians, cost 0.19 ms)
        __shared__ uint32_t s_front;
        if (threadIdx.x == 0) s_front = 0u;
        __syncthreads();
        front_n = wave_reduce_add(front_n);
        if (lane_id() == 0 && front_n) atomicAdd(&s_front, front_n);
        __syncthreads();
        if (threadIdx.x == 0 && s_front) atomicAdd(reinterpret_cast<unsigned long long*>(blk + GS_ROUND_FRONT_N), (unsigned long long)s_front);
    }
}

}  // namespace gs

using namespace gs;

extern "C" int gs_round_split(void* stream, int64_t N, const float* depths, const int32_t* tiles_per_gauss, float fraction,
                              uint32_t* hist_ws, int64_t* rounds_dev) {
    GS_REQUIRE(N >= 0 && hist_ws && rounds_dev && (N == 0 || (depths && tiles_per_gauss)), "N>=0, non-null pointers");
    GS_REQUIRE(fraction > 0.f, "fraction > 0");
    hipStream_t st = (hipStream_t)stream;
    if (N > 0) {
        const unsigned grid = (unsigned)std::min<int64_t>(512, (N + kRoundThreads - 1) / kRoundThreads);
        hipLaunchKernelGGL(round_hist_kernel, dim3(grid), dim3(kRoundThreads), 0, st, N, depths, tiles_per_gauss, hist_ws);
        GS_LAUNCH_CHECK("round_hist_kernel");
    }
    hipLaunchKernelGGL(round_select_kernel, dim3(1), dim3(kRoundThreads), 0, st, hist_ws, fraction, rounds_dev);
    GS_LAUNCH_CHECK("round_select_kernel");
    return GS_OK;
}

namespace gs {
__global__ void round_status_kernel(const int64_t* __restrict__ blk, volatile int64_t* __restrict__ host) {
    if (threadIdx.x < GS_ROUND_WORDS) host[threadIdx.x] = blk[threadIdx.x];
    __threadfence_system();
}
}  // namespace gs

extern "C" int gs_round_status(void* stream, const int64_t* rounds_dev, int64_t* status_host_mapped) {
    GS_REQUIRE(rounds_dev && status_host_mapped, "null pointer");
    hipLaunchKernelGGL(round_status_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, rounds_dev, status_host_mapped);
    GS_LAUNCH_CHECK("round_status_kernel");
    return GS_OK;
}

extern "C" int gs_round_footprints(void* stream, int64_t N, int tile_w, int tile_h, const uint32_t* bbox, const float* depths,
                                   uint32_t* bbox_round, int32_t* tiles_per_gauss_round) {
    const Rounds R = current_rounds();
    GS_REQUIRE(R.phase == 1 || R.phase == 2 || R.phase == 4, "gs_round_footprints works for a round: gs_rounds_set phase 1, 2 or 4 first");
    GS_REQUIRE(N >= 0 && tile_w > 0 && tile_h > 0 && tile_w < 65536 && tile_h < 65536, "N>=0, tile grid within 16 bits");
    if (N == 0) return GS_OK;
    GS_REQUIRE(bbox && depths && bbox_round && tiles_per_gauss_round, "null pointer");
    const int W64 = (tile_w + 63) / 64;
    const size_t lds = sizeof(unsigned long long) * (size_t)W64 * tile_h;
    GS_REQUIRE(lds <= 60 * 1024, "tile grid too large for the live-tile bitmap in LDS");
    // (back round: every block builds the tile bitmap first -- th x W64 dependent byte gathers per wave --, so one block per CU)
    const unsigned grid = (unsigned)std::min<int64_t>(R.phase == 2 ? 256 : 1024, (N + kRoundThreads - 1) / kRoundThreads);
    hipLaunchKernelGGL(round_footprints_kernel, dim3(grid), dim3(kRoundThreads), lds, (hipStream_t)stream, N, tile_w, tile_h, W64,
                       (const uint4*)bbox, depths, (const uint8_t*)R.live, R.blk, R.phase == 4 ? 1 : R.phase, (uint4*)bbox_round, tiles_per_gauss_round);
    GS_LAUNCH_CHECK("round_footprints_kernel");
    return GS_OK;
}
