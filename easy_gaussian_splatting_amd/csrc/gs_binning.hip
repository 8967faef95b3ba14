// gs_binning.hip -- tile lists for gfx950 without global atomics and without a global sort.
//
// gsplat builds the per-tile lists with a count pass, a cumsum, an emit pass and ONE global
// 64-bit radix sort over (camera | tile | depth) keys.  Here the tile
// id is never sorted on: the 160 KB LDS of a CU holds a full per-tile histogram
// (1080p: 8160 tiles = 32 KB, 4K: 32400 tiles = 127 KB), so the tile partition is a counting
// sort with LDS atomics only:
//   bin_hist_kernel     one block per (camera, Gaussian group): LDS histogram over tiles -> matrix
//   bin_colscan_kernel  per tile: exclusive scan down the group axis (in place), tile totals
//   bin_tilescan_kernel one block: exclusive scan of tile totals -> isect_offsets, bucket_offsets,
//                       group bases, {I, n_buckets, max_tile}
//   bin_emit_kernel     same blocks as hist: per-tile write cursor in LDS (returning ds_add),
//                       emits (depth bits << 32 | row slot) into its tile's segment
//   tile_radix_sort_kernel  one block per list of <= 1024 keys: four stable 8-bit LDS counting passes on the depth word ->
//                       depth order; ties by flatten index (slot order == flatten order), as the stable global sort of
//                       the reference yields, restored by a bitonic network in the rare list that has out-of-order
//                       equal depths.  Longer lists: the same sort over compacted work lists (class_items_kernel,
//                       tile_radix_sort_items_kernel: <= 4096 and <= 8192 keys), 8192-key segments + a rank merge up to
//                       65536 keys (seg_merge_kernel), a bitonic network in place in global memory beyond.
// Second half of the file: the two-level binning (gs_bins_count / gs_bins_lists) for scenes whose Gaussians cover many
// tiles -- coarse bins sorted with the same kernels, tiles refined out of them by ordered compaction.
// The "slot" carried in the key's low word is the index of this intersection's gradient row
// (cum_tiles[f] + k): rows of one Gaussian are contiguous, which lets the backward reduce them
// with plain coalesced loads instead of float atomics.
#include <map>
#include <mutex>
#include <utility>

#include "gs_common.h"

namespace gs {

#define GS_BIN_PER_GROUP 4096     // Gaussians per binning block (histogram / emit)
#define GS_BIN_MAX_GROUPS 256     // (bin_colscan_kernel takes up to kColChunks * 16 groups in its batched path)
constexpr int kBinThreads = 1024;
constexpr int kCoopTiles = 32;  // footprints above this are spread over the whole wave

BinLayout bin_layout(int C, int64_t N, int tiles) {
    BinLayout L;
    int64_t g = (N + GS_BIN_PER_GROUP - 1) / GS_BIN_PER_GROUP;
    if (g < 1) g = 1;
    if (g > GS_BIN_MAX_GROUPS) g = GS_BIN_MAX_GROUPS;
    L.groups = (int)g;
    L.per_group = (N + g - 1) / g;
    auto align = [](size_t x) { return (x + 255) & ~(size_t)255; };
    size_t off = 0;
    L.hist_off = off; off = align(off + sizeof(uint32_t) * (size_t)C * L.groups * tiles);
    L.tile_cnt_off = off; off = align(off + sizeof(uint32_t) * (size_t)C * tiles);
    L.grp_tot_off = off; off = align(off + sizeof(uint32_t) * (size_t)C * L.groups);
    L.grp_base_off = off; off = align(off + sizeof(uint32_t) * (size_t)C * L.groups);
    L.items_off = off; off = align(off + sizeof(int32_t) * (64 + (size_t)10 * (size_t)C * tiles));   // == sort_items_bytes(C * tiles)
    L.total = off;
    return L;
}

// footprint word layout written by project_fwd_kernel: x0 | x1<<16, y0 | y1<<16, tile mask, count
__device__ __forceinline__ void unpack_bbox(uint4 b, int& x0, int& x1, int& y0, int& y1) {
    x0 = b.x & 0xffff; x1 = b.x >> 16; y0 = b.y & 0xffff; y1 = b.y >> 16;
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBinThreads) void bin_hist_kernel(int64_t N, int tw, int tiles,
                                                               int64_t per_group,
                                                               const uint4* __restrict__ bbox,
                                                               uint32_t* __restrict__ hist_mat,
                                                               uint32_t* __restrict__ grp_tot, const int64_t* __restrict__ rblk, int phase) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    if (round_idle(rblk, phase)) return;
    uint32_t* hist = lds;               // [tiles]
    uint32_t* scratch = lds + tiles;    // [32]
    const int grp = blockIdx.x, c = blockIdx.y, G = gridDim.x;
    for (int t = threadIdx.x; t < tiles; t += blockDim.x) hist[t] = 0;
    __syncthreads();
    const int64_t g0 = grp * per_group, g1 = min(N, g0 + per_group);
    uint32_t local = 0;
    for (int64_t base = g0; base < g1; base += blockDim.x) {
        const int64_t n = base + threadIdx.x;
        int x0 = 0, x1 = 0, y0 = 0, y1 = 0;
        uint4 fp = make_uint4(0u, 0u, 0u, 0u);
        if (n < g1) { fp = bbox[(int64_t)c * N + n]; unpack_bbox(fp, x0, x1, y0, y1); }
        const int w = x1 - x0, rect = w * (y1 - y0), cnt = (int)fp.w;
        const float inv_w = 1.0f / (float)max(w, 1);
        local += cnt;
        if (rect <= kCoopTiles)   // small footprints: one bit per tile of the rectangle
            for (uint32_t mb = fp.z; mb; mb &= mb - 1) {
                const int i = __ffs((int)mb) - 1, yy = div_by_width(i, inv_w);
                atomicAdd(&hist[(y0 + yy) * tw + x0 + (i - yy * w)], 1u);
            }
        unsigned long long big = __ballot(rect > kCoopTiles);
        while (big) {
            const int src = __ffsll((long long)big) - 1;
            big &= big - 1;
            const int bx0 = __shfl(x0, src, 64), by0 = __shfl(y0, src, 64), bw = __shfl(w, src, 64),
                      bcnt = __shfl(cnt, src, 64);
            const float binv = __shfl(inv_w, src, 64);
            for (int i = lane_id(); i < bcnt; i += 64) {
                const int yy = div_by_width(i, binv);
                atomicAdd(&hist[(by0 + yy) * tw + bx0 + (i - yy * bw)], 1u);
            }
        }
    }
    __syncthreads();
    uint32_t* out = hist_mat + ((size_t)c * G + grp) * tiles;
    for (int t = threadIdx.x; t < tiles; t += blockDim.x) out[t] = hist[t];
    // block total
    uint32_t wsum = wave_reduce_add(local);
    if (lane_id() == 0) scratch[threadIdx.x >> 6] = wsum;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t s = 0;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += scratch[i];
        grp_tot[c * G + grp] = s;
    }
}

// per (camera, tile): exclusive scan over groups, in place.  One thread per (tile, chunk of kColChunk consecutive groups):
// 64 tiles x 16 chunks per block, the chunk sums scanned across the block through LDS.  (Round 2 walked a whole column -- up
// to 256 groups -- with one thread per tile: 8160 threads for the chip, 17 us of dependent batches; this form: 130 k threads.)
constexpr int kColTiles = GS_BIN_MAX_GROUPS > 256 ? 32 : 64, kColChunks = GS_BIN_MAX_GROUPS > 256 ? 32 : 16;
__global__ __launch_bounds__(kColTiles * kColChunks) void bin_colscan_kernel(int C, int G, int tiles, uint32_t* __restrict__ hist_mat,
                                                                              uint32_t* __restrict__ tile_cnt) {
    __shared__ uint32_t part[kColChunks][kColTiles + 1];
    const int tl = threadIdx.x % kColTiles, ch = threadIdx.x / kColTiles;
    const int64_t i = (int64_t)blockIdx.x * kColTiles + tl;   // (camera, tile)
    const bool in = i < (int64_t)C * tiles;
    const int c = in ? (int)(i / tiles) : 0, t = in ? (int)(i % tiles) : 0;
    const int per = (G + kColChunks - 1) / kColChunks;        // groups per chunk (<= 16 for G <= 256)
    const int g0 = ch * per, g1 = min(G, g0 + per);
    uint32_t* col = hist_mat + (size_t)c * G * tiles + t;
    constexpr int kMax = 16;
    uint32_t v[kMax];
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < kMax; ++k) {   // every load first (the scan is in place)
        v[k] = (in && g0 + k < g1) ? col[(size_t)(g0 + k) * tiles] : 0u;
        sum += v[k];
    }
    uint32_t extra = 0;
    for (int g = g0 + kMax; in && g < g1; ++g) extra += col[(size_t)g * tiles];   // (G > 256 never happens: bin_layout)
    part[ch][tl] = sum + extra;
    __syncthreads();
    uint32_t run = 0, total = 0;
#pragma unroll
    for (int k = 0; k < kColChunks; ++k) {
        const uint32_t p = part[k][tl];
        run += k < ch ? p : 0u;
        total += p;
    }
    if (in) {
#pragma unroll
        for (int k = 0; k < kMax; ++k) {
            if (g0 + k < g1) col[(size_t)(g0 + k) * tiles] = run;
            run += v[k];
        }
        for (int g = g0 + kMax; g < g1; ++g) { const uint32_t x = col[(size_t)g * tiles]; col[(size_t)g * tiles] = run; run += x; }
        if (ch == 0) tile_cnt[i] = total;
    }
}

// Three independent single-block jobs in one launch (they used to run one after the other in a single block, 23 us of
// barrier and memory latency on the critical path of every forward):
//   block 0  exclusive scans of the tile counts -> isect_offsets, bucket_offsets; {I, n_buckets, longest list}, flags
//   block 1  launch order of the blend forward (longest lists first)
//   block 2  group bases (exclusive scan of the groups' intersection totals)
constexpr int kScanItems = 16;
__global__ __launch_bounds__(kBinThreads) void bin_tilescan_kernel(
    int n_tiles_total, int n_groups_total, const uint32_t* __restrict__ tile_cnt,
    const uint32_t* __restrict__ grp_tot, int32_t* __restrict__ isect_offsets,
    int32_t* __restrict__ bucket_offsets, uint32_t* __restrict__ grp_base,
    int64_t* __restrict__ info, int32_t* __restrict__ tile_order, int64_t cap_isects, int64_t cap_tile,
    int64_t keep_mask, int32_t* __restrict__ sort_counts, int64_t* __restrict__ info_mirror, int per_call,
    int64_t* __restrict__ rblk, int phase) {
    __shared__ unsigned long long scratch[17];
    const int chunk = kBinThreads * kScanItems;
    // depth rounds: the back round's lists (and slots) continue behind the front round's; a back round with no live tile
    // leaves everything as the front round left it (info[0] = its total)
    if (round_idle(rblk, phase)) return;
    if (blockIdx.x == 2) {   // group bases (few thousand values at most): serial chunks of kBinThreads
        if (sort_counts && threadIdx.x < 64) sort_counts[threadIdx.x] = 0;   // work-list counters of the sort that follows
        unsigned long long gcarry = 0;
        for (int base = 0; base < n_groups_total; base += kBinThreads) {
            const int i = base + threadIdx.x;
            unsigned long long v = i < n_groups_total ? grp_tot[i] : 0ull, total;
            unsigned long long ex = block_excl_scan_add(v, scratch, &total);
            if (i < n_groups_total) grp_base[i] = (uint32_t)(gcarry + ex);
            gcarry += total;
        }
        return;
    }
    if (blockIdx.x == 1) {
        // Launch order for the blend forward (one wave per tile): longest lists first, so that the short
        // ones fill the gaps at the end instead of the long ones sticking out (-10 % on its run time).
        // Counting sort into 64 quarter-octave length classes with LDS atomics -- the order inside a class
        // is arbitrary, which only permutes independent work.  One register-resident chunk only.
        if (!tile_order) return;
        if (n_tiles_total > chunk) {
            for (int i = threadIdx.x; i < n_tiles_total; i += kBinThreads) tile_order[i] = i;
            return;
        }
        // per-wave counters: 16 x fewer lanes contend for one LDS word than with a single set of 64
        constexpr int kW = kBinThreads / 64;
        __shared__ uint32_t cls_cnt[64][kW + 1];   // [class][wave] (+1: stride 17, no bank aliasing down a class)
        auto cls_of = [](uint32_t x) -> int {   // 4 * floor(log2 x) + the next two bits, 0 for x < 2
            if (x < 2u) return 0;
            const int lg = 31 - __clz((int)x);
            return min(63, (lg << 2) | (int)((x << (31 - lg)) >> 29 & 3u));
        };
        const int wv = threadIdx.x >> 6;
        const int first = threadIdx.x * kScanItems;
        uint32_t v[kScanItems];
#pragma unroll
        for (int k = 0; k < kScanItems; ++k) v[k] = (first + k < n_tiles_total) ? tile_cnt[first + k] : 0u;
        for (int i = threadIdx.x; i < 64 * (kW + 1); i += kBinThreads) (&cls_cnt[0][0])[i] = 0u;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kScanItems; ++k)
            if (first + k < n_tiles_total) atomicAdd(&cls_cnt[cls_of(v[k])][wv], 1u);
        __syncthreads();
        if (threadIdx.x < 64) {   // descending classes: lane l owns class 63 - l; waves in order inside a class
            const int c = 63 - (int)threadIdx.x;
            uint32_t n = 0, cw[kW];
#pragma unroll
            for (int w = 0; w < kW; ++w) { cw[w] = cls_cnt[c][w]; n += cw[w]; }
            uint32_t run = wave_incl_scan_add(n) - n;
#pragma unroll
            for (int w = 0; w < kW; ++w) { cls_cnt[c][w] = run; run += cw[w]; }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kScanItems; ++k)
            if (first + k < n_tiles_total) tile_order[atomicAdd(&cls_cnt[cls_of(v[k])][wv], 1u)] = first + k;
        return;
    }
    unsigned long long carry_i = phase == 2 ? (unsigned long long)rblk[GS_ROUND_BASE] : 0ull, carry_b = 0;
    uint32_t max_cnt = 0;
    for (int base = 0; base < n_tiles_total; base += chunk) {
        const int first = base + threadIdx.x * kScanItems;
        uint32_t v[kScanItems];
        unsigned long long si = 0, sb = 0;
#pragma unroll
        for (int k = 0; k < kScanItems; ++k) {
            v[k] = (first + k < n_tiles_total) ? tile_cnt[first + k] : 0u;
            si += v[k];
            sb += (v[k] + GS_BUCKET - 1) / GS_BUCKET;
            max_cnt = max(max_cnt, v[k]);
        }
        // pack both running sums in one 64-bit scan: buckets (<= 2^31/64+tiles) in the high word
        unsigned long long packed = si | (sb << 36), total;
        unsigned long long ex = block_excl_scan_add(packed, scratch, &total);
        unsigned long long ri = carry_i + (ex & ((1ull << 36) - 1)), rb = carry_b + (ex >> 36);
#pragma unroll
        for (int k = 0; k < kScanItems; ++k) {
            if (first + k < n_tiles_total) {
                isect_offsets[first + k] = (int32_t)ri;
                bucket_offsets[first + k] = (int32_t)rb;
            }
            ri += v[k];
            rb += (v[k] + GS_BUCKET - 1) / GS_BUCKET;
        }
        carry_i += total & ((1ull << 36) - 1);
        carry_b += total >> 36;
    }
    // max over block
    uint32_t m = max_cnt;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d, 64));
    __shared__ uint32_t smax[16];
    if (lane_id() == 0) smax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t mm = 0;
        for (int i = 0; i < kBinThreads / 64; ++i) mm = max(mm, smax[i]);
        isect_offsets[n_tiles_total] = (int32_t)carry_i;
        bucket_offsets[n_tiles_total] = (int32_t)carry_b;
        if (phase == 1) rblk[GS_ROUND_BASE] = (int64_t)carry_i;
        if (phase != 0) {   // (depth rounds: what one round would have listed, and how many tiles the front round left live -- 0 when
            info[7] = rblk[GS_ROUND_LISTED_ALL];   //  this record is the front round's and no back round's follows: it was idle)
            info[6] = phase == 2 ? rblk[GS_ROUND_LIVE] : 0;
        }
        if (phase == 2) mm = max(mm, (uint32_t)info[2]);   // (the longest list of either round)
        info[0] = (int64_t)carry_i; info[1] = (int64_t)carry_b; info[2] = (int64_t)mm;
        if (cap_isects > 0) {
            // guarded step (gs_guard_set): flags are sticky -- once a step does not fit, this and every later
            // step is a no-op until the host has re-sized the buffers and cleared the word
            const int64_t f = ((int64_t)carry_i > cap_isects ? 1 : 0) | ((int64_t)mm > cap_tile ? 2 : 0);
            // (gs_guard_set_call: whatever an earlier CALL left is overwritten -- by the front round when there are two)
            if (per_call && phase != 2) info[3] = (info[3] & keep_mask) | f;
            else if (f) info[3] |= f;
        } else {
            info[3] = keep_mask ? (info[3] & keep_mask) : 0;   // (two-level binning: the coarse stage's overflow bit survives)
        }
        if (info_mirror) {   // page-locked host memory (gs_info_mirror_set): the sizes reach the host without a copy on the stream
#pragma unroll
            for (int i = 0; i < 8; ++i) info_mirror[i] = info[i];
            __threadfence_system();
        }
    }
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBinThreads) void bin_emit_kernel(
    int64_t N, int tw, int tiles, int64_t per_group, const uint4* __restrict__ bbox,
    const float* __restrict__ depths, const uint32_t* __restrict__ hist_mat,
    const int32_t* __restrict__ isect_offsets, const uint32_t* __restrict__ grp_base,
    unsigned long long* __restrict__ keys, int32_t* __restrict__ slot_gid,
    int32_t* __restrict__ cum_tiles, const int64_t* __restrict__ guard, const int64_t* __restrict__ rblk, int phase) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    if (guard_tripped(guard) || round_idle(rblk, phase)) return;
    uint32_t* cursor = lds;                                    // [tiles]
    uint32_t* scratch = lds + tiles;                           // [32]
    const int grp = blockIdx.x, c = blockIdx.y, G = gridDim.x;
    const uint32_t* mine = hist_mat + ((size_t)c * G + grp) * tiles;
    const int32_t* toff = isect_offsets + (size_t)c * tiles;
    // cursor init, loads batched ahead of the LDS stores (8 tiles per thread and round)
    for (int t0 = 0; t0 < tiles; t0 += 8 * (int)blockDim.x) {
        uint32_t v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int t = t0 + k * (int)blockDim.x + (int)threadIdx.x;
            v[k] = t < tiles ? mine[t] + (uint32_t)toff[t] : 0u;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int t = t0 + k * (int)blockDim.x + (int)threadIdx.x;
            if (t < tiles) cursor[t] = v[k];
        }
    }
    __syncthreads();
    // (depth rounds: the back round's gradient-row slots continue behind the front round's)
    uint32_t running = grp_base[c * G + grp] + (phase == 2 ? (uint32_t)rblk[GS_ROUND_BASE] : 0u);
    const int64_t g0 = grp * per_group, g1 = min(N, g0 + per_group);
    // the next round's footprint and depth are requested before this round's scatter
    uint4 fp_next = make_uint4(0u, 0u, 0u, 0u);
    float d_next = 0.f;
    if (g0 + threadIdx.x < g1) { fp_next = bbox[(int64_t)c * N + g0 + threadIdx.x]; d_next = depths[(int64_t)c * N + g0 + threadIdx.x]; }
    for (int64_t base = g0; base < g1; base += blockDim.x) {
        const int64_t n = base + threadIdx.x;
        const int64_t f = (int64_t)c * N + n;
        int x0 = 0, x1 = 0, y0 = 0, y1 = 0;
        const uint4 fp = fp_next;
        const float dcur = d_next;
        if (n < g1) unpack_bbox(fp, x0, x1, y0, y1);
        const int w = x1 - x0, rect = w * (y1 - y0), cnt = (int)fp.w;
        const float inv_w = 1.0f / (float)max(w, 1);
        uint32_t total;
        const uint32_t slot0 = running + block_excl_scan_add((uint32_t)cnt, scratch, &total);
        running += total;
        // (after the scan: its barriers drain the vector-memory counter)
        fp_next = make_uint4(0u, 0u, 0u, 0u);
        if (n + blockDim.x < g1) { fp_next = bbox[f + blockDim.x]; d_next = depths[f + blockDim.x]; }
        if (n < g1 && (phase != 2 || cnt > 0)) cum_tiles[f] = (int32_t)slot0;   // (a Gaussian of the front round keeps its slots)
        const uint32_t dbits = cnt > 0 ? __float_as_uint(dcur) : 0u;
        if (rect <= kCoopTiles) {
            uint32_t k = 0;
            for (uint32_t mb = fp.z; mb; mb &= mb - 1, ++k) {
                const int i = __ffs((int)mb) - 1, yy = div_by_width(i, inv_w);
                const uint32_t pos = atomicAdd(&cursor[(y0 + yy) * tw + x0 + (i - yy * w)], 1u);
                keys[pos] = ((unsigned long long)dbits << 32) | (slot_gid ? slot0 + k : (uint32_t)f);
                if (slot_gid) slot_gid[slot0 + k] = (int32_t)f;
            }
        }
        unsigned long long big = __ballot(rect > kCoopTiles);
        while (big) {
            const int src = __ffsll((long long)big) - 1;
            big &= big - 1;
            const int bx0 = __shfl(x0, src, 64), by0 = __shfl(y0, src, 64), bw = __shfl(w, src, 64),
                      bcnt = __shfl(cnt, src, 64);
            const uint32_t bslot = __shfl((int)slot0, src, 64), bd = __shfl((int)dbits, src, 64);
            const int32_t bf = (int32_t)(c * N + base) + (src + (threadIdx.x & ~63));
            const float binv = __shfl(inv_w, src, 64);
            for (int i = lane_id(); i < bcnt; i += 64) {
                const int yy = div_by_width(i, binv);
                const uint32_t pos = atomicAdd(&cursor[(by0 + yy) * tw + bx0 + (i - yy * bw)], 1u);
                keys[pos] = ((unsigned long long)bd << 32) | (slot_gid ? bslot + i : (uint32_t)bf);
                if (slot_gid) slot_gid[bslot + i] = bf;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Ascending-only bitonic network (first step of each stage mirrors, the rest are half-cleaners);
// indices >= n behave as +inf and are never touched, so no padding is materialised.
template <typename KeyPtr>
__device__ __forceinline__ void bitonic_sort_keys(KeyPtr key, int n, int P) {
    for (int k = 2; k <= P; k <<= 1) {
        const int half = k >> 1;
        for (int idx = threadIdx.x; idx < (P >> 1); idx += blockDim.x) {
            const int blk = idx / half, off = idx - blk * half;
            const int a = blk * k + off, b = blk * k + k - 1 - off;
            if (b < n) {
                const unsigned long long ka = key[a], kb = key[b];
                if (ka > kb) { key[a] = kb; key[b] = ka; }
            }
        }
        __syncthreads();
        for (int j = half >> 1; j > 0; j >>= 1) {
            for (int idx = threadIdx.x; idx < (P >> 1); idx += blockDim.x) {
                const int a = ((idx & ~(j - 1)) << 1) | (idx & (j - 1)), b = a + j;
                if (b < n) {
                    const unsigned long long ka = key[a], kb = key[b];
                    if (ka > kb) { key[a] = kb; key[b] = ka; }
                }
            }
            __syncthreads();
        }
    }
}

struct SortArgs {
    int tiles, tile_bits, lo_excl, hi_incl;  // this launch sorts tiles with lo_excl < n <= hi_incl
    const int32_t* isect_offsets;
    unsigned long long* keys;
    const int32_t* slot_gid;
    int64_t* isect_ids;
    int32_t* flatten_ids;
    int32_t* slots;
    const int64_t* guard;
    int coarse;   // two-level binning: the lists are coarse-bin lists of (depth bits << 32 | flatten id) keys, sorted in place
    int seg;      // > 0: segment mode -- block b sorts segment b % kSegMax (kSegLen keys) of list b / kSegMax, in place
    unsigned long long* merged;   // segment mode, coarse lists: where seg_merge_kernel leaves the merged list
    const int64_t* rblk;          // depth rounds (gs_rounds_set): a back round with no live tile sorts nothing
    int phase;
};
__device__ __forceinline__ bool sort_off(const SortArgs& a) { return guard_tripped(a.guard) || round_idle(a.rblk, a.phase); }

constexpr int kSegLen = 8192;   // the largest LDS radix class
constexpr int kSegMax = 8;      // lists of up to 65536 keys: sorted segment by segment, then rank-merged

template <bool IN_LDS>
__global__ void tile_sort_kernel(const SortArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long skeys[];
    const int t = blockIdx.x;
    if (sort_off(a)) return;
    const int lo = a.isect_offsets[t], hi = a.isect_offsets[t + 1];
    const int n = hi - lo;
    if (n <= a.lo_excl || n > a.hi_incl) return;
    int P = 1;
    while (P < n) P <<= 1;
    unsigned long long* gk = a.keys + lo;
    if (IN_LDS) {
        for (int i = threadIdx.x; i < n; i += blockDim.x) skeys[i] = gk[i];
        __syncthreads();
        bitonic_sort_keys(skeys, n, P);
    } else {
        bitonic_sort_keys(gk, n, P);
    }
    if (a.coarse) {
        if (IN_LDS) for (int i = threadIdx.x; i < n; i += blockDim.x) gk[i] = skeys[i];
        return;
    }
    const int cam = t / a.tiles, tid = t - cam * a.tiles;
    const long long hi_bits = ((long long)cam << (32 + a.tile_bits)) | ((long long)tid << 32);
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const unsigned long long k = IN_LDS ? skeys[i] : gk[i];
        const uint32_t slot = (uint32_t)k;
        if (a.slot_gid) { a.slots[lo + i] = (int32_t)slot; a.flatten_ids[lo + i] = a.slot_gid[slot]; }
        else a.flatten_ids[lo + i] = (int32_t)slot;   // inference lists: the key's low word is the flatten id itself
        if (a.isect_ids) a.isect_ids[lo + i] = hi_bits | (long long)(k >> 32);
    }
}

// ------------------------------------------------------------------------------------------------
// LDS radix sort of one tile's keys (depth bits << 32 | slot): four stable 8-bit counting passes
// on the depth word, ping-ponging between two LDS buffers.  The bitonic network above moves every
// key through LDS ~log^2(n)/2 times and sits on the LDS store rate (ds_write_b64 is a third of the
// read rate on gfx950); this moves each key 4 times.
//   per pass:  counts[wave][digit] by LDS atomics -> exclusive scan (digit-major, wave-minor) ->
//              every wave re-walks ITS contiguous slice in order, 64 keys at a time, ranks each key
//              among the equal digits of the 64 (eight ballots) and scatters.
// A pass whose digit is the same for all keys is skipped (the exponent byte of the depths, mostly).
// Equal depths must come out in slot order (the stable-sort contract of the reference, A.3) but
// arrive in arbitrary order: a tile in which an equal-depth pair is out of order -- practically
// never -- is re-sorted on the full 64-bit key by the bitonic network.
template <int T>
__device__ __forceinline__ void radix_sort_list(const SortArgs& a, const int vblock) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long skeys[];
    constexpr int W = T / 64;
    const int tile = a.seg ? vblock / kSegMax : vblock;
    int lo = a.isect_offsets[tile], n = a.isect_offsets[tile + 1] - lo;
    if (n <= a.lo_excl || n > a.hi_incl) return;
    if (a.seg) {   // this block's slice of a list in (kSegLen, kSegMax * kSegLen]
        const int sg = vblock - tile * kSegMax;
        lo += sg * kSegLen;
        n = min(kSegLen, n - sg * kSegLen);
        if (n <= 0) return;
    }
    const int cap = a.seg ? kSegLen : a.hi_incl;
    unsigned long long* src = skeys;
    unsigned long long* dst = skeys + cap;
    uint32_t* cnt = reinterpret_cast<uint32_t*>(skeys + 2 * cap);   // [W][256]: counts, then offsets, of this pass
    uint32_t* nxt = cnt + W * 256;                                  // [W][256]: counts of the next pass
    __shared__ int s_flag;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long* gk = a.keys + lo;
    // every wave owns a contiguous slice of positions (stable passes need position order)
    const int per_wave = ((n + W * 64 - 1) / (W * 64)) * 64;
    const int w_lo = min(n, wave * per_wave), w_hi = min(n, w_lo + per_wave);
    // slice of a position without an integer division: positions and slice length in units of 64 are
    // small integers, (a + 0.5) / b truncates exactly in fp32
    const float inv_pw64 = 1.0f / (float)(per_wave >> 6);
    auto owner = [inv_pw64](int pos) { return (int)(((float)(pos >> 6) + 0.5f) * inv_pw64); };
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    for (int i = tid; i < 2 * W * 256; i += T) cnt[i] = 0u;
    __syncthreads();
    for (int i = tid; i < n; i += T) {   // load + digit counts of pass 0 for the slice the key sits in
        const unsigned long long k = gk[i];
        src[i] = k;
        atomicAdd(&cnt[owner(i) * 256 + (int)((k >> 32) & 255ull)], 1u);
    }
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 32 + 8 * pass;
        __syncthreads();   // counts of this pass complete, keys in place
        if (wave == 0) {
            // digit-major, wave-minor exclusive scan by one wave: lane l owns digits 4l .. 4l+3
            uint32_t c[4][W], tot[4], mine = 0;
            bool all_in_one = false;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                tot[q] = 0;
#pragma unroll
                for (int w = 0; w < W; ++w) { c[q][w] = cnt[w * 256 + 4 * lane + q]; tot[q] += c[q][w]; }
                all_in_one |= (int)tot[q] == n;
                mine += tot[q];
            }
            uint32_t run = wave_incl_scan_add(mine) - mine;
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int w = 0; w < W; ++w) { cnt[w * 256 + 4 * lane + q] = run; run += c[q][w]; }
            const bool skip = __any(all_in_one);   // one digit holds every key: nothing to move in this pass
            if (lane == 0) s_flag = skip ? 1 : 0;
        } else {
            for (int i = tid - 64; i < W * 256; i += T - 64) nxt[i] = 0u;
        }
        __syncthreads();
        const bool skip = s_flag != 0;
        const int nshift = shift + 8;
        if (skip) {
            if (pass < 3)   // keys stay where they are: count the next digit in place
                for (int i = w_lo + lane; i < w_hi; i += 64) atomicAdd(&nxt[wave * 256 + (int)((src[i] >> nshift) & 255ull)], 1u);
        } else {
            for (int base = w_lo; base < w_hi; base += 64) {
                const int i = base + lane;
                const bool valid = i < w_hi;
                const unsigned long long k = valid ? src[i] : 0ull;
                const int d = (int)((k >> shift) & 255ull);
                // lanes whose digit differs from this lane's in bit b: ballot ^ (-bit) -- one sign-extending bit-field
                // extract, one compare, two xor and two or per bit (the select form costs two cndmask + two and + a not more)
                uint32_t mlo = 0u, mhi = 0u;
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    const int e = __builtin_amdgcn_sbfe(d, b, 1);   // 0 or -1
                    const unsigned long long bm = __ballot(e != 0);
                    mlo |= (uint32_t)bm ^ (uint32_t)e;
                    mhi |= (uint32_t)(bm >> 32) ^ (uint32_t)e;
                }
                const unsigned long long peers = __ballot(valid) & ~(((unsigned long long)mhi << 32) | (unsigned long long)mlo);
                if (valid) {
                    const int rank = __popcll(peers & lt_mask);
                    const uint32_t off = cnt[wave * 256 + d];
                    const uint32_t pos = off + (uint32_t)rank;
                    dst[pos] = k;
                    if (rank == 0) cnt[wave * 256 + d] = off + (uint32_t)__popcll(peers);   // after every peer's read (in-order LDS)
                    if (pass < 3) atomicAdd(&nxt[owner((int)pos) * 256 + (int)((k >> nshift) & 255ull)], 1u);
                }
            }
            unsigned long long* t = src; src = dst; dst = t;
        }
        uint32_t* tc = cnt; cnt = nxt; nxt = tc;
    }
    // equal depths out of slot order?  (keys are unique, so "not ascending" can only mean that)
    __syncthreads();
    if (tid == 0) s_flag = 0;
    __syncthreads();
    bool bad = false;
    for (int i = tid + 1; i < n; i += T) bad |= src[i] < src[i - 1];
    if (bad) s_flag = 1;
    __syncthreads();
    if (s_flag) {
        int P = 1;
        while (P < n) P <<= 1;
        bitonic_sort_keys(src, n, P);
    }
    if (a.coarse || a.seg) {
        unsigned long long* out = a.keys + lo;
        for (int i = tid; i < n; i += T) out[i] = src[i];
        return;
    }
    const int cam = tile / a.tiles, tix = tile - cam * a.tiles;
    const long long hi_bits = ((long long)cam << (32 + a.tile_bits)) | ((long long)tix << 32);
    for (int i = tid; i < n; i += T) {
        const unsigned long long k = src[i];
        const uint32_t slot = (uint32_t)k;
        if (a.slot_gid) { a.slots[lo + i] = (int32_t)slot; a.flatten_ids[lo + i] = a.slot_gid[slot]; }
        else a.flatten_ids[lo + i] = (int32_t)slot;   // inference lists: the key's low word is the flatten id itself
        if (a.isect_ids) a.isect_ids[lo + i] = hi_bits | (long long)(k >> 32);
    }
}

template <int T>
__global__ __launch_bounds__(T) void tile_radix_sort_kernel(const SortArgs a) {
    if (sort_off(a)) return;
    radix_sort_list<T>(a, blockIdx.x);
}

// The same sort over a compacted work list (class_items_kernel): the classes above 1024 keys need 72-152 KB of LDS per
// block, so a block that only finds out that its list belongs to another class still has to wait for that much LDS to
// come free -- thousands of them serialise behind the real work.  Here a fixed number of blocks strides over the items.
template <int T>
__global__ __launch_bounds__(T) void tile_radix_sort_items_kernel(const SortArgs a, const int32_t* __restrict__ items,
                                                                  const int32_t* __restrict__ n_items) {
    if (sort_off(a)) return;
    const int n = *n_items;
    for (int it = blockIdx.x; it < n; it += gridDim.x) {
        radix_sort_list<T>(a, items[it]);
        __syncthreads();
    }
}

// Work lists of the sort classes above 1024 keys: items[0] lists of (1024, 4096], items[1] of (4096, 8192], items[2] the
// (list * kSegMax + segment) pairs of the lists of (8192, 65536].  counts[3] zeroed by the caller.
__global__ __launch_bounds__(256) void class_items_kernel(int n_lists, const int32_t* __restrict__ offsets,
                                                          int32_t* __restrict__ items, int32_t* __restrict__ counts,
                                                          const int64_t* __restrict__ guard, const int64_t* __restrict__ rblk, int phase) {
    if (guard_tripped(guard) || round_idle(rblk, phase)) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_lists) return;
    const int n = offsets[i + 1] - offsets[i];
    if (n > 1024 && n <= 4096) items[atomicAdd(&counts[0], 1)] = i;
    else if (n > 4096 && n <= kSegLen) items[n_lists + atomicAdd(&counts[1], 1)] = i;
    else if (n > kSegLen && n <= kSegLen * kSegMax) {
        const int ns = (n + kSegLen - 1) / kSegLen;
        const int p = atomicAdd(&counts[2], ns);
        for (int sg = 0; sg < ns; ++sg) items[2 * n_lists + p + sg] = i * kSegMax + sg;
    }
}

static size_t sort_items_bytes(int64_t n_lists) {   // work lists of launch_list_sorts: 3 counters + items of three classes
    return sizeof(int32_t) * (64 + (size_t)(2 + kSegMax) * (size_t)n_lists);
}


// ================================================================================================
// Two-level binning (gs_bins_count / gs_bins_lists).  The per-tile pipeline above sorts every one of the I list
// entries (3-4 LDS passes each); with footprints of tens to hundreds of tiles (real captures) that is most of the
// forward.  Here only the COARSE lists are sorted: a bin is 4x4 (or 2x2) tiles, a Gaussian enters every bin its
// rectangle touches (I' entries, 1.4-17 per Gaussian instead of 3-190), each bin's list is depth-sorted by the same
// LDS radix kernels on (depth bits << 32 | flatten id) keys, and every tile then takes its own list out of its bin's
// sorted list by an ORDERED compaction (ballot + popcount: no atomics, no second sort, depth and tie order inherited):
//   bins_hist_kernel    per (camera, Gaussian group): LDS histogram over bins
//   bin_colscan_kernel  (shared)  per bin: exclusive scan down the group axis
//   bins_scan_kernel    one block: bin offsets, group bases, {I', longest bin list}, coarse overflow flag
//   bins_emit_kernel    per group: LDS cursors -> keys into the bin segments; cum_tiles (gradient-row slot bases)
//   tile_radix_sort_kernel / tile_sort_kernel (shared, coarse mode): bins sorted in place
//   bins_refine_kernel<false>  one block per bin, one wave per tile: per-tile counts
//   bin_tilescan_kernel (shared)  isect_offsets, bucket_offsets, tile_order, {I, n_buckets, max list}
//   bins_refine_kernel<true>   the same walk, writing flatten_ids / isect_ids / slots at the scanned offsets
struct BinsLayout {
    int groups, shift, bw, bh, nbins;
    int64_t per_group;
    size_t hist_off;      // u32 [C*groups][nbins]
    size_t bin_cnt_off;   // u32 [C*nbins]
    size_t grp_tot_off;   // u32 [C*groups]   fine intersections of each group (for cum_tiles)
    size_t grp_base_off;  // u32 [C*groups]
    size_t coff_off;      // i32 [C*nbins+1]  bin offsets into the coarse key buffer
    size_t choff_off;     // i32 [C*nbins+1]  exclusive scan of the bins' chunk counts (chunk = 64 << 2*shift entries)
    size_t tile_cnt_off;  // u32 [C*tiles]
    size_t chunk_bin_off; // int4 [max_chunks]  {bin, first entry, entries, -} of each chunk
    int chunk_shift;
    size_t rec_off;       // uint4 [C*N]  footprint with the slot base in place of the count (one gather instead of two)
    size_t staged_off;    // uint4 [coarse_cap]  the same records in sorted bin order (written by pass 1, read by pass 2)
    size_t cnt_ct_off;    // u32 [max_chunks][tiles per bin]  per-chunk tile counts -> exclusive prefix inside a tile
    size_t items_off;     // i32 work lists of the sort classes above 1024 keys
    int64_t max_chunks;
    size_t total;
};

static BinsLayout bins_layout(int C, int64_t N, int tw, int th, int bin_shift, int64_t coarse_cap) {
    BinsLayout L;
    int64_t g = (N + GS_BIN_PER_GROUP - 1) / GS_BIN_PER_GROUP;
    if (g < 1) g = 1;
    if (g > GS_BIN_MAX_GROUPS) g = GS_BIN_MAX_GROUPS;
    L.groups = (int)g;
    L.per_group = (N + g - 1) / g;
    // 4x4-tile bins unless that puts more than ~2500 Gaussians per bin on average (sort classes above 4096 entries
    // run one block per CU): then 2x2
    const int64_t nb2 = (int64_t)((tw + 3) / 4) * ((th + 3) / 4);
    L.shift = bin_shift ? bin_shift : ((N / nb2 > 2500) ? 1 : 2);
    const int B = 1 << L.shift;
    L.bw = (tw + B - 1) / B; L.bh = (th + B - 1) / B; L.nbins = L.bw * L.bh;
    auto align = [](size_t x) { return (x + 255) & ~(size_t)255; };
    size_t off = 0;
    L.hist_off = off; off = align(off + sizeof(uint32_t) * (size_t)C * L.groups * L.nbins);
    L.bin_cnt_off = off; off = align(off + sizeof(uint32_t) * (size_t)C * L.nbins);
    L.grp_tot_off = off; off = align(off + sizeof(uint32_t) * (size_t)C * L.groups);
    L.grp_base_off = off; off = align(off + sizeof(uint32_t) * (size_t)C * L.groups);
    L.coff_off = off; off = align(off + sizeof(int32_t) * ((size_t)C * L.nbins + 1));
    L.choff_off = off; off = align(off + sizeof(int32_t) * ((size_t)C * L.nbins + 1));
    L.tile_cnt_off = off; off = align(off + sizeof(uint32_t) * (size_t)C * tw * th);
    L.chunk_shift = 10 + (L.shift - 1);   // chunk = 1024 entries (2x2-tile bins) / 2048 (4x4)
    L.max_chunks = (coarse_cap >> L.chunk_shift) + (int64_t)C * L.nbins;   // sum of ceil(n_b / chunk) can not exceed this
    L.chunk_bin_off = off; off = align(off + sizeof(int4) * (size_t)L.max_chunks);
    L.rec_off = off; off = align(off + sizeof(uint4) * (size_t)C * (size_t)N);
    L.staged_off = off; off = align(off + sizeof(uint4) * (size_t)coarse_cap);
    L.cnt_ct_off = off; off = align(off + sizeof(uint32_t) * (size_t)L.max_chunks * B * B);
    L.items_off = off; off = align(off + sort_items_bytes((int64_t)C * L.nbins));
    L.total = off;
    return L;
}

// bins touched by a footprint (tile rectangle, possibly with a tile mask: the rectangle is what counts here -- a bin
// none of whose tiles is in the mask just yields no entry in the refinement)
__device__ __forceinline__ void coarse_rect(uint4 fp, int shift, int& cx0, int& cw, int& cy0, int& crect) {
    int x0, x1, y0, y1;
    unpack_bbox(fp, x0, x1, y0, y1);
    if (fp.w == 0u) { cx0 = cy0 = cw = crect = 0; return; }
    const int r = (1 << shift) - 1;
    cx0 = x0 >> shift; cy0 = y0 >> shift;
    cw = ((x1 + r) >> shift) - cx0;
    crect = cw * (((y1 + r) >> shift) - cy0);
}

__global__ __launch_bounds__(kBinThreads) void bins_hist_kernel(int64_t N, int shift, int bw, int nbins, int64_t per_group,
                                                                const uint4* __restrict__ bbox,
                                                                uint32_t* __restrict__ hist_mat,
                                                                uint32_t* __restrict__ grp_tot, const int64_t* __restrict__ rblk, int phase) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    if (round_idle(rblk, phase)) return;
    uint32_t* hist = lds;               // [nbins]
    uint32_t* scratch = lds + nbins;    // [32]
    const int grp = blockIdx.x, c = blockIdx.y, G = gridDim.x;
    for (int t = threadIdx.x; t < nbins; t += blockDim.x) hist[t] = 0;
    __syncthreads();
    const int64_t g0 = grp * per_group, g1 = min(N, g0 + per_group);
    uint32_t local = 0;
    for (int64_t base = g0; base < g1; base += blockDim.x) {
        const int64_t n = base + threadIdx.x;
        uint4 fp = make_uint4(0u, 0u, 0u, 0u);
        if (n < g1) fp = bbox[(int64_t)c * N + n];
        int cx0, cw, cy0, crect;
        coarse_rect(fp, shift, cx0, cw, cy0, crect);
        local += fp.w;
        if (crect <= kCoopTiles)
            for (int i = 0, xx = 0, row = cy0 * bw + cx0; i < crect; ++i) {
                atomicAdd(&hist[row + xx], 1u);
                if (++xx == cw) { xx = 0; row += bw; }
            }
        unsigned long long big = __ballot(crect > kCoopTiles);
        while (big) {
            const int src = __ffsll((long long)big) - 1;
            big &= big - 1;
            const int bx0 = __shfl(cx0, src, 64), by0 = __shfl(cy0, src, 64), bcw = __shfl(cw, src, 64), bcnt = __shfl(crect, src, 64);
            for (int i = lane_id(); i < bcnt; i += 64) {
                const int yy = i / bcw;
                atomicAdd(&hist[(by0 + yy) * bw + bx0 + (i - yy * bcw)], 1u);
            }
        }
    }
    __syncthreads();
    uint32_t* out = hist_mat + ((size_t)c * G + grp) * nbins;
    for (int t = threadIdx.x; t < nbins; t += blockDim.x) out[t] = hist[t];
    uint32_t wsum = wave_reduce_add(local);
    if (lane_id() == 0) scratch[threadIdx.x >> 6] = wsum;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t s = 0;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += scratch[i];
        grp_tot[c * G + grp] = s;
    }
}

// per (camera, bin): exclusive scan over the groups, one wave per bin (G <= 256: four strided elements per lane)
__global__ __launch_bounds__(256) void bins_colscan_kernel(int C, int G, int nbins, uint32_t* __restrict__ hist_mat,
                                                          uint32_t* __restrict__ bin_cnt) {
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= (int64_t)C * nbins) return;
    const int c = (int)(i / nbins), t = (int)(i % nbins), lane = lane_id();
    uint32_t* col = hist_mat + (size_t)c * G * nbins + t;
    uint32_t v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int g = 64 * k + lane; v[k] = g < G ? col[(size_t)g * nbins] : 0u; }
    uint32_t carry = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t incl = wave_incl_scan_add(v[k]);
        const int g = 64 * k + lane;
        if (g < G) col[(size_t)g * nbins] = carry + incl - v[k];
        carry += __shfl(incl, 63, 64);
    }
    if (lane == 0) bin_cnt[i] = carry;
}

// block 0: bin offsets (exclusive scan of the bin totals), chunk descriptors, info[4] = I', info[5] = longest bin list,
// coarse overflow -> flags bit 4 (sticky under a step guard, fresh otherwise); block 1: group bases
__global__ __launch_bounds__(kBinThreads) void bins_scan_kernel(int n_bins_total, int n_groups_total,
                                                                const uint32_t* __restrict__ bin_cnt,
                                                                const uint32_t* __restrict__ grp_tot,
                                                                int32_t* __restrict__ coff, int32_t* __restrict__ choff,
                                                                int4* __restrict__ chunk_desc, int64_t max_chunks,
                                                                int chunk_shift, uint32_t* __restrict__ grp_base,
                                                                int64_t* __restrict__ info, int64_t coarse_cap, int64_t list_cap,
                                                                int guarded, int32_t* __restrict__ sort_counts,
                                                                const int64_t* __restrict__ rblk, int phase) {
    __shared__ unsigned long long scratch[17];
    __shared__ uint32_t smax[16];
    if (round_idle(rblk, phase)) return;
    if (blockIdx.x == 1) {   // second block of the launch: group bases, beside the bin scan
        if (threadIdx.x < 64) sort_counts[threadIdx.x] = 0;   // work-list counters of the coarse sort
        unsigned long long gcarry = 0;
        for (int base = 0; base < n_groups_total; base += kBinThreads) {
            const int i = base + threadIdx.x;
            unsigned long long v = i < n_groups_total ? grp_tot[i] : 0ull, total;
            unsigned long long ex = block_excl_scan_add(v, scratch, &total);
            if (i < n_groups_total) grp_base[i] = (uint32_t)(gcarry + ex);
            gcarry += total;
        }
        return;
    }
    unsigned long long carry = 0, ccarry = 0;
    uint32_t max_cnt = 0;
    // (8 bins per thread, not kScanItems = 16: with the chunk-descriptor loop in the second pass 16 spilled to scratch, and a
    //  kernel that needs scratch can stall its queue for milliseconds, see gs_blend.hip)
    constexpr int kItems = 8;
    const int chunk = kBinThreads * kItems;
    for (int base = 0; base < n_bins_total; base += chunk) {
        const int first = base + threadIdx.x * kItems;
        uint32_t v[kItems];
        unsigned long long si = 0, sc = 0;
        const uint32_t cmask = (1u << chunk_shift) - 1u;
#pragma unroll
        for (int k = 0; k < kItems; ++k) {
            v[k] = (first + k < n_bins_total) ? bin_cnt[first + k] : 0u;
            si += v[k];
            sc += (v[k] + cmask) >> chunk_shift;
            max_cnt = max(max_cnt, v[k]);
        }
        // entries (< 2^31) in the low word, chunks in the high word of one 64-bit scan
        unsigned long long total;
        const unsigned long long ex = block_excl_scan_add(si | (sc << 36), scratch, &total);
        unsigned long long run = carry + (ex & ((1ull << 36) - 1)), crun = ccarry + (ex >> 36);
#pragma unroll
        for (int k = 0; k < kItems; ++k) {
            const uint32_t nch = (v[k] + cmask) >> chunk_shift;
            if (first + k < n_bins_total) {
                coff[first + k] = (int32_t)min(run, (unsigned long long)0x7fffffff);
                choff[first + k] = (int32_t)crun;
                for (uint32_t j = 0; j < nch; ++j)   // (bounded by the buffer even when the coarse capacity is exceeded)
                    if ((int64_t)(crun + j) < max_chunks)
                        chunk_desc[crun + j] = make_int4(first + k, (int)(run + ((unsigned long long)j << chunk_shift)),
                                                         (int)min(v[k] - (j << chunk_shift), 1u << chunk_shift), 0);
            }
            run += v[k];
            crun += nch;
        }
        carry += total & ((1ull << 36) - 1);
        ccarry += total >> 36;
    }
    uint32_t m = max_cnt;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d, 64));
    if (lane_id() == 0) smax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t mm = 0;
        for (int i = 0; i < kBinThreads / 64; ++i) mm = max(mm, smax[i]);
        coff[n_bins_total] = (int32_t)min(carry, (unsigned long long)0x7fffffff);
        choff[n_bins_total] = (int32_t)ccarry;
        if (phase == 2) {   // (depth rounds: the status words show what the larger of the two rounds needed)
            info[4] = max(info[4], (int64_t)carry); info[5] = max(info[5], (int64_t)mm); info[6] = (int64_t)ccarry;
        } else {
            info[4] = (int64_t)carry; info[5] = (int64_t)mm; info[6] = (int64_t)ccarry;
        }
        const int64_t f = ((int64_t)carry > coarse_cap ? 4 : 0) | ((int64_t)mm > list_cap ? 8 : 0);
        if (guarded || phase == 2) { if (f) info[3] |= f; }   // (the back round of a call adds to the front round's flags)
        else info[3] = f;
    }
}

__global__ __launch_bounds__(kBinThreads) void bins_emit_kernel(
    int64_t N, int shift, int bw, int nbins, int64_t per_group, const uint4* __restrict__ bbox,
    const float* __restrict__ depths, const uint32_t* __restrict__ hist_mat, const int32_t* __restrict__ coff,
    const uint32_t* __restrict__ grp_base, unsigned long long* __restrict__ keys, int32_t* __restrict__ cum_tiles,
    uint4* __restrict__ rec, const int64_t* __restrict__ info, const int64_t* __restrict__ rblk, int phase) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    if (info[3] != 0 || round_idle(rblk, phase)) return;
    uint32_t* cursor = lds;             // [nbins]
    uint32_t* scratch = lds + nbins;    // [32]
    const int grp = blockIdx.x, c = blockIdx.y, G = gridDim.x;
    const uint32_t* mine = hist_mat + ((size_t)c * G + grp) * nbins;
    const int32_t* boff = coff + (size_t)c * nbins;
    for (int t0 = 0; t0 < nbins; t0 += 8 * (int)blockDim.x) {   // loads batched ahead of the LDS stores
        uint32_t v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int t = t0 + k * (int)blockDim.x + (int)threadIdx.x;
            v[k] = t < nbins ? mine[t] + (uint32_t)boff[t] : 0u;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int t = t0 + k * (int)blockDim.x + (int)threadIdx.x;
            if (t < nbins) cursor[t] = v[k];
        }
    }
    __syncthreads();
    uint32_t running = grp_base[c * G + grp] + (phase == 2 ? (uint32_t)rblk[GS_ROUND_BASE] : 0u);   // (the back round's slots continue)
    const int64_t g0 = grp * per_group, g1 = min(N, g0 + per_group);
    uint4 fp_next = make_uint4(0u, 0u, 0u, 0u);
    float d_next = 0.f;
    if (g0 + threadIdx.x < g1) { fp_next = bbox[(int64_t)c * N + g0 + threadIdx.x]; d_next = depths[(int64_t)c * N + g0 + threadIdx.x]; }
    for (int64_t base = g0; base < g1; base += blockDim.x) {
        const int64_t n = base + threadIdx.x;
        const int64_t f = (int64_t)c * N + n;
        const uint4 fp = fp_next;
        const float dcur = d_next;
        int cx0, cw, cy0, crect;
        coarse_rect(n < g1 ? fp : make_uint4(0u, 0u, 0u, 0u), shift, cx0, cw, cy0, crect);
        uint32_t total;
        const uint32_t slot0 = running + block_excl_scan_add(n < g1 ? fp.w : 0u, scratch, &total);
        running += total;
        fp_next = make_uint4(0u, 0u, 0u, 0u);
        if (n + blockDim.x < g1) { fp_next = bbox[f + blockDim.x]; d_next = depths[f + blockDim.x]; }
        if (n < g1) {
            if (phase != 2 || fp.w != 0u) cum_tiles[f] = (int32_t)slot0;   // (a Gaussian of the front round keeps its slots)
            rec[f] = make_uint4(fp.x, fp.y, fp.z, slot0);
        }
        const unsigned long long key = ((unsigned long long)__float_as_uint(dcur) << 32) | (unsigned long long)(uint32_t)f;
        if (crect <= kCoopTiles)
            for (int i = 0, xx = 0, row = cy0 * bw + cx0; i < crect; ++i) {
                keys[atomicAdd(&cursor[row + xx], 1u)] = key;
                if (++xx == cw) { xx = 0; row += bw; }
            }
        unsigned long long big = __ballot(crect > kCoopTiles);
        while (big) {
            const int src = __ffsll((long long)big) - 1;
            big &= big - 1;
            const int bx0 = __shfl(cx0, src, 64), by0 = __shfl(cy0, src, 64), bcw = __shfl(cw, src, 64), bcnt = __shfl(crect, src, 64);
            const unsigned long long bkey = __shfl(key, src, 64);
            for (int i = lane_id(); i < bcnt; i += 64) {
                const int yy = i / bcw;
                keys[atomicAdd(&cursor[(by0 + yy) * bw + bx0 + (i - yy * bcw)], 1u)] = bkey;
            }
        }
    }
}

struct RefineArgs {
    int tw, th, tiles, tile_bits, bw, nbins, n_bins_total;
    const int32_t* coff;
    const int32_t* choff;
    const int4* chunk_desc;
    const unsigned long long* keys;   // sorted bin lists
    const uint4* rec;                 // [C*N] footprint + slot base
    uint4* staged;                    // [I'] the same in sorted bin order
    uint32_t* cnt_ct;                 // [chunk][tiles per bin]
    uint32_t* tile_cnt;
    const int32_t* isect_offsets;
    int64_t* isect_ids;
    int32_t* flatten_ids;
    int32_t* slots;
    const int64_t* info;
    const int64_t* rblk;   // depth rounds (gs_rounds_set)
    int phase;
};

// One block per CHUNK (64 << 2*SHIFT consecutive entries of one bin's sorted list), one wave per tile of the bin.  The
// chunk is staged in LDS (key, footprint, slot base: each read from memory once); every wave walks it 64 entries per
// step and keeps those whose footprint holds ITS tile.  Pass 1 (WRITE = false) leaves the per-(chunk, tile) counts;
// bins_chunkscan_kernel turns them into each chunk's start inside the tile's list; pass 2 repeats the walk and writes:
// position = tile offset + chunk start + running count + popcount of the lower lanes' hits, so the bin's (depth, flatten
// id) order is the tile's.  The slot of an entry is the Gaussian's first gradient row + the rank of this tile inside its
// footprint (row-major over the rectangle / the set mask bits).  Chunks are independent: no serial walk down a long bin.
template <bool WRITE, int SHIFT>
__global__ __launch_bounds__(64 << (2 * SHIFT)) void bins_refine_kernel(const RefineArgs a) {
    constexpr int B = 1 << SHIFT, T = 64 * B * B, E = 1024 << (SHIFT - 1), PER = E / T;
    __shared__ uint4 s_box[E];
    __shared__ uint32_t s_d[WRITE ? E : 1], s_cum[WRITE ? E : 1];
    const int chunk = blockIdx.x;
    // (three independent loads, then the tests: a block's life is a chain of memory latencies, keep it short)
    const int64_t flags = a.info[3];
    const int n_chunks = a.choff[a.n_bins_total];
    const int4 desc = a.chunk_desc[chunk];
    if (flags != 0 || chunk >= n_chunks || round_idle(a.rblk, a.phase)) return;
    const int bin = desc.x, base = desc.y, m = desc.z;
    const int cam = bin / a.nbins, b = bin - cam * a.nbins;
    const int by = b / a.bw, bx = b - by * a.bw;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tx = bx * B + (wave & (B - 1)), ty = by * B + (wave >> SHIFT);
    const bool tile_ok = tx < a.tw && ty < a.th;
    const int tile = cam * a.tiles + ty * a.tw + tx;
    int64_t out = 0;
    if (WRITE && tile_ok) out = (int64_t)a.isect_offsets[tile] + (int64_t)a.cnt_ct[(size_t)chunk * (B * B) + wave];
    // pass 1 gathers the footprint records by flatten id (the one random access of the refinement: a 128-byte line per
    // entry) and leaves them in sorted order; pass 2 streams them
    unsigned long long key[PER];
    uint4 fp[PER];
#pragma unroll
    for (int p = 0; p < PER; ++p) {
        const int e = p * T + (int)threadIdx.x;
        key[p] = e < m ? a.keys[base + e] : 0ull;
        if (WRITE) fp[p] = e < m ? a.staged[base + e] : make_uint4(0u, 0u, 0u, 0u);
    }
    if (!WRITE) {
#pragma unroll
        for (int p = 0; p < PER; ++p) {
            const int e = p * T + (int)threadIdx.x;
            fp[p] = e < m ? a.rec[(uint32_t)key[p]] : make_uint4(0u, 0u, 0u, 0u);
        }
    }
#pragma unroll
    for (int p = 0; p < PER; ++p) {
        const int e = p * T + (int)threadIdx.x;
        if (e < m) {
            if (!WRITE) a.staged[base + e] = fp[p];
            if (WRITE) { s_cum[e] = fp[p].w; s_d[e] = (uint32_t)(key[p] >> 32); }
            fp[p].w = (uint32_t)key[p];   // the flatten id rides in the fourth word from here on
            s_box[e] = fp[p];
        }
    }
    __syncthreads();
    if (!tile_ok) return;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const long long hi_bits = ((long long)cam << (32 + a.tile_bits)) | ((long long)(ty * a.tw + tx) << 32);
    uint32_t count = 0;
    for (int j = 0; j < m; j += 64) {
        const int e = j + lane;
        bool hit = false;
        int k = 0;
        uint32_t f = 0;
        if (e < m) {
            const uint4 q = s_box[e];
            int x0, x1, y0, y1;
            unpack_bbox(q, x0, x1, y0, y1);
            f = q.w;
            if (tx >= x0 && tx < x1 && ty >= y0 && ty < y1) {
                const int w = x1 - x0, idx = (ty - y0) * w + (tx - x0);
                if (w * (y1 - y0) <= kCoopTiles) { hit = (q.z >> idx) & 1u; k = __popc(q.z & ((1u << idx) - 1u)); }
                else { hit = true; k = idx; }
            }
        }
        const unsigned long long bal = __ballot(hit);
        if (WRITE) {
            if (hit) {
                const int64_t pos = out + __popcll(bal & lt_mask);
                a.flatten_ids[pos] = (int32_t)f;
                if (a.isect_ids) a.isect_ids[pos] = hi_bits | (long long)s_d[e];
                if (a.slots) a.slots[pos] = (int32_t)(s_cum[e] + (uint32_t)k);
            }
            out += __popcll(bal);
        } else {
            count += (uint32_t)__popcll(bal);
        }
    }
    if (!WRITE && lane == 0) a.cnt_ct[(size_t)chunk * (B * B) + wave] = count;
}

// per tile: exclusive scan of its counts down the chunks of its bin (in place) -> tile_cnt
template <int SHIFT>
__global__ __launch_bounds__(256) void bins_chunkscan_kernel(const RefineArgs a, int C) {
    constexpr int B = 1 << SHIFT;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C * a.tiles) return;
    if (a.info[3] != 0 || round_idle(a.rblk, a.phase)) return;
    const int cam = i / a.tiles, t = i - cam * a.tiles;
    const int ty = t / a.tw, tx = t - ty * a.tw;
    const int bin = cam * a.nbins + (ty >> SHIFT) * a.bw + (tx >> SHIFT);
    const int w = ((ty & (B - 1)) << SHIFT) | (tx & (B - 1));
    const int c0 = a.choff[bin], c1 = a.choff[bin + 1];
    uint32_t run = 0;
    constexpr int kBatch = 8;   // loads batched ahead of the in-place stores (see bin_colscan_kernel)
    for (int c = c0; c < c1; c += kBatch) {
        uint32_t v[kBatch];
#pragma unroll
        for (int k = 0; k < kBatch; ++k) v[k] = c + k < c1 ? a.cnt_ct[(size_t)(c + k) * (B * B) + w] : 0u;
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
            if (c + k < c1) a.cnt_ct[(size_t)(c + k) * (B * B) + w] = run;
            run += v[k];
        }
    }
    a.tile_cnt[i] = run;
}

// Lists of (kSegLen, kSegMax * kSegLen] keys: every kSegLen-segment has been sorted in place (tile_radix_sort_kernel in
// segment mode); the keys are unique, so the position of a key in the merged list is its index in its own segment plus,
// for every other segment, the number of keys there that are smaller (one binary search each).  Per-tile lists are
// written out directly (slots / flatten ids / isect ids); coarse-bin lists go to `merged` and seg_copy_kernel takes
// them back.  (The bitonic network this replaces moved every key ~100 times through LDS: 229 us for 9 k-key bins.)
__global__ __launch_bounds__(1024) void seg_merge_kernel(const SortArgs a, const int32_t* __restrict__ items,
                                                         const int32_t* __restrict__ n_items) {
    if (sort_off(a)) return;
    const int n_it = *n_items;
    for (int it = blockIdx.x; it < n_it; it += gridDim.x) {
    const int vblock = items[it];
    const int list = vblock / kSegMax, sg = vblock - list * kSegMax;
    const int lo = a.isect_offsets[list], n = a.isect_offsets[list + 1] - lo;
    const int s_lo = sg * kSegLen, s_n = min(kSegLen, n - s_lo);
    const int n_seg = (n + kSegLen - 1) / kSegLen;
    const unsigned long long* gk = a.keys + lo;
    const int cam = list / max(a.tiles, 1), tix = list - cam * a.tiles;
    const long long hi_bits = ((long long)cam << (32 + a.tile_bits)) | ((long long)tix << 32);
    for (int i = threadIdx.x; i < s_n; i += blockDim.x) {
        const unsigned long long k = gk[s_lo + i];
        int pos = i;
        for (int o = 0; o < n_seg; ++o) {
            if (o == sg) continue;
            const unsigned long long* seg = gk + o * kSegLen;
            int l = 0, r = min(kSegLen, n - o * kSegLen);   // first index with seg[idx] > k  (== count of smaller keys)
            while (l < r) {
                const int mid = (l + r) >> 1;
                if (seg[mid] < k) l = mid + 1; else r = mid;
            }
            pos += l;
        }
        if (a.coarse) {
            a.merged[lo + pos] = k;
        } else {
            const uint32_t slot = (uint32_t)k;
            if (a.slot_gid) { a.slots[lo + pos] = (int32_t)slot; a.flatten_ids[lo + pos] = a.slot_gid[slot]; }
            else a.flatten_ids[lo + pos] = (int32_t)slot;
            if (a.isect_ids) a.isect_ids[lo + pos] = hi_bits | (long long)(k >> 32);
        }
    }
    }
}

__global__ __launch_bounds__(1024) void seg_copy_kernel(const SortArgs a, const int32_t* __restrict__ items,
                                                        const int32_t* __restrict__ n_items) {
    if (sort_off(a)) return;
    const int n_it = *n_items;
    for (int it = blockIdx.x; it < n_it; it += gridDim.x) {
        const int vblock = items[it];
        const int list = vblock / kSegMax, sg = vblock - list * kSegMax;
        const int lo = a.isect_offsets[list], n = a.isect_offsets[list + 1] - lo;
        const int s_lo = sg * kSegLen, s_n = min(kSegLen, n - s_lo);
        for (int i = threadIdx.x; i < s_n; i += blockDim.x) a.keys[lo + s_lo + i] = a.merged[lo + s_lo + i];
    }
}

constexpr int kSortLarge = kSegLen * kSegMax;   // beyond: bitonic network in place in global memory

}  // namespace gs

using namespace gs;

extern "C" int gs_bin_groups(int64_t N) { return bin_layout(1, N, 1).groups; }

// the depth round the list stages work for (gs_rounds_set): 1 front, 2 back, 0 none
static int list_round(int64_t*& rblk) {
    const Rounds R = current_rounds();
    const int phase = (R.phase == 1 || R.phase == 4) ? 1 : (R.phase == 2 ? 2 : 0);   // (4: the front round alone)
    rblk = phase ? R.blk : nullptr;
    return phase;
}

extern "C" size_t gs_bin_workspace_bytes(int C, int64_t N, int tile_w, int tile_h) {
    return bin_layout(C, N, tile_w * tile_h).total;
}

// (the attribute is raised once per kernel and size: nothing but launches reaches the stream afterwards, which
//  keeps a warmed-up pipeline capturable into a hipGraph)
static int ensure_lds(const void* fn, size_t bytes) {
    if (bytes > 64 * 1024) {
        static std::mutex mu;
        static std::map<std::pair<int, const void*>, size_t> done;
        int dev = 0;
        GS_HIP_CHECK(hipGetDevice(&dev));
        std::lock_guard<std::mutex> lock(mu);
        size_t& have = done[{dev, fn}];
        if (have < bytes) {
            GS_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
            have = bytes;
        }
    }
    return GS_OK;
}

#define GS_SORT1K_THREADS 256
#define GS_SORT4K_THREADS 256
// Size classes of the per-list sort (max_count: the longest list, known to the host, or an upper bound):
//   <= 1024 keys: one block per list, 256 threads, 24 KB of LDS
//   (1024, 4096] 72 KB | (4096, 8192] 768 threads, 152 KB | (8192, 65536] 8192-key segments + rank merge: over compacted
//   work lists (class_items_kernel), a fixed number of blocks striding over the items
//   beyond: bitonic network in place in global memory
static int launch_list_sorts(hipStream_t st, SortArgs& a, unsigned grid, int64_t max_tile_count, int32_t* items_ws) {
    int lo_excl = 0;
    int32_t* counts = items_ws;
    int32_t* items = items_ws + 64;
    const size_t radix_lds = 2 * sizeof(uint32_t) * 256;
    a.lo_excl = 0; a.hi_incl = 1024;
    if (max_tile_count > 0) {   // lists of up to 1024 keys: one block per list (24 KB of LDS: a block of another class costs nothing)
        const size_t lds_k = 2 * sizeof(uint64_t) * 1024 + radix_lds * (GS_SORT1K_THREADS / 64);
        hipLaunchKernelGGL(tile_radix_sort_kernel<GS_SORT1K_THREADS>, dim3(grid), dim3(GS_SORT1K_THREADS), lds_k, st, a);
        GS_LAUNCH_CHECK("tile_radix_sort_kernel");
    }
    if (max_tile_count <= 1024) return GS_OK;
    // (counts were zeroed by the scan kernel of the count stage: bin_tilescan_kernel block 2 / bins_scan_kernel block 1)
    hipLaunchKernelGGL(class_items_kernel, dim3((grid + 255) / 256), dim3(256), 0, st, (int)grid, a.isect_offsets, items, counts, a.guard, a.rblk, a.phase);
    GS_LAUNCH_CHECK("class_items_kernel");
    {   // (1024, 4096]: 72 KB -> two blocks per CU
        a.lo_excl = 1024; a.hi_incl = 4096;
        const size_t lds_k = 2 * sizeof(uint64_t) * 4096 + radix_lds * (GS_SORT4K_THREADS / 64);
        if (int rc = ensure_lds((const void*)tile_radix_sort_items_kernel<GS_SORT4K_THREADS>, lds_k)) return rc;
        hipLaunchKernelGGL(tile_radix_sort_items_kernel<GS_SORT4K_THREADS>, dim3(grid < 512u ? grid : 512u), dim3(GS_SORT4K_THREADS), lds_k, st, a,
                           (const int32_t*)items, (const int32_t*)counts);
        GS_LAUNCH_CHECK("tile_radix_sort_items_kernel<4096>");
    }
    if (max_tile_count > 4096) {   // (4096, 8192]: 152 KB -> one block per CU
        a.lo_excl = 4096; a.hi_incl = kSegLen;
        const size_t lds_k = 2 * sizeof(uint64_t) * (size_t)kSegLen + radix_lds * (768 / 64);
        if (int rc = ensure_lds((const void*)tile_radix_sort_items_kernel<768>, lds_k)) return rc;
        hipLaunchKernelGGL(tile_radix_sort_items_kernel<768>, dim3(grid < 256u ? grid : 256u), dim3(768), lds_k, st, a,
                           (const int32_t*)(items + grid), (const int32_t*)(counts + 1));
        GS_LAUNCH_CHECK("tile_radix_sort_items_kernel<8192>");
    }
    if (max_tile_count > kSegLen) {   // (8192, 65536]: segments sorted in LDS, then rank-merged
        a.lo_excl = kSegLen; a.hi_incl = kSortLarge; a.seg = 1;
        const size_t lds_k = 2 * sizeof(uint64_t) * (size_t)kSegLen + radix_lds * (768 / 64);
        if (int rc = ensure_lds((const void*)tile_radix_sort_items_kernel<768>, lds_k)) return rc;
        const int32_t* it = items + 2 * (size_t)grid;
        const int32_t* nit = counts + 2;
        hipLaunchKernelGGL(tile_radix_sort_items_kernel<768>, dim3(256), dim3(768), lds_k, st, a, it, nit);
        GS_LAUNCH_CHECK("tile_radix_sort_items_kernel<segments>");
        hipLaunchKernelGGL(seg_merge_kernel, dim3(512), dim3(1024), 0, st, a, it, nit);
        GS_LAUNCH_CHECK("seg_merge_kernel");
        if (a.coarse) {
            hipLaunchKernelGGL(seg_copy_kernel, dim3(512), dim3(1024), 0, st, a, it, nit);
            GS_LAUNCH_CHECK("seg_copy_kernel");
        }
        a.seg = 0;
    }
    if (max_tile_count > kSortLarge) {
        a.lo_excl = kSortLarge; a.hi_incl = 0x7fffffff;
        hipLaunchKernelGGL(tile_sort_kernel<false>, dim3(grid), dim3(1024), 0, st, a);
        GS_LAUNCH_CHECK("tile_sort_kernel<global>");
    }
    (void)lo_excl;
    return GS_OK;
}

extern "C" int gs_bin_count(void* stream, int C, int64_t N, int tile_w, int tile_h, const uint32_t* bbox,
                            void* workspace, size_t workspace_bytes, int32_t* isect_offsets,
                            int32_t* bucket_offsets, int32_t* tile_order, int64_t* info_dev, int64_t* info_host) {
    GS_REQUIRE(C >= 1 && N >= 0 && tile_w > 0 && tile_h > 0, "C>=1, N>=0, positive tile grid");
    const int tiles = tile_w * tile_h;
    GS_REQUIRE((size_t)tiles * 4 + 128 <= 160 * 1024, "tile grid too large for the LDS histogram (max 40928 tiles per camera)");
    GS_REQUIRE((int64_t)C * tiles < (1ll << 31), "too many tiles");
    const BinLayout L = bin_layout(C, N, tiles);
    GS_REQUIRE(workspace && workspace_bytes >= L.total, "workspace too small (see gs_bin_workspace_bytes)");
    GS_REQUIRE(isect_offsets && bucket_offsets && info_dev, "null output pointer");
    GS_REQUIRE(N == 0 || bbox, "null bbox");
    int64_t* rblk;
    const int phase = list_round(rblk);
    GS_REQUIRE(phase == 0 || C == 1, "depth rounds: one camera per call");
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    uint32_t* hist = (uint32_t*)(ws + L.hist_off);
    uint32_t* tile_cnt = (uint32_t*)(ws + L.tile_cnt_off);
    uint32_t* grp_tot = (uint32_t*)(ws + L.grp_tot_off);
    uint32_t* grp_base = (uint32_t*)(ws + L.grp_base_off);
    const size_t lds = sizeof(uint32_t) * ((size_t)tiles + 32);
    if (int rc = ensure_lds((const void*)bin_hist_kernel, lds)) return rc;
    hipLaunchKernelGGL(bin_hist_kernel, dim3(L.groups, C), dim3(kBinThreads), lds, st, N, tile_w, tiles,
                       L.per_group, (const uint4*)bbox, hist, grp_tot, (const int64_t*)rblk, phase);
    GS_LAUNCH_CHECK("bin_hist_kernel");
    const int64_t ct = (int64_t)C * tiles;
    hipLaunchKernelGGL(bin_colscan_kernel, dim3((unsigned)((ct + kColTiles - 1) / kColTiles)), dim3(kColTiles * kColChunks), 0, st, C, L.groups,
                       tiles, hist, tile_cnt);
    GS_LAUNCH_CHECK("bin_colscan_kernel");
    const Guard gd = current_guard();
    GS_REQUIRE(gd.info == nullptr || gd.info == info_dev, "the guard set by gs_guard_set must be this call's info_dev");
    hipLaunchKernelGGL(bin_tilescan_kernel, dim3(3), dim3(kBinThreads), 0, st, (int)ct, C * L.groups, tile_cnt,
                       grp_tot, isect_offsets, bucket_offsets, grp_base, info_dev, tile_order, gd.cap_isects, gd.cap_tile,
                       (int64_t)0, (int32_t*)(ws + L.items_off), current_info_mirror(), gd.per_call, rblk, phase);
    GS_LAUNCH_CHECK("bin_tilescan_kernel");
    if (info_host) {
        GS_HIP_CHECK(hipMemcpyAsync(info_host, info_dev, 4 * sizeof(int64_t), hipMemcpyDeviceToHost, st));
        GS_HIP_CHECK(hipStreamSynchronize(st));
    }
    return GS_OK;
}

extern "C" int gs_bin_emit_sort(void* stream, int C, int64_t N, int tile_w, int tile_h, const uint32_t* bbox,
                                const float* depths, void* workspace, size_t workspace_bytes,
                                const int32_t* isect_offsets, int64_t n_isects, int64_t max_tile_count,
                                uint64_t* keys_tmp, int32_t* slot_gid, int32_t* cum_tiles,
                                int64_t* isect_ids, int32_t* flatten_ids, int32_t* slots) {
    GS_REQUIRE(C >= 1 && N >= 0 && tile_w > 0 && tile_h > 0, "C>=1, N>=0, positive tile grid");
    const int tiles = tile_w * tile_h;
    const BinLayout L = bin_layout(C, N, tiles);
    GS_REQUIRE(workspace && workspace_bytes >= L.total, "workspace too small (see gs_bin_workspace_bytes)");
    GS_REQUIRE(n_isects >= 0 && n_isects < (1ll << 31), "intersection count must fit int32");
    if (N == 0) return GS_OK;
    GS_REQUIRE(bbox && depths && isect_offsets && cum_tiles, "null pointer");
    GS_REQUIRE(n_isects == 0 || (keys_tmp && flatten_ids), "null intersection buffer");
    GS_REQUIRE((slot_gid == nullptr) == (slots == nullptr), "slot_gid and slots: both (training lists) or neither (inference lists)");
    int64_t* rblk;
    const int phase = list_round(rblk);
    GS_REQUIRE(phase == 0 || C == 1, "depth rounds: one camera per call");
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    const uint32_t* hist = (const uint32_t*)(ws + L.hist_off);
    const uint32_t* grp_base = (const uint32_t*)(ws + L.grp_base_off);
    const size_t lds = sizeof(uint32_t) * ((size_t)tiles + 32);
    if (int rc = ensure_lds((const void*)bin_emit_kernel, lds)) return rc;
    hipLaunchKernelGGL(bin_emit_kernel, dim3(L.groups, C), dim3(kBinThreads), lds, st, N, tile_w, tiles,
                       L.per_group, (const uint4*)bbox, depths, hist, isect_offsets, grp_base,
                       (unsigned long long*)keys_tmp, slot_gid, cum_tiles, current_guard().info, (const int64_t*)rblk, phase);
    GS_LAUNCH_CHECK("bin_emit_kernel");
    if (n_isects == 0) return GS_OK;
    SortArgs a;
    a.tiles = tiles;
    int tb = 0;
    for (int v = tiles; v > 0; v >>= 1) ++tb;
    a.tile_bits = tb;
    a.isect_offsets = isect_offsets; a.keys = (unsigned long long*)keys_tmp; a.slot_gid = slot_gid;
    a.isect_ids = isect_ids; a.flatten_ids = flatten_ids; a.slots = slots;
    a.guard = current_guard().info;
    a.coarse = 0; a.seg = 0; a.merged = nullptr; a.rblk = rblk; a.phase = phase;
    static_assert(2 + kSegMax == 10, "bin_layout sizes the work lists for kSegMax == 8");
    return launch_list_sorts(st, a, (unsigned)(C * tiles), max_tile_count, (int32_t*)(ws + L.items_off));
}

// ------------------------------------------------------------------------------------------------ two-level binning
extern "C" size_t gs_bins_workspace_bytes(int C, int64_t N, int tile_w, int tile_h, int bin_shift, int64_t coarse_cap) {
    if (bin_shift < 0 || bin_shift > 2 || coarse_cap < 0) return 0;
    return bins_layout(C, N, tile_w, tile_h, bin_shift, coarse_cap).total;
}

template <bool WRITE>
static int launch_refine(hipStream_t st, const BinsLayout& L, const RefineArgs& a) {
    const unsigned grid = (unsigned)L.max_chunks;   // upper bound; blocks past the last chunk return at once
    if (L.shift == 2) hipLaunchKernelGGL((bins_refine_kernel<WRITE, 2>), dim3(grid), dim3(1024), 0, st, a);
    else hipLaunchKernelGGL((bins_refine_kernel<WRITE, 1>), dim3(grid), dim3(256), 0, st, a);
    GS_LAUNCH_CHECK("bins_refine_kernel");
    return GS_OK;
}

static void fill_refine_args(RefineArgs& r, const BinsLayout& L, int C, int tile_w, int tile_h, char* ws, const uint64_t* coarse_keys,
                             const uint32_t* bbox, const int64_t* info_dev) {
    r.tw = tile_w; r.th = tile_h; r.tiles = tile_w * tile_h; r.n_bins_total = C * L.nbins;
    r.choff = (const int32_t*)(ws + L.choff_off);
    r.chunk_desc = (const int4*)(ws + L.chunk_bin_off);
    r.cnt_ct = (uint32_t*)(ws + L.cnt_ct_off);
    int tb = 0;
    for (int v = r.tiles; v > 0; v >>= 1) ++tb;
    r.tile_bits = tb; r.bw = L.bw; r.nbins = L.nbins;
    r.coff = (const int32_t*)(ws + L.coff_off);
    r.keys = (const unsigned long long*)coarse_keys;
    r.rec = (const uint4*)(ws + L.rec_off);
    r.staged = (uint4*)(ws + L.staged_off);
    r.tile_cnt = (uint32_t*)(ws + L.tile_cnt_off);
    r.isect_offsets = nullptr; r.isect_ids = nullptr; r.flatten_ids = nullptr; r.slots = nullptr;
    r.info = info_dev;
    int64_t* rblk;
    r.phase = list_round(rblk);
    r.rblk = rblk;
}

extern "C" int gs_bins_count(void* stream, int C, int64_t N, int tile_w, int tile_h, int bin_shift, const uint32_t* bbox,
                             const float* depths, void* workspace, size_t workspace_bytes, uint64_t* coarse_keys,
                             int64_t coarse_cap, int64_t coarse_list_cap,
                             int32_t* cum_tiles, int32_t* isect_offsets, int32_t* bucket_offsets, int32_t* tile_order,
                             int64_t* info_dev, int64_t* info_host) {
    GS_REQUIRE(C >= 1 && N >= 0 && tile_w > 0 && tile_h > 0, "C>=1, N>=0, positive tile grid");
    const int tiles = tile_w * tile_h;
    GS_REQUIRE((int64_t)C * tiles < (1ll << 31) && (int64_t)C * N < (1ll << 31), "too many tiles or flatten ids");
    GS_REQUIRE(bin_shift >= 0 && bin_shift <= 2, "bin_shift: 0 (library default), 1 (2x2 tiles) or 2 (4x4 tiles)");
    GS_REQUIRE(coarse_cap >= 0 && coarse_cap < (1ll << 31) && (coarse_cap == 0 || coarse_keys), "coarse key buffer");
    const BinsLayout L = bins_layout(C, N, tile_w, tile_h, bin_shift, coarse_cap);
    GS_REQUIRE((size_t)L.nbins * 4 + 128 <= 160 * 1024, "tile grid too large for the LDS bin histogram");
    GS_REQUIRE(workspace && workspace_bytes >= L.total, "workspace too small (see gs_bins_workspace_bytes)");
    GS_REQUIRE(isect_offsets && bucket_offsets && info_dev, "null output pointer");
    GS_REQUIRE(N == 0 || (bbox && depths && cum_tiles), "null bbox / depths / cum_tiles");
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    uint32_t* hist = (uint32_t*)(ws + L.hist_off);
    uint32_t* bin_cnt = (uint32_t*)(ws + L.bin_cnt_off);
    uint32_t* grp_tot = (uint32_t*)(ws + L.grp_tot_off);
    uint32_t* grp_base = (uint32_t*)(ws + L.grp_base_off);
    int32_t* coff = (int32_t*)(ws + L.coff_off);
    uint32_t* tile_cnt = (uint32_t*)(ws + L.tile_cnt_off);
    const Guard gd = current_guard();
    GS_REQUIRE(gd.info == nullptr || gd.info == info_dev, "the guard set by gs_guard_set must be this call's info_dev");
    int64_t* rblk;
    const int phase = list_round(rblk);
    GS_REQUIRE(phase == 0 || C == 1, "depth rounds: one camera per call");
    // sort classes launched: up to the one that holds coarse_list_cap (<= 0: all of them); a longer bin list raises
    // flags bit 8 and nothing is emitted
    int64_t list_cap = 0x7fffffff;
    if (coarse_list_cap > 0) list_cap = coarse_list_cap <= 1024 ? 1024 : coarse_list_cap <= 4096 ? 4096 : coarse_list_cap <= kSegLen ? kSegLen
                                        : coarse_list_cap <= kSortLarge ? kSortLarge : 0x7fffffff;
    const size_t lds = sizeof(uint32_t) * ((size_t)L.nbins + 32);
    if (int rc = ensure_lds((const void*)bins_hist_kernel, lds)) return rc;
    hipLaunchKernelGGL(bins_hist_kernel, dim3(L.groups, C), dim3(kBinThreads), lds, st, N, L.shift, L.bw, L.nbins, L.per_group,
                       (const uint4*)bbox, hist, grp_tot, (const int64_t*)rblk, phase);
    GS_LAUNCH_CHECK("bins_hist_kernel");
    const int64_t cb = (int64_t)C * L.nbins;
    hipLaunchKernelGGL(bins_colscan_kernel, dim3((unsigned)((cb + 3) / 4)), dim3(256), 0, st, C, L.groups, L.nbins, hist, bin_cnt);
    GS_LAUNCH_CHECK("bins_colscan_kernel");
    hipLaunchKernelGGL(bins_scan_kernel, dim3(2), dim3(kBinThreads), 0, st, (int)cb, C * L.groups, bin_cnt, grp_tot, coff,
                       (int32_t*)(ws + L.choff_off), (int4*)(ws + L.chunk_bin_off), L.max_chunks, L.chunk_shift, grp_base,
                       info_dev, coarse_cap, list_cap,
                       (gd.info != nullptr && !gd.per_call) ? 1 : 0, (int32_t*)(ws + L.items_off), (const int64_t*)rblk, phase);   // (per-call guard: overwrite)
    GS_LAUNCH_CHECK("bins_scan_kernel");
    if (N > 0) {
        if (int rc = ensure_lds((const void*)bins_emit_kernel, lds)) return rc;
        hipLaunchKernelGGL(bins_emit_kernel, dim3(L.groups, C), dim3(kBinThreads), lds, st, N, L.shift, L.bw, L.nbins, L.per_group,
                           (const uint4*)bbox, depths, hist, coff, grp_base, (unsigned long long*)coarse_keys, cum_tiles,
                           (uint4*)(ws + L.rec_off), (const int64_t*)info_dev, (const int64_t*)rblk, phase);
        GS_LAUNCH_CHECK("bins_emit_kernel");
        SortArgs a;
        a.tiles = L.nbins; a.tile_bits = 0;
        a.isect_offsets = coff; a.keys = (unsigned long long*)coarse_keys; a.slot_gid = nullptr;
        a.isect_ids = nullptr; a.flatten_ids = nullptr; a.slots = nullptr;
        a.guard = info_dev; a.coarse = 1; a.seg = 0; a.rblk = rblk; a.phase = phase;
        a.merged = (unsigned long long*)(ws + L.staged_off);   // (free until the refinement's first pass fills it)
        // (the longest bin list is not known to the host here: every class is launched, the blocks of the classes a
        //  list does not belong to return at once; a list can never be longer than the key buffer)
        if (int rc = launch_list_sorts(st, a, (unsigned)cb, list_cap < coarse_cap ? list_cap : coarse_cap, (int32_t*)(ws + L.items_off))) return rc;
    }
    RefineArgs r;
    fill_refine_args(r, L, C, tile_w, tile_h, ws, coarse_keys, bbox, info_dev);
    if (int rc = launch_refine<false>(st, L, r)) return rc;
    {
        const unsigned g = (unsigned)(((int64_t)C * tiles + 255) / 256);
        if (L.shift == 2) hipLaunchKernelGGL(bins_chunkscan_kernel<2>, dim3(g), dim3(256), 0, st, r, C);
        else hipLaunchKernelGGL(bins_chunkscan_kernel<1>, dim3(g), dim3(256), 0, st, r, C);
        GS_LAUNCH_CHECK("bins_chunkscan_kernel");
    }
    hipLaunchKernelGGL(bin_tilescan_kernel, dim3(2), dim3(kBinThreads), 0, st, C * tiles, 0, tile_cnt, (const uint32_t*)nullptr,
                       isect_offsets, bucket_offsets, (uint32_t*)nullptr, info_dev, tile_order, gd.cap_isects,
                       (int64_t)0x7fffffffffffffffll, (int64_t)12, (int32_t*)nullptr, current_info_mirror(), gd.per_call, rblk, phase);
    GS_LAUNCH_CHECK("bin_tilescan_kernel");
    if (info_host) {
        GS_HIP_CHECK(hipMemcpyAsync(info_host, info_dev, 8 * sizeof(int64_t), hipMemcpyDeviceToHost, st));
        GS_HIP_CHECK(hipStreamSynchronize(st));
    }
    return GS_OK;
}

extern "C" int gs_bins_lists(void* stream, int C, int64_t N, int tile_w, int tile_h, int bin_shift, const uint32_t* bbox, void* workspace,
                             size_t workspace_bytes, const uint64_t* coarse_keys, int64_t coarse_cap, const int32_t* cum_tiles,
                             const int32_t* isect_offsets, int64_t* isect_ids, int32_t* flatten_ids, int32_t* slots,
                             const int64_t* info_dev) {
    GS_REQUIRE(C >= 1 && N >= 0 && tile_w > 0 && tile_h > 0, "C>=1, N>=0, positive tile grid");
    GS_REQUIRE(bin_shift >= 0 && bin_shift <= 2 && coarse_cap >= 0, "bin_shift: 0, 1 or 2; coarse_cap as given to gs_bins_count");
    const BinsLayout L = bins_layout(C, N, tile_w, tile_h, bin_shift, coarse_cap);
    GS_REQUIRE(workspace && workspace_bytes >= L.total, "workspace too small (see gs_bins_workspace_bytes)");
    GS_REQUIRE(info_dev && isect_offsets, "null pointer");
    if (N == 0) return GS_OK;
    GS_REQUIRE(cum_tiles && bbox && flatten_ids, "null list buffer");
    RefineArgs r;
    fill_refine_args(r, L, C, tile_w, tile_h, (char*)workspace, coarse_keys, bbox, info_dev);
    r.isect_offsets = isect_offsets; r.isect_ids = isect_ids; r.flatten_ids = flatten_ids; r.slots = slots;
    return launch_refine<true>((hipStream_t)stream, L, r);
}
