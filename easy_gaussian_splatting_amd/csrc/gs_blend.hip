// gs_blend.hip -- alpha-blend forward and backward for gfx950.
//
// Forward: ONE wavefront per 16x16 tile, four pixels per lane -- lane (lx, ly) of an 8x8 grid
// owns the pixel at that position in each of the tile's four 8x8 QUADRANTS.  The tile's
// depth-sorted list is walked in buckets of 64 entries: each lane gathers one packed 48-byte
// record, tests its opacity-aware extent against the four quadrants and the wave ballots the
// results into four 64-bit scalar masks (a quadrant whose 64 pixels are all saturated drops out).
// The inner loop then runs on scalar control flow only: it visits just the entries that can touch
// the tile at all (s_ff1 over the OR of the masks), reads the record back at a wave-uniform LDS
// address (broadcast) and evaluates only the quadrants whose bit is set.  The per-pixel
// arithmetic is straight-line predicated code (v_cndmask, no divergent branches).
//
// In training mode the forward also emits, per quadrant, the COMPACTED depth-ordered sublist of
// the entries that touch it (ranks from popcounts of the ballots), a checkpoint of the quadrant's
// 64 pixel states (T, accumulated rgb) every 32 sublist entries, and one work-unit descriptor per
// such 32-entry work unit.  All three live in storage allocated PER WORK UNIT as the walk opens it
// (round 6): a tile takes chunks of kChunk storage units from one device counter, a unit's checkpoint
// is row `storage` of ckpt, its 32 sublist pairs block `storage` of qlist -- what the training forward
// leaves scales with what it WALKED, not with what is listed (a saturated tile abandons the rest of
// its list; on realistic footprints 1-3 % of the listed entries are ever walked).  A last pass scans
// the quadrant masks into row_base[slot]: the gradient rows of the backward are compact as well.
//
// Backward: GAUSSIAN-parallel, no cross-lane reductions, no atomics.  A wavefront runs eight
// independent 8-lane systolic pipelines (two per 16-lane DPP row); a pipeline owns one work unit =
// 32 consecutive entries of a quadrant sublist: lane r keeps entries 4r..4r+3 and their 4x11 gradient
// sums in registers while the quadrant's 64 pixels stream through -- at step s lane r treats pixel
// s-r against its four entries in depth order, receives that pixel's running (T, P = prefix colour .
// v_colour) from lane r-1 through one DPP row_shr:1 each and hands it on; the pipeline's head lane
// is fed from the checkpoint.  71 steps cover 64 pixels x 32 entries (10 % pipeline fill; the 16-lane,
// 64-entry form it replaces paid 19 % and wasted half a bucket per sublist on average instead of a
// quarter), and only (entry, quadrant) pairs the forward actually walked are ever evaluated.
// Each lane finally stores 48-byte gradient rows at rows[row_base[slot] + rank of the quadrant among
// the slot's existing rows]: the rows of a Gaussian (contiguous slots) are contiguous, and so are the
// rows of consecutive Gaussians -- gs_project_bwd sums them with plain coalesced loads.
//
// Forward and backward evaluate alpha, w = alpha*T and T' = fma(-alpha, T, T) with the same
// instruction sequence on the same inputs, so the backward re-derives the forward's contributor
// set exactly (no last_ids).
//
// Semantics: SURVEY.md Appendix A.4 / A.5 (gsplat 1.0.0 rasterize_to_pixels fwd/bwd).
#include <algorithm>

#include "gs_common.h"
#include "gs_math.h"

namespace gs {

// -DGS_BWD_CHECK (tests/test_gpu_contributors.py; never in the product build): the invariant of the comment on top -- the
// backward re-derives the forward's contributor set exactly, with no last_ids -- made observable.  The training forward also
// leaves every pixel's FINAL transmittance and its number of contributors; every pipeline of the backward leaves, per (work
// unit, pixel), the transmittance behind the unit's last contributing entry (-1: the pixel was finished before the unit) and
// the number of entries of the unit the pixel took.  Chained over a sublist's units the two must agree bit for bit.
#ifdef GS_BWD_CHECK
struct BwdCheck {
    float* fwd_T;        // [C*H*W]
    int32_t* fwd_cnt;    // [C*H*W]
    float2* unit_out;    // [units][64]  (T behind the unit, contributors inside the unit)
    int2* unit_hdr;      // [units]      (tile * 4 + quadrant, position of the unit in its sublist)
};
static BwdCheck g_bwd_check = {nullptr, nullptr, nullptr, nullptr};
#define GS_IF_CHECK(...) __VA_ARGS__
#else
#define GS_IF_CHECK(...)
#endif

// Work unit of the backward: kUnit consecutive entries of one quadrant sublist (half a 64-entry bucket), with a
// checkpoint of the quadrant's 64 pixel states in front of it.
constexpr int kUnit = GS_UNIT;
// Storage units a tile takes from its range's counter at a time (one returning atomic per kChunk work units; what a tile
// leaves unused of its last chunk costs memory only -- the work-unit descriptors are published densely, see unit log).
// Round 6, same box, blend_fwd stage at 1 M / 1080p: ONE counter for all tiles, chunks of 8 / 32 / 64: 0.47 / 0.395 / 0.398 ms
// (30 k / 10 k / 8 k same-address device-scope atomics per frame) -- hence GS_WALK_RANGES counters, a cache line each.
constexpr int kChunk = 8;
// Work units a tile-wave has opened and not yet published, in LDS: (storage unit, position in the sublist * 4 + quadrant).
// Published -- one atomic for the lot, descriptors dense in launch order -- when the log fills up and at the end of the tile.
constexpr int kUnitLog = 128;
// words of the walk state (include/gs_raster.h: GS_WALK_*)
constexpr int kWalkUnits = GS_WALK_UNITS, kWalkStorage = GS_WALK_STORAGE, kWalkRows = GS_WALK_ROWS,
              kWalkFlags = GS_WALK_FLAGS, kWalkSkip = GS_WALK_SKIP;
constexpr int kScanThreads = 256, kScanPerThread = 32, kScanChunk = kScanThreads * kScanPerThread;   // 8192 slots per block

struct BlendFwdArgs {
    const int32_t* tile_order;   // optional (gs_bin_count)
    int C, W, H, tw, tiles;
    const float4* rec;
    const float* bg;
    const int32_t *isect_offsets, *flatten_ids, *slots;
    float *out_colors, *out_alphas;
    // training-mode outputs
    float4* ckpt;          // [cap_units][64]    the pixel states in front of a work unit
    int2* qlist;           // [cap_units][32]    (flatten id, row slot) pairs of a work unit, in sublist order
    int32_t* qcnt;         // [C*tiles*4]        sublist lengths
    uint8_t* qmask;        // [I] by slot        which quadrant rows of an intersection exist
    int4* unit_desc;       // [cap_units]        (tile*4+quadrant, position of the unit in its sublist, storage unit, 0)
    int32_t* walk;         // walk state: counters (GS_WALK_*) + the chunk counts of the row-base scan
    int cap_units;
    unsigned long long* flags;   // the guard's flag word (overflow bits are ORed in) or nullptr
    const int64_t* guard;  // step guard (gs_guard_set) or nullptr
    // depth rounds (gs_rounds_set; blend_fwd_kernel's ROUND 1 / 2)
    int64_t* rblk;
    uint8_t* live;         // [tiles]
    float4* tstate;        // [tiles][4][64]
    int4* trec;            // [tiles][2]  {sublist lengths}, {part-filled work units' storage}
    int solo;              // front round alone (gs_rounds_set phase 4): a tile left live voids the step instead of leaving its state
    GS_IF_CHECK(BwdCheck chk;)
};

// -DGS_CLOCK_PROBE (tools/clock_probe.sh; never in the product build): every wave of the two blend kernels adds the shader-clock
// cycles (s_memtime) and the constant-rate ticks (s_memrealtime) between its first and last instruction to two device
// words -- their ratio is the shader clock the kernel actually ran at (profiles/r04_clock.json).
#ifdef GS_CLOCK_PROBE
__device__ unsigned long long gs_clk_acc[4];   // {fwd cycles, fwd ticks, bwd cycles, bwd ticks}
struct ClockProbe {
    long long c0, w0; int slot;
    __device__ ClockProbe(int s) : c0(clock64()), w0(wall_clock64()), slot(s) {}
    __device__ ~ClockProbe() {
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&gs_clk_acc[slot], (unsigned long long)(clock64() - c0));
            atomicAdd(&gs_clk_acc[slot + 1], (unsigned long long)(wall_clock64() - w0));
        }
    }
};
#define GS_CLOCK_PROBE_SCOPE(slot) ClockProbe clock_probe_(slot)
#else
#define GS_CLOCK_PROBE_SCOPE(slot)
#endif

// -DGS_EXACT_MATH (tools/acc64_ab.py; never in the product build): correctly rounded exp2 and division in BOTH blend kernels
// instead of the hardware approximations (v_exp_f32 / v_rcp_f32: ~1 ulp) -- the other half of the A/B on rows that cancel.
#ifdef GS_EXACT_MATH
__device__ __forceinline__ float fast_exp2(float x) { return (float)exp2((double)x); }
__device__ __forceinline__ float fast_rcp(float x) { return (float)(1.0 / (double)x); }
#else
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
#endif

// One (pixel, Gaussian) pair of the forward (alpha and the take-it test come from the caller, which
// skips entries no pixel takes); straight-line, predicated.  The entry's alpha is
// masked once (v_cndmask costs more than an fp32 multiply on gfx950, tools/micro/): a masked alpha
// of 0 leaves T and the colour sums bit-for-bit unchanged (fma(-0, T, T) == T, fma(r, 0, c) == c).
// A pixel's state is ONE register: T in (1e-4, 1] = live with transmittance T; T > 2^32 = finished (the stop rule fired, or
// the pixel lies outside the image) with final transmittance T * 2^-64 -- the scaling by a power of two is exact, and it
// makes every test a single compare: "live" is T <= 1, and the stop test Tn <= 1e-4 needs no mask, because a finished
// pixel's Tn = fma(-0, T, T) = T is huge and a live pixel that does not take the entry keeps Tn = T > 1e-4.  (Round 3: a
// separate per-lane flag cost an and + compare + mask inversion per pair and two selects per stop; a sign flag still needed
// `ok && Tn <= 1e-4`, which hipcc turns into a select + compare + s_nop in front of the ballot.)
// The stop rule (T would fall to 1e-4: the entry is NOT blended, the pixel is finished) fires a
// handful of times per pixel at most, so it lives behind a wave-uniform branch on the ballot of that one compare.
constexpr float kDoneScale = 18446744073709551616.f;   // 2^64
constexpr float kDoneInv = 5.421010862427522e-20f;      // 2^-64
__device__ __forceinline__ bool px_live(float T) { return T <= 1.f; }
__device__ __forceinline__ float px_final_T(float T) { return T > 2.f ? T * kDoneInv : T; }
__device__ __forceinline__ void blend_pair(const float alpha, const bool ok, const float r,
                                           const float g, const float b, float& T, float& cr,
                                           float& cg, float& cb) {
    const float am = ok ? alpha : 0.f;
    float w = am * T;
    float Tn = fmaf(-am, T, T);   // explicit: fwd and bwd must round identically
    const bool stop = Tn <= kTMin;
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(stop) != 0ull, 0)) {
        w = stop ? 0.f : w;
        Tn = stop ? T * kDoneScale : Tn;
    }
    cr = fmaf(r, w, cr); cg = fmaf(g, w, cg); cb = fmaf(b, w, cb);
    T = Tn;
}

// The training instantiation is held to 5 waves / SIMD (the 8160 tile-waves of a 1080p frame then run in 1.6 rounds); held
// to 80 VGPRs (5 spilled to scratch) it keeps 6 resident and is 2-3 % faster back to back -- but a kernel
// that needs SCRATCH makes the runtime (re-)provision scratch memory for the queue it is launched on: once another
// stream of the process had run the captured step, every eager launch of this kernel on the caller's stream stalled 0.5-2 ms
// behind that (bench.py's eager stage profile read 0.76-2.4 ms for a 0.28 ms kernel on some boxes; HISTORY.md section 8b).
// No kernel of the library uses scratch (tests/test_capi.py checks the compiler's resource report).  Inference: 8 waves.
#define GS_FWD_ATTR __attribute__((amdgpu_waves_per_eu(CKPT ? 5 : 1, 8)))

// The number of list entries the training passes over the slots cover: the caller's n (a capacity under the step guard), cut
// to the count the tile scan left in the guard's info block.
__device__ __forceinline__ int64_t slots_in_use(int64_t n, const int64_t* guard) {
    return guard != nullptr ? min(n, guard[0]) : n;
}

// Streaming clear of the quadrant masks (16-byte stores; `n` bytes from an arbitrarily aligned pointer) + of the walk state
// (counters; the chunk counts of the row-base scan are written before they are read).
__global__ __launch_bounds__(256) void qmask_clear_kernel(uint8_t* __restrict__ q, int64_t n_cap, int32_t* __restrict__ walk, int walk_ints,
                                                          const int64_t* __restrict__ guard, const int64_t* __restrict__ rblk, int phase) {
    // The step guard as it stands when the CALL starts decides for all three of its kernels (walk[kWalkSkip]): the overflow flags
    // this call's own blend raises must not stop its later tiles (the image stays complete) nor its scan (which reports what
    // the walk needed).  Depth rounds: the two calls of a frame are one walk -- the back round goes on unless the front round
    // was skipped or the list stages of the back round overflowed (the front round's own blend does not touch the guard).
    bool skip = guard_tripped(guard);
    if (phase == 2) skip = skip || walk[kWalkSkip] != 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) walk[kWalkSkip] = skip ? 1 : 0;
    if (skip) return;
    // the slots this call's lists own: all of them, or (back round) those behind the front round's
    const int64_t first = phase == 2 ? min(rblk[GS_ROUND_BASE], slots_in_use(n_cap, guard)) : 0;
    const int64_t n = slots_in_use(n_cap, guard) - first;
    q += first;
    if (phase != 2)
        for (int i = (int)blockIdx.x * 256 + (int)threadIdx.x; i < walk_ints; i += (int)gridDim.x * 256)
            if (i != kWalkSkip) walk[i] = 0;
    const int64_t head = min(n, (int64_t)((16 - ((uintptr_t)q & 15)) & 15));
    const int64_t n16 = (n - head) >> 4;
    uint4* q16 = reinterpret_cast<uint4*>(q + head);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) q16[i] = make_uint4(0u, 0u, 0u, 0u);
    if (blockIdx.x == 0) {
        if ((int64_t)threadIdx.x < head) q[threadIdx.x] = 0;
        const int64_t tail0 = head + (n16 << 4);
        if (tail0 + (int64_t)threadIdx.x < n) q[tail0 + threadIdx.x] = 0;
    }
}

// ROUND (depth rounds, include/gs_raster.h): 0 = the frame's one list per tile; 1 = front round: a tile that still has live pixels
// behind its list leaves them (and, training, where its quadrant sublists stand) for the back round; 2 = back round: only such
// tiles, from those states -- the same walk as over one list, cut in two.
template <bool CKPT, int WAVES, int ROUND>
__global__ __launch_bounds__(64 * WAVES) GS_FWD_ATTR void blend_fwd_kernel(const BlendFwdArgs a) {
    __shared__ float4 srec_all[WAVES][GS_BUCKET * 3];
    __shared__ int2 ulog_all[CKPT ? WAVES : 1][CKPT ? kUnitLog : 1];
    float4* srec = srec_all[threadIdx.x >> 6];
    int2* ulog = ulog_all[CKPT ? (threadIdx.x >> 6) : 0];
    const int ti = (int)blockIdx.x * WAVES + (int)(threadIdx.x >> 6);
    if (ti >= a.C * a.tiles) return;   // wave-uniform
    if (CKPT ? a.walk[kWalkSkip] != 0 : guard_tripped(a.guard)) return;   // (training: the guard at the start of the call, qmask_clear_kernel)
    if (ROUND == 2 && a.rblk[GS_ROUND_LIVE] == 0) return;   // the front round finished every tile
    GS_CLOCK_PROBE_SCOPE(0);
    const int t = a.tile_order ? a.tile_order[ti] : ti;   // launch slot -> tile (longest lists first)
    const int cam = t / a.tiles, tt = t - cam * a.tiles;
    const int tyi = tt / a.tw, txi = tt - tyi * a.tw;
    const int lane = threadIdx.x & 63;
    const int x0 = txi * GS_TILE, y0 = tyi * GS_TILE;
    const int px0 = x0 + (lane & 7), py0 = y0 + (lane >> 3);   // quadrant 0 pixel; +8 for the others
    const float fx0 = (float)px0 + 0.5f, fy0 = (float)py0 + 0.5f, fx1 = fx0 + 8.f, fy1 = fy0 + 8.f;
    const int lo = a.isect_offsets[t], hi = a.isect_offsets[t + 1];
    const int len = hi - lo;
    const int nb = (len + GS_BUCKET - 1) / GS_BUCKET;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    // back round: a tile the front round finished, or one with nothing behind the split, keeps what the front round wrote
    if (ROUND == 2 && (len == 0 || a.live[t] == 0)) return;   // wave-uniform

    float T[4], cr[4], cg[4], cb[4];   // T <= 1: live; T > 2^32: finished, T * 2^-64 final (blend_pair)
    int cnt[4] = {0, 0, 0, 0};   // wave-uniform sublist lengths
    // storage of the work units a bucket's sublist entries can fall into (wave-uniform): [k][0] the unit that holds position
    // cnt[k] at the start of the bucket (carried over when it is part-filled), [k][1..2] the units opened behind it
    int us[4][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    int pool_next = 0, pool_left = 0, log_n = 0;   // this tile's chunk of storage units; unpublished work units in `ulog`
    // the storage range this tile draws from: 1/32 of [0, cap_units) with a counter in a cache line of its own (launch slots
    // take the ranges in turn: the tiles come longest list first, every range sees the same mix)
    const int range_len = (a.cap_units / GS_WALK_RANGES) & ~(kChunk - 1), range0 = (ti & (GS_WALK_RANGES - 1)) * range_len;
    int32_t* range_ctr = a.walk + GS_WALK_RANGE0 + 32 * (ti & (GS_WALK_RANGES - 1));
    GS_IF_CHECK(int taken[4] = {0, 0, 0, 0};)   // per pixel: entries blended
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        cr[k] = cg[k] = cb[k] = 0.f;
        T[k] = ((px0 + 8 * (k & 1)) < a.W && (py0 + 8 * (k >> 1)) < a.H) ? 1.f : kDoneScale;
    }
    if (ROUND == 2) {   // the states the front round left, in its lanes' own layout
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float4 s4 = a.tstate[((size_t)t * 4 + k) * 64 + lane];
            T[k] = s4.x; cr[k] = s4.y; cg[k] = s4.z; cb[k] = s4.w;
        }
        if (CKPT) {
            const int4 rc = a.trec[2 * (size_t)t], ru = a.trec[2 * (size_t)t + 1];
            cnt[0] = __builtin_amdgcn_readfirstlane(rc.x); cnt[1] = __builtin_amdgcn_readfirstlane(rc.y);
            cnt[2] = __builtin_amdgcn_readfirstlane(rc.z); cnt[3] = __builtin_amdgcn_readfirstlane(rc.w);
            us[0][0] = __builtin_amdgcn_readfirstlane(ru.x); us[1][0] = __builtin_amdgcn_readfirstlane(ru.y);
            us[2][0] = __builtin_amdgcn_readfirstlane(ru.z); us[3][0] = __builtin_amdgcn_readfirstlane(ru.w);
        }
    }
    // pixel-centre bounds of the four quadrants
    const float qxlo[2] = {(float)x0 + 0.5f, (float)x0 + 8.5f}, qxhi[2] = {(float)x0 + 7.5f, (float)x0 + 15.5f};
    const float qylo[2] = {(float)y0 + 0.5f, (float)y0 + 8.5f}, qyhi[2] = {(float)y0 + 7.5f, (float)y0 + 15.5f};

    // publishes the logged work units: one atomic for the lot, descriptors dense behind it
    auto publish = [&]() {
        if (log_n == 0) return;
        int base = 0;
        if (lane == 0) base = atomicAdd(a.walk + kWalkUnits, log_n);
        base = __builtin_amdgcn_readfirstlane(base);
        __builtin_amdgcn_wave_barrier();   // (the log was written by lane 0; LDS operations of one wave complete in issue order)
        for (int j = lane; j < log_n; j += 64) {
            const int2 e = ulog[j];
            if (base + j < a.cap_units) a.unit_desc[base + j] = make_int4(4 * t + (e.y & 3), e.y >> 2, e.x, 0);
        }
        __builtin_amdgcn_wave_barrier();
        log_n = 0;
    };

    for (int b = 0; b < nb; ++b) {
        bool qa[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) qa[k] = __builtin_amdgcn_ballot_w64(px_live(T[k])) != 0ull;
        // every pixel of the tile is finished: the rest of the list contributes nothing (its quadrant masks already
        // read "no rows": the mask array is cleared by one streaming pass in front of the kernel)
        if (!(qa[0] || qa[1] || qa[2] || qa[3])) break;
        if (CKPT && log_n > kUnitLog - 8) publish();   // (a bucket opens at most two units per quadrant)
        const int first = lo + b * GS_BUCKET;
        const int m = min(GS_BUCKET, hi - first);
        bool hx[2] = {false, false}, hy[2] = {false, false};
        int my_gid = 0, my_slot = 0;
        if (lane < m) {
            my_gid = a.flatten_ids[first + lane];
            if (CKPT) my_slot = a.slots[first + lane];
            const float4* r = a.rec + 3 * (size_t)my_gid;
            const float4 q0 = r[0], q1 = r[1], q2 = r[2];
            srec[lane * 3] = q0; srec[lane * 3 + 1] = q1; srec[lane * 3 + 2] = q2;
            const float ex = q1.z, ey = q1.w;   // negative when alpha can never reach 1/255
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                hx[h] = ex >= 0.f && q0.x + ex >= qxlo[h] && q0.x - ex <= qxhi[h];
                hy[h] = ey >= 0.f && q0.y + ey >= qylo[h] && q0.y - ey <= qyhi[h];
            }
        }
        bool bit[4] = {qa[0] && hx[0] && hy[0], qa[1] && hx[1] && hy[0], qa[2] && hx[0] && hy[1], qa[3] && hx[1] && hy[1]};
        if (bit[0] || bit[1] || bit[2] || bit[3]) {
            // exact test: the smallest exponent over the quadrant's pixel-centre rectangle must allow
            // alpha >= 1/255, i.e. min sigma' <= log2(255 * opacity)  (conservative margin; the
            // per-pixel test in blend_pair stays authoritative)
            const float4 q0 = srec[lane * 3], q1 = srec[lane * 3 + 1];
            const float tau = __log2f(255.f * q1.y) + 0.02f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (bit[k]) {
                    const float ms = quad_min_on_rect(q0.z, q0.w, q1.x, qxlo[k & 1] - q0.x, qxhi[k & 1] - q0.x,
                                                      qylo[k >> 1] - q0.y, qyhi[k >> 1] - q0.y);
                    bit[k] = ms <= tau + 1e-4f * fabsf(tau);
                }
            }
        }
        // mq[k]: entries whose footprint can reach quadrant k (they are evaluated);  cq[k]: the subset for
        // which at least one pixel of the quadrant takes the entry (alpha >= 1/255, pixel not finished).
        // Only those enter the backward's sublist -- 13 % of the evaluated (entry, quadrant) pairs on the
        // bench workload turn out to contribute to no pixel centre, and skipping them costs the backward
        // nothing in accuracy: they have no gradient.
        unsigned long long mq[4], cq[4] = {0ull, 0ull, 0ull, 0ull};
#pragma unroll
        for (int k = 0; k < 4; ++k) mq[k] = __ballot(bit[k]);
        // single-wave tile: LDS operations of one wave complete in issue order, so the staged records
        // are visible without a workgroup barrier
        __builtin_amdgcn_wave_barrier();
        for (unsigned long long rem = mq[0] | mq[1] | mq[2] | mq[3]; rem; rem &= rem - 1) {
            const int j = __builtin_ctzll(rem);
            const float4 q0 = srec[j * 3], q1 = srec[j * 3 + 1], q2 = srec[j * 3 + 2];
            const float op = q1.y, r = q2.x, g = q2.y, bl = q2.z;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if ((mq[k] >> j) & 1) {
                    // the offsets are formed per quadrant (an entry touches 1.7 of the 4 on average);
                    // same operation sequence as the backward: (hA dx) dx, B dx, two fma
                    const float dx = q0.x - ((k & 1) ? fx1 : fx0), dy = q0.y - ((k >> 1) ? fy1 : fy0);
                    const float sigma = fmaf(dy, fmaf(q1.x, dy, q0.w * dx), q0.z * dx * dx);
                    const float alpha = fminf(kAlphaMax, op * fast_exp2(-sigma));
                    const bool ok = px_live(T[k]) && sigma >= 0.f && alpha >= kAlphaMin;
                    if (CKPT) {
                        if (__builtin_amdgcn_ballot_w64(ok) == 0ull) continue;   // no pixel of the quadrant takes it: nothing to blend, nothing to list
                        // the sublist entry that opens a new work unit takes a storage unit and saves the pixel states before it
                        const int pos = cnt[k] + (int)__popcll(cq[k]);
                        if ((pos & (kUnit - 1)) == 0) {
                            if (pool_left == 0) {
                                int base = 0;
                                if (lane == 0) base = atomicAdd(range_ctr, kChunk);
                                base = __builtin_amdgcn_readfirstlane(base);
                                if (base + kChunk > range_len) {
                                    // out of storage: the call is void (flag; every kernel behind it is a no-op under the step guard,
                                    // the caller re-sizes from the counters, which keep counting) -- its stores land in the range's
                                    // first chunk
                                    // (front round: the step guard learns of it behind the back round -- row_chunk_scan_kernel --, whose
                                    //  list stages run under the same guard and must not find it tripped by the walk in between)
                                    if (lane == 0) {
                                        atomicOr(a.walk + kWalkFlags, GS_FLAG_UNITS);
                                        if (ROUND != 1 && a.flags) atomicOr(a.flags, (unsigned long long)GS_FLAG_UNITS);
                                    }
                                    base = 0;
                                }
                                pool_next = range0 + base; pool_left = kChunk;
                            }
                            const int su = pool_next++;
                            --pool_left;
                            a.ckpt[(size_t)su * 64 + lane] = make_float4(px_live(T[k]) ? T[k] : -1.f, cr[k], cg[k], cb[k]);   // (the backward's "finished" is T < 0)
                            if (lane == 0) ulog[log_n] = make_int2(su, (pos / kUnit) * 4 + k);
                            ++log_n;
                            const int w = pos / kUnit - cnt[k] / kUnit;
                            if (w == 0) us[k][0] = su; else if (w == 1) us[k][1] = su; else us[k][2] = su;
                        }
                        cq[k] |= 1ull << j;
                    }
                    GS_IF_CHECK(const float T_before = T[k];)
                    blend_pair(alpha, ok, r, g, bl, T[k], cr[k], cg[k], cb[k]);
                    GS_IF_CHECK(taken[k] += T[k] < T_before ? 1 : 0;)   // (blended: T shrinks and stays live; stop rule: T jumps beyond 2^32; not taken: unchanged)
                }
            }
        }
        if (CKPT) {   // compacted quadrant sublists of (flatten id, slot) pairs, in list order, into the units' blocks
            int mybits = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int n_new = (int)__popcll(cq[k]);
                if ((cq[k] >> lane) & 1) {
                    const int p = cnt[k] + (int)__popcll(cq[k] & lt_mask);
                    const int w = p / kUnit - cnt[k] / kUnit;
                    const int su = w == 0 ? us[k][0] : (w == 1 ? us[k][1] : us[k][2]);
                    a.qlist[(size_t)su * kUnit + (p & (kUnit - 1))] = make_int2(my_gid, my_slot);
                    mybits |= 1 << k;
                }
                if (n_new) {   // (wave-uniform) the unit the next bucket's first entry falls into, if it is part-filled
                    const int w = (cnt[k] + n_new - 1) / kUnit - cnt[k] / kUnit;
                    us[k][0] = w == 0 ? us[k][0] : (w == 1 ? us[k][1] : us[k][2]);
                    cnt[k] += n_new;
                }
            }
            // a scattered one-byte store leaves L2 as a 32-byte partial write (profiles/r03_traffic_calibration.json): only the
            // entries some quadrant takes store their mask, the others keep the clear's zero
            if (mybits) a.qmask[my_slot] = (uint8_t)mybits;
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (CKPT) {
        // publish the sublist lengths and the work units still in the log
        if (lane < 4) a.qcnt[4 * t + lane] = lane == 0 ? cnt[0] : (lane == 1 ? cnt[1] : (lane == 2 ? cnt[2] : cnt[3]));
        publish();
    }
    if (ROUND == 1) {
        const bool lv = __builtin_amdgcn_ballot_w64(px_live(T[0]) || px_live(T[1]) || px_live(T[2]) || px_live(T[3])) != 0ull;
        if (lane == 0) {
            a.live[t] = lv ? 1 : 0;
            if (lv) atomicAdd(reinterpret_cast<unsigned long long*>(a.rblk + GS_ROUND_LIVE), 1ull);
            if (lv && a.solo && a.flags) atomicOr(a.flags, (unsigned long long)GS_FLAG_BACK);   // nobody will come for this tile
        }
        if (lv && !a.solo) {
#pragma unroll
            for (int k = 0; k < 4; ++k) a.tstate[((size_t)t * 4 + k) * 64 + lane] = make_float4(T[k], cr[k], cg[k], cb[k]);
            if (CKPT && lane == 0) {
                a.trec[2 * (size_t)t] = make_int4(cnt[0], cnt[1], cnt[2], cnt[3]);
                a.trec[2 * (size_t)t + 1] = make_int4(us[0][0], us[1][0], us[2][0], us[3][0]);
            }
        }
    }
    float bgr = 0.f, bgg = 0.f, bgb = 0.f;
    if (a.bg) { bgr = a.bg[3 * cam]; bgg = a.bg[3 * cam + 1]; bgb = a.bg[3 * cam + 2]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int px = px0 + 8 * (k & 1), py = py0 + 8 * (k >> 1);
        if (px < a.W && py < a.H) {
            const size_t o = ((size_t)cam * a.H + py) * a.W + px;
            const float Tf = px_final_T(T[k]);
            a.out_colors[3 * o] = cr[k] + Tf * bgr;
            a.out_colors[3 * o + 1] = cg[k] + Tf * bgg;
            a.out_colors[3 * o + 2] = cb[k] + Tf * bgb;
            a.out_alphas[o] = 1.f - Tf;
            GS_IF_CHECK(if (CKPT && a.chk.fwd_T) { a.chk.fwd_T[o] = Tf; a.chk.fwd_cnt[o] = taken[k]; })
        }
    }
}

// ------------------------------------------------------------------------------------------------
// row_base[g] = number of gradient rows in front of slot 16 g = exclusive scan of popcount(qmask) sampled every 16 slots, g = 0 ..
// n / 16 (readers add the popcounts inside the group: rows_before, gs_common.h).  Three small launches over chunks of 8192 slots:
// per-chunk counts, a one-block scan of the counts (which also publishes the total, checks it against cap_rows and writes the
// call's walk record), per-chunk scan + offset.
// (A chained single-pass scan with decoupled look-back was measured first: one thread walking back over the descriptors of
//  400 co-resident blocks cost 0.25 ms at 3.3 M slots and 3.7 ms at 57 M -- every hop an uncached device-scope load.)
struct RowScanArgs {
    const uint8_t* qmask;
    int32_t* row_base;
    int32_t* walk;       // counters; the chunk counts / offsets live behind them (walk + GS_WALK_WORDS)
    int64_t n_cap, cap_rows;
    unsigned long long* flags;
    const int64_t* guard;
    volatile int64_t* mirror;   // page-locked host memory (gs_walk_mirror_set) or nullptr
};

// the 32 mask bytes of a thread as 8 words of per-byte popcounts (bytes at or beyond n count as 0: they may never have been
// cleared); returns their sum
__device__ __forceinline__ int32_t row_counts32(const uint8_t* __restrict__ qmask, int64_t i0, int64_t n, uint32_t (&w)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) w[j] = 0u;
    if (i0 + kScanPerThread <= n) {
        const uint4 x = reinterpret_cast<const uint4*>(qmask + i0)[0], y = reinterpret_cast<const uint4*>(qmask + i0)[1];
        w[0] = x.x; w[1] = x.y; w[2] = x.z; w[3] = x.w; w[4] = y.x; w[5] = y.y; w[6] = y.z; w[7] = y.w;
    } else {
        for (int j = 0; j < kScanPerThread; ++j)
            if (i0 + j < n) w[j >> 2] |= (uint32_t)qmask[i0 + j] << (8 * (j & 3));
    }
    int32_t tot = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        // bytes hold 4-bit masks: per-byte popcounts by two fold steps; their sum by one multiply
        uint32_t v = w[j] & 0x0f0f0f0fu;
        v = v - ((v >> 1) & 0x55555555u);
        v = (v & 0x33333333u) + ((v >> 2) & 0x33333333u);
        w[j] = v;
        tot += (int32_t)((v * 0x01010101u) >> 24);
    }
    return tot;
}

__global__ __launch_bounds__(kScanThreads) void row_count_kernel(const RowScanArgs a) {
    __shared__ int32_t scratch[20];
    if (a.walk[kWalkSkip] != 0) return;
    const int64_t n = slots_in_use(a.n_cap, a.guard);
    const int64_t c0 = (int64_t)blockIdx.x * kScanChunk;
    if (c0 > n) return;   // (block-uniform; entry n -- the total -- belongs to the chunk that holds it)
    uint32_t w[8];
    const int32_t tot = row_counts32(a.qmask, c0 + (int64_t)threadIdx.x * kScanPerThread, n, w);
    int32_t block_tot;
    block_excl_scan_add<int32_t>(tot, scratch, &block_tot);
    if (threadIdx.x == 0) a.walk[GS_WALK_WORDS + blockIdx.x] = block_tot;
}

__global__ __launch_bounds__(1024) void row_chunk_scan_kernel(const RowScanArgs a) {
    __shared__ int32_t scratch[20];
    if (a.walk[kWalkSkip] != 0) return;
    const int64_t n = slots_in_use(a.n_cap, a.guard);
    const int nb = (int)(n / kScanChunk) + 1;
    int32_t* part = a.walk + GS_WALK_WORDS;
    // thread t owns the contiguous run [t * per, (t + 1) * per) of chunk counts
    const int per = (nb + 1023) / 1024;
    const int j0 = (int)threadIdx.x * per, j1 = min(nb, j0 + per);
    int32_t mine = 0;
    for (int j = j0; j < j1; ++j) mine += part[j];
    int32_t total;
    int32_t run = block_excl_scan_add<int32_t>(mine, scratch, &total);
    for (int j = j0; j < j1; ++j) { const int32_t c = part[j]; part[j] = run; run += c; }
    if (threadIdx.x == 0) {
        // what the walk needed of cap_units: every range as long as the fullest one
        int32_t fullest = 0;
        for (int r = 0; r < GS_WALK_RANGES; ++r) fullest = max(fullest, a.walk[GS_WALK_RANGE0 + 32 * r]);
        a.walk[kWalkStorage] = GS_WALK_RANGES * fullest;
        a.walk[kWalkRows] = total;
        int fl = a.walk[kWalkFlags];   // (the blend's tiles have all finished: its flag is final)
        if (a.flags && (fl & GS_FLAG_UNITS)) atomicOr(a.flags, (unsigned long long)GS_FLAG_UNITS);   // (depth rounds: the front round's, see blend_fwd_kernel)
        if ((int64_t)total > a.cap_rows) {
            fl |= GS_FLAG_ROWS;
            a.walk[kWalkFlags] = fl;
            if (a.flags) atomicOr(a.flags, (unsigned long long)GS_FLAG_ROWS);
        }
        if (a.mirror) {   // the call's walk record, for a host that waits on an event behind this launch
            a.mirror[0] = a.walk[kWalkUnits]; a.mirror[1] = a.walk[kWalkStorage]; a.mirror[2] = total; a.mirror[3] = fl;
            __threadfence_system();
        }
    }
}

__global__ __launch_bounds__(kScanThreads) void row_base_kernel(const RowScanArgs a) {
    __shared__ int32_t scratch[20];
    if (a.walk[kWalkSkip] != 0) return;
    const int64_t n = slots_in_use(a.n_cap, a.guard);
    const int64_t c0 = (int64_t)blockIdx.x * kScanChunk;
    if (c0 > n) return;
    const int64_t i0 = c0 + (int64_t)threadIdx.x * kScanPerThread;
    uint32_t w[8];
    const int32_t tot = row_counts32(a.qmask, i0, n, w);
    int32_t block_tot;
    const int32_t run = block_excl_scan_add<int32_t>(tot, scratch, &block_tot) + a.walk[GS_WALK_WORDS + blockIdx.x];
    // this thread's 32 slots are two groups of 16: the base of the first, and of the second behind the first's 16 masks
    int32_t first16 = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) first16 += (int32_t)((w[j] * 0x01010101u) >> 24);
    if (i0 <= n) a.row_base[i0 >> 4] = run;
    if (i0 + 16 <= n) a.row_base[(i0 >> 4) + 1] = run + first16;
}

// ------------------------------------------------------------------------------------------------
struct BlendBwdArgs {
    int C, W, H, tw, tiles;
    const float4* rec;
    const int32_t* walk;
    const int2* qlist;
    const int32_t* qcnt;
    const int4* unit_desc;
    const float4* ckpt;
    const uint8_t* qmask;
    const int32_t* row_base;
    int cap_units;
    // fill classes (unit_classes_kernel), or nullptr: every unit runs as class 4 straight from unit_desc
    const int32_t* cls_hdr;   // [kClsHdrInts]: units of class c at [c], c = 1 .. 4
    const int4* cls_desc;     // class 4 at [0, cap_units), classes 3, 2, 1 behind it, cls_cap descriptors each
    int cls_cap;
    const float *out_colors, *out_alphas, *v_colors, *v_alphas;
    float4* rows;   // [row pairs][3]
    const int64_t* guard;
    GS_IF_CHECK(BwdCheck chk;)
};

constexpr int kBwdWaves = 4;
constexpr int kPipeLanes = 8;                       // lanes per systolic pipeline (two pipelines share a 16-lane DPP row)
constexpr int kPerLane = kUnit / kPipeLanes;        // 4 entries per lane in a full unit (bwd_wave<E>: E = 1 .. 4)
static_assert(kPerLane == 4, "bwd_wave is instantiated for 1 .. 4 entries per lane");
constexpr int kUnitsPerWave = 64 / kPipeLanes;      // 8
// (Round 5 bounded half-quadrant -- 8 x 4 pixel -- work units with a timing build: -15 to -23 us for this kernel at the 1.55 x
//  unit count they would have, before twice the rows in the row sum; not built.  HISTORY.md, profiles/r05_halfq_bound.txt.)
constexpr int kUnitPixels = 64;
constexpr int kBwdSteps = kUnitPixels + kPipeLanes - 1;      // 71

__device__ __forceinline__ float dpp_row_shr1(float v) {
    // lane r of each 16-lane row receives lane r-1's value; lane 0 keeps its own
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, false));
}

// per pixel in LDS: float4 (v_r, v_g, v_b, E) and float2 (checkpoint T, P = checkpoint colour . v) in two arrays
// (24 B; the pixel centre is recomputed from the pixel index)

// -DGS_BWD_ACC64 (tools/acc64_ab.py; never in the product build): the 11 per-entry sums in double -- the A/B that separates the
// rounding of the fp32 ACCUMULATION from everything else in a row whose pixel terms cancel.
#ifdef GS_BWD_ACC64
typedef double acc_t;
#define GS_ACC_FMA(a, b, c) ((double)(a) * (double)(b) + (c))
#else
typedef float acc_t;
#define GS_ACC_FMA(a, b, c) fmaf((a), (b), (c))
#endif
struct EntryState {
    float mx, my, hA, Bc, hC, op, colr, colg, colb, At, Bt, Ct;
    acc_t s_mx, s_my, s_ax, s_ay, s_A, s_B, s_C, s_vs, s_r, s_g, s_b;
    bool has;
};

// Round 5 tried the forward's own state encoding here (T alone: 0 = finished; alpha masked once; the stop rule behind a
// wave-uniform branch on newly stopping pixels, or branch-free): 197 / 204 instead of 208 VALU per step in the ISA, and the
// kernel 0.445 / 0.410 ms against 0.414 for this form on the same box -- a scalar branch per pair stalls three waves per SIMD more
// than eleven selects cost them, and four instructions fewer are within the noise.  Not kept.
// (pixels outside the pipeline window arrive with T < 0; slots past the end of the sublist have
// opacity 0, hence alpha 0: neither needs a flag of its own)
__device__ __forceinline__ void bwd_pair(EntryState& e, const float4 d0, const float2 d1, float& T, float& P) {
    const float dx = e.mx - d1.x, dy = e.my - d1.y;
    const float sigma = fmaf(dy, fmaf(e.hC, dy, e.Bc * dx), e.hA * dx * dx);   // same op sequence as the forward
    const float vis = fast_exp2(-sigma);
    const float ov = e.op * vis;
    const float alpha = fminf(kAlphaMax, ov);
    const bool ok = T > 0.f && sigma >= 0.f && alpha >= kAlphaMin;
    const float w = alpha * T;
    const float Tn = fmaf(-alpha, T, T);   // identical to the forward's update
    const bool stop = ok && Tn <= kTMin;
    const bool contrib = ok && !stop;
    const float cv = e.colr * d0.x + e.colg * d0.y + e.colb * d0.z;
    const float wm = contrib ? w : 0.f;
    const float Pn = fmaf(wm, cv, P);   // == P exactly when the entry does not contribute: no select for P below
    const float ra = fast_rcp(1.f - alpha);
    const float v_alpha = fmaf(T, cv, ra * (d0.w + Pn));
    e.s_r = GS_ACC_FMA(wm, d0.x, e.s_r); e.s_g = GS_ACC_FMA(wm, d0.y, e.s_g); e.s_b = GS_ACC_FMA(wm, d0.z, e.s_b);
    const float vs = (contrib && ov <= kAlphaMax) ? -ov * v_alpha : 0.f;   // d loss / d sigma
    const float hx = vs * dx, hy = vs * dy;
    e.s_A = GS_ACC_FMA(hx, dx, e.s_A); e.s_B = GS_ACC_FMA(hx, dy, e.s_B); e.s_C = GS_ACC_FMA(hy, dy, e.s_C);
    const float gx = fmaf(e.At, hx, e.Bt * hy), gy = fmaf(e.Bt, hx, e.Ct * hy);
    e.s_mx += (acc_t)gx; e.s_my += (acc_t)gy; e.s_ax += (acc_t)fabsf(gx); e.s_ay += (acc_t)fabsf(gy);
    e.s_vs += (acc_t)vs;
    T = contrib ? Tn : (stop ? -1.f : T);
    P = Pn;
}

// One wave of the backward: eight work units of ONE fill class E = entries per lane.  A unit of class E holds at most 8 E
// entries -- full units and tails (the last, part-filled unit of a quadrant sublist) of 25-32 entries are class 4, tails of 1-8 /
// 9-16 / 17-24 entries classes 1 / 2 / 3 -- and the systolic loop passes over 8 E entry slots per unit instead of all 32 (the
// bench workload: 9 % of the slots of its 169 k units are empty, heavy-tailed footprints 16 %; tools/tail_stats.py).
template <int E>
__device__ __forceinline__ void bwd_wave(const BlendBwdArgs& a, const int4 ud, const bool valid, const int unit, float4* sd0, float2* sck, const int r) {
    const int tq = ud.x, su = ud.z;
    const bool first_unit = ud.y == 0;   // the sublist starts here: every pixel inside the image has T = 1, no colour yet
    const int t = tq >> 2, q = tq & 3;
    const int cam = t / a.tiles, tt = t - cam * a.tiles;
    const int tyi = tt / a.tw, txi = tt - tyi * a.tw;
    const int qx0 = txi * GS_TILE + 8 * (q & 1), qy0 = tyi * GS_TILE + 8 * (q >> 1);
    const float fx0 = (float)qx0 + 0.5f, fy0 = (float)qy0 + 0.5f;
    // the unit's checkpoint: 64 pixel states in front of its first entry (T < 0: pixel finished or outside the image)
    const float4* ckp = a.ckpt + (size_t)su * 64;

    // ---- prologue, three dependent memory round trips in all (unit descriptor above; everything below in two batches).
    // Written branch-free on purpose: with the loads inside `if (inside the image)` / `if (entry exists)` blocks hipcc waits
    // for each pixel's and each entry's loads before it issues the next ones -- 8 + 2 x 4 serialised round trips per wave in
    // front of a main loop of comparable length, with three waves per SIMD to hide them (round 3: 0.45 -> see DESIGN.md).
    // Out-of-range pixels / entries load from a clamped, valid address and are masked afterwards.
    // batch 1: the sublist's length, this lane's E (flatten id, slot) pairs -- 8 E contiguous bytes of the unit's block; pairs
    // past the end of the sublist are whatever the block held before -- and the quadrant's 64 pixels, 8 per lane of the pipeline
    const int n_sub = valid ? a.qcnt[tq] : 0;
    int2 gs[E];
    {
        const int2* qp2 = a.qlist + (size_t)su * kUnit + E * r;
        if constexpr (E == 4) {
            const int4* qp = reinterpret_cast<const int4*>(qp2);
            const int4 g0 = qp[0], g1 = qp[1];
            gs[0] = make_int2(g0.x, g0.y); gs[1] = make_int2(g0.z, g0.w); gs[2] = make_int2(g1.x, g1.y); gs[3] = make_int2(g1.z, g1.w);
        } else if constexpr (E == 2) {
            const int4 g0 = *reinterpret_cast<const int4*>(qp2);
            gs[0] = make_int2(g0.x, g0.y); gs[1] = make_int2(g0.z, g0.w);
        } else {
#pragma unroll
            for (int i = 0; i < E; ++i) gs[i] = qp2[i];
        }
    }
    constexpr int kPix = kUnitPixels / kPipeLanes;
    float l_vr[kPix], l_vg[kPix], l_vb[kPix], l_oa[kPix], l_cr[kPix], l_cg[kPix], l_cb[kPix], l_va[kPix];
    float4 l_ck[kPix];
    const float* vap = a.v_alphas ? a.v_alphas : a.out_alphas;   // (one load either way; masked below)
#pragma unroll
    for (int i = 0; i < kPix; ++i) {
        const int p = r + kPipeLanes * i;
        const int px = min(qx0 + (p & 7), a.W - 1), py = min(qy0 + (p >> 3), a.H - 1);
        const size_t o = ((size_t)cam * a.H + py) * a.W + px;
        l_vr[i] = a.v_colors[3 * o]; l_vg[i] = a.v_colors[3 * o + 1]; l_vb[i] = a.v_colors[3 * o + 2];
        l_oa[i] = a.out_alphas[o];
        l_cr[i] = a.out_colors[3 * o]; l_cg[i] = a.out_colors[3 * o + 1]; l_cb[i] = a.out_colors[3 * o + 2];
        l_va[i] = vap[o];
        l_ck[i] = ckp[p];
    }
    // batch 2: the entries' packed records (addresses from batch 1), in flight while the pixels are staged
    const int n_in = min(kPipeLanes * E, n_sub - ud.y * kUnit);   // entries of the sublist that fall into this unit (a unit of class E: <= 8 E)
    float4 rq[E][3];
#pragma unroll
    for (int i = 0; i < E; ++i) {
        const bool has = E * r + i < n_in;
        const float4* rp = a.rec + 3 * (size_t)(has ? gs[i].x : 0);
        rq[i][0] = rp[0]; rq[i][1] = rp[1]; rq[i][2] = rp[2];
    }
#pragma unroll
    for (int i = 0; i < kPix; ++i) {
        const int p = r + kPipeLanes * i;
        const bool inside = valid && (qx0 + (p & 7)) < a.W && (qy0 + (p >> 3)) < a.H;
        const float vr = l_vr[i], vg = l_vg[i], vb = l_vb[i];
        const float Tf = 1.f - l_oa[i];
        const float va = a.v_alphas ? l_va[i] : 0.f;
        // E = T_final * v_alpha - render_colour . v_colour  (background terms cancel)
        const float Ev = Tf * va - (l_cr[i] * vr + l_cg[i] * vg + l_cb[i] * vb);
        // the unit's checkpoint: the pixel's state in front of its first entry (T < 0: finished or outside the image); the
        // first unit of a sublist starts from T = 1, nothing accumulated
        const float4 ck = first_unit ? make_float4(1.f, 0.f, 0.f, 0.f) : l_ck[i];
        sd0[p] = inside ? make_float4(vr, vg, vb, Ev) : make_float4(0.f, 0.f, 0.f, 0.f);
        sck[p] = inside ? make_float2(ck.x, ck.y * vr + ck.z * vg + ck.w * vb) : make_float2(-1.f, 0.f);
    }

    EntryState e[E];
    int slot[E];
#pragma unroll
    for (int i = 0; i < E; ++i) {
        const int en = E * r + i;
        e[i].has = en < n_in;
        slot[i] = e[i].has ? gs[i].y : 0;
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 q0 = e[i].has ? rq[i][0] : z4, q1 = e[i].has ? rq[i][1] : z4, q2 = e[i].has ? rq[i][2] : z4;
        e[i].mx = q0.x; e[i].my = q0.y; e[i].hA = q0.z; e[i].Bc = q0.w; e[i].hC = q1.x; e[i].op = q1.y;
        e[i].colr = q2.x; e[i].colg = q2.y; e[i].colb = q2.z;
        // true conic entries for the mean gradient (the record stores them scaled by log2(e))
        e[i].At = 2.f * kLn2 * q0.z; e[i].Bt = kLn2 * q0.w; e[i].Ct = 2.f * kLn2 * q1.x;
        e[i].s_mx = e[i].s_my = e[i].s_ax = e[i].s_ay = e[i].s_A = e[i].s_B = e[i].s_C = e[i].s_vs = (acc_t)0;
        e[i].s_r = e[i].s_g = e[i].s_b = (acc_t)0;
    }
    __builtin_amdgcn_wave_barrier();

    float T_out = -1.f, P_out = 0.f;
    // (check build: D and C travel with the pixel like T and P -- last live transmittance, entries taken in this unit)
    GS_IF_CHECK(float D_out = -1.f, C_out = 0.f; if (valid && r == 0 && a.chk.unit_hdr) a.chk.unit_hdr[unit] = make_int2(tq, ud.y);)
    for (int s = 0; s < kBwdSteps; ++s) {
        float T = dpp_row_shr1(T_out), P = dpp_row_shr1(P_out);
        const int p = s - r;
        const bool act = (unsigned)p < (unsigned)kUnitPixels;
        const int pc = min(max(p, 0), kUnitPixels - 1);
        const float4 d0 = sd0[pc];
        const float2 ck = sck[pc];
        const float2 d1 = make_float2(fx0 + (float)(pc & 7), fy0 + (float)(pc >> 3));   // pixel centre
        if (r == 0) { T = ck.x; P = ck.y; }   // head of the pipeline: fed from the checkpoint, not from the lane below
        T = act ? T : -1.f;   // pipeline fill / drain: nothing contributes
        GS_IF_CHECK(float D = dpp_row_shr1(D_out), Cn = dpp_row_shr1(C_out); if (r == 0) { D = ck.x; Cn = 0.f; } if (!act) { D = -1.f; Cn = 0.f; })
#pragma unroll
        for (int i = 0; i < E; ++i) {
            GS_IF_CHECK(const float T_before = T;)
            bwd_pair(e[i], d0, d1, T, P);
            GS_IF_CHECK(if (T > 0.f && T < T_before) { D = T; Cn += 1.f; })   // (contributed; the stop rule leaves T = -1 and D at the final value)
        }
        GS_IF_CHECK(D_out = D; C_out = Cn;
                    if (valid && act && r == kPipeLanes - 1 && a.chk.unit_out) a.chk.unit_out[(size_t)unit * 64 + p] = make_float2(D, Cn);)
        T_out = T; P_out = P;
    }

    // the row of (slot, quadrant): the slot's first row (rows_before: a base per 16 slots + the masks of its predecessors in the
    // group) + the number of its existing rows in front of this quadrant.  Looked up HERE, not in the prologue: four 16-byte mask
    // loads in flight next to the records and the pixels cost the kernel its third wave per SIMD (178 VGPRs).
    int row[E];
#pragma unroll
    for (int i = 0; i < E; ++i) {
        int m;
        row[i] = rows_before(a.row_base, a.qmask, slot[i], &m);
        row[i] += __popc((unsigned)m & ((1u << q) - 1u));
    }
#pragma unroll
    for (int i = 0; i < E; ++i) {
        if (e[i].has) {
            float4* rp = a.rows + (GS_ROW_FLOATS / 4) * (size_t)row[i];
            // v_opacity = sum vis * v_alpha = -sum(vs) / opacity   (vs = -opacity*vis*v_alpha)
            const float v_op = e[i].op > 0.f ? (float)(-e[i].s_vs / (acc_t)e[i].op) : 0.f;
            rp[0] = make_float4((float)e[i].s_mx, (float)e[i].s_my, (float)e[i].s_ax, (float)e[i].s_ay);
            rp[1] = make_float4((float)((acc_t)0.5f * e[i].s_A), (float)e[i].s_B, (float)((acc_t)0.5f * e[i].s_C), v_op);
            rp[2] = make_float4((float)e[i].s_r, (float)e[i].s_g, (float)e[i].s_b, 0.f);
        }
    }
}

// Groups the published work units by fill class in front of blend_bwd_kernel: cls[0, cap_units) class 4, then classes 3, 2, 1
// in regions of cls_cap descriptors (a frame has at most one tail per quadrant sublist: 4 C tiles).  One atomic per block and
// class (a returning same-address atomic per WAVE cost 0.19 ms at 31 k waves, HISTORY.md); the order inside a class is whatever
// the blocks' atomics make it -- every unit writes rows of its own, no result depends on it.  .w = the unit's published index.
constexpr int kClsHdrInts = 16, kClsThreads = 1024;
__global__ __launch_bounds__(64) void unit_classes_clear_kernel(int32_t* __restrict__ hdr) {
    if (threadIdx.x < kClsHdrInts) hdr[threadIdx.x] = 0;
}
__global__ __launch_bounds__(kClsThreads) void unit_classes_kernel(const int32_t* __restrict__ walk, const int4* __restrict__ unit_desc,
                                                                   const int32_t* __restrict__ qcnt, int cap_units, int cls_cap,
                                                                   int32_t* __restrict__ hdr, int4* __restrict__ cls, const int64_t* __restrict__ guard) {
    __shared__ int s_cnt[kClsThreads / 64][4], s_base[kClsThreads / 64][4];
    if (guard_tripped(guard)) return;
    const int n_units = min(walk[kWalkUnits], cap_units);
    if ((int)blockIdx.x * kClsThreads >= n_units) return;   // block-uniform
    const int u = (int)blockIdx.x * kClsThreads + (int)threadIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int c = 0;
    int4 ud = make_int4(0, 0, 0, 0);
    if (u < n_units) {
        ud = unit_desc[u];
        const int n_in = min(kUnit, qcnt[ud.x] - ud.y * kUnit);
        c = min(max((n_in + kPipeLanes - 1) / kPipeLanes, 1), 4);
        ud.w = u;
    }
    unsigned long long m[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) m[k] = __ballot(c == k + 1);
    if (lane < 4) s_cnt[wave][lane] = (int)__popcll(lane == 0 ? m[0] : (lane == 1 ? m[1] : (lane == 2 ? m[2] : m[3])));
    __syncthreads();
    if (threadIdx.x < 4) {
        int tot = 0;
        for (int w = 0; w < kClsThreads / 64; ++w) { s_base[w][threadIdx.x] = tot; tot += s_cnt[w][threadIdx.x]; }
        const int base = tot ? atomicAdd(hdr + 1 + threadIdx.x, tot) : 0;
        for (int w = 0; w < kClsThreads / 64; ++w) s_base[w][threadIdx.x] += base;
    }
    __syncthreads();
    if (c) {
        const unsigned long long mine = c == 1 ? m[0] : (c == 2 ? m[1] : (c == 3 ? m[2] : m[3]));
        const int i = s_base[wave][c - 1] + (int)__popcll(mine & ((1ull << lane) - 1ull));
        // (class 4 cannot outgrow cap_units; a tail region can only where the caller passed a cls_cap below 4 C tiles: dropped
        //  descriptors would lose gradient rows, so gs_blend_bwd refuses such a cls_cap)
        if (i < (c == 4 ? cap_units : cls_cap)) cls[(c == 4 ? 0 : cap_units + (3 - c) * cls_cap) + i] = ud;
    }
}

__global__ __launch_bounds__(kBwdWaves * 64) void blend_bwd_kernel(const BlendBwdArgs a) {
    __shared__ float4 sd0_all[kBwdWaves][kUnitsPerWave][64];   // 8 KB per wave: v_r, v_g, v_b, E of the unit's 64 pixels
    __shared__ float2 sck_all[kBwdWaves][kUnitsPerWave][64];   // 4 KB per wave: checkpoint T, P = checkpoint colour . v
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int pipe = lane / kPipeLanes, r = lane & (kPipeLanes - 1);
    if (guard_tripped(a.guard)) return;
    // the block's class and its place in it: class 4 over [0, cap_units) first (the longest waves), then 3, 2, 1
    constexpr int kPerBlock = kBwdWaves * kUnitsPerWave;
    int cls = 4, blk = (int)blockIdx.x, n_units, first = 0;
    if (a.cls_hdr == nullptr) {
        n_units = min(a.walk[kWalkUnits], a.cap_units);
    } else {
        const int b4 = (a.cap_units + kPerBlock - 1) / kPerBlock, bt = (a.cls_cap + kPerBlock - 1) / kPerBlock;
        if (blk >= b4) {
            const int k = (blk - b4) / bt;   // 0, 1, 2 -> class 3, 2, 1
            cls = 3 - k; blk -= b4 + k * bt; first = a.cap_units + k * a.cls_cap;
        }
        n_units = min(a.cls_hdr[cls], cls == 4 ? a.cap_units : a.cls_cap);
    }
    const int in_class = (blk * kBwdWaves + wave) * kUnitsPerWave + pipe;
    if ((blk * kBwdWaves + wave) * kUnitsPerWave >= n_units) return;   // wave-uniform
    GS_CLOCK_PROBE_SCOPE(2);
    const bool valid = in_class < n_units;
    float4* sd0 = sd0_all[wave][pipe];
    float2* sck = sck_all[wave][pipe];

    int4 ud = make_int4(0, 0, 0, 0);
    if (valid) ud = a.cls_hdr ? a.cls_desc[first + in_class] : a.unit_desc[in_class];
    const int unit = a.cls_hdr ? ud.w : in_class;   // the unit's published index (the check build's arrays are indexed by it)
    if (cls == 4) bwd_wave<4>(a, ud, valid, unit, sd0, sck, r);
    else if (cls == 3) bwd_wave<3>(a, ud, valid, unit, sd0, sck, r);
    else if (cls == 2) bwd_wave<2>(a, ud, valid, unit, sd0, sck, r);
    else bwd_wave<1>(a, ud, valid, unit, sd0, sck, r);
}

}  // namespace gs

using namespace gs;

#ifdef GS_BWD_CHECK
extern "C" int gs_debug_bwd_check_set(float* fwd_T, int32_t* fwd_cnt, float* unit_out, int32_t* unit_hdr) {
    g_bwd_check.fwd_T = fwd_T; g_bwd_check.fwd_cnt = fwd_cnt;
    g_bwd_check.unit_out = reinterpret_cast<float2*>(unit_out); g_bwd_check.unit_hdr = reinterpret_cast<int2*>(unit_hdr);
    return GS_OK;
}
#endif

#ifdef GS_CLOCK_PROBE
extern "C" int gs_debug_clock_probe(int64_t* out4, int reset) {
    unsigned long long h[4] = {0, 0, 0, 0};
    if (out4) { GS_HIP_CHECK(hipMemcpyFromSymbol(h, HIP_SYMBOL(gs_clk_acc), sizeof(h))); for (int i = 0; i < 4; ++i) out4[i] = (int64_t)h[i]; }
    if (reset) { unsigned long long z[4] = {0, 0, 0, 0}; GS_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(gs_clk_acc), z, sizeof(z))); }
    return GS_OK;
}
#endif

extern "C" size_t gs_walk_state_ints(int64_t n_isects) {
    if (n_isects < 0) return 0;
    return (size_t)GS_WALK_WORDS + (size_t)(n_isects / kScanChunk + 1);
}

extern "C" int gs_blend_fwd(void* stream, int C, int width, int height, const float* rec,
                            const float* backgrounds, const int32_t* isect_offsets, const int32_t* tile_order,
                            const int32_t* flatten_ids, const int32_t* slots, int64_t n_isects, float* render_colors,
                            float* render_alphas, float* ckpt, int32_t* qlist, int32_t* qcnt, uint8_t* qmask,
                            int32_t* unit_desc, int64_t cap_units, int32_t* row_base, int64_t cap_rows, int32_t* walk_state) {
    GS_REQUIRE(C >= 1 && width > 0 && height > 0 && n_isects >= 0, "C>=1, positive image size, n_isects>=0");
    GS_REQUIRE(isect_offsets && render_colors && render_alphas, "null pointer");
    const bool train = ckpt != nullptr;
    GS_REQUIRE(!train || (qlist && qcnt && qmask && unit_desc && row_base && walk_state && slots), "training mode needs every list output");
    GS_REQUIRE(!train || (cap_units >= kChunk * GS_WALK_RANGES && cap_units < (1ll << 26) && cap_rows >= 0 && cap_rows < (1ll << 31)),
               "training mode: 256 <= cap_units < 2^26 work units, cap_rows < 2^31 gradient rows");
    GS_REQUIRE(!train || n_isects < (1ll << 31) - kScanChunk, "training mode: the slots of a call are indexed by int32");
    static_assert(kScanPerThread == 32, "row_base_kernel writes two groups of 16 slots per thread");
    GS_REQUIRE(!train || (((uintptr_t)qmask & 15) == 0 && ((uintptr_t)row_base & 15) == 0 && ((uintptr_t)walk_state & 7) == 0),
               "training mode: qmask / row_base 16-byte aligned, walk_state 8-byte aligned");
    BlendFwdArgs a;
    a.C = C; a.W = width; a.H = height;
    a.tw = (width + GS_TILE - 1) / GS_TILE;
    a.tiles = a.tw * ((height + GS_TILE - 1) / GS_TILE);
    a.rec = reinterpret_cast<const float4*>(rec); a.bg = backgrounds; a.isect_offsets = isect_offsets;
    a.flatten_ids = flatten_ids; a.slots = slots;
    a.out_colors = render_colors; a.out_alphas = render_alphas;
    a.ckpt = reinterpret_cast<float4*>(ckpt); a.qlist = reinterpret_cast<int2*>(qlist); a.qcnt = qcnt; a.qmask = qmask;
    a.unit_desc = reinterpret_cast<int4*>(unit_desc); a.walk = walk_state; a.cap_units = (int)cap_units;
    a.tile_order = tile_order;
    a.guard = current_guard().info;
    a.flags = a.guard ? reinterpret_cast<unsigned long long*>(const_cast<int64_t*>(a.guard) + 3) : nullptr;
    const Rounds R = current_rounds();
    const int phase = (R.phase == 1 || R.phase == 4) ? 1 : (R.phase == 2 ? 2 : 0);
    a.solo = R.phase == 4 ? 1 : 0;
    GS_REQUIRE(R.phase != 4 || a.flags != nullptr, "depth rounds, front round alone: needs a step guard (gs_guard_set)");
    GS_REQUIRE(phase == 0 || C == 1, "depth rounds: one camera per call");
    GS_REQUIRE(phase == 0 || !train || R.tile_rec, "depth rounds, training mode: gs_rounds_set needs tile_rec");
    a.rblk = R.blk; a.live = R.live; a.tstate = R.state; a.trec = reinterpret_cast<int4*>(R.tile_rec);
    GS_IF_CHECK(a.chk = g_bwd_check;)
    const unsigned n_tiles = (unsigned)(C * a.tiles);
    hipStream_t st = (hipStream_t)stream;
    // 4 tiles (waves) per workgroup: measured equal to single-wave workgroups (0.33-0.36 ms at the bench
    // size).  One box of the pool ran the single-wave form at 0.63 ms with every other kernel at its usual
    // time; the cause was not established, a quarter of the workgroups is the conservative launch shape.
    constexpr int kFwdWaves = 4;
    const dim3 grid((n_tiles + kFwdWaves - 1) / kFwdWaves), block(64 * kFwdWaves);
    if (train) {
        // one streaming clear of the quadrant masks (I bytes, 16-byte stores) and of the walk state: entries no quadrant
        // takes, and the tails of lists a saturated tile abandons, then need no store at all
        const int walk_ints = (int)gs_walk_state_ints(n_isects);
        const unsigned cg = (unsigned)std::min<int64_t>(1024, std::max<int64_t>(1, (n_isects / 16 + 255) / 256));
        hipLaunchKernelGGL(qmask_clear_kernel, dim3(cg), dim3(256), 0, st, qmask, n_isects, walk_state, walk_ints, a.guard,
                           (const int64_t*)R.blk, phase);
        GS_LAUNCH_CHECK("qmask_clear_kernel");
        if (phase == 1) hipLaunchKernelGGL((blend_fwd_kernel<true, kFwdWaves, 1>), grid, block, 0, st, a);
        else if (phase == 2) hipLaunchKernelGGL((blend_fwd_kernel<true, kFwdWaves, 2>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((blend_fwd_kernel<true, kFwdWaves, 0>), grid, block, 0, st, a);
        GS_LAUNCH_CHECK("blend_fwd_kernel");
        if (phase == 1 && !a.solo) return GS_OK;   // (the row bases are scanned once, behind the back round)
        RowScanArgs s;
        s.qmask = qmask; s.row_base = row_base; s.walk = walk_state; s.n_cap = n_isects; s.cap_rows = cap_rows;
        s.flags = a.flags; s.guard = a.guard; s.mirror = current_walk_mirror();
        const unsigned chunks = (unsigned)(n_isects / kScanChunk + 1);
        hipLaunchKernelGGL(row_count_kernel, dim3(chunks), dim3(kScanThreads), 0, st, s);
        hipLaunchKernelGGL(row_chunk_scan_kernel, dim3(1), dim3(1024), 0, st, s);
        hipLaunchKernelGGL(row_base_kernel, dim3(chunks), dim3(kScanThreads), 0, st, s);
        GS_LAUNCH_CHECK("row_base_kernel");
    } else {
        if (phase == 1) hipLaunchKernelGGL((blend_fwd_kernel<false, kFwdWaves, 1>), grid, block, 0, st, a);
        else if (phase == 2) hipLaunchKernelGGL((blend_fwd_kernel<false, kFwdWaves, 2>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((blend_fwd_kernel<false, kFwdWaves, 0>), grid, block, 0, st, a);
        GS_LAUNCH_CHECK("blend_fwd_kernel");
    }
    return GS_OK;
}

// descriptors a tail region holds: one tail per quadrant sublist at most, rounded up to whole blocks of the backward
static int64_t cls_region(int C, int width, int height) {
    const int64_t tiles = (int64_t)((width + GS_TILE - 1) / GS_TILE) * ((height + GS_TILE - 1) / GS_TILE);
    const int64_t per_block = kBwdWaves * kUnitsPerWave;
    return (4 * (int64_t)C * tiles + per_block - 1) / per_block * per_block;
}

extern "C" size_t gs_unit_classes_ints(int64_t cap_units, int C, int width, int height) {
    if (cap_units < 0 || C < 1 || width <= 0 || height <= 0) return 0;
    return (size_t)kClsHdrInts + 4 * ((size_t)cap_units + 3 * (size_t)cls_region(C, width, height));
}

extern "C" int gs_blend_bwd(void* stream, int C, int width, int height, const float* rec,
                            const int32_t* qlist, const int32_t* qcnt, const int32_t* unit_desc, int64_t cap_units,
                            const float* ckpt, const uint8_t* qmask, const int32_t* row_base, const int32_t* walk_state,
                            const float* render_colors, const float* render_alphas, const float* v_render_colors,
                            const float* v_render_alphas, float* rows, int32_t* unit_classes) {
    GS_REQUIRE(C >= 1 && width > 0 && height > 0 && cap_units >= 0 && cap_units < (1ll << 26), "C>=1, positive image size, 0 <= cap_units < 2^26");
    if (cap_units == 0) return GS_OK;
    GS_REQUIRE(rec && qlist && qcnt && unit_desc && ckpt && qmask && row_base && walk_state, "null list pointer");
    GS_REQUIRE(render_colors && render_alphas && v_render_colors && rows, "null image pointer");
    BlendBwdArgs a;
    a.C = C; a.W = width; a.H = height;
    a.tw = (width + GS_TILE - 1) / GS_TILE;
    a.tiles = a.tw * ((height + GS_TILE - 1) / GS_TILE);
    a.rec = reinterpret_cast<const float4*>(rec);
    a.qlist = reinterpret_cast<const int2*>(qlist); a.qcnt = qcnt; a.walk = walk_state;
    a.unit_desc = reinterpret_cast<const int4*>(unit_desc); a.cap_units = (int)cap_units;
    a.ckpt = reinterpret_cast<const float4*>(ckpt); a.qmask = qmask; a.row_base = row_base;
    a.out_colors = render_colors;
    a.out_alphas = render_alphas; a.v_colors = v_render_colors; a.v_alphas = v_render_alphas;
    a.rows = reinterpret_cast<float4*>(rows);
    a.guard = current_guard().info;
    GS_IF_CHECK(a.chk = g_bwd_check;)
    // one pipeline per work unit, eight per wave: the grid covers the capacity, waves past the published count return at once
    unsigned grid = (unsigned)((cap_units + kUnitsPerWave * kBwdWaves - 1) / (kUnitsPerWave * kBwdWaves));
    a.cls_hdr = nullptr; a.cls_desc = nullptr; a.cls_cap = 0;
    if (unit_classes) {
        // the units grouped by fill class first (16-byte aligned buffer of gs_unit_classes_ints int32: counters, then descriptors)
        GS_REQUIRE(((uintptr_t)unit_classes & 15) == 0, "unit_classes must be 16-byte aligned");
        a.cls_hdr = unit_classes; a.cls_desc = reinterpret_cast<const int4*>(unit_classes + kClsHdrInts);
        a.cls_cap = (int)cls_region(C, width, height);
        // (the counters are cleared by a launch, not by hipMemsetAsync: inside a captured step a memset node ahead of the kernel
        //  that counts ended in GPU memory faults on this ROCm -- the eager sequence of the same calls did not)
        hipLaunchKernelGGL(unit_classes_clear_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, unit_classes);
        hipLaunchKernelGGL(unit_classes_kernel, dim3((unsigned)((cap_units + kClsThreads - 1) / kClsThreads)), dim3(kClsThreads), 0, (hipStream_t)stream,
                           walk_state, a.unit_desc, qcnt, (int)cap_units, a.cls_cap, unit_classes, reinterpret_cast<int4*>(unit_classes + kClsHdrInts),
                           a.guard);
        GS_LAUNCH_CHECK("unit_classes_kernel");
        grid += 3u * (unsigned)(a.cls_cap / (kUnitsPerWave * kBwdWaves));
    }
    hipLaunchKernelGGL(blend_bwd_kernel, dim3(grid), dim3(kBwdWaves * 64), 0, (hipStream_t)stream, a);
    GS_LAUNCH_CHECK("blend_bwd_kernel");
    return GS_OK;
}
