/*
 * gs_raster.h -- C ABI of the MI355X (gfx950) Gaussian-splat rasterizer.
 *
 * This is the drop-in boundary for the ONE call the reference makes into its rasterizer:
 *
 *     from gsplat.rendering import rasterization            (/root/reference/model/gaussian.py:8)
 *     rasterization(means, quats, scales, opacities, colors, viewmats, Ks, width, height,
 *                   sh_degree=..., backgrounds=..., absgrad=True, packed=False)
 *                                                           (/root/reference/model/gaussian.py:353-367)
 *
 * gsplat 1.0.0 reaches its CUDA kernels through per-stage torch ops; each entry point below is
 * the stage a binding for this path would call instead (stage names: SURVEY.md section 2.2).
 * Plain pointers and sizes only: every pointer is a DEVICE pointer unless its name ends in
 * `_host`; `stream` is a hipStream_t passed as void*.  All tensors are dense row-major fp32 /
 * int32 / int64.  Every function returns 0 on success and a negative code on failure, in which
 * case gs_last_error() (thread-local) describes it.  Nothing is allocated inside; the caller owns
 * all memory (ownership contract: SURVEY.md section 8b).  Entry points may be called from any
 * host thread.
 *
 * Index vocabulary: C cameras, N Gaussians, K stored SH coefficients per Gaussian, flatten id
 * f = c*N + n, tiles = tile_w*tile_h (16x16 pixel tiles), I = number of (tile, Gaussian)
 * intersections, bucket = 64 consecutive entries of one tile's depth-sorted list.
 */
#ifndef GS_RASTER_H_
#define GS_RASTER_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GS_TILE 16           /* tile edge in pixels (gsplat default tile_size=16) */
#define GS_BUCKET 64         /* entries per bucket = one wavefront */
#define GS_UNIT 32           /* quadrant-sublist entries per work unit of gs_blend_bwd (one checkpoint each) */
#define GS_REC_FLOATS 12     /* packed per-(camera,Gaussian) blend record */
#ifndef GS_ROW_FLOATS
#define GS_ROW_FLOATS 12     /* per-intersection gradient row written by gs_blend_bwd */
#endif

/* Walk state of a training forward (gs_blend_fwd -> gs_blend_bwd): int32 words, followed by the chunk counts of the row-base scan.
 * gs_walk_state_ints(n_isects) = GS_WALK_WORDS + n_isects / 8192 + 1 words; 8-byte aligned; cleared by gs_blend_fwd itself. */
#define GS_WALK_UNITS 0      /* work units published (descriptors written) */
#define GS_WALK_STORAGE 1    /* storage units needed = GS_WALK_RANGES x the fullest range's count: what cap_units has to hold */
#define GS_WALK_ROWS 3       /* gradient rows = (intersection, quadrant) pairs some pixel took: what cap_rows has to hold */
#define GS_WALK_FLAGS 4      /* GS_FLAG_UNITS / GS_FLAG_ROWS of this call */
#define GS_WALK_SKIP 5       /* 1: the step guard was tripped when the call started -- its kernels did nothing */
#define GS_WALK_RANGES 32    /* the storage units are handed out from 32 counters, each over its own 1/32 of [0, cap_units) and in */
#define GS_WALK_RANGE0 32    /* a cache line of its own: word GS_WALK_RANGE0 + 32 * r (one counter took 30 k same-address atomics per frame) */
#define GS_WALK_WORDS (GS_WALK_RANGE0 + 32 * GS_WALK_RANGES)
/* Flag bits of an info block (info_dev[3], gs_guard_set) */
#define GS_FLAG_ISECTS 1     /* I > cap_isects */
#define GS_FLAG_TILE 2       /* a tile list longer than cap_tile */
#define GS_FLAG_COARSE 4     /* two-level binning: coarse entries > coarse_cap */
#define GS_FLAG_COARSE_LIST 8 /* two-level binning: a bin list longer than coarse_list_cap */
#define GS_FLAG_UNITS 16     /* training forward: the walk opened more work units than cap_units */
#define GS_FLAG_ROWS 32      /* training forward: more gradient rows than cap_rows */
#define GS_FLAG_BACK 64      /* depth rounds, phase 4: the front round left live tiles and no back round was enqueued */
#define GS_FLAG_PEER 128     /* view-parallel step: another rank's guard was tripped (gs_guard_merge) */

#define GS_OK 0
#define GS_ERR_ARG (-1)
#define GS_ERR_HIP (-2)
#define GS_ERR_UNSUPPORTED (-3)

/* Library identity / diagnostics. */
int gs_version(void);
const char* gs_build_flags(void); /* preprocessor flags beyond the product build's ("" = the product library; a diagnostic variant names its own) */
const char* gs_last_error(void);
const char* gs_arch(void); /* "gfx950" */

/* Step guard: lets a whole step (forward, backward, statistics, Adam) be enqueued -- and captured into a
 * hipGraph -- without the host ever reading the list sizes back (SURVEY.md section 8b "Sync").  The caller
 * sizes every intersection-indexed buffer for cap_isects entries (bucket-indexed ones for
 * cap_isects/64 + C*tiles + 1 buckets), launches the sort classes for lists of up to cap_tile entries
 * (gs_bin_emit_sort's max_tile_count = cap_tile) and zeroes info_dev[3] once.  While the guard is set (per host
 * thread; pass NULL to clear), gs_bin_count ORs into info_dev[3]: 1 if I > cap_isects, 2 if a tile list exceeds
 * cap_tile; a training gs_blend_fwd ORs 16 if its walk opens more work units than its cap_units and 32 if it leaves more
 * gradient rows than its cap_rows; and gs_bin_emit_sort, gs_blend_fwd, gs_blend_bwd, gs_project_bwd, gs_update_statistics and
 * gs_adam_step[_dev] return at once on the device when info_dev[3] != 0: a step that does not fit -- and every
 * step enqueued behind it -- is a no-op the host can detect later (read info_dev), re-size for, clear and replay.
 * n_isects / max_tile_count arguments of those entry points may then be the capacities (the passes over the slots -- the clear
 * of the quadrant masks, the row-base scan -- stop at the I the tile scan left in info_dev[0]). */
int gs_guard_set(const int64_t* info_dev, int64_t cap_isects, int64_t cap_tile);
/* The same guard for ONE call (the eager seam: every rasterization() enqueues its list stages and blend speculatively under it):
 * the flags are not sticky across calls -- the call's first flag writer (gs_bins_count's bin scan, else gs_bin_count's tile scan)
 * OVERWRITES info_dev[3], so the info block need not be zeroed between calls. */
int gs_guard_set_call(const int64_t* info_dev, int64_t cap_isects, int64_t cap_tile);

/* Host mirror of the info block (per host thread; NULL clears it): page-locked host memory the device can address
 * (hipHostMalloc / torch pin_memory), int64[8].  While set, gs_bin_count / gs_bins_count have their last kernel write the eight
 * info words there as well, followed by a system-scope fence: a caller that records an event behind the call and waits for it
 * reads the list sizes and flags from plain memory -- no device-to-host copy (a 4 us transfer kernel and the idle gap behind
 * it) on the stream.  The blocking `info_host` argument of those calls is independent of this. */
int gs_info_mirror_set(int64_t* info_host_mapped);
/* The same for the walk record of a training gs_blend_fwd (per host thread; NULL clears it): int64[4] of page-locked, device-
 * addressable host memory; the call's last kernel writes {work units, storage units taken, gradient rows, GS_FLAG_UNITS |
 * GS_FLAG_ROWS} there, followed by a system-scope fence -- the eager seam's backward learns whether the walk fitted its
 * capacities (and what it needed) from plain memory. */
int gs_walk_mirror_set(int64_t* walk_host_mapped);

/* The step guard across ranks (row e: one view per rank, the captured view-parallel step).  A rank whose lists or walk outgrew its
 * capacities must not be the only replica that skips the step.  gs_guard_flag_out writes 1.0f / 0.0f (info_dev[3] != 0) to dst0
 * and dst1 (either may be NULL): the word of this rank's all-gather record and the word of the SUM all-reduce bucket that carry
 * it.  gs_guard_merge ORs GS_FLAG_PEER into info_dev[3] when any of the n floats flags[i * stride] is non-zero (the gathered
 * records' words; the reduced bucket's word): in front of the kernels that apply the step, on every rank -- all replicas skip,
 * or none.  gs_step_applied: applied_dev[0] += 1 unless the guard is tripped (the applied-step count of gs_step_status, for a
 * step whose update is not one of the fused kernels that count themselves). */
int gs_guard_flag_out(void* stream, const int64_t* info_dev, float* dst0, float* dst1);
int gs_guard_merge(void* stream, int64_t* info_dev, const float* flags, int n, int64_t stride);
int gs_step_applied(void* stream, const int64_t* info_dev, int64_t* applied_dev);

/* Depth rounds: refine-on-demand list stages (a per-tile order equal to the stable global sort is
 * contract-equivalent).  On realistic footprints 97-98 % of what the list stages count, emit and sort is never read: a tile
 * saturates a few hundred entries into a list of thousands.  With rounds the frame's Gaussians are split at a depth quantile
 * (gs_round_split): the FRONT round lists, sorts and blends the nearest slab only; the BACK round lists the rest only into the
 * tiles the front round left with live pixels (gs_round_footprints windows every footprint to them) and resumes those tiles'
 * blend from their saved pixel states.  Every depth of the front slab is smaller than every depth behind it, so a tile walks
 * the same entries in the same order as over one list: images, sublists, checkpoints and gradient rows are bit for bit those
 * of the one-round pipeline; only lists nobody reads are never built.  A Gaussian lives in exactly one round; the back round's
 * list entries and gradient-row slots continue behind the front round's (isect_offsets, flatten_ids, slots, qmask, row_base,
 * unit_desc, ckpt, qlist are ONE set of buffers for both), gs_blend_bwd needs no change and the row gather of gs_project_bwd* /
 * gs_row_sums reads a wave's rows as two ranges.
 * gs_rounds_set (per host thread, like gs_guard_set; phase 0 clears it) tells the entry points which round they work for:
 *   phase 1 (front round)  gs_round_footprints, gs_bin_count / gs_bin_emit_sort or gs_bins_count / gs_bins_lists, gs_blend_fwd
 *   phase 2 (back round)   the same calls again, on the same buffers
 *   phase 3 (behind both)  gs_project_bwd, gs_project_bwd_adam, gs_row_sums
 * Per round the caller passes the footprints gs_round_footprints wrote (bbox_round) to the list stages, and to the backward the
 * tiles_per_gauss it wrote (a Gaussian's count in ITS round).  Under phase 1 / 2 the list entry points require C == 1.
 *   rounds_dev[GS_ROUND_WORDS] i64   device block: see GS_ROUND_*; written by gs_round_split and the list stages
 *   tile_live[tiles] u8             1: the front round left the tile with live pixels
 *   tile_state[tiles*4*64*4] f32    pixel states (T, r, g, b) of the live tiles
 *   tile_rec[tiles*8] i32           training: lengths of the quadrant sublists and their part-filled work units (NULL: inference)
 * gs_blend_fwd in phase 1 clears the quadrant masks and the walk state and leaves the row-base scan to phase 2; render_colors /
 * render_alphas are complete after phase 2 (phase 1 writes every tile, phase 2 re-writes the tiles it resumes).  When the
 * front round leaves no live tile (rounds_dev[GS_ROUND_LIVE] == 0) the kernels of the back round return at once.
 *   phase 4 (front round ALONE, under a step guard)  the same calls as phase 1, speculating that the front round finishes the frame:
 * gs_blend_fwd runs its row-base scan itself, and a tile that still has live pixels ORs GS_FLAG_BACK into the guard's flag word --
 * the step is then void like one that outgrew a capacity (every kernel behind is a no-op), and the caller repeats it with both
 * rounds.  What a captured step saves: the ~18 launches of a back round that has nothing to do. */
#define GS_ROUND_BASE 0      /* list entries (= gradient-row slots) of the front round: the back round continues here */
#define GS_ROUND_SPLIT 1     /* depth split as float bits: a Gaussian is in the front round when its depth bits are below */
#define GS_ROUND_LIVE 2      /* tiles the front round left with live pixels */
#define GS_ROUND_FRONT_N 3   /* visible Gaussians in the front round (diagnostic) */
#define GS_ROUND_LISTED_ALL 4 /* what the frame would list in ONE round (sum of tiles_per_gauss); the tile scans of both rounds copy it to info_dev[7] */
#define GS_ROUND_WORDS 8
int gs_rounds_set(int64_t* rounds_dev, uint8_t* tile_live, float* tile_state, int32_t* tile_rec, int phase);
/* Depth split of a frame (C == 1): the smallest depth bin edge d such that the Gaussians nearer than d hold at least `fraction`
 * of the frame's listed intersections (histogram over the float bits >> 19: 16 bins per octave; weights tiles_per_gauss).
 * Writes rounds_dev[GS_ROUND_SPLIT] and zeroes GS_ROUND_BASE / GS_ROUND_LIVE.  hist_ws: 4096 u32 of scratch, ZERO on first use
 * (the call leaves it zero).  fraction >= 1: everything is front. */
int gs_round_split(void* stream, int64_t N, const float* depths, const int32_t* tiles_per_gauss, float fraction,
                   uint32_t* hist_ws, int64_t* rounds_dev);
/* Footprints of the current round (gs_rounds_set phase 1 or 2), in gs_project_fwd's bbox layout: phase 1 keeps the footprints
 * of the front slab and empties the others; phase 2 keeps those of the Gaussians behind the split, windowed to the live tiles
 * (footprints of <= 32 tiles: dead tiles leave the mask; larger ones shrink to the bounding rectangle of their live tiles, and
 * get a mask when that holds <= 32 tiles).  tiles_per_gauss_round[N]: phase 1 writes every count (0 behind the split), phase 2
 * only those of the Gaussians behind the split -- afterwards it holds every Gaussian's count in its own round. */
int gs_round_footprints(void* stream, int64_t N, int tile_w, int tile_h, const uint32_t* bbox, const float* depths,
                        uint32_t* bbox_round, int32_t* tiles_per_gauss_round);

/* The round block into page-locked, device-addressable host memory (int64[GS_ROUND_WORDS]; one tiny launch + a system-scope fence,
 * like gs_step_status): a host that waits for an event behind it -- behind the front round's gs_blend_fwd -- reads
 * GS_ROUND_LIVE from plain memory and does not enqueue the back round at all when the front round left no tile live (the eager
 * seam; a captured step cannot branch on the host and lets the back round's kernels return at once instead). */
int gs_round_status(void* stream, const int64_t* rounds_dev, int64_t* status_host_mapped);

/* Publishes a guarded step's outcome without a copy or an event: one tiny launch writes
 * status[0..3] = info_dev[0..3] ({I, n_buckets, max tile, flags}), status[4] = applied_dev[0] (may be NULL) and -- walk_state
 * (gs_blend_fwd's, may be NULL) -- status[5] = storage units taken, status[6] = gradient rows: what the walk needed; status[7] =
 * tiles the front round of the step left live (under gs_rounds_set phase 3; -1 otherwise).
 * `status` may be page-locked HOST memory (hipHostMalloc'ed, device-accessible): the host then polls plain memory
 * -- the flags are sticky and the applied-step counter monotonic, so a torn read is harmless.
 * loss_ring_dev (optional, with loss3_dev = gs_l1_ssim_fwd's out3): a device ring of ring_len x 3 floats; an APPLIED step
 * n (counted from 1) logs its {l1, 1-ssim, total} in slot (n-1) mod ring_len, a skipped step logs nothing. */
int gs_step_status(void* stream, const int64_t* info_dev, const int64_t* applied_dev, int64_t* status,
                   const float* loss3_dev, float* loss_ring_dev, int ring_len, const int32_t* walk_state);

/* Persistent workspace of the eager seam (SURVEY.md section 8b "Ownership" / "Sync"; replaces the ~25 per-call allocations a
 * binding would otherwise make and lets the list stages be enqueued BEFORE the list sizes have reached the host).
 * gs_workspace_query: layout of every intermediate of one rasterization() call that does not escape to the caller, for a call
 * shape and a CAPACITY of intersections.  offsets[GS_WS_SLOTS]: byte offset of each buffer inside its arena (-1: not needed for
 * these flags); slots below GS_WS_LIST_FIRST live in the fixed arena (sized by C, N, image), those below GS_WS_WALK_FIRST in the
 * list arena (sized by cap_isects / coarse_cap: what is LISTED), the others in the walk arena (sized by cap_units / cap_rows:
 * what a training forward WALKS -- checkpoints, quadrant sublists, work units, gradient rows); arena_bytes[3] = bytes of the
 * three arenas.  Every offset is 256-byte aligned.
 * gs_workspace_bind: validates three caller-owned device arenas against a layout and zeroes the info block on `stream`; needed
 * once per (arena triple, layout) -- calls made under gs_guard_set_call leave nothing behind that the next call could trip over.
 * The caller keeps one triple per call in flight on a (device, stream) and hands the sub-pointers to the stage entry points
 * below; a call whose lists outgrow the capacity (info flags, see gs_guard_set) replaces the list arena only and repeats
 * gs_bin_count .. gs_blend_fwd; a training call whose walk outgrows cap_units / cap_rows replaces the walk arena only and repeats
 * gs_blend_fwd. */
#define GS_WS_INFO 0            /* int64[8]   {I, n_buckets, longest tile list, flags, I', longest bin list, chunks (depth rounds: tiles the front round left live), one-round I of a call in depth rounds} */
#define GS_WS_REC 1             /* f32 [C*N][12] */
#define GS_WS_BBOX 2            /* u32 [C*N][4] */
#define GS_WS_TILES_PER_GAUSS 3 /* i32 [C*N] */
#define GS_WS_CUM_TILES 4       /* i32 [C*N] */
#define GS_WS_COLORS_POST 5     /* f32 [C*N][3] */
#define GS_WS_ISECT_OFFSETS 6   /* i32 [C*tiles+1] */
#define GS_WS_BUCKET_OFFSETS 7  /* i32 [C*tiles+1] */
#define GS_WS_TILE_ORDER 8      /* i32 [C*tiles] */
#define GS_WS_QCNT 9            /* i32 [C*tiles*4]              (training) */
#define GS_WS_SH_JAC 10         /* f32 [C*N*9]                  (training: gs_project_fwd -> gs_project_bwd) */
#define GS_WS_LIST_FIRST 11     /* ---- list arena ---- */
#define GS_WS_BIN 11            /* gs_bin_workspace_bytes / gs_bins_workspace_bytes */
#define GS_WS_COARSE_KEYS 12    /* u64 [coarse_cap]             (two-level binning) */
#define GS_WS_KEYS_TMP 13       /* u64 [cap]                    (per-tile pipeline) */
#define GS_WS_SLOT_GID 14       /* i32 [cap]                    (per-tile pipeline, training) */
#define GS_WS_FLATTEN_IDS 15    /* i32 [cap] */
#define GS_WS_SLOTS_BUF 16      /* i32 [cap]                    (training) */
#define GS_WS_ISECT_IDS_BUF 17  /* i64 [cap]                    (GS_WS_ISECT_IDS) */
#define GS_WS_QMASK 18          /* u8  [cap]                    (training) */
#define GS_WS_ROW_BASE 19       /* i32 [cap / 16 + 2]           (training) */
#define GS_WS_WALK_STATE 20     /* i32 [gs_walk_state_ints(cap)] (training: counters of the walk + scan descriptors) */
#define GS_WS_WALK_FIRST 21     /* ---- walk arena (training) ---- */
#define GS_WS_CKPT 21           /* f32 [cap_units][64][4] */
#define GS_WS_QLIST 22          /* i32 [cap_units][32][2] */
#define GS_WS_UNIT_DESC 23      /* i32 [cap_units][4] */
#define GS_WS_ROWS 24           /* f32 [cap_rows][12]           (gs_blend_bwd -> gs_project_bwd) */
#define GS_WS_SLOTS 25
#define GS_WS_TRAIN 1           /* flags: the backward's lists, checkpoints and rows */
#define GS_WS_TWO_LEVEL 2       /*        two-level binning (coarse_cap, bin_shift) instead of the per-tile pipeline */
#define GS_WS_ISECT_IDS 4       /*        gsplat's isect_ids written eagerly */
int gs_workspace_query(int C, int64_t N, int width, int height, int64_t cap_isects, int64_t coarse_cap, int64_t cap_units,
                       int64_t cap_rows, int bin_shift, int flags, int64_t* offsets, int64_t* arena_bytes);
int gs_workspace_bind(void* stream, void* fixed_base, int64_t fixed_bytes, void* list_base, int64_t list_bytes,
                      void* walk_base, int64_t walk_bytes, const int64_t* offsets, const int64_t* arena_bytes);

/* Number of Gaussian groups per camera used by the binning kernels, and the bytes of scratch
 * `workspace` gs_bin_count / gs_bin_emit_sort need for (C, N, tiles). */
int gs_bin_groups(int64_t N);
size_t gs_bin_workspace_bytes(int C, int64_t N, int tile_w, int tile_h);

/* P-fwd + SH-fwd fused (replaces gsplat fully_fused_projection + spherical_harmonics +
 * clamp_min(rgb+0.5, 0); call site /root/reference/model/gaussian.py:353-367).
 * sh_degree >= 0: `colors_in` is shs[N,K,3] and sh_rest is NULL, or -- the reference model's own
 * parameter layout (/root/reference/model/gaussian.py:49-50, cat at :105-107) -- `colors_in` is
 * sh_0[N,1,3] and `sh_rest` is [N,K-1,3] (no concatenated copy is ever made).
 * sh_degree < 0: `colors_in` is already post-activation colour, [N,3] (colors_per_camera=0) or
 * [C,N,3] (=1); sh_rest is ignored.
 * Outputs: radii[C,N] i32 (0 = culled), means2d[C,N,2], depths[C,N], conics[C,N,3],
 * colors_out[C,N,3], rec[C*N*12] (mx, my, A*log2e/2, B*log2e | C*log2e/2, opacity, ext_x, ext_y |
 * r, g, b, 0: conic pre-scaled for exp2, opacity-aware half extents in pixels),
 * bbox[C*N*4] u32 (x0 | x1<<16, y0 | y1<<16: tile rectangle, min inclusive / max exclusive;
 * row-major tile bit mask for rectangles of <= 32 tiles; tile count),
 * tiles_per_gauss[C,N] i32.
 * tile_culling: 0 = gsplat's 3-sigma square (A.3; lists identical to the reference's), 1 = that
 * rectangle intersected with the opacity-aware extent and, for footprints of <= 32 tiles, an exact
 * ellipse-vs-tile test (tiles in which no pixel can reach alpha >= 1/255 are dropped; the rendered
 * image and all gradients are unchanged).
 * rect_ref[C*N*2] u32 (optional, may be NULL; written by stages 0 and 1): gsplat's 3-sigma tile rectangle itself (x0 | x1<<16,
 * y0 | y1<<16; 0 for culled Gaussians) whatever tile_culling did to bbox -- with tile_culling = 1 the caller can render from
 * the short lists and still build gsplat's exact list arrays (a function of this rectangle and `depths`) when somebody reads them.
 * stage: 0 = everything; 1 = geometry only (all outputs except colors_out and the colour quad of
 * rec); 2 = colour only, for the Gaussians a previous stage-1 call marked visible in radii.  Calling
 * 1, then the gs_bin_count kernels, then 2 lets the colour pass overlap the host read-back of I.
 * sh_jac[C*N*9] (optional, may be NULL; written by stages 0 and 2 for visible Gaussians when sh_degree >= 1; opaque to the
 * caller): the 3 x 3 Jacobian d(pre-clamp colour)/d(unit view direction), eight entries as [C*N][8] and the ninth as a plane
 * [C*N] behind them.  Handed to gs_project_bwd /
 * gs_project_bwd_adam it makes the backward independent of the SH coefficients: v_sh = Y(u) (x) v_pre, and the direction
 * term of v_means is sh_jac^T v_pre -- the backward then never streams the 48 coefficients per Gaussian. */
int gs_project_fwd(void* stream, int C, int64_t N, int K, int sh_degree, const float* means,
                   const float* quats, const float* scales, const float* opacities,
                   const float* colors_in, const float* sh_rest, int colors_per_camera,
                   const float* viewmats, const float* Ks, int width, int height, float eps2d,
                   float near_plane, float far_plane, float radius_clip, int tile_culling, int stage, int activations,
                   int32_t* radii,
                   float* means2d, float* depths, float* conics, float* colors_out, float* rec, uint32_t* bbox,
                   int32_t* tiles_per_gauss, uint32_t* rect_ref, float* sh_jac);

/* I-count (replaces the counting half of gsplat isect_tiles + its cumsum and
 * isect_offset_encode).  Writes isect_offsets[C*tiles+1] (exclusive; last = I),
 * bucket_offsets[C*tiles+1] (exclusive scan of ceil(count/64)), info_dev[4] =
 * {I, n_buckets, max entries in one tile, 0}.  If info_host != NULL the four values are copied
 * there and the stream is synchronised (the one host sync of the exact mode).
 * tile_order[C*tiles] (optional, may be NULL): a launch order for gs_blend_fwd, tiles with the longest
 * lists first (a permutation of 0 .. C*tiles-1; the order among lists of similar length is arbitrary). */
int gs_bin_count(void* stream, int C, int64_t N, int tile_w, int tile_h, const uint32_t* bbox,
                 void* workspace, size_t workspace_bytes, int32_t* isect_offsets,
                 int32_t* bucket_offsets, int32_t* tile_order, int64_t* info_dev, int64_t* info_host);

/* I-emit + per-tile depth sort (replaces the emitting half of isect_tiles and the global
 * cub::DeviceRadixSort).  Needs the workspace as left by gs_bin_count and consumes it (the sort's work-list counters
 * are zeroed by the count): one gs_bin_emit_sort per gs_bin_count, in that order.  keys_tmp[I] u64 and
 * slot_gid[I] i32 are scratch.  Outputs: cum_tiles[C*N] (exclusive scan of tiles_per_gauss =
 * first gradient-row slot of each flatten id), isect_ids[I] i64 (cam | tile | depth bits, sorted; may be NULL: it is
 * a function of the other outputs -- cam | tile from isect_offsets, depth bits from depths[flatten_ids] -- and half of the list bytes),
 * flatten_ids[I] i32 (sorted), slots[I] i32 (gradient-row slot of each sorted entry).  Inference lists: pass slot_gid =
 * slots = NULL -- the keys then carry the flatten id itself (no slot map is written or gathered; same order). */
int gs_bin_emit_sort(void* stream, int C, int64_t N, int tile_w, int tile_h, const uint32_t* bbox,
                     const float* depths, void* workspace, size_t workspace_bytes,
                     const int32_t* isect_offsets, int64_t n_isects, int64_t max_tile_count,
                     uint64_t* keys_tmp, int32_t* slot_gid, int32_t* cum_tiles, int64_t* isect_ids,
                     int32_t* flatten_ids, int32_t* slots);

/* Two-level binning: the same outputs as gs_bin_count + gs_bin_emit_sort (bit for bit), built differently.  A bin is
 * 2x2 (bin_shift 1) or 4x4 (2) tiles (0: the library picks from N and the tile grid).  gs_bins_count emits every
 * Gaussian into the bins its tile rectangle touches as a (depth bits << 32 | flatten id) key -- I' entries, 1.4-17 per
 * Gaussian where the tile lists hold 3-190 --, depth-sorts each bin's list in LDS, and counts each tile's entries out
 * of its bin's sorted list; gs_bins_lists repeats that walk and writes the per-tile lists by ordered compaction (ballot
 * + popcount: no atomics, no second sort; depth and tie order are inherited from the bin).  The per-tile pipeline sorts
 * I entries, this one I'; it pays from ~6 tile-list entries per Gaussian upwards (real captures: tens to hundreds).
 *   coarse_keys[coarse_cap] u64   scratch for the bin lists (kept between the two calls)
 *   coarse_list_cap               longest bin list the sort classes launched must take (<= 0: launch every class)
 *   info_dev[8]                   {I, n_buckets, longest tile list, flags, I', longest bin list, chunks, -}
 *                                 flags: 1 I > guard capacity | 4 I' > coarse_cap | 8 a bin list > coarse_list_cap;
 *                                 with flags != 0 nothing was emitted: repeat gs_bins_count with the sizes reported
 *   cum_tiles[C*N]                exclusive scan of tiles_per_gauss (first gradient-row slot of each flatten id)
 *   isect_ids                     may be NULL (see gs_bin_emit_sort)
 * workspace: gs_bins_workspace_bytes(C, N, tile_w, tile_h, bin_shift, coarse_cap), same arguments in both calls.
 * If info_host != NULL the eight values are copied there and the stream is synchronised. */
size_t gs_bins_workspace_bytes(int C, int64_t N, int tile_w, int tile_h, int bin_shift, int64_t coarse_cap);
int gs_bins_count(void* stream, int C, int64_t N, int tile_w, int tile_h, int bin_shift, const uint32_t* bbox,
                  const float* depths, void* workspace, size_t workspace_bytes, uint64_t* coarse_keys,
                  int64_t coarse_cap, int64_t coarse_list_cap, int32_t* cum_tiles, int32_t* isect_offsets,
                  int32_t* bucket_offsets, int32_t* tile_order, int64_t* info_dev, int64_t* info_host);
int gs_bins_lists(void* stream, int C, int64_t N, int tile_w, int tile_h, int bin_shift, const uint32_t* bbox,
                  void* workspace, size_t workspace_bytes, const uint64_t* coarse_keys, int64_t coarse_cap,
                  const int32_t* cum_tiles, const int32_t* isect_offsets, int64_t* isect_ids, int32_t* flatten_ids,
                  int32_t* slots, const int64_t* info_dev);

/* B-fwd (replaces gsplat rasterize_to_pixels forward).  backgrounds[C,3] may be NULL.
 * Outputs render_colors[C,H,W,3], render_alphas[C,H,W,1].  One wavefront per 16x16 tile, each
 * lane owning one pixel of each 8x8 quadrant.  Inference: pass ckpt = NULL (and NULL for every
 * list output).  Training (ckpt != NULL) additionally emits what gs_blend_bwd consumes -- sized by what the forward WALKS (a
 * saturated tile abandons the rest of its list), not by what is listed:
 *   qcnt[C*tiles*4]     lengths of the four compacted, depth-ordered quadrant sublists of every tile
 *   unit_desc[cap_units*4] i32   one work unit per GS_UNIT entries of a sublist: (tile*4+quadrant, position of the unit in its
 *                       sublist, storage unit, 0), dense in [0, walk_state[GS_WALK_UNITS])
 *   ckpt[cap_units*64*4] f32     row `storage unit`: the quadrant's 64 pixel states in front of the unit (T -- negative once
 *                       saturated / outside the image -- and accumulated rgb)
 *   qlist[cap_units*32*2] i32    block `storage unit`: the unit's (flatten id, gradient-row slot) pairs
 *   qmask[n_isects] u8  by gradient-row slot: which quadrant rows of an intersection exist.  Cleared by this call (one
 *                       streaming pass), then only the non-zero masks are stored: a scattered one-byte
 *                       store leaves L2 as a 32-byte partial write (profiles/r03_traffic_calibration.json)
 *   row_base[n_isects/16+2] i32  exclusive scan of popcount(qmask) over the slots, sampled every 16 slots (three small launches
 *                       behind the blend): rows_before(s) = row_base[s / 16] + popcount of the masks of slots [16 (s / 16), s);
 *                       the gradient rows of slot s are rows [rows_before(s), rows_before(s + 1)), in quadrant order -- the
 *                       rows of a Gaussian (contiguous slots) and of consecutive Gaussians are contiguous
 *   walk_state[gs_walk_state_ints(n_isects)] i32   counters (GS_WALK_*) and the scan's chunk counts; cleared by this call.
 * Storage units are taken in chunks of 8 per tile from GS_WALK_RANGES counters (a tile's launch slot picks one), each over its
 * own 1/32 of [0, cap_units).  When a range runs out -- the walk needs more than cap_units units, give or take the imbalance -- or
 * leaves more than cap_rows rows, the call is void: GS_FLAG_UNITS / GS_FLAG_ROWS are ORed into walk_state[GS_WALK_FLAGS] and
 * into the step guard's flag word (gs_guard_set), and walk_state[GS_WALK_STORAGE] / [GS_WALK_ROWS] hold what was needed
 * (render_colors / render_alphas are complete either way).  qmask / row_base 16-byte aligned; cap_units >= 256.
 * Inside the forward a pixel's state is one float: T in (1e-4, 1] = live; a finished pixel (stop rule fired / outside the image)
 * carries its final transmittance scaled by 2^64 (exact in fp32; "live" is T <= 1).  Checkpoints do NOT store that form: a live
 * pixel's checkpoint holds T, a finished one holds -1 (gs_blend_bwd looks at the sign only; the final T of a finished pixel comes
 * from render_alphas). */
size_t gs_walk_state_ints(int64_t n_isects);
int gs_blend_fwd(void* stream, int C, int width, int height, const float* rec,
                 const float* backgrounds, const int32_t* isect_offsets, const int32_t* tile_order,
                 const int32_t* flatten_ids, const int32_t* slots, int64_t n_isects, float* render_colors,
                 float* render_alphas, float* ckpt, int32_t* qlist, int32_t* qcnt, uint8_t* qmask,
                 int32_t* unit_desc, int64_t cap_units, int32_t* row_base, int64_t cap_rows, int32_t* walk_state);

/* B-bwd (replaces rasterize_to_pixels backward incl. absgrad).  Gaussian-parallel: eight 8-lane
 * systolic pipelines per wavefront, one GS_UNIT-entry work unit each; no atomics.  Writes one
 * 12-float row per (intersection, quadrant) some pixel took, at rows[(row_base[slot] + rank of the quadrant among the slot's
 * rows)*12]: (v_mx, v_my, |v_mx|, |v_my|, v_A, v_B, v_C, v_opacity, v_r, v_g, v_b, 0); rows[cap_rows*12].
 * The launch covers cap_units work units (pipelines past walk_state[GS_WALK_UNITS] return at once).  v_render_alphas may be NULL.
 * unit_classes (may be NULL): scratch of gs_unit_classes_ints(cap_units, C, width, height) int32, 16-byte aligned.  With it the
 * call first groups the published units by FILL CLASS -- the last unit of a quadrant sublist is part-filled; units of at most 8 /
 * 16 / 24 entries run 1 / 2 / 3 entries per lane instead of 4 -- and every wave takes eight units of one class.  Same rows,
 * bit for bit (a unit writes rows of its own; the order units run in decides nothing). */
size_t gs_unit_classes_ints(int64_t cap_units, int C, int width, int height);
int gs_blend_bwd(void* stream, int C, int width, int height, const float* rec,
                 const int32_t* qlist, const int32_t* qcnt, const int32_t* unit_desc, int64_t cap_units,
                 const float* ckpt, const uint8_t* qmask, const int32_t* row_base, const int32_t* walk_state,
                 const float* render_colors, const float* render_alphas, const float* v_render_colors,
                 const float* v_render_alphas, float* rows, int32_t* unit_classes);

/* Row reduction + SH-bwd + P-bwd fused (replaces the atomics of the blend backward,
 * spherical_harmonics backward and fully_fused_projection backward).  Sums each Gaussian's rows
 * (slots [cum_tiles[f], cum_tiles[f]+tiles_per_gauss[f]) = rows [rows_before(first slot), rows_before(last slot + 1)): row_base + qmask) and
 * pushes the result through the colour and projection VJPs.  Outputs (all fully written):
 * v_means[N,3], v_quats[N,4], v_scales[N,3], v_opacities[N], v_colors: v_shs[N,K,3]
 * (sh_degree>=0; with the split layout v_colors is v_sh_0[N,1,3] and v_sh_rest[N,K-1,3]) or
 * v_colors[N,3]/[C,N,3]; v_means2d_abs[C,N,2] (the `.absgrad` side channel,
 * /root/reference/model/gaussian.py:191).
 * Optional (may be NULL): v_means2d[C,N,2], v_conics[C,N,3], v_colors_post[C,N,3] (gradient of the
 * post-clamp colour), v_colors_pre[C,N,3] (SH colours only: gradient of the pre-clamp colour, zero
 * for culled Gaussians).  With SH colours v_colors (and v_sh_rest) may be NULL: the SH-parameter
 * gradients are then left to gs_sh_grad_views / gs_sh_adam_views (fed by v_colors_pre from here or from
 * gs_row_sums).
 * activations != 0 (both directions): `scales` / `opacities` are the reference model's log-scales and
 * logit opacities (/root/reference/model/gaussian.py:98-103); exp / sigmoid are applied inside and
 * v_scales / v_opacities are gradients w.r.t. those raw parameters.  0 = gsplat's contract.
 * sh_jac (optional, may be NULL): gs_project_fwd's direction Jacobian of the same inputs; with it colors_in / sh_rest are not
 * read (same gradients to rounding: the direction term of v_means is then summed as J^T v_pre instead of per coefficient).
 * row_sums[C*N][12] (optional, may be NULL): the row sums gs_row_sums left -- rows / row_base / qmask are then not read (may be NULL);
 * same results bit for bit.  stat_grad_norm[N] / stat_count[N] (optional, both or neither; C = 1): this view's two additive
 * statistics of /root/reference/model/gaussian.py:188-197, WRITTEN not accumulated -- |absgrad|_2 * max(width, height) and 1 for
 * visible Gaussians, 0 for culled ones (the segments of the view-parallel step's SUM all-reduce, see gs_pack_view_step). */
int gs_project_bwd(void* stream, int C, int64_t N, int K, int sh_degree, const float* means,
                   const float* quats, const float* scales, const float* colors_in,
                   const float* sh_rest, int colors_per_camera, const float* viewmats,
                   const float* Ks, int width, int height, float eps2d, float near_plane, float far_plane,
                   const int32_t* radii, const float* colors_post, const int32_t* tiles_per_gauss,
                   const int32_t* cum_tiles, const float* rows, const int32_t* row_base, const uint8_t* qmask,
                   float* v_means, float* v_quats, float* v_scales, float* v_opacities,
                   float* v_colors, float* v_sh_rest, float* v_means2d_abs, float* v_means2d,
                   float* v_conics, float* v_colors_post, float* v_colors_pre,
                   const float* opacities, int activations, const float* sh_jac, const float* row_sums,
                   float* stat_grad_norm, float* stat_count);

/* Row e (view sharding): the row sums of every Gaussian as a pass of its own, and from them everything another rank needs of
 * this view before the long projection backward runs.  row_sums[C*N][12] = the 11 sums gs_project_bwd forms first (same
 * function, same bits; 12th float 0; rows of culled Gaussians are left unwritten) -> gs_project_bwd(..., row_sums);
 * v_colors_pre[C*N*3] = the clamp-masked pre-clamp colour gradient (identical to gs_project_bwd's optional output);
 * radii_norm[C*N] (optional) = radius / max_hw, 0 for culled Gaussians; cam_out[16] (optional, C = 1) = a
 * copy of viewmats[0].  With v_colors_pre / radii_norm / cam_out pointing into one buffer [3N | N | 16] this launch fills a
 * rank's whole all-gather payload of the view-parallel step (gs_sh_adam_views).  Honours the step guard. */
int gs_row_sums(void* stream, int C, int64_t N, const int32_t* radii, const float* colors_post,
                const int32_t* tiles_per_gauss, const int32_t* cum_tiles, const float* rows, const int32_t* row_base, const uint8_t* qmask,
                float* row_sums, float* v_colors_pre, float* radii_norm, float max_hw, const float* viewmats, float* cam_out);

/* Row e (view sharding): dense SH-parameter gradients of R views rebuilt from the per-view
 * pre-clamp colour gradients,  v_sh[n][k][:] = sum_r Y_k(dir(means[n], camera r)) * v_colors_pre[r][n][:]
 * (r ascending; the SH VJP of spherical_harmonics backward).  Ranks exchange
 * v_colors_pre[N,3] per view (12 B per Gaussian) instead of the 48 SH gradients (192 B).
 * viewmats[R,4,4]; output layout as gs_project_bwd: v_colors[N,K,3], or split v_colors = v_sh_0[N,1,3]
 * and v_sh_rest[N,K-1,3].  R <= 64. */
int gs_sh_grad_views(void* stream, int R, int64_t N, int K, int sh_degree, const float* means,
                     const float* viewmats, const float* v_colors_pre, float* v_colors,
                     float* v_sh_rest);

/* Row e (view sharding): gs_sh_grad_views + the SH half of gs_adam_step in one pass -- the dense SH gradient (192 B per
 * Gaussian at SH3) is never written or re-read.  payload: the all-gathered per-view records, view r at payload +
 * r * payload_stride floats: [3N pre-clamp colour gradients | N normalised radii | 16 floats of the view matrix] (what
 * gs_row_sums fills; payload_stride >= 4N + 16).  sh_0[N,1,3] / sh_rest[N,K-1,3] and their moments are updated in place with
 * g = grad_scale * sum_r Y(dir(means[n], camera r)) (x) v_colors_pre[r][n] (r ascending: bitwise identical replicas), Adam
 * arithmetic and bias corrections of gs_adam_step(step); max_radii[N] (optional) = max(max_radii, max_r radii[r]).
 * Same update bit for bit as gs_sh_grad_views followed by gs_adam_step over the two SH segments.  Honours the step guard. */
int gs_sh_adam_views(void* stream, int R, int64_t N, int K, int sh_degree, const float* means, const float* payload,
                     int64_t payload_stride, float* sh_0, float* sh_0_exp_avg, float* sh_0_exp_avg_sq, float* sh_rest,
                     float* sh_rest_exp_avg, float* sh_rest_exp_avg_sq, float lr_sh_0, float lr_sh_rest, float beta1, float beta2,
                     float eps, int64_t step, float grad_scale, float* max_radii);

/* ---- "next" row f-1: the loss that feeds v_render_colors ----
 * Fused L1 + (1 - SSIM) of /root/reference/model/gaussian.py:415-453 (torchmetrics SSIM:
 * 11x11 Gaussian window sigma 1.5, K1 0.01, K2 0.03, data_range 1, mean over the interior).
 * render / gt are [H,W,3]; mask [H,W] may be NULL (render := mask*gt + (1-mask)*render).
 * workspace holds gs_loss_workspace_floats(H,W) floats and carries the SSIM derivative maps from
 * the forward to the backward.  out3 = {l1, 1-ssim, (1-lambda)*l1 + lambda*(1-ssim)}.
 * v_total: device scalar d(loss)/d(out3[2]);  v_render[H,W,3] is fully written.
 * clamp_input != 0: `render` is the rasterizer's un-clamped image; torch.clamp(render, 0, 1) of
 * GaussianModel.forward (/root/reference/model/gaussian.py:368) and its backward are applied inside.
 * Sizes: height, width > 10 (the window), height * width <= 2^28, width <= 2^20 and height <= 2^24 (32-bit byte offsets, 24-bit row multiplies);
 * anything else is refused with GS_ERR_ARG before a launch. */
size_t gs_loss_workspace_floats(int height, int width);
int gs_l1_ssim_fwd(void* stream, int height, int width, float lambda_ssim, const float* render,
                   const float* gt, const float* mask, int clamp_input, float* workspace, float* out3);
int gs_l1_ssim_bwd(void* stream, int height, int width, float lambda_ssim, const float* render,
                   const float* gt, const float* mask, int clamp_input, const float* workspace,
                   const float* v_total, float* v_render);

/* The same two entries for a CAPTURED step (the graph's kernel arguments are frozen at capture, the image a step trains on is
 * not): ground truth and mask are read through device memory, slots_dev[0] = gt, slots_dev[1] = mask, which gs_step_inputs
 * writes in front of every replay.  has_mask (known at capture) selects the forward's instantiation; the backward tests the slot. */
int gs_l1_ssim_fwd_slots(void* stream, int height, int width, float lambda_ssim, const float* render,
                         const float* const* slots_dev, int has_mask, int clamp_input, float* workspace, float* out3);
int gs_l1_ssim_bwd_slots(void* stream, int height, int width, float lambda_ssim, const float* render,
                         const float* const* slots_dev, int clamp_input, const float* workspace,
                         const float* v_total, float* v_render);

/* Row a-2: `torch.clamp(render, 0, 1)` of /root/reference/model/gaussian.py:368 as one pass.
 * v_out == NULL: out = clamp(x, 0, 1).  v_out != NULL: out = v_out where 0 <= x <= 1, else 0 (the
 * backward of that clamp).  n floats, 16-byte aligned buffers. */
int gs_clamp01(void* stream, int64_t n, const float* x, const float* v_out, float* out);

/* ---- "next" row f-2: fused Adam over flat buffers (one torch.optim.Adam with six groups in the
 * reference, /root/reference/model/gaussian.py:389-412; defaults: no weight decay / amsgrad).
 * params, exp_avg, exp_avg_sq: flat fp32 device buffers of n elements holding n_segments parameter
 * tensors back to back, each padded to a multiple of 4 elements.  HOST arrays per segment:
 * seg_ends_host (exclusive padded end), seg_lens_host (true element count), seg_grads_host (device
 * pointer of that tensor's contiguous, 16-byte aligned gradient; NULL = skip the segment, as torch
 * skips `p.grad is None`), seg_lrs_host.  step counts from 1.  Gradients are multiplied by
 * grad_scale first (1.0 = torch semantics; 1/world turns a sum over ranks into the mean). */
int gs_adam_step(void* stream, int64_t n, float* params, float* exp_avg, float* exp_avg_sq,
                 int n_segments, const int64_t* seg_ends_host, const int64_t* seg_lens_host,
                 const float* const* seg_grads_host, const float* seg_lrs_host, float beta1,
                 float beta2, float eps, int64_t step, float grad_scale);

/* The same step with two additive statistics riding along in the launch (the view-parallel step's geometry half, row e):
 * stat_dst0[i] += stat_src0[i], stat_dst1[i] += stat_src1[i] for i < stat_n -- the all-reduced |absgrad| norm and visibility
 * count of /root/reference/model/gaussian.py:188-197 into grad_norm_accum / collecting_counts.  stat_n = 0: exactly gs_adam_step.
 * (Both entry points walk only the segments that have a gradient.) */
int gs_adam_step_stats(void* stream, int64_t n, float* params, float* exp_avg, float* exp_avg_sq,
                       int n_segments, const int64_t* seg_ends_host, const int64_t* seg_lens_host,
                       const float* const* seg_grads_host, const float* seg_lrs_host, float beta1, float beta2,
                       float eps, int64_t step, float grad_scale, int64_t stat_n, const float* stat_src0,
                       const float* stat_src1, float* stat_dst0, float* stat_dst1);

/* Replayable form of gs_adam_step for a captured step: bias corrections and learning rates come from the device
 * array hyper_dev[1 + n_segments] = {1/sqrt(1-beta2^t), lr_k/(1-beta1^t)...}, which gs_adam_hyper writes (values
 * travel as kernel arguments of a one-thread launch, so successive steps cannot race on a host buffer).
 * applied_dev (optional): device counter incremented by every launch the step guard did not skip. */
int gs_adam_hyper(void* stream, int n_segments, const float* seg_lrs_host, float beta1, float beta2, int64_t step,
                  float* hyper_dev);
/* gs_adam_hyper + everything else that changes from one replay of a captured step to the next, in ONE launch (values travel
 * as kernel arguments): viewmat_dst[16] <- viewmat_src[16] and K_dst[9] <- K_src[9] (device pointers; a NULL source leaves
 * its destination alone), slots_dev[0] = gt, slots_dev[1] = mask (the POINTERS gs_l1_ssim_*_slots read through; slots_dev
 * may be NULL).  A step on another view costs no copy of its ground-truth image. */
int gs_step_inputs(void* stream, int n_segments, const float* seg_lrs_host, float beta1, float beta2, int64_t step,
                   float* hyper_dev, const float* viewmat_src, float* viewmat_dst, const float* K_src, float* K_dst,
                   const float* gt, const float* mask, const float** slots_dev);
int gs_adam_step_dev(void* stream, int64_t n, float* params, float* exp_avg, float* exp_avg_sq, int n_segments,
                     const int64_t* seg_ends_host, const int64_t* seg_lens_host, const float* const* seg_grads_host,
                     float beta1, float beta2, float eps, float grad_scale, const float* hyper_dev, int64_t* applied_dev);

/* ---- "next" row f-3: densify / prune on the device (/root/reference/model/gaussian.py:199-349) ----
 * gs_refine_flags: per Gaussian, the reference's decisions.  flags[3][n] (0/1): row 0 the old Gaussian survives the
 * prune (low opacity | max_radii > ratio | largest scale > prune_scale | it was split); row 1 it is split AND its
 * children survive (children: same opacity, scales / (0.8 S)); row 2 it is cloned AND the clone survives.
 * counters[5] (device, zeroed here) = {split, clone, then the reference's cumulative prune counts over [old | new]:
 * low opacity, + large radii, + large scale} -- what densify_and_prune returns as tb_info.
 * gs_refine_apply: given the INCLUSIVE prefix scans of the three flag rows (gs_scan_rows_i32) and
 * their totals (the one host read of the path: they size the new buffers), fills the new flat parameter / exp_avg /
 * exp_avg_sq buffers in the reference's order [surviving old | split children, copy-major | clones]: survivors keep
 * their moments, new Gaussians start at zero; split children get mean + R(q)(s * noise[copy][parent]) and
 * log(s / (0.8 S)).  Flat layout as gs_adam_step: the six tensors of param_names (means, log_scales, quats, sh_0,
 * sh_rest, logit_opacities) at old_offsets_host[6] / new_offsets_host[6] floats.  src_scratch[n_new] i32 and
 * tag_scratch[n_new] i8 are scratch.  noise: [S][n_old][3] standard normal. */
/* Inclusive prefix scan of every row of an int32 [rows][n] array.  workspace: gs_scan_rows_workspace_ints(rows, n) int32. */
size_t gs_scan_rows_workspace_ints(int rows, int64_t n);
int gs_scan_rows_i32(void* stream, int rows, int64_t n, const int32_t* in, int32_t* out, int32_t* workspace);
int gs_refine_flags(void* stream, int64_t n, int num_splits, float densify_grad_thresh, float densify_scale_thresh,
                    float prune_radii_ratio_thresh, float prune_scale_thresh, float min_opacity,
                    const float* grad_norm_accum, const float* counts, const float* max_radii, const float* log_scales,
                    const float* logit_opacities, int32_t* flags, int64_t* counters);
int gs_refine_apply(void* stream, int64_t n_old, int num_splits, int K, const int32_t* flags, const int32_t* flags_incl,
                    int64_t tot_old, int64_t tot_child, int64_t tot_clone, const float* noise, const float* old_params,
                    const float* old_exp_avg, const float* old_exp_avg_sq, const int64_t* old_offsets_host, float* new_params,
                    float* new_exp_avg, float* new_exp_avg_sq, const int64_t* new_offsets_host, int32_t* src_scratch,
                    int8_t* tag_scratch);

/* gs_project_bwd + gs_adam_step_dev in one pass, for the reference's own single-camera step with the model's raw
 * parameters (log-scales, logit opacities, split SH: activations as in gs_project_bwd(..., activations = 1)): no
 * gradient is written; each parameter element and its two moments are updated in place where its gradient is
 * formed (saves writing and re-reading 59 floats per Gaussian).  params / exp_avg / exp_avg_sq and offsets_host[6]
 * (means, log_scales, quats, sh_0, sh_rest, logit_opacities; floats) describe the flat buffers of gs_adam_step.
 * Same update, bit for bit, as gs_project_bwd followed by gs_adam_step_dev.  Honours the step guard.
 * max_radii / grad_norm_accum / counts (optional, all three or none): gs_update_statistics is applied in the same pass.
 * sh_jac (optional): as in gs_project_bwd. */
int gs_project_bwd_adam(void* stream, int64_t N, int K, int sh_degree, float* params, float* exp_avg, float* exp_avg_sq,
                        const int64_t* offsets_host, const float* viewmats, const float* Ks, int width, int height, float eps2d,
                        float near_plane, float far_plane, const int32_t* radii, const float* colors_post,
                        const int32_t* tiles_per_gauss, const int32_t* cum_tiles, const float* rows, const int32_t* row_base, const uint8_t* qmask,
                        float* v_means2d_abs, float beta1, float beta2, float eps, const float* hyper_dev, int64_t* applied_dev,
                        float* max_radii, float* grad_norm_accum, float* counts, const float* sh_jac);

/* Row e: this rank's contribution to the SUM all-reduce of the view-parallel step in one pass: the four
 * geometry gradients and this view's two additive statistics (|absgrad|_2 * max_hw, visibility count)
 * packed into flat = [means 3N | log_scales 3N | quats 4N | logit_opacities N | grad_norm N | count N],
 * every segment padded to a multiple of 4 floats (flat holds 4*ceil(3N/4)*2 + 4N + 3*4*ceil(N/4) floats).
 * v_means = v_scales = v_quats = v_opacities = NULL: gs_project_bwd has written its four gradients into those segments of
 * `flat` itself (they are plain output pointers there); only the two statistics segments are filled in. */
int gs_pack_view_step(void* stream, int64_t n, float max_hw, const float* v_means, const float* v_scales,
                      const float* v_quats, const float* v_opacities, const int32_t* radii, const float* absgrad,
                      float* flat);

/* Row a-3: the consumer of the side channels, `GaussianModel.update_statistics`
 * (/root/reference/model/gaussian.py:188-197), as one launch for the reference's single camera:
 * for radii[i] > 0: max_radii = max(max_radii, radii/max_hw); grad_norm_accum += |absgrad[i]|_2 * max_hw;
 * counts += 1.  absgrad is [N,2]. */
int gs_update_statistics(void* stream, int64_t n, float max_hw, const int32_t* radii, const float* absgrad,
                         float* max_radii, float* grad_norm_accum, float* counts);

#ifdef __cplusplus
}
#endif
#endif /* GS_RASTER_H_ */
