// gs_workspace.hip -- the persistent, caller-owned scratch of the eager seam (SURVEY.md section 8b "Ownership": "the native
// layer ... allocates nothing that outlives [the call] except a per-device, per-stream reusable workspace"; the entry points
// `gs_workspace_query/bind` that row names).  The library owns the LAYOUT -- which intermediates exist for a call shape,
// their sizes and alignment --, the caller owns the MEMORY: three arenas per call in flight,
//   fixed arena: everything sized by (C, N, image) -- packed records, footprints, offsets, counters;
//   list arena:  everything sized by a CAPACITY of intersections (and of coarse-bin entries) -- keys, sorted lists, the
//                quadrant masks and row bases of a training call: what is LISTED;
//   walk arena:  (training) everything sized by capacities of work units and gradient rows -- checkpoints, quadrant
//                sublists, work-unit descriptors, rows: what the forward WALKS
// so that a call whose lists outgrow the capacity replaces only the list arena and repeats the list stages, and a training
// call whose walk outgrows its capacities replaces only the walk arena and repeats the blend.
#include "gs_common.h"

namespace gs {
static inline int64_t align256(int64_t x) { return (x + 255) & ~(int64_t)255; }
}  // namespace gs

extern "C" int gs_workspace_query(int C, int64_t N, int width, int height, int64_t cap_isects, int64_t coarse_cap, int64_t cap_units,
                                  int64_t cap_rows, int bin_shift, int flags, int64_t* offsets, int64_t* arena_bytes) {
    GS_REQUIRE(C >= 1 && N >= 0 && width > 0 && height > 0, "C>=1, N>=0, positive image size");
    GS_REQUIRE(cap_isects >= 0 && cap_isects < (1ll << 31) && coarse_cap >= 0 && coarse_cap < (1ll << 31), "capacities must fit int32");
    GS_REQUIRE(offsets && arena_bytes, "null output pointer");
    const bool train = flags & GS_WS_TRAIN, two_level = flags & GS_WS_TWO_LEVEL;
    GS_REQUIRE(!train || (cap_units >= 256 && cap_units < (1ll << 26) && cap_rows >= 0 && cap_rows < (1ll << 31)),
               "training mode: 256 <= cap_units < 2^26 work units, cap_rows < 2^31 gradient rows");
    const int tw = (width + GS_TILE - 1) / GS_TILE, th = (height + GS_TILE - 1) / GS_TILE;
    const int64_t tiles = (int64_t)tw * th, CN = (int64_t)C * N, CT = (int64_t)C * tiles;
    const int64_t cap = cap_isects;
    int64_t size[GS_WS_SLOTS];
    for (int i = 0; i < GS_WS_SLOTS; ++i) size[i] = 0;
    size[GS_WS_INFO] = 8 * sizeof(int64_t);
    size[GS_WS_REC] = CN * GS_REC_FLOATS * 4;
    size[GS_WS_BBOX] = CN * 16;
    size[GS_WS_TILES_PER_GAUSS] = CN * 4;
    size[GS_WS_CUM_TILES] = CN * 4;
    size[GS_WS_COLORS_POST] = CN * 12;
    size[GS_WS_ISECT_OFFSETS] = (CT + 1) * 4;
    size[GS_WS_BUCKET_OFFSETS] = (CT + 1) * 4;
    size[GS_WS_TILE_ORDER] = CT * 4;
    if (train) {
        size[GS_WS_QCNT] = CT * 4 * 4;
        size[GS_WS_SH_JAC] = CN * 9 * 4;
    }
    if (two_level) {
        const size_t b = gs_bins_workspace_bytes(C, N, tw, th, bin_shift, coarse_cap);
        GS_REQUIRE(b > 0, "bin_shift: 0, 1 or 2");
        size[GS_WS_BIN] = (int64_t)b;
        size[GS_WS_COARSE_KEYS] = coarse_cap * 8;
    } else {
        size[GS_WS_BIN] = (int64_t)gs_bin_workspace_bytes(C, N, tw, th);
        size[GS_WS_KEYS_TMP] = cap * 8;
        if (train) size[GS_WS_SLOT_GID] = cap * 4;
    }
    size[GS_WS_FLATTEN_IDS] = cap * 4;
    if (flags & GS_WS_ISECT_IDS) size[GS_WS_ISECT_IDS_BUF] = cap * 8;
    if (train) {
        size[GS_WS_SLOTS_BUF] = cap * 4;
        size[GS_WS_QMASK] = cap + 16;
        size[GS_WS_ROW_BASE] = (cap / 16 + 2) * 4;
        size[GS_WS_WALK_STATE] = (int64_t)gs_walk_state_ints(cap) * 4;
        size[GS_WS_CKPT] = cap_units * 64 * 16;
        size[GS_WS_QLIST] = cap_units * GS_UNIT * 8;
        size[GS_WS_UNIT_DESC] = cap_units * 16;
        size[GS_WS_ROWS] = (cap_rows > 0 ? cap_rows : 1) * GS_ROW_FLOATS * 4;
    }
    int64_t off[3] = {0, 0, 0};
    for (int i = 0; i < GS_WS_SLOTS; ++i) {
        const int arena = i >= GS_WS_WALK_FIRST ? 2 : (i >= GS_WS_LIST_FIRST ? 1 : 0);
        if (size[i] == 0) { offsets[i] = -1; continue; }
        offsets[i] = off[arena];
        off[arena] = gs::align256(off[arena] + size[i]);
    }
    for (int k = 0; k < 3; ++k) arena_bytes[k] = off[k] > 0 ? off[k] : 256;
    return GS_OK;
}

extern "C" int gs_workspace_bind(void* stream, void* fixed_base, int64_t fixed_bytes, void* list_base, int64_t list_bytes,
                                 void* walk_base, int64_t walk_bytes, const int64_t* offsets, const int64_t* arena_bytes) {
    GS_REQUIRE(fixed_base && list_base && walk_base && offsets && arena_bytes, "null pointer");
    GS_REQUIRE(((uintptr_t)fixed_base & 255) == 0 && ((uintptr_t)list_base & 255) == 0 && ((uintptr_t)walk_base & 255) == 0,
               "arenas must be 256-byte aligned");
    GS_REQUIRE(fixed_bytes >= arena_bytes[0] && list_bytes >= arena_bytes[1] && walk_bytes >= arena_bytes[2],
               "arena smaller than gs_workspace_query reported");
    GS_REQUIRE(offsets[GS_WS_INFO] >= 0, "layout without an info block");
    hipStream_t st = (hipStream_t)stream;
    // the control words a call expects at zero: the info block (flags are sticky ORs)
    GS_HIP_CHECK(hipMemsetAsync((char*)fixed_base + offsets[GS_WS_INFO], 0, 8 * sizeof(int64_t), st));
    return GS_OK;
}
