"""Persistent per-(device, stream) workspace of the eager `rasterization()` seam (SURVEY.md section 8b "Ownership").

gsplat lets every stage allocate its outputs; a binding that mirrors that makes ~25 allocator calls per forward and
has to wait for the list sizes before it can size half of them.  Here the native library owns the LAYOUT
(`gs_workspace_query`: which intermediates a call shape needs, sizes, alignment) and this module owns the MEMORY: a
pool of leases per (device, stream), each lease three torch byte tensors --

  fixed arena   everything sized by (C, N, image): packed records, footprints, offsets, counters;
  list arena    everything sized by a CAPACITY of intersections: keys, sorted lists, quadrant masks, row bases;
  walk arena    (training) everything sized by what the forward WALKS -- capacities of work units and gradient rows:
                checkpoints, quadrant sublists, work-unit descriptors, rows (round 6: 1-3 % of the listed entries on realistic
                footprints; rounds 1-5 sized them by the list capacity, 355 bytes per listed entry).

A forward takes a free lease (or makes one), the autograd node and the returned `meta` keep it, and it goes back to
the pool when both are gone -- in a training loop that is ONE lease, re-used every step with no allocator traffic.
Two forwards in flight before a backward (several views accumulated into one loss) simply hold two leases.  A call
whose lists outgrow the capacity replaces the list arena only (`Lease.grow_lists`) and repeats the list stages; a training
call whose walk outgrows its capacities replaces the walk arena only (`Lease.grow_walk`) and repeats the blend.
Only tensors that escape to the caller (image, alphas, radii, means2d, depths, conics, gradients) are `torch.empty`.
"""
from __future__ import annotations

import ctypes as ct
import threading
from typing import Dict, List, Optional, Tuple

import torch

from . import _native as nat

# slot indices / flags of include/gs_raster.h
INFO, REC, BBOX, TILES_PER_GAUSS, CUM_TILES, COLORS_POST, ISECT_OFFSETS, BUCKET_OFFSETS, TILE_ORDER, QCNT, SH_JAC = range(11)
LIST_FIRST = 11
BIN, COARSE_KEYS, KEYS_TMP, SLOT_GID, FLATTEN_IDS, SLOTS, ISECT_IDS, QMASK, ROW_BASE, WALK_STATE = range(11, 21)
WALK_FIRST = 21
CKPT, QLIST, UNIT_DESC, ROWS = range(21, 25)
N_SLOTS = 25
WALK_UNITS, WALK_STORAGE, _, WALK_ROWS, WALK_FLAGS = range(5)   # words of the walk state (GS_WALK_*)
FLAG_UNITS, FLAG_ROWS = 16, 32
F_TRAIN, F_TWO_LEVEL, F_ISECT_IDS = 1, 2, 4

_DTYPES = {INFO: torch.int64, REC: torch.float32, BBOX: torch.int32, TILES_PER_GAUSS: torch.int32, CUM_TILES: torch.int32,
           COLORS_POST: torch.float32, ISECT_OFFSETS: torch.int32, BUCKET_OFFSETS: torch.int32, TILE_ORDER: torch.int32,
           QCNT: torch.int32, WALK_STATE: torch.int32, ROW_BASE: torch.int32, SH_JAC: torch.float32, FLATTEN_IDS: torch.int32, SLOTS: torch.int32, ISECT_IDS: torch.int64,
           QMASK: torch.uint8, ROWS: torch.float32, QLIST: torch.int32, UNIT_DESC: torch.int32,
           CKPT: torch.float32}

stats = {"leases_created": 0, "acquires": 0, "fixed_allocs": 0, "list_allocs": 0, "walk_allocs": 0, "list_grows": 0, "walk_grows": 0, "binds": 0}


class Layout:
    """Result of one gs_workspace_query call."""
    __slots__ = ("offsets", "arena_bytes", "key", "_c_offsets", "_c_bytes", "cap_units", "cap_rows")

    def __init__(self, C: int, N: int, W: int, H: int, cap: int, coarse_cap: int, bin_shift: int, flags: int,
                 cap_units: int = 256, cap_rows: int = 0):
        self._c_offsets = (ct.c_int64 * N_SLOTS)()
        self._c_bytes = (ct.c_int64 * 3)()
        nat.check(nat.lib().gs_workspace_query(C, N, W, H, cap, coarse_cap, cap_units, cap_rows, bin_shift, flags, self._c_offsets,
                                               self._c_bytes), "gs_workspace_query")
        self.offsets = list(self._c_offsets)
        self.arena_bytes = (int(self._c_bytes[0]), int(self._c_bytes[1]), int(self._c_bytes[2]))
        self.key = (C, N, W, H, cap, coarse_cap, bin_shift, flags, cap_units, cap_rows)
        self.cap_units, self.cap_rows = int(cap_units), int(cap_rows)


class Lease:
    """Three arenas + the layout they are currently bound to.  `ptr(slot)` is the device address of a buffer (None when
    the layout does not hold it), `view(slot, n)` a typed 1-D tensor view of its first n elements."""

    def __init__(self, pool: "_Pool", device: torch.device):
        self.pool, self.device = pool, device
        self.fixed: Optional[torch.Tensor] = None
        self.lists: Optional[torch.Tensor] = None
        self.walk: Optional[torch.Tensor] = None
        self.layout: Optional[Layout] = None
        self.cap = 0
        self.busy = False
        self.refs = 0
        self._bound = None

    # -- reference counting by the objects that read the arenas after the forward returned (autograd ctx, meta).
    # Under the pool's (re-entrant) lock: meta is dropped on the caller's thread, the autograd ctx on the engine's -- an
    # unsynchronised `refs -= 1` could lose an update and strand the lease.
    def retain(self) -> "Lease":
        with self.pool.lock:
            self.refs += 1
        return self

    def release(self) -> None:
        with self.pool.lock:
            self.refs -= 1
            if self.refs <= 0:
                self.refs = 0
                self.pool.give_back(self)

    def _ensure(self, which: str, nbytes: int) -> None:
        cur = getattr(self, which)
        if cur is None or cur.numel() < nbytes:
            # (grown with head-room: the next, slightly larger frame must not allocate again)
            setattr(self, which, torch.empty((nbytes + (nbytes >> 3) + 4096,), dtype=torch.uint8, device=self.device))
            stats[{"fixed": "fixed_allocs", "lists": "list_allocs", "walk": "walk_allocs"}[which]] += 1

    def bind(self, layout: Layout, stream: int) -> None:
        self._ensure("fixed", layout.arena_bytes[0])
        self._ensure("lists", layout.arena_bytes[1])
        self._ensure("walk", layout.arena_bytes[2])
        bound = (self.fixed.data_ptr(), self.lists.data_ptr(), self.walk.data_ptr(), layout.key)
        self.layout = layout
        self.cap = layout.key[4]
        if bound == self._bound:
            # same arenas, same layout as the last call: nothing to validate, and nothing to zero -- the call runs under
            # gs_guard_set_call (its first flag writer overwrites the flags word) and the blend clears its own counter
            return
        nat.check(nat.lib().gs_workspace_bind(stream, self.fixed.data_ptr(), self.fixed.numel(), self.lists.data_ptr(),
                                              self.lists.numel(), self.walk.data_ptr(), self.walk.numel(), layout._c_offsets,
                                              layout._c_bytes), "gs_workspace_bind")
        self._bound = bound
        stats["binds"] += 1

    def grow_lists(self, layout: Layout, stream: int) -> None:
        """Capacity exceeded: same call shape, larger list arena (the fixed arena -- records, offsets -- is kept as it is)."""
        assert layout.arena_bytes[0] <= self.fixed.numel()
        self._ensure("lists", layout.arena_bytes[1])
        self._ensure("walk", layout.arena_bytes[2])
        self.layout = layout
        self.cap = layout.key[4]
        self._bound = (self.fixed.data_ptr(), self.lists.data_ptr(), self.walk.data_ptr(), layout.key)   # (the caller zeroes the info block itself)
        stats["list_grows"] += 1

    def grow_walk(self, layout: Layout) -> None:
        """The walk of a training forward outgrew cap_units / cap_rows: same lists, larger walk arena (the caller repeats the
        blend into it)."""
        assert layout.arena_bytes[0] <= self.fixed.numel() and layout.arena_bytes[1] <= self.lists.numel()
        self._ensure("walk", layout.arena_bytes[2])
        self.layout = layout
        self._bound = (self.fixed.data_ptr(), self.lists.data_ptr(), self.walk.data_ptr(), layout.key)
        stats["walk_grows"] += 1

    def ptr(self, slot: int) -> Optional[int]:
        off = self.layout.offsets[slot]
        if off < 0:
            return None
        base = self.fixed if slot < LIST_FIRST else (self.lists if slot < WALK_FIRST else self.walk)
        return base.data_ptr() + off

    def view(self, slot: int, n: int) -> torch.Tensor:
        off = self.layout.offsets[slot]
        assert off >= 0, f"workspace slot {slot} is not part of this layout"
        base = self.fixed if slot < LIST_FIRST else (self.lists if slot < WALK_FIRST else self.walk)
        dt = _DTYPES[slot]
        isz = torch.empty((), dtype=dt).element_size()
        return base[off:off + n * isz].view(dt)


class _Pool:
    def __init__(self):
        # re-entrant: LeaseRef.__del__ -> release() -> give_back() can run on the very thread that holds the lock, when a
        # cyclic-GC pass triggered by an allocation inside acquire() / give_back() finalises a meta dict or an autograd ctx
        self.lock = threading.RLock()
        self.free: Dict[Tuple[int, int], List[Lease]] = {}

    def acquire(self, device: torch.device, stream: int) -> Lease:
        key = (device.index if device.index is not None else torch.cuda.current_device(), stream)
        with self.lock:
            stats["acquires"] += 1
            lst = self.free.setdefault(key, [])
            lease = lst.pop() if lst else None
            if lease is None:
                lease = Lease(self, device)
                lease.key = key
                stats["leases_created"] += 1
            lease.busy = True
            lease.refs = 0
            return lease

    def give_back(self, lease: Lease) -> None:
        with self.lock:
            if lease.busy:
                lease.busy = False
                self.free.setdefault(lease.key, []).append(lease)

    def clear(self) -> None:
        """Drops every idle lease (their arenas return to torch's caching allocator)."""
        with self.lock:
            self.free.clear()


pool = _Pool()


class LeaseRef:
    """Holds one reference on a lease; dropping the object (CPython refcount) releases it."""
    __slots__ = ("lease",)

    def __init__(self, lease: Lease):
        self.lease = lease.retain()

    def __del__(self):
        try:
            self.lease.release()
        except Exception:   # interpreter shutdown
            pass
