"""`rasterization()` -- the drop-in for `gsplat.rendering.rasterization` on MI355X.

Mirrors the gsplat 1.0.0 signature that /root/reference/model/gaussian.py:353-367 calls
(`packed=False, absgrad=True, sh_degree=active_sh_degree, backgrounds=[1,3]`) and the two side
channels the reference reads afterwards (`meta["means2d"].absgrad`, `meta["radii"]`,
/root/reference/model/gaussian.py:188-197, 371-372).  All arithmetic runs in the hand-written
gfx950 kernels behind the C ABI of include/gs_raster.h; PyTorch only owns the memory, the
stream and the autograd graph edge.  There is no CPU or eager fallback.
"""
from __future__ import annotations

import ctypes as _ct
import math
import os
import threading
import time
import weakref
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

from . import _native as nat
from . import workspace as WS

_TILE = nat.GS_TILE
# GS_BWD_CLASSES=0: the blend backward takes its work units as the forward published them (A/B switch; the rows are the same bits)
_BWD_CLASSES = os.environ.get("GS_BWD_CLASSES", "1") != "0"


def _ptr(t: Optional[Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _stream(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


# Optional per-stage device timing (bench.py only): when `_prof` is a dict, every native stage call
# is bracketed by HIP events recorded on the stream the kernels are launched on.
_prof: Optional[Dict] = None
_prof_repeat: Dict[str, int] = {}


def _stage(name: str, device, thunk):
    if _prof is None:
        return thunk()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    stream = torch.cuda.current_stream(device)
    n = max(1, int(_prof_repeat.get(name, 1)))
    s.record(stream)
    r = thunk()
    for _ in range(n - 1):   # (idempotent stages only: the same launch again, so that the launch gap is paid once per n)
        thunk()
    e.record(stream)
    _prof.setdefault(name, []).append((s, e, n))
    return r


def profile_stages(enable: bool, repeat: Optional[Dict[str, int]] = None) -> Optional[Dict]:
    """Start (True) or stop (False) stage timing; stopping returns {stage: [ms, ...]}.  `repeat={"gs_blend_bwd": 10}`: that
    stage (it must be idempotent: the blend kernels are) is launched 10 times back to back inside its pair of events and the
    time divided by 10 -- the kernel's duration without the launch gap a single eager launch carries (what a kernel trace
    reports)."""
    global _prof, _prof_repeat
    if enable:
        _prof, _prof_repeat = {}, dict(repeat or {})
        return None
    out, _prof, _prof_repeat = _prof, None, {}
    if out is None:
        return None
    torch.cuda.synchronize()
    return {k: [s.elapsed_time(e) / n for s, e, n in v] for k, v in out.items()}


# GS_SH_JAC=0: the backward stages the SH coefficients itself instead of using the forward's direction Jacobian (A/B, tests)
_SH_JAC = os.environ.get("GS_SH_JAC", "1") != "0"
# GS_FWD_SPLIT=1: projection and SH colour as two launches around the tile count (the form that hid the size read-back before
# the list stages became speculative)
_SPLIT_PROJECT = os.environ.get("GS_FWD_SPLIT") == "1"

_tls = threading.local()
_state_lock = threading.Lock()   # guards the two module-level dicts below (entry points are called from any thread)
_hints: Dict[tuple, dict] = {}   # per (device, C, W, H, training, list mode): capacities the next call of this shape starts from
_coarse_hint: Dict[int, dict] = {}   # per device: pipeline and mean footprint of the last call (last_binning)


def reset_hints() -> None:
    """Forgets the capacities learnt from earlier calls (the next call of every shape starts from the first-call guess
    again) and drops the idle workspaces."""
    with _state_lock:
        _hints.clear()
        _coarse_hint.clear()
        _round_bufs.clear()
    WS.pool.clear()


def bin_shift_for(footprint) -> int:
    """Bin size of the two-level binning from the mean footprint (tile-list entries per Gaussian) of the previous call:
    4x4-tile bins (2) from the footprint at which the two-level binning is chosen at all, 2x2 (1) below (only reached
    when GS_BINNING forces it); 0 = library default (first call).  GS_BINS_SHIFT overrides."""
    env = os.environ.get("GS_BINS_SHIFT")
    if env:
        return int(env)
    if footprint is None:
        return 0
    return 1 if footprint < BINS_FROM_FOOTPRINT else 2   # (tools/binning_sweep.py: 4x4 wins wherever the two-level binning does)


def binning_mode() -> str:
    """GS_BINNING: "auto" (default), "bins" (two-level binning: coarse-bin lists sorted, tiles refined out of them in order)
    or "tiles" (every tile list emitted and sorted on its own).  Both yield bit-identical lists."""
    mode = os.environ.get("GS_BINNING", "auto")
    if mode not in ("auto", "bins", "tiles"):
        raise ValueError(f"GS_BINNING must be 'auto', 'bins' or 'tiles', got {mode!r}")
    return mode


# mean tile-list entries per Gaussian from which sorting coarse bins beats sorting tiles (tools/binning_sweep.py on MI355X,
# 1080p, per-tile vs two-level: 0.22 vs 0.23 ms at 4.8, 0.27 vs 0.23 at 9.5, 0.42 vs 0.23 at 21, 1.10 vs 0.30 at 106;
# round 3, HISTORY.md: 2 M Gaussians at 5.4 0.54 vs 0.47, 5 M at 4K and 7.6 1.86 vs 0.89)
BINS_FROM_FOOTPRINT = 5.0


def last_binning(device=None) -> Optional[str]:
    """"tiles" / "bins": the pipeline the last eager rasterization on `device` went through (None before the first)."""
    key = torch.cuda.current_device() if device is None or torch.device(device).index is None else torch.device(device).index
    with _state_lock:
        return _coarse_hint.get(key, {}).get("mode")


MAX_TILES_PER_TILE_PIPELINE = 40928   # the per-tile pipeline keeps a histogram over every tile of a camera in 160 KB of LDS


def binning_choice(footprint, tiles: int = 0) -> str:
    """The pipeline for a scene whose Gaussians cover `footprint` tiles on average (None: unknown -> per-tile).  Images of
    more than 40928 tiles per camera (beyond ~4K x 2.5K) always take the two-level binning, whose LDS histogram is over bins."""
    mode = binning_mode()
    if tiles > MAX_TILES_PER_TILE_PIPELINE and mode != "tiles":
        return "bins"
    if mode != "auto":
        return mode
    return "bins" if footprint is not None and footprint >= BINS_FROM_FOOTPRINT else "tiles"


# host-side diagnostics: time spent blocked on the list-size read-back (bench.py reports it; a wait
# near zero means the host, not the GPU, paces the loop)
stats = {"sync_wait_ns": 0, "calls": 0, "coarse_retries": 0, "overflow_reruns": 0, "walk_reruns": 0, "deferred_calls": 0, "late_overflows": 0,
         "round_calls": 0}


# Depth rounds in the eager seam (include/gs_raster.h "Depth rounds"; the captured train step has its own switch,
# TrainStepGraph(rounds=...)): INFERENCE calls with one camera whose lists are long enough for the two-level binning run the list
# stages and the blend in two rounds -- the front slab by depth, then the rest only into tiles it has not finished.  Same image bit
# for bit; the lists nobody reads are never built.  GS_ROUNDS: "auto" (default: on for a call shape whose first, one-round call
# listed >= ROUNDS_MIN_LISTED entries through the two-level binning), "on", "off".  GS_ROUND_FRACTION: share of the listed
# intersections in front of the depth split.
ROUNDS_MIN_LISTED = 4_000_000
ROUNDS_MAX_LIVE = 0.05   # "auto": share of the tiles the front round may leave live, three calls in a row, before rounds are given up
ROUND_FRACTION = float(os.environ.get("GS_ROUND_FRACTION", "0.125"))
_round_bufs: Dict[tuple, dict] = {}   # per (device, stream, N, tiles): what lives between the rounds of a call


def rounds_mode() -> str:
    mode = os.environ.get("GS_ROUNDS", "auto")
    if mode not in ("auto", "on", "off"):
        raise ValueError(f"GS_ROUNDS must be 'auto', 'on' or 'off', got {mode!r}")
    return mode


def _round_buffers(dev: torch.device, st: int, N: int, tiles: int) -> dict:
    """The round block, the depth histogram (zero between calls), the tiles' liveness and pixel states, the footprints and
    counts of the round at hand: one set per (device, stream) -- calls on one stream follow one another --, re-made when the
    call shape changes."""
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), st)
    with _state_lock:
        b = _round_bufs.get(key)
    if b is None or b["N"] != N or b["tiles"] != tiles:
        b = {"N": N, "tiles": tiles,
             "blk": torch.zeros((nat.GS_ROUND_WORDS,), dtype=torch.int64, device=dev),
             "hist": torch.zeros((4096,), dtype=torch.int32, device=dev),
             "live": torch.zeros((tiles,), dtype=torch.uint8, device=dev),
             "state": torch.empty((tiles, 4, 64, 4), dtype=torch.float32, device=dev),
             "bbox": torch.empty((N, 4), dtype=torch.int32, device=dev),
             "tpg": torch.empty((N,), dtype=torch.int32, device=dev),
             "host": torch.zeros((nat.GS_ROUND_WORDS,), dtype=torch.int64, pin_memory=True)}
        with _state_lock:
            _round_bufs[key] = b
    return b


def _quantize_up(x: int) -> int:
    g = 1 << max(12, int(x).bit_length() - 5)
    return (int(x) + g - 1) // g * g


_INFO_RING = 8   # size records in flight per (host thread, device) with the deferred size check


def _pinned_info(device: torch.device) -> Tensor:
    """Page-locked 8 x int64 landing buffer for the list sizes: the next of a ring of `_INFO_RING` per (host thread, device).
    (With the immediate size check a call has read its record before it returns; with the deferred one -- `_size_check=
    "deferred"` -- up to `_INFO_RING - 1` later calls may be issued before it is looked at, `_pending_checks`.)"""
    cache = getattr(_tls, "pinned", None)
    if cache is None:
        cache = _tls.pinned = {}
    key = device.index if device.index is not None else torch.cuda.current_device()
    ring = cache.get(key)
    if ring is None:
        ring = cache[key] = [torch.empty((_INFO_RING, 8), dtype=torch.int64, pin_memory=True), 0]
    ring[1] = (ring[1] + 1) % _INFO_RING
    return ring[0][ring[1]]


class _WalkRecord:
    """Page-locked int64[4] landing buffer {work units, storage units, gradient rows, flags} of ONE training forward: the row-base
    scan behind its blend writes it (gs_walk_mirror_set), the call's backward reads it -- as plain memory, behind an event that
    has long passed -- before it launches anything that walks the units.  Buffers are recycled per host thread (a pinned
    allocation per call would cost more than the forward)."""
    __slots__ = ("host", "event", "_pool")

    def __init__(self):
        pool = getattr(_tls, "walk_pool", None)
        if pool is None:
            pool = _tls.walk_pool = []
        self._pool = pool
        self.host = pool.pop() if pool else torch.zeros((4,), dtype=torch.int64, pin_memory=True)
        self.host.fill_(-1)   # (a record the device never wrote -- every attempt of the call skipped by the guard -- is told apart)
        self.event = None

    def __del__(self):
        try:
            self._pool.append(self.host)
        except Exception:   # interpreter shutdown
            pass


def _pending_checks() -> list:
    q = getattr(_tls, "pending_checks", None)
    if q is None:
        q = _tls.pending_checks = []
    return q


def flush_size_checks() -> int:
    """Resolves every deferred size check of this host thread (`_size_check="deferred"`): waits for the size records, repairs
    -- in place, into the very tensors that were handed out -- any forward whose list capacities did not hold.  Returns the
    number of repaired calls.  Called implicitly by the next `rasterization()` on the thread, by `backward()` of the call
    itself and by the first read of one of `meta`'s list arrays."""
    q = _pending_checks()
    late = 0
    while q:
        late += int(q.pop(0).resolve())
    return late


class _PendingCheck:
    """A forward whose size record {I, buckets, longest list, flags} has not been looked at yet."""

    def __init__(self, resolve_fn):
        self._fn, self.done, self.late = resolve_fn, False, False
        self._lock = threading.Lock()   # (backward() runs on the autograd engine's thread, the next forward on the caller's)

    def resolve(self) -> bool:
        with self._lock:
            if not self.done:
                self.late = bool(self._fn())
                self.done = True
                self._fn = None
                self.lease_ref = None
        return self.late


class _LazyMeta(dict):
    """gsplat's meta dict with `isect_ids` built on first access.  The sorted (camera | tile | depth bits) keys are a function
    of the other entries -- camera | tile from `isect_offsets`, depth bits from `depths[flatten_ids]` -- and writing them was
    half of the list bytes of every forward although nothing downstream of the reference reads them
    (/root/reference/model/gaussian.py:368-375 takes `means2d` and `radii`)."""
    PENDING = object()

    class Lazy:
        """A value produced on first access (list arrays copied out of the leased workspace)."""
        __slots__ = ("fn",)

        def __init__(self, fn):
            self.fn = fn

    _lease = None   # workspace.LeaseRef: the arenas the lazies read stay leased for as long as this dict lives

    def _isect_ids(self) -> Tensor:
        C, tiles = self["n_cameras"], self["tile_width"] * self["tile_height"]
        fid = self["flatten_ids"]
        off = self["isect_offsets"].reshape(-1).long()
        counts = torch.diff(off, append=off.new_tensor([fid.numel()]))
        tile = torch.repeat_interleave(torch.arange(C * tiles, device=fid.device), counts, output_size=fid.numel())
        tile_bits = int(tiles).bit_length()
        dbits = dict.__getitem__(self, "depths").reshape(-1).view(torch.int32)[fid.long()].long() & 0xFFFFFFFF
        return ((tile // tiles) << (32 + tile_bits)) | ((tile % tiles) << 32) | dbits

    def __getitem__(self, key):
        v = dict.__getitem__(self, key)
        if v is _LazyMeta.PENDING:
            v = self._isect_ids()
            dict.__setitem__(self, key, v)
        elif isinstance(v, _LazyMeta.Lazy):
            v = v.fn()
            dict.__setitem__(self, key, v)
        return v

    def get(self, key, default=None):
        return self[key] if key in self else default

    def __iter__(self):   # (defined so that dict(meta) / {**meta} go through keys() + __getitem__, not the raw storage)
        return dict.__iter__(self)

    def items(self):
        return [(k, self[k]) for k in dict.keys(self)]

    def values(self):
        return [self[k] for k in dict.keys(self)]

    def pop(self, key, *default):
        if key in self:
            self[key]
        return dict.pop(self, key, *default)

    def copy(self):
        return dict(self.items())


class _Holder:
    """Carries non-tensor state between `rasterization()` and the autograd node without making
    the node own its own output (the weak reference lets backward attach `.absgrad` to the very
    tensor object that was handed out in `meta`)."""

    __slots__ = ("meta", "means2d_ref", "absgrad", "debug", "on_colors_pre", "grad_out", "view_payload", "lease_ref", "out_ptrs")

    def __init__(self, absgrad: bool):
        self.meta: Dict = {}
        self.means2d_ref = None
        self.absgrad = absgrad
        self.debug: Optional[Dict] = None
        self.on_colors_pre = None
        self.grad_out = None
        self.view_payload = None
        self.lease_ref = None
        self.out_ptrs = ()   # data pointers of the node's OUTPUTS (render_colors, render_alphas): see _lease_pack_hook


class _ReferenceLists:
    """gsplat's list arrays for a call that rendered from the short lists (`_tile_culling="gsplat"`, the default): a function
    of the 3-sigma tile rectangles the projection kept (`rect_ref`) and of `depths`, built -- all four at once, by the same
    count / emit / sort kernels, in fresh buffers -- when one of them is first read.  The reference reads none of them
    (/root/reference/model/gaussian.py:368-375 takes the image, `means2d` and `radii`), so its training loop never pays for
    the ~40 % longer lists; a caller that does read them gets exactly what `_tile_culling="gsplat_eager"` walks."""

    def __init__(self, rect_ref: Tensor, depths: Tensor, C: int, N: int, tw: int, th: int, eager_ids: bool):
        self.rect_ref, self.depths, self.C, self.N, self.tw, self.th, self.eager_ids = rect_ref, depths, C, N, tw, th, eager_ids
        self.out: Optional[Dict[str, Tensor]] = None
        self.lock = threading.Lock()
        # the lists are built on whatever stream is current when they are first READ: ordered behind the projection that
        # produced `rect_ref` / `depths` by this event (a reader on another stream raced the producer)
        self.produced = torch.cuda.Event()
        self.produced.record(torch.cuda.current_stream(depths.device))

    def get(self, key: str) -> Tensor:
        with self.lock:
            if self.out is None:
                self.out = self._build()
        return self.out[key]

    def _build(self) -> Dict[str, Tensor]:
        L = nat.lib()
        C, N, tw, th = self.C, self.N, self.tw, self.th
        dev = self.depths.device
        tiles = tw * th
        i32 = dict(dtype=torch.int32, device=dev)
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("meta's list arrays of a `_tile_culling='gsplat'` call are built on first access with host read-backs: "
                               "read them before the capture starts, or render with _tile_culling='gsplat_eager' / 'tight'")
        with torch.cuda.device(dev), torch.no_grad():
            torch.cuda.current_stream(dev).wait_event(self.produced)
            st = _stream(dev)
            # footprint records of the binning kernels: rectangle, tile bit mask (all tiles of the rectangle), tile count
            r = self.rect_ref.long()
            x0, x1, y0, y1 = r[:, 0] & 0xFFFF, (r[:, 0] >> 16) & 0xFFFF, r[:, 1] & 0xFFFF, (r[:, 1] >> 16) & 0xFFFF
            cnt = (x1 - x0) * (y1 - y0)
            mask = torch.where(cnt >= 32, torch.full_like(cnt, 0xFFFFFFFF), (torch.ones_like(cnt) << cnt.clamp(max=31)) - 1)
            to_i32 = lambda v: torch.where(v >= (1 << 31), v - (1 << 32), v).to(torch.int32)   # (uint32 bit patterns)
            bbox = torch.stack([to_i32(r[:, 0] & 0xFFFFFFFF), to_i32(r[:, 1] & 0xFFFFFFFF), to_i32(mask), cnt.to(torch.int32)], dim=1).contiguous()
            tiles_per_gauss = cnt.to(torch.int32).view(C, N)
            info_dev = torch.zeros((8,), dtype=torch.int64, device=dev)
            isect_offsets = torch.empty((C * tiles + 1,), **i32)
            bucket_offsets = torch.empty((C * tiles + 1,), **i32)
            cum_tiles = torch.empty((C * N,), **i32)
            footprint = float(cnt.sum().item()) / max(1, C * N)
            two_level = binning_choice(footprint, tiles) == "bins"
            if two_level:
                shift = bin_shift_for(footprint) or 2
                coarse_cap, info = 2 * C * N + 1024, None
                while True:
                    keys = torch.empty((coarse_cap,), dtype=torch.int64, device=dev)
                    ws = torch.empty((int(L.gs_bins_workspace_bytes(C, N, tw, th, shift, coarse_cap)),), dtype=torch.uint8, device=dev)
                    nat.check(L.gs_bins_count(st, C, N, tw, th, shift, _ptr(bbox), _ptr(self.depths), _ptr(ws), ws.numel(), _ptr(keys), coarse_cap,
                                              0, _ptr(cum_tiles), _ptr(isect_offsets), _ptr(bucket_offsets), None, _ptr(info_dev), None),
                              "gs_bins_count")
                    info = info_dev.tolist()
                    if not int(info[3]) & 12:
                        break
                    coarse_cap = int(info[4]) + (int(info[4]) >> 2) + 1024
                    info_dev.zero_()
                n_isects = int(info[0])
                flatten_ids = torch.empty((max(n_isects, 1),), **i32)
                isect_ids = torch.empty((max(n_isects, 1),), dtype=torch.int64, device=dev) if self.eager_ids else None
                nat.check(L.gs_bins_lists(st, C, N, tw, th, shift, _ptr(bbox), _ptr(ws), ws.numel(), _ptr(keys), coarse_cap, _ptr(cum_tiles),
                                          _ptr(isect_offsets), _ptr(isect_ids), _ptr(flatten_ids), None, _ptr(info_dev)), "gs_bins_lists")
            else:
                ws = torch.empty((int(L.gs_bin_workspace_bytes(C, N, tw, th)),), dtype=torch.uint8, device=dev)
                nat.check(L.gs_bin_count(st, C, N, tw, th, _ptr(bbox), _ptr(ws), ws.numel(), _ptr(isect_offsets), _ptr(bucket_offsets), None,
                                         _ptr(info_dev), None), "gs_bin_count")
                info = info_dev.tolist()
                n_isects, max_tile = int(info[0]), int(info[2])
                flatten_ids = torch.empty((max(n_isects, 1),), **i32)
                isect_ids = torch.empty((max(n_isects, 1),), dtype=torch.int64, device=dev) if self.eager_ids else None
                keys_tmp = torch.empty((max(n_isects, 1),), dtype=torch.int64, device=dev)
                nat.check(L.gs_bin_emit_sort(st, C, N, tw, th, _ptr(bbox), _ptr(self.depths), _ptr(ws), ws.numel(), _ptr(isect_offsets), n_isects,
                                             max_tile, _ptr(keys_tmp), None, _ptr(cum_tiles), _ptr(isect_ids), _ptr(flatten_ids), None),
                          "gs_bin_emit_sort")
        out = {"tiles_per_gauss": tiles_per_gauss, "flatten_ids": flatten_ids[:n_isects],
               "isect_offsets": isect_offsets[: C * tiles].view(C, th, tw)}
        if isect_ids is not None:
            out["isect_ids"] = isect_ids[:n_isects]
        return out


class _OneRoundLists:
    """The list arrays of a call that rendered in two depth rounds, built when somebody reads them: the per-tile pipeline, in one
    round, on the footprints and depths the projection left (blocking; the arrays a one-round call of the same mode hands out)."""

    def __init__(self, lease, depths: Tensor, C: int, N: int, tw: int, th: int):
        self.lease, self.depths, self.shape, self.out = lease, depths, (C, N, tw, th), None

    def get(self, key: str) -> Tensor:
        if self.out is None:
            self.out = self._build()
        return self.out[key]

    def _build(self) -> Dict[str, Tensor]:
        C, N, tw, th = self.shape
        tiles = tw * th
        if tiles > MAX_TILES_PER_TILE_PIPELINE:
            raise NotImplementedError("list arrays of a two-round call at this image size: render with GS_ROUNDS=off to read them")
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("meta's list arrays of a two-round call are built on first access with a host read-back: read them "
                               "before the capture starts, or render with GS_ROUNDS=off")
        L, dev = nat.lib(), self.depths.device
        with torch.cuda.device(dev):
            st = _stream(dev)
            i32 = dict(dtype=torch.int32, device=dev)
            ws_bytes = int(L.gs_bin_workspace_bytes(C, N, tw, th))
            ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
            offs, bko, info = torch.empty((C * tiles + 1,), **i32), torch.empty((C * tiles + 1,), **i32), torch.zeros((8,), dtype=torch.int64, device=dev)
            host = torch.zeros((4,), dtype=torch.int64)
            bbox = self.lease.ptr(WS.BBOX)
            nat.check(L.gs_bin_count(st, C, N, tw, th, bbox, ws.data_ptr(), ws_bytes, offs.data_ptr(), bko.data_ptr(), None, info.data_ptr(),
                                     host.data_ptr()), "gs_bin_count")
            n_isects, max_tile = int(host[0]), int(host[2])
            keys = torch.empty((max(n_isects, 1),), dtype=torch.int64, device=dev)
            flat, cum = torch.empty((max(n_isects, 1),), **i32), torch.empty((C * N,), **i32)
            nat.check(L.gs_bin_emit_sort(st, C, N, tw, th, bbox, self.depths.data_ptr(), ws.data_ptr(), ws_bytes, offs.data_ptr(), n_isects,
                                         max(max_tile, 1), keys.data_ptr(), None, cum.data_ptr(), None, flat.data_ptr(), None), "gs_bin_emit_sort")
        return {"flatten_ids": flat[:n_isects], "isect_offsets": offs[: C * tiles].view(C, th, tw)}


def _sort_class(n: int) -> int:
    """Capacity of the smallest per-tile sort class that takes a list of n entries (gs_binning.hip: launch_list_sorts)."""
    for c in (1024, 4096, 8192, 16384):
        if n <= c:
            return c
    return 1 << 30


def _forward_stages(means, quats, scales, opacities, colors, colors_rest, viewmats, Ks, backgrounds, cfg, need_grad):
    """Runs P/SH-fwd, binning, per-tile sort and B-fwd.  Returns (outputs, meta, saved-state dict).

    Sync-free up to its last line (SURVEY.md 8b "Sync"): every intermediate lives in a leased workspace sized by a
    CAPACITY learnt from earlier calls, and the list stages + the blend are enqueued under the library's step guard
    BEFORE the list sizes {I, n_buckets, longest list} have reached the host; the host then waits on the event behind
    the count kernels while the GPU is busy with everything queued after it -- the stream never drains because of the
    read-back.  Lists that outgrew the capacity made the guarded kernels device-side no-ops: the list arena is replaced
    and count .. blend are repeated with the sizes the count reported (rare: the capacity follows the largest frame)."""
    L = nat.lib()
    dev = means.device
    C, N = viewmats.shape[0], means.shape[0]
    W, H = cfg["width"], cfg["height"]
    sh_degree = cfg["sh_degree"]
    tw, th = math.ceil(W / _TILE), math.ceil(H / _TILE)
    tiles = tw * th
    st = _stream(dev)
    f32 = dict(dtype=torch.float32, device=dev)
    i32 = dict(dtype=torch.int32, device=dev)

    if sh_degree is None:
        deg, K = -1, 0
        per_cam = 1 if colors.dim() == 3 else 0
    else:
        deg, per_cam = int(sh_degree), 0
        K = colors.shape[1] + (0 if colors_rest is None else colors_rest.shape[1])

    # what escapes to the caller through `meta` is allocated per call; everything else is workspace
    radii = torch.empty((C, N), **i32)
    means2d = torch.empty((C, N, 2), **f32)
    depths = torch.empty((C, N), **f32)
    conics = torch.empty((C, N, 3), **f32)
    render_colors = torch.empty((C, H, W, 3), **f32)
    render_alphas = torch.empty((C, H, W, 1), **f32)

    # "gsplat" lists, lazily: render from the short ("tight") lists -- image, alphas, radii, means2d bitwise the same, gradients
    # to rounding -- and keep gsplat's 3-sigma rectangles, from which its exact list arrays are built if somebody reads them
    lazy_ref = bool(cfg.get("lazy_ref_lists"))
    rect_ref = torch.empty((C * N, 2), **i32) if lazy_ref else None
    dev_index = dev.index if dev.index is not None else torch.cuda.current_device()
    hint_key = (dev_index, C, W, H, bool(need_grad), cfg["tile_culling"])
    with _state_lock:
        hint = dict(_hints.get(hint_key, {}))
    two_level = binning_choice(hint.get("footprint"), tiles) == "bins"
    shift = bin_shift_for(hint.get("footprint")) if two_level else 0
    # depth rounds: inference, one camera, long lists (see ROUNDS_MIN_LISTED above); the decision of a call shape is taken from
    # what its one-round calls listed and kept (a two-round call reports the short total of its two rounds)
    rmode = cfg.get("rounds") or rounds_mode()
    rounds = (not need_grad) and C == 1 and N > 0 and rmode != "off" and (rmode == "on" or (
        two_level and int(hint.get("n", -1)) == N and int(hint.get("listed_one_round", 0)) >= ROUNDS_MIN_LISTED and not hint.get("rounds_off")))
    rb = _round_buffers(dev, st, N, tiles) if rounds else None
    eager_ids = os.environ.get("GS_EAGER_ISECT_IDS") == "1"   # (default: meta builds isect_ids on first access, _LazyMeta)
    factorised = cfg.get("sh_grads") == "colors_pre"
    flags = (WS.F_TRAIN if need_grad else 0) | (WS.F_TWO_LEVEL if two_level else 0) | (WS.F_ISECT_IDS if eager_ids else 0)
    # capacities: what earlier calls of this shape needed (+25 %), or a first guess
    cap = max(int(hint.get("cap", 0)), 2 * C * N + 4096 if not hint else 0, 4096)
    cap_tile = int(hint.get("cap_tile", 1024)) if not two_level else (1 << 30)
    coarse_cap = max(int(hint.get("entries", 0)), 2 * C * N + 1024) if two_level else 0
    coarse_list_cap = int(hint.get("longest", 0)) if two_level else 0   # 0: launch every sort class
    # (capacities in steps of ~3 %: the hint drifts a little with every frame, the LAYOUT should not -- a lease whose arenas
    #  and layout are those of the last call is not re-bound: no validation, no memset on the stream)
    cap, coarse_cap = _quantize_up(cap), (_quantize_up(coarse_cap) if two_level else 0)
    # training: what the forward WALKS -- work units (checkpoints, quadrant sublists) and gradient rows -- has capacities of its
    # own, learnt from the walk records of earlier calls (first call: a guess; a walk that outgrows them is repeated by the
    # call's backward, `_settle_walk`)
    cap_units = cap_rows = 0
    if need_grad:
        cap_units = _quantize_up(max(int(hint.get("cap_units", 0)), (cap // 16 + 12 * C * tiles) if "cap_units" not in hint else 0, 256))
        cap_rows = _quantize_up(max(int(hint.get("cap_rows", 0)), cap if "cap_rows" not in hint else 0, 4096))
    walk = _WalkRecord() if need_grad else None

    lease = WS.pool.acquire(dev, st)
    lease.bind(WS.Layout(C, N, W, H, cap, coarse_cap, shift, flags, max(cap_units, 256), cap_rows), st)
    P = lease.ptr
    info_dev = lease.view(WS.INFO, 8)
    info_host = _pinned_info(dev)
    bin_bytes = lambda: int(L.gs_bins_workspace_bytes(C, N, tw, th, shift, coarse_cap) if two_level else L.gs_bin_workspace_bytes(C, N, tw, th))

    # training with SH colours: the forward leaves d colour / d view direction per visible Gaussian, and the backward never
    # reads the coefficients (include/gs_raster.h, gs_project_fwd: sh_jac)
    use_jac = bool(need_grad and _SH_JAC and deg >= 1)

    def project(stage: int, tag: str):
        _stage(tag, dev, lambda: nat.check(L.gs_project_fwd(
            st, C, N, K, deg, _ptr(means), _ptr(quats), _ptr(scales), _ptr(opacities), _ptr(colors), _ptr(colors_rest),
            per_cam, _ptr(viewmats), _ptr(Ks), W, H, cfg["eps2d"], cfg["near_plane"], cfg["far_plane"],
            cfg["radius_clip"], cfg["tile_culling"], stage, cfg.get("activations", 0), _ptr(radii), _ptr(means2d), _ptr(depths), _ptr(conics),
            P(WS.COLORS_POST), P(WS.REC), P(WS.BBOX), P(WS.TILES_PER_GAUSS), _ptr(rect_ref),
            P(WS.SH_JAC) if use_jac else None), "gs_project_fwd"))

    def bbox_ptr():   # the footprints the list stages read: the projection's, or (depth rounds) those of the round at hand
        return rb["bbox"].data_ptr() if rounds else P(WS.BBOX)

    def count():
        if two_level:
            nat.check(L.gs_bins_count(st, C, N, tw, th, shift, bbox_ptr(), _ptr(depths), P(WS.BIN), bin_bytes(),
                                      P(WS.COARSE_KEYS), coarse_cap, coarse_list_cap, P(WS.CUM_TILES), P(WS.ISECT_OFFSETS),
                                      P(WS.BUCKET_OFFSETS), P(WS.TILE_ORDER), P(WS.INFO), None), "gs_bins_count")
        else:
            nat.check(L.gs_bin_count(st, C, N, tw, th, bbox_ptr(), P(WS.BIN), bin_bytes(), P(WS.ISECT_OFFSETS),
                                     P(WS.BUCKET_OFFSETS), P(WS.TILE_ORDER), P(WS.INFO), None), "gs_bin_count")

    def lists_and_blend():
        if two_level:
            _stage("gs_bin_emit_sort", dev, lambda: nat.check(L.gs_bins_lists(
                st, C, N, tw, th, shift, bbox_ptr(), P(WS.BIN), bin_bytes(), P(WS.COARSE_KEYS), coarse_cap,
                P(WS.CUM_TILES), P(WS.ISECT_OFFSETS), P(WS.ISECT_IDS), P(WS.FLATTEN_IDS), P(WS.SLOTS), P(WS.INFO)), "gs_bins_lists"))
        else:
            _stage("gs_bin_emit_sort", dev, lambda: nat.check(L.gs_bin_emit_sort(
                st, C, N, tw, th, bbox_ptr(), _ptr(depths), P(WS.BIN), bin_bytes(), P(WS.ISECT_OFFSETS), cap,
                min(cap_tile, cap), P(WS.KEYS_TMP), P(WS.SLOT_GID), P(WS.CUM_TILES), P(WS.ISECT_IDS), P(WS.FLATTEN_IDS),
                P(WS.SLOTS)), "gs_bin_emit_sort"))
        if walk is not None:
            nat.check(L.gs_walk_mirror_set(walk.host.data_ptr()), "gs_walk_mirror_set")
        try:
            _stage("gs_blend_fwd", dev, lambda: nat.check(L.gs_blend_fwd(
                st, C, W, H, P(WS.REC), _ptr(backgrounds), P(WS.ISECT_OFFSETS), P(WS.TILE_ORDER),
                P(WS.FLATTEN_IDS), P(WS.SLOTS), cap, _ptr(render_colors), _ptr(render_alphas), P(WS.CKPT), P(WS.QLIST), P(WS.QCNT),
                P(WS.QMASK), P(WS.UNIT_DESC), lease.layout.cap_units, P(WS.ROW_BASE), lease.layout.cap_rows, P(WS.WALK_STATE)),
                "gs_blend_fwd"))
        finally:
            if walk is not None:
                L.gs_walk_mirror_set(None)
                walk.event = torch.cuda.Event()
                walk.event.record(tstream)

    # 1. geometry; 2. tile counts under the guard (flags = capacity exceeded) and their 64-byte copy to the host;
    # 3. SH colours; 4. lists + blend, speculatively; 5. only now the host looks at the sizes.
    # (one launch for geometry + colour: since the list stages no longer wait for the host, a colour pass of its own behind
    #  the tile count hides nothing, and the fused launch is 25 us shorter than the two -- GS_FWD_SPLIT=1 keeps the split)
    project(1 if _SPLIT_PROJECT else 0, "gs_project_fwd")
    sizes = {"attempt": 0, "waited": 0, "info": None}   # (mutable: the deferred check finishes this call after it returned)
    tstream = torch.cuda.current_stream(dev)            # (the stream `st` belongs to: a late repair runs on it, whoever calls)

    def enqueue_attempt():
        """Tile counts under the guard (flags = capacity exceeded), their 64-byte record straight into page-locked memory,
        then lists + blend, speculatively."""
        nat.check(L.gs_guard_set_call(P(WS.INFO), cap, min(cap_tile, cap)), "gs_guard_set_call")
        # (the tile scan writes the eight info words straight into the page-locked landing buffer: no copy on the stream)
        nat.check(L.gs_info_mirror_set(info_host.data_ptr()), "gs_info_mirror_set")
        try:
            if rounds:
                # depth split, then count .. blend per round on the round's footprints.  Immediate size check: the host waits for the
                # front round anyway (its size record), so it also looks at how many tiles that round left live -- and does not
                # enqueue the back round at all when there are none (18 launches that would return at once: 0.1 ms per frame).
                # Deferred check: both rounds are enqueued, the size record is the back round's (the total of both; a back round
                # with no live tile leaves the front round's record standing).
                if sizes["attempt"] == 0 and _SPLIT_PROJECT:
                    project(2, "gs_project_fwd_color")
                nat.check(L.gs_round_split(st, N, _ptr(depths), P(WS.TILES_PER_GAUSS), ROUND_FRACTION, rb["hist"].data_ptr(), rb["blk"].data_ptr()),
                          "gs_round_split")
                ev = None
                for phase in (1, 2):
                    nat.check(L.gs_rounds_set(rb["blk"].data_ptr(), rb["live"].data_ptr(), rb["state"].data_ptr(), None, phase), "gs_rounds_set")
                    nat.check(L.gs_info_mirror_set(info_host.data_ptr()), "gs_info_mirror_set")
                    _stage("gs_round_footprints", dev, lambda: nat.check(L.gs_round_footprints(
                        st, N, tw, th, P(WS.BBOX), _ptr(depths), rb["bbox"].data_ptr(), rb["tpg"].data_ptr()), "gs_round_footprints"))
                    _stage("gs_bin_count", dev, count)
                    nat.check(L.gs_info_mirror_set(None), "gs_info_mirror_set")
                    if phase == 2:
                        ev = torch.cuda.Event()
                        ev.record(tstream)
                    lists_and_blend()
                    if phase == 1 and not deferred:
                        nat.check(L.gs_round_status(st, rb["blk"].data_ptr(), rb["host"].data_ptr()), "gs_round_status")
                        ev = torch.cuda.Event()
                        ev.record(tstream)
                        t_wait = time.perf_counter_ns()
                        ev.synchronize()
                        sizes["waited"] += time.perf_counter_ns() - t_wait
                        if int(info_host[3]) != 0 or int(rb["host"][nat.GS_ROUND_LIVE]) == 0:
                            break   # (the front round did not fit: `settle` re-sizes and the attempt is repeated; or it finished the frame)
                return ev
            _stage("gs_bin_count", dev, count)
            nat.check(L.gs_info_mirror_set(None), "gs_info_mirror_set")
            ev = torch.cuda.Event()
            ev.record(tstream)
            if sizes["attempt"] == 0 and _SPLIT_PROJECT:
                project(2, "gs_project_fwd_color")
            lists_and_blend()
        finally:
            L.gs_guard_set(None, 0, 0)
            L.gs_info_mirror_set(None)
            L.gs_rounds_set(None, None, None, None, 0)
        return ev

    def settle(ev, whole_stream: bool) -> bool:
        """Waits for the size record; True when the capacities held.  Otherwise nothing was emitted or blended: re-sizes from
        what the count reported (the caller repeats the attempt)."""
        nonlocal cap, cap_tile, coarse_cap, coarse_list_cap, info_dev
        t_wait = time.perf_counter_ns()
        if whole_stream:
            tstream.synchronize()
        else:
            ev.synchronize()
        sizes["waited"] += time.perf_counter_ns() - t_wait
        info = sizes["info"] = [int(v) for v in info_host.tolist()]
        n_isects, max_tile, fl = info[0], info[2], info[3]
        if fl == 0:
            return True
        sizes["attempt"] += 1
        if sizes["attempt"] > 6:
            raise nat.NativeLibraryError(f"rasterization: list capacities did not settle (info {info})")
        if two_level and fl & 12:
            # the coarse stage did not fit: the tile counts (I, longest list) were never formed -- only its own
            # sizes {I', longest bin list} are meaningful; the tile-list capacity is checked by the repeat
            if fl & 4:
                coarse_cap = info[4] + (info[4] >> 2) + 1024
            coarse_list_cap = 0
            with _state_lock:
                stats["coarse_retries"] += 1
        else:
            if fl & 1:
                cap = n_isects + (n_isects >> 3) + 1024
            if fl & 2:
                cap_tile = _sort_class(max_tile)
        with _state_lock:
            stats["overflow_reruns"] += 1
        lease.grow_lists(WS.Layout(C, N, W, H, cap, coarse_cap, shift, flags, max(cap_units, 256), cap_rows), st)
        info_dev = lease.view(WS.INFO, 8)
        info_dev.zero_()
        return False

    def learn():
        """The capacity hints follow what this call needed."""
        info = sizes["info"]
        n_isects, max_tile = info[0], info[2]
        with _state_lock:
            stats["sync_wait_ns"] += sizes["waited"]
            stats["calls"] += 1
            stats["round_calls"] += 1 if rounds else 0
            old = _hints.get(hint_key, {})
            # the capacity follows the largest recent frame (slow decay), so alternating views do not overflow every time
            new = dict(cap=max(n_isects + (n_isects >> 2) + 1024, int(old.get("cap", 0) * 0.995)),
                       cap_tile=max(_sort_class(max_tile + (max_tile >> 2)), int(old.get("cap_tile", 1024))), footprint=n_isects / max(1, C * N),
                       mode="bins" if two_level else "tiles", n=N, listed_one_round=n_isects)
            if rounds:   # (what one round would have listed, info[7]: the pipeline choice and the rounds decision follow the frame, not the rounds)
                new["footprint"], new["listed_one_round"] = info[7] / max(1, C * N), info[7]
                # ... and "auto" gives rounds up for a call shape whose front slab keeps leaving tiles to the back round (info[6]: a
                # back round that has work pays the fixed costs of the list stages twice -- train_graph.TrainStepGraph.ROUNDS_MAX_LIVE)
                strikes = int(old.get("live_strikes", 0)) + 1 if info[6] > ROUNDS_MAX_LIVE * C * tiles else 0
                new["live_strikes"], new["rounds_off"] = strikes, strikes >= 3
            elif old.get("rounds_off") and int(old.get("n", -1)) == N:
                new["rounds_off"] = True
            if two_level:
                new.update(entries=max(info[4] + (info[4] >> 2) + 1024, int(old.get("entries", 0) * 0.995)),
                           longest=max(info[5] + (info[5] >> 2) + 64, int(old.get("longest", 0) * 0.995)))
            _hints[hint_key] = new
            _coarse_hint[dev_index] = dict(mode=new["mode"], footprint=new["footprint"])

    # SURVEY.md 8b "Sync".  Immediate (default): the host reads the size record before the call returns -- it never drains the
    # stream (everything above is queued behind the count already), but it cannot run ahead of the count either.  Deferred
    # (`_size_check="deferred"`, opt-in): with capacities learnt from earlier calls of this shape the call returns at once; the
    # record is looked at by the next call on this thread, by this call's own backward or by the first read of a list array
    # (`flush_size_checks`).  An overflow found then is repaired IN PLACE -- count .. blend repeated into the very tensors
    # that were handed out, bit-identical to what the immediate check produces -- but whatever the caller enqueued in
    # between has consumed unwritten memory: inference loops see the repaired frame, `backward()` refuses (its upstream
    # gradient came from that memory).  That is why it is not the default.
    deferred = bool(cfg.get("defer_size_check")) and bool(hint) and int(hint.get("n", N)) == N
    ev0 = enqueue_attempt()
    state_late = {}
    if deferred:
        n_buckets_bound = cap // nat.GS_BUCKET + C * tiles + 1

        def resolve_late() -> bool:
            late = False
            ev, whole = ev0, False
            with torch.cuda.device(dev):
                while not settle(ev, whole):
                    late = True
                    ev, whole = enqueue_attempt(), True
            learn()
            state_late["n_isects"], state_late["n_buckets"] = sizes["info"][0], sizes["info"][1]
            if late:
                with _state_lock:
                    stats["late_overflows"] += 1
            return late

        pending = _PendingCheck(resolve_late)
        _pending_checks().append(pending)
        with _state_lock:
            stats["deferred_calls"] += 1
        n_isects, n_buckets, max_tile = None, n_buckets_bound, None
    else:
        pending = None
        ev, whole = ev0, False
        while not settle(ev, whole):
            ev, whole = enqueue_attempt(), True
        learn()
        n_isects, n_buckets, max_tile = sizes["info"][0], sizes["info"][1], sizes["info"][2]

    def isects() -> int:
        if pending is not None:
            pending.resolve()
            return state_late["n_isects"]
        return n_isects

    ref = WS.LeaseRef(lease)
    lazy = _LazyMeta.Lazy
    if lazy_ref:
        ref_lists = _ReferenceLists(rect_ref, depths, C, N, tw, th, eager_ids)
        list_entries = {"tiles_per_gauss": lazy(lambda: ref_lists.get("tiles_per_gauss")),
                        "isect_ids": lazy(lambda: ref_lists.get("isect_ids")) if eager_ids else _LazyMeta.PENDING,
                        "flatten_ids": lazy(lambda: ref_lists.get("flatten_ids")),
                        "isect_offsets": lazy(lambda: ref_lists.get("isect_offsets"))}
    elif rounds:
        # a two-round call never builds the frame's one-round lists: a caller that reads them has them built then, from the
        # footprints the projection left (blocking; same arrays as a one-round call's)
        one = _OneRoundLists(lease, depths, C, N, tw, th)
        list_entries = {"tiles_per_gauss": lazy(lambda: lease.view(WS.TILES_PER_GAUSS, C * N).clone().view(C, N)),
                        "isect_ids": _LazyMeta.PENDING, "flatten_ids": lazy(lambda: one.get("flatten_ids")),
                        "isect_offsets": lazy(lambda: one.get("isect_offsets"))}
    else:
        list_entries = {
            # list arrays: copied out of the workspace on first access (the lease is kept alive by this dict)
            "tiles_per_gauss": lazy(lambda: (isects(), lease.view(WS.TILES_PER_GAUSS, C * N).clone().view(C, N))[1]),
            "isect_ids": lazy(lambda: lease.view(WS.ISECT_IDS, isects()).clone()) if eager_ids else _LazyMeta.PENDING,
            "flatten_ids": lazy(lambda: lease.view(WS.FLATTEN_IDS, isects()).clone()),
            "isect_offsets": lazy(lambda: (isects(), lease.view(WS.ISECT_OFFSETS, C * tiles).clone().view(C, th, tw))[1])}
    meta = _LazyMeta({
        "camera_ids": None, "gaussian_ids": None,
        "radii": radii, "means2d": means2d, "depths": depths, "conics": conics,
        "opacities": opacities[None, :].expand(C, N),
        "tile_width": tw, "tile_height": th, **list_entries,
        "width": W, "height": H, "tile_size": _TILE, "n_cameras": C,
    })
    if not lazy_ref:
        meta._lease = ref   # (the list lazies above read the arenas; in the "gsplat" mode nothing in meta does)
    state = dict(C=C, N=N, K=K, deg=deg, per_cam=per_cam, sh_jac=use_jac, n_isects=n_isects, n_buckets=n_buckets, radii=radii, lease=lease,
                 lease_ref=WS.LeaseRef(lease) if need_grad else None, factorised=factorised, pending=pending, late=state_late,
                 walk=walk, hint_key=hint_key, backgrounds=backgrounds)
    if pending is not None:
        pending.lease_ref = WS.LeaseRef(lease)   # (a repair needs the arenas: leased until the check has run)
    del ref   # (the lease goes back to its pool here unless meta or the autograd node holds it)
    return render_colors, render_alphas, meta, state


def _settle_walk(s: dict, cfg: dict, render_colors: Tensor, render_alphas: Tensor) -> int:
    """Reads the walk record of a training forward (page-locked memory behind an event recorded in the forward: no stream
    synchronisation) before its backward launches anything that follows the work units.  A walk that outgrew its capacities left
    the image complete but the checkpoints / sublists / row bases void: the walk arena is replaced and the blend repeated (into
    scratch images: same bits) with what the record says was needed; the capacity hints of the call shape follow.  Returns the
    number of gradient rows."""
    walk, lease = s.get("walk"), s["lease"]
    if walk is None:
        return 0
    L = nat.lib()
    dev = render_colors.device
    walk.event.synchronize()
    units, storage, rows, fl = (int(v) for v in walk.host.tolist())
    if fl < 0:
        raise nat.NativeLibraryError("rasterization: the forward of this call left no walk record (was it skipped by a step guard?)")
    tries = 0
    while fl & (WS.FLAG_UNITS | WS.FLAG_ROWS):
        tries += 1
        if tries > 3:
            raise nat.NativeLibraryError(f"rasterization: the walk capacities did not settle (units {storage}, rows {rows}, flags {fl})")
        lay = lease.layout
        cu = _quantize_up(storage + (storage >> 2) + 512) if fl & WS.FLAG_UNITS else lay.cap_units
        cr = _quantize_up(rows + (rows >> 2) + 4096) if fl & WS.FLAG_ROWS else lay.cap_rows
        C, N, W, H, cap, coarse_cap, shift, flags = lay.key[:8]
        lease.grow_walk(WS.Layout(C, N, W, H, cap, coarse_cap, shift, flags, cu, cr))
        P = lease.ptr
        with _state_lock:
            stats["walk_reruns"] += 1
        with torch.cuda.device(dev):
            st = _stream(dev)
            sc, sa = torch.empty_like(render_colors), torch.empty_like(render_alphas)
            nat.check(L.gs_walk_mirror_set(walk.host.data_ptr()), "gs_walk_mirror_set")
            try:
                nat.check(L.gs_blend_fwd(st, C, W, H, P(WS.REC), _ptr(s["backgrounds"]), P(WS.ISECT_OFFSETS), P(WS.TILE_ORDER),
                                         P(WS.FLATTEN_IDS), P(WS.SLOTS), s["n_isects"], _ptr(sc), _ptr(sa), P(WS.CKPT), P(WS.QLIST),
                                         P(WS.QCNT), P(WS.QMASK), P(WS.UNIT_DESC), cu, P(WS.ROW_BASE), cr, P(WS.WALK_STATE)), "gs_blend_fwd")
            finally:
                L.gs_walk_mirror_set(None)
            torch.cuda.current_stream(dev).synchronize()   # (rare: the capacities follow the largest recent frame)
        units, storage, rows, fl = (int(v) for v in walk.host.tolist())
    with _state_lock:
        old = _hints.get(s["hint_key"])
        if old is not None:
            old["cap_units"] = max(storage + (storage >> 2) + 512, int(old.get("cap_units", 0) * 0.995))
            old["cap_rows"] = max(rows + (rows >> 2) + 4096, int(old.get("cap_rows", 0) * 0.995))
    return rows


class _Rasterize(torch.autograd.Function):
    """One autograd node for the whole path (P-fwd .. B-fwd | B-bwd .. P-bwd)."""

    @staticmethod
    def forward(ctx, means, quats, scales, opacities, colors, colors_rest, viewmats, Ks, backgrounds, cfg, holder):
        # (needs_input_grad reflects requires_grad of the inputs even under torch.no_grad(); the grad mode
        #  is captured by the caller, forward() itself always runs with grad disabled)
        need_grad = bool(cfg.get("grad_enabled", True)) and any(ctx.needs_input_grad[:6])
        ctx.set_materialize_grads(False)   # an unused render_alphas must not cost a zero-filled image
        render_colors, render_alphas, meta, state = _forward_stages(
            means, quats, scales, opacities, colors, colors_rest, viewmats, Ks, backgrounds, cfg, need_grad)
        holder.meta = meta
        if holder.debug is not None and need_grad:   # work-unit counters of the backward (bench.py's compute roofline): copies
            lease, tiles = state["lease"], meta["tile_width"] * meta["tile_height"]
            wstate = lease.view(WS.WALK_STATE, 8).clone()   # {work units, storage units, -, gradient rows, flags}
            holder.debug.update(unit_counter=wstate[WS.WALK_UNITS:WS.WALK_UNITS + 1], walk_state=wstate,
                                qcnt=lease.view(WS.QCNT, state["C"] * tiles * 4).clone(), unit_entries=nat.GS_UNIT)
            if state["n_isects"] is not None:   # intersections the backward holds gradient rows for (the forward walked them and some pixel took them)
                holder.debug["walked_isects"] = lease.view(WS.QMASK, max(state["n_isects"], 1))[: state["n_isects"]].count_nonzero()
        # The backward reads the forward's workspace (lists, checkpoints, records).  Its lease is NOT held by this ctx -- that
        # would keep it until the OUTPUT tensors die, and a loop that holds the previous image while the next forward runs
        # would pin two full workspaces -- but by the saved tensors (`rasterization()` installs a pack hook that
        # attaches it): the engine drops those when a backward without retain_graph has finished, or when the graph dies.
        holder.lease_ref = state.pop("lease_ref")
        holder.out_ptrs = (render_colors.data_ptr(), render_alphas.data_ptr())
        ctx.cfg, ctx.holder, ctx.state = cfg, holder, state
        ctx.split = colors_rest is not None
        if need_grad:
            extra = (colors_rest,) if ctx.split else ()
            ctx.save_for_backward(means, quats, scales, colors, viewmats, Ks, render_colors, render_alphas, opacities, *extra)
        return render_colors, render_alphas

    @staticmethod
    def backward(ctx, v_render_colors, v_render_alphas):
        L = nat.lib()
        means, quats, scales, colors, viewmats, Ks, render_colors, render_alphas, opacities = ctx.saved_tensors[:9]
        colors_rest = ctx.saved_tensors[9] if ctx.split else None
        s, cfg, holder = ctx.state, ctx.cfg, ctx.holder
        dev = means.device
        st = _stream(dev)
        C, N, K = s["C"], s["N"], s["K"]
        W, H = cfg["width"], cfg["height"]
        f32 = dict(dtype=torch.float32, device=dev)
        v_rc = torch.zeros_like(render_colors) if v_render_colors is None else v_render_colors.contiguous()
        v_ra = None if v_render_alphas is None else v_render_alphas.contiguous()
        lease, factorised = s["lease"], s["factorised"]   # (the forward's workspace: kept leased by the saved tensors, see forward)
        if s.get("pending") is not None:
            # deferred size check: this call's own record (backward may run on the autograd engine's thread)
            if s["pending"].resolve():
                raise RuntimeError("rasterization(_size_check='deferred'): the forward of this call outgrew its list capacities and was "
                                   "repaired only now -- the loss and the upstream gradient of this backward were computed from unwritten "
                                   "memory.  Re-run the step (the capacities have been raised), or use the default immediate size check.")
            s["n_isects"], s["n_buckets"] = s["late"]["n_isects"], s["late"]["n_buckets"]
        P = lease.ptr
        n_rows = _settle_walk(s, cfg, render_colors, render_alphas)
        # scratch for the backward's fill classes (gs_raster.h; a caching-allocator block, re-used stream-ordered)
        ucls = torch.empty((int(L.gs_unit_classes_ints(lease.layout.cap_units, C, W, H)),), dtype=torch.int32, device=dev) if _BWD_CLASSES else None
        _stage("gs_blend_bwd", dev, lambda: nat.check(L.gs_blend_bwd(st, C, W, H, P(WS.REC), P(WS.QLIST), P(WS.QCNT),
                                 P(WS.UNIT_DESC), lease.layout.cap_units, P(WS.CKPT), P(WS.QMASK), P(WS.ROW_BASE), P(WS.WALK_STATE),
                                 _ptr(render_colors), _ptr(render_alphas), _ptr(v_rc), _ptr(v_ra), P(WS.ROWS), _ptr(ucls)), "gs_blend_bwd"))
        go = holder.grad_out or {}   # (`_grad_out`: caller-owned gradient tensors; autograd then receives None for those inputs)
        for k_, shp in (("means", (N, 3)), ("quats", (N, 4)), ("scales", (N, 3)), ("opacities", (N,)), ("grad_norm", (N,)), ("count", (N,))):
            if k_ in go and not (go[k_].shape == shp and go[k_].is_contiguous() and go[k_].dtype == torch.float32 and go[k_].device == dev):
                raise ValueError(f"_grad_out['{k_}'] must be a contiguous float32 tensor of shape {shp} on {dev}")
        if ("grad_norm" in go) != ("count" in go) or ("count" in go and not (factorised and C == 1)):
            raise ValueError("_grad_out['grad_norm'] / ['count'] come together, with _sh_grads='colors_pre' and a single camera")
        v_means = go["means"] if "means" in go else torch.empty((N, 3), **f32)
        v_quats = go["quats"] if "quats" in go else torch.empty((N, 4), **f32)
        v_scales = go["scales"] if "scales" in go else torch.empty((N, 3), **f32)
        v_opac = go["opacities"] if "opacities" in go else torch.empty((N,), **f32)
        v_colors = None if factorised else torch.empty(colors.shape, **f32)
        v_rest = torch.empty(colors_rest.shape, **f32) if (ctx.split and not factorised) else None
        v_pre = row_sums = None
        if factorised:
            # Row sums as a pass of their own (gs_row_sums): they yield the colour gradient other ranks need -- and, with
            # `_view_payload`, this rank's whole all-gather record [3N colour gradients | N normalised radii | view matrix] --
            # BEFORE the long projection backward, so that the exchange started by holder.on_colors_pre overlaps it;
            # gs_project_bwd then takes the sums instead of walking the rows again.  (Rounds 2-4: gs_colors_pre_grad walked the
            # rows' colour lanes a second time, 0.10 ms.)
            row_sums = torch.empty((C * N, 12), **f32)
            pay = holder.view_payload
            if pay is not None:
                if not (C == 1 and pay.dim() == 1 and pay.numel() >= 4 * N + 16 and pay.is_contiguous() and pay.dtype == torch.float32 and pay.device == dev):
                    raise ValueError(f"_view_payload must be a contiguous 1-D float32 tensor of >= 4 N + 16 elements on {dev} (single camera)")
                v_pre, rad_out, cam_out = pay[:3 * N].view(1, N, 3), pay[3 * N:4 * N], pay[4 * N:4 * N + 16]
            else:
                v_pre, rad_out, cam_out = torch.empty((C, N, 3), **f32), None, None
            _stage("gs_row_sums", dev, lambda: nat.check(L.gs_row_sums(
                st, C, N, _ptr(s["radii"]), P(WS.COLORS_POST), P(WS.TILES_PER_GAUSS), P(WS.CUM_TILES), P(WS.ROWS), P(WS.ROW_BASE), P(WS.QMASK),
                _ptr(row_sums), _ptr(v_pre), _ptr(rad_out), float(max(W, H)), _ptr(viewmats), _ptr(cam_out)), "gs_row_sums"))
            if holder.means2d_ref is not None and holder.means2d_ref() is not None:
                holder.means2d_ref().colors_pre_grad = v_pre
            if holder.on_colors_pre is not None:
                holder.on_colors_pre(v_pre)
        v_abs = torch.empty((C, N, 2), **f32)
        dbg = holder.debug
        v_m2 = v_cn = v_cp = None
        m2_out = holder.means2d_ref() if holder.means2d_ref is not None else None
        if dbg is not None or (m2_out is not None and m2_out.requires_grad):
            v_m2 = torch.empty((C, N, 2), **f32)   # dL/d means2d: gsplat's `meta["means2d"].grad` (8 B per Gaussian)
        if dbg is not None:
            v_cn = torch.empty((C, N, 3), **f32)
            v_cp = torch.empty((C, N, 3), **f32)
        _stage("gs_project_bwd", dev, lambda: nat.check(L.gs_project_bwd(st, C, N, K, s["deg"], _ptr(means), _ptr(quats), _ptr(scales), _ptr(colors),
                                   _ptr(colors_rest), s["per_cam"], _ptr(viewmats), _ptr(Ks), W, H, cfg["eps2d"],
                                   cfg["near_plane"], cfg["far_plane"], _ptr(s["radii"]),
                                   P(WS.COLORS_POST), P(WS.TILES_PER_GAUSS), P(WS.CUM_TILES),
                                   P(WS.ROWS), P(WS.ROW_BASE), P(WS.QMASK), _ptr(v_means), _ptr(v_quats), _ptr(v_scales), _ptr(v_opac),
                                   _ptr(v_colors), _ptr(v_rest), _ptr(v_abs), _ptr(v_m2), _ptr(v_cn), _ptr(v_cp), None,
                                   _ptr(opacities), cfg.get("activations", 0), P(WS.SH_JAC) if s.get("sh_jac") else None,
                                   _ptr(row_sums), _ptr(go.get("grad_norm")), _ptr(go.get("count"))), "gs_project_bwd"))
        if holder.grad_out is not None:
            holder.grad_out["_written"] = True   # (the caller's own dict: it can tell that its tensors were filled by THIS backward)
        if dbg is not None:
            dbg.update(v_means2d=v_m2, v_conics=v_cn, v_colors_post=v_cp,
                       rows=lease.view(WS.ROWS, max(n_rows, 1) * nat.GS_ROW_FLOATS)[: n_rows * nat.GS_ROW_FLOATS].clone().view(-1, nat.GS_ROW_FLOATS))
        if m2_out is not None:
            if holder.absgrad:
                m2_out.absgrad = v_abs
            if m2_out.requires_grad and v_m2 is not None:
                # gsplat hands `means2d` out as a graph tensor: after `meta["means2d"].retain_grad()` its `.grad` holds
                # dL/d means2d.  Here the projection and the blend are ONE autograd node, so the tensor is a leaf that
                # requires grad (retain_grad() is then a no-op) and the node deposits the same quantity itself.
                m2_out.grad = v_m2 if m2_out.grad is None else m2_out.grad + v_m2
        ni = ctx.needs_input_grad
        return (v_means if (ni[0] and "means" not in go) else None, v_quats if (ni[1] and "quats" not in go) else None,
                v_scales if (ni[2] and "scales" not in go) else None,
                v_opac if (ni[3] and "opacities" not in go) else None, v_colors if (ni[4] and not factorised) else None,
                v_rest if (ctx.split and ni[5] and not factorised) else None,
                None, None, None, None, None)


def _lease_pack_hook(holder):
    """Pack hook of the node's saved tensors: the INPUTS it saves carry the workspace lease (the engine drops saved tensors
    when a backward without retain_graph has finished, or when the graph dies -- that is when the lease returns to its pool).
    The node's own OUTPUTS (render_colors / render_alphas are saved for the backward) must not: a payload that holds the output
    tensor itself closes a reference cycle output -> grad_fn -> saved payload -> output that Python's collector cannot see
    through (THPFunction does not traverse hook payloads), and a grad-enabled forward whose backward never runs -- an eval
    render, an exception in backward -- would leak the image, the node and the whole workspace.  Outputs are
    therefore saved detached (backward only reads their values) and without the lease."""
    def pack(t):
        if t.data_ptr() in holder.out_ptrs:
            return (t.detach(), None)
        return (t, holder.lease_ref)
    return pack


def sh_grad_views(means: Tensor, viewmats: Tensor, colors_pre_grad: Tensor, sh_degree: int, K: int,
                  split: bool = True):
    """SH-coefficient gradients of R views from their pre-clamp colour gradients (`gs_sh_grad_views`):
    `v_sh[n,k,:] = sum_r Y_k(dir(means[n], camera r)) * colors_pre_grad[r,n,:]`.
    means [N,3], viewmats [R,4,4], colors_pre_grad [R,N,3] -> (v_sh_0 [N,1,3], v_sh_rest [N,K-1,3])
    if `split` else v_shs [N,K,3]."""
    if not means.is_cuda:
        raise RuntimeError("sh_grad_views() runs on the GPU only (there is no CPU fallback in this package)")
    L = nat.lib()
    R, N = colors_pre_grad.shape[0], means.shape[0]
    assert colors_pre_grad.shape == (R, N, 3) and viewmats.shape == (R, 4, 4), (colors_pre_grad.shape, viewmats.shape)
    f32 = dict(dtype=torch.float32, device=means.device)
    means_c, vm_c, v_c = means.detach().contiguous(), viewmats.contiguous(), colors_pre_grad.contiguous()
    if split:
        v0 = torch.empty((N, 1, 3), **f32)
        vr = torch.empty((N, K - 1, 3), **f32)
    else:
        v0, vr = torch.empty((N, K, 3), **f32), None
    with torch.cuda.device(means.device):
        nat.check(L.gs_sh_grad_views(_stream(means.device), R, N, K, sh_degree, _ptr(means_c), _ptr(vm_c), _ptr(v_c),
                                     _ptr(v0), _ptr(vr if (vr is not None and K > 1) else None)), "gs_sh_grad_views")
    return (v0, vr) if split else v0


def quat_to_rotmat_torch(quats: Tensor) -> Tensor:
    """wxyz quaternion (normalised here) -> rotation matrices [...,3,3]; the convention of
    /root/reference/model/utils.py:31-55, used by the split sampling of densify_and_prune."""
    q = torch.nn.functional.normalize(quats, dim=-1)
    w, x, y, z = q.unbind(-1)
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                     2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                     2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=-1)
    return R.reshape(quats.shape[:-1] + (3, 3))


def rasterization(
    means: Tensor,  # [N, 3]
    quats: Tensor,  # [N, 4]  wxyz, need not be normalised
    scales: Tensor,  # [N, 3]
    opacities: Tensor,  # [N]
    colors,  # Tensor [N, K, 3] SH coefficients (sh_degree given) or [N, 3] / [C, N, 3]; or (sh_0, sh_rest)
    viewmats: Tensor,  # [C, 4, 4] world -> camera
    Ks: Tensor,  # [C, 3, 3]
    width: int,
    height: int,
    near_plane: float = 0.01,
    far_plane: float = 1e10,
    radius_clip: float = 0.0,
    eps2d: float = 0.3,
    sh_degree: Optional[int] = None,
    packed: bool = True,
    tile_size: int = 16,
    backgrounds: Optional[Tensor] = None,
    render_mode: str = "RGB",
    sparse_grad: bool = False,
    absgrad: bool = False,
    rasterize_mode: str = "classic",
    channel_chunk: int = 32,
    _debug: Optional[Dict] = None,
    _tile_culling: str = "gsplat",
    _sh_grads: str = "dense",
    _on_colors_pre=None,
    _activations: str = "none",
    _size_check: Optional[str] = None,
    _rounds: Optional[str] = None,
    _grad_out: Optional[Dict[str, Tensor]] = None,
    _view_payload: Optional[Tensor] = None,
) -> Tuple[Tensor, Tensor, Dict]:
    """Rasterize 3D Gaussians to images; same tensor signature and return value as
    `gsplat.rendering.rasterization` (gsplat 1.0.0).

    Returns `(render_colors [C,H,W,3], render_alphas [C,H,W,1], meta)`.  `meta["means2d"]`
    receives the attribute `.absgrad` ([C,N,2]) during backward when `absgrad=True`, and -- a leaf that requires grad whenever
    the render does -- `.grad` = dL/d means2d ([C,N,2]; gsplat: after `meta["means2d"].retain_grad()`, a no-op here);
    `meta["radii"]` is int32 [C,N] with `> 0` marking visible Gaussians.

    Only the configuration the reference exercises is implemented natively; anything else raises
    `NotImplementedError` instead of silently computing something different.

    `_tile_culling="gsplat"` (default): `meta["tiles_per_gauss"]`, `["isect_ids"]`, `["flatten_ids"]`, `["isect_offsets"]` are
    bit-for-bit gsplat's -- built when one of them is first read (`_ReferenceLists`): they are a function of the 3-sigma tile
    rectangles and the depths, nothing in the reference reads them (/root/reference/model/gaussian.py:368-375 takes the image,
    `means2d` and `radii`), and the render itself walks the shorter lists below (image, alphas, radii, means2d bitwise the
    same, gradients to rounding; tested).  `"gsplat_eager"`: the render pipeline walks gsplat's own lists (rounds 1-2's
    default; 1.4-1.7x the list entries).  `"tight"` (what `model.GaussianModel` passes): from gsplat's 3-sigma tile rectangle
    of each Gaussian, the tiles in which no pixel can reach alpha >= 1/255 are dropped; the four list arrays in `meta` are
    then that render-equivalent subset.

    `_sh_grads="colors_pre"` (SH colours only; used by `distributed.ViewParallelStep`) leaves the
    gradients of the SH coefficients to `gs_sh_grad_views`: backward returns `None` for `colors`
    and attaches `meta["means2d"].colors_pre_grad` ([C,N,3], gradient w.r.t. the pre-clamp
    colour), which is all another rank needs to rebuild this view's SH-gradient term.  It is computed
    right after the blend backward; `_on_colors_pre(tensor)` is called at that point, before the
    projection backward is queued, so that an exchange started there overlaps it.

    `_activations="exp_sigmoid"`: `scales` and `opacities` are the reference model's raw parameters
    (log-scales, logit opacities, /root/reference/model/gaussian.py:98-103); exp and sigmoid are applied
    inside the projection kernels and the returned gradients are w.r.t. the raw parameters, which
    removes the model's four activation kernels per step.  `meta["opacities"]` then holds the logits.

    `_grad_out` (used by `distributed.ViewParallelStep`): {"means" | "quats" | "scales" | "opacities": tensor} -- gradient tensors
    the caller owns (e.g. segments of its all-reduce bucket); backward writes those gradients there and hands autograd `None`
    for the corresponding inputs (their `.grad` stays untouched), so no pack / copy pass stands between backward and exchange.
    With `_sh_grads="colors_pre"` and one camera also "grad_norm" / "count" ([N] each): this view's two additive statistics of
    /root/reference/model/gaussian.py:188-197, written (not accumulated) -- the last two segments of that bucket.

    `_view_payload` (`_sh_grads="colors_pre"`, one camera): a flat float32 tensor of >= 4 N + 16 elements that backward fills,
    in ONE launch right after the blend backward, with this rank's record of the view-parallel all-gather: [3N pre-clamp colour
    gradients | N radii / max(W, H) (0 for culled Gaussians) | the 16 floats of the view matrix] (`gs_row_sums`;
    `meta["means2d"].colors_pre_grad` is then a view of its first segment).

    `_rounds` ("auto" | "on" | "off"; default: env GS_ROUNDS or "auto"): depth rounds of the list stages for INFERENCE calls with
    one camera -- the front slab by depth is listed, sorted and blended first, the rest only into tiles it has not finished
    (include/gs_raster.h "Depth rounds"; same image bit for bit).  "auto": on for a call shape whose one-round call listed
    >= 4 M intersections through the two-level binning.  With `_tile_culling="tight"` / `"gsplat_eager"` the list arrays in
    `meta` of such a call are built (in one round, blocking) when they are first read.

    `_size_check` ("immediate" | "deferred"; default: env GS_SIZE_CHECK or "immediate"): when the host looks at the list sizes
    the count kernels reported.  "immediate": before the call returns (one host wait per forward, never a stream drain).
    "deferred": at the next call on this host thread / in this call's backward / at the first read of a list array in `meta`
    (`flush_size_checks()`); the forward then returns without blocking once a call of the same shape has primed the
    capacities.  See `_forward_stages` for what a late-discovered overflow means -- opt-in for that reason.
    """
    N = means.shape[0]
    C = viewmats.shape[0]
    # Extension over gsplat: `colors=(sh_0[N,1,3], sh_rest[N,K-1,3])` hands over the reference model's
    # two SH parameters (/root/reference/model/gaussian.py:49-50) without the per-forward torch.cat.
    colors_rest = None
    if isinstance(colors, (tuple, list)):
        assert sh_degree is not None and len(colors) == 2, "a (sh_0, sh_rest) pair needs sh_degree"
        colors, colors_rest = colors
        assert colors.shape == (N, 1, 3) and colors_rest.dim() == 3 and colors_rest.shape[0] == N and colors_rest.shape[2] == 3
        if colors_rest.shape[1] == 0:
            colors_rest = None
    assert means.shape == (N, 3), means.shape
    assert quats.shape == (N, 4), quats.shape
    assert scales.shape == (N, 3), scales.shape
    assert opacities.shape == (N,), opacities.shape
    assert viewmats.shape == (C, 4, 4), viewmats.shape
    assert Ks.shape == (C, 3, 3), Ks.shape
    assert render_mode in ["RGB", "D", "ED", "RGB+D", "RGB+ED"], render_mode
    if sh_degree is None:
        assert (colors.dim() == 2 and colors.shape[0] == N) or (
            colors.dim() == 3 and colors.shape[:2] == (C, N)), colors.shape
        if colors.shape[-1] != 3:
            raise NotImplementedError("only 3-channel colours are implemented on the HIP path")
    else:
        assert colors.dim() == 3 and colors.shape[0] == N and colors.shape[2] == 3, colors.shape
        k_store = colors.shape[1] + (0 if colors_rest is None else colors_rest.shape[1])
        assert (sh_degree + 1) ** 2 <= k_store, (colors.shape, k_store)
        if sh_degree > 3 or k_store > 16:
            raise NotImplementedError("SH degree > 3 is not implemented")
    if backgrounds is not None:
        assert backgrounds.shape == (C, 3), backgrounds.shape
    if packed:
        raise NotImplementedError("packed=True is not implemented (the reference passes packed=False)")
    if render_mode != "RGB":
        raise NotImplementedError("only render_mode='RGB' is implemented")
    if rasterize_mode != "classic":
        raise NotImplementedError("only rasterize_mode='classic' is implemented")
    if tile_size != _TILE:
        raise NotImplementedError(f"only tile_size={_TILE} is implemented")
    if sparse_grad:
        raise NotImplementedError("sparse_grad requires packed=True")
    if viewmats.requires_grad or Ks.requires_grad:
        # gsplat returns camera gradients; the reference never asks for them and this path
        # does not compute them -- refuse instead of handing back None silently
        raise NotImplementedError("gradients w.r.t. viewmats / Ks are not implemented (the reference's cameras are constants)")
    if not means.is_cuda:
        raise RuntimeError("rasterization() runs on the GPU only: tensors must live on a HIP device "
                           "(there is no CPU fallback in this package)")
    nat.lib()  # fail loudly here if the extension is missing
    size_check = _size_check or os.environ.get("GS_SIZE_CHECK", "immediate")
    if size_check not in ("immediate", "deferred"):
        raise ValueError("_size_check: 'immediate' or 'deferred'")
    if _rounds not in (None, "auto", "on", "off"):
        raise ValueError("_rounds: 'auto', 'on' or 'off'")
    flush_size_checks()   # (size records of earlier deferred calls on this thread: looked at -- and repaired -- in order)

    def prep(t: Tensor) -> Tensor:
        if t.dtype != torch.float32:
            raise TypeError(f"expected float32 tensors, got {t.dtype}")
        if t.device != means.device:
            raise RuntimeError("all tensors must be on the same device")
        return t.contiguous()

    means_c, quats_c, scales_c, opac_c, colors_c = map(prep, (means, quats, scales, opacities, colors))
    rest_c = None if colors_rest is None else prep(colors_rest)
    viewmats_c, Ks_c = prep(viewmats), prep(Ks)
    bg_c = None if backgrounds is None else prep(backgrounds)
    cfg = dict(width=int(width), height=int(height), near_plane=float(near_plane),
               far_plane=float(far_plane), radius_clip=float(radius_clip), eps2d=float(eps2d),
               sh_degree=sh_degree, tile_culling={"gsplat": 1, "tight": 1, "gsplat_eager": 0}[_tile_culling],
               lazy_ref_lists=_tile_culling == "gsplat", sh_grads=_sh_grads,
               activations={"none": 0, "exp_sigmoid": 1}[_activations], grad_enabled=torch.is_grad_enabled(),
               defer_size_check=size_check == "deferred", rounds=_rounds)
    if _sh_grads not in ("dense", "colors_pre") or (_sh_grads == "colors_pre" and sh_degree is None):
        raise ValueError("_sh_grads: 'dense', or 'colors_pre' together with sh_degree")
    holder = _Holder(absgrad)
    holder.debug = _debug
    holder.on_colors_pre = _on_colors_pre
    holder.grad_out = _grad_out
    holder.view_payload = _view_payload
    if _view_payload is not None and _sh_grads != "colors_pre":
        raise ValueError("_view_payload needs _sh_grads='colors_pre'")
    # (pack hook: every tensor the node saves for backward carries the workspace lease -- see _Rasterize.forward)
    with torch.cuda.device(means.device), torch.autograd.graph.saved_tensors_hooks(_lease_pack_hook(holder), lambda p: p[0]):
        render_colors, render_alphas = _Rasterize.apply(means_c, quats_c, scales_c, opac_c, colors_c, rest_c,
                                                        viewmats_c, Ks_c, bg_c, cfg, holder)
    holder.lease_ref = None   # (held by the saved tensors now, if anything was saved)
    meta = holder.meta
    holder.meta = {}
    if render_colors.requires_grad:
        meta["means2d"].requires_grad_(True)   # (gsplat's graph tensor: `.retain_grad()` / `.grad` work, see backward)
    holder.means2d_ref = weakref.ref(meta["means2d"])
    return render_colors, render_alphas, meta
