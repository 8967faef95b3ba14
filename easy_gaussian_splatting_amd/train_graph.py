"""The reference's train step as ONE replayable hipGraph (SURVEY.md section 8b "Sync").

`TrainStepGraph.step(data, gt_img)` runs, for the reference's single camera per iteration
(/root/reference/train.py:36-58, 93-157):

    activations + projection + SH colour -> tile lists -> per-tile sort -> blend forward ->
    clamp + L1 + (1 - SSIM) forward / backward -> blend backward -> projection / SH backward ->
    update_statistics -> Adam step (-> gradients dropped)

through the C ABI of include/gs_raster.h directly -- the same kernels, in the same order, on the same
inputs as `model(data)` / `LossComputer` / `loss.backward()` / `update_statistics` / `FusedAdam.step` issue
them, so the two paths agree bit for bit (tests/test_gpu_train_graph.py) -- but

* every buffer lives in a persistent workspace sized by CAPACITY -- of LISTED intersections (keys, sorted lists, quadrant
  masks, row bases: 13-25 bytes each) and of what the forward WALKS (work units with their checkpoints and sublists, gradient
  rows; round 6: rounds 1-5 sized those by the list capacity, 355 bytes per listed entry, 25 GB at 2 M Gaussians on a realistic
  footprint of which 1.6 % was ever walked) -- high-water marks + margin; nothing is allocated per step and the host never
  reads a size back: the list kernels, the
  statistics and Adam run under the library's step guard (`gs_guard_set`) -- a step whose lists outgrow the
  capacity is a device-side no-op, and so is every step queued behind it;
* the whole sequence is captured once into a hipGraph and replayed: per step the host enqueues the input
  copies, one tiny launch that carries Adam's bias corrections / learning rates as kernel arguments, and the
  graph;
* overflow is detected lazily and without any runtime call: the last launch of every step writes {I, max tile
  list, flags, applied-step count} into page-locked host memory the device can address, which the host polls as
  plain memory; the runner then grows the workspace, re-captures and replays the steps that were skipped, in
  order, with the inputs it kept for them -- the trajectory is exactly the one an unlimited workspace would
  have taken.  (Copy + event based polling between replays of the same graph faulted on this ROCm build.)

A change of N (densify_and_prune), of the parameter storage (reset_opacities), of the active SH degree or of
the image size re-builds the workspace and re-captures.
"""
from __future__ import annotations

import ctypes as ct
import math
import time
from collections import deque
from collections.abc import Mapping
from typing import Any, Dict, Optional

import torch
from torch import Tensor

from . import _native as nat
from .optim import FusedAdam
from . import rendering
from .rendering import _SH_JAC


_TILE = nat.GS_TILE
_SORT_CLASSES = (1024, 4096, 8192, 16384)


def _p(t: Optional[Tensor]):
    return None if t is None else t.data_ptr()


class TrainStepGraph:
    def __init__(self, model, optimizer: FusedAdam, loss_computer, data: Dict[str, Any], gt_img: Tensor,
                 mask: Optional[Tensor] = None, margin: float = 1.3, use_graph: bool = True, check_every: int = 16,
                 fuse_adam: bool = True, handback: str = "eager", copy_targets: bool = False, rounds: str = "auto",
                 round_fraction: float = 0.125):
        """`handback`: when the caller's stream is ordered behind a step.
        "eager" (default): on return from every `step()` -- whatever the caller enqueues next (a read of the outputs, an eval
        render that reads the parameters, `densify_and_prune`) sees the finished step, no thought required.
        "lazy": only when the caller touches the returned outputs (`out["loss3"]`, ...), calls `fence()` or `finish()`.
        A loop that looks at nothing between steps then leaves its own stream idle, and the wait every step performs on
        entry (below) costs nothing; with the eager form that wait sits behind the previous step's hand-back on the caller's
        stream -- two cross-queue signal hops, 24 us of idle GPU per step in the kernel trace of the bench loop.  Parameters
        and statistics are NOT intercepted: a lazy caller that reads them between steps calls `fence()` first."""
        if handback not in ("eager", "lazy"):
            raise ValueError("handback: 'eager' or 'lazy'")
        self.handback = handback
        # rounds: depth rounds of the list stages (include/gs_raster.h "Depth rounds": the front slab of the frame is listed, sorted
        # and blended first, the rest only into tiles it has not finished; same images, sublists and gradient rows bit for bit).
        # "auto" (default): on when the build-time probe finds that the frame lists several times what its blend walks
        # (realistic footprints: 20-50 x) -- the two-round pipeline costs ~0.1 ms of extra launches, which a scene that walks most
        # of what it lists never earns back; "on" / "off" force it.  round_fraction: the share of the listed intersections in
        # front of the depth split.
        rounds = __import__("os").environ.get("GS_TG_ROUNDS", rounds)
        if rounds not in ("auto", "on", "off"):
            raise ValueError("rounds: 'auto', 'on' or 'off'")
        self.rounds_mode, self.round_fraction = rounds, float(__import__("os").environ.get("GS_TG_ROUND_FRACTION", round_fraction))
        self.rounds_on = rounds == "on"
        self._live_sum = self._live_n = 0   # tiles the front round left live, summed over the polled steps since the last build
        # "auto" only: the captured step holds the FRONT round alone (gs_rounds_set phase 4), speculating that it finishes every
        # frame -- a hipGraph cannot skip the ~18 launches of a back round that has nothing to do (0.09 ms).  A frame that does
        # leave tiles live voids its step on the device (GS_FLAG_BACK) like a capacity overflow; the runner then re-builds with
        # both rounds and replays.
        self.front_only = False
        # copy_targets: every step takes a private copy of its target image (and mask) instead of reading the caller's tensors in
        # place -- for loaders that recycle ONE device staging buffer (`gt_buf.copy_(next)`): a target is read until its step is
        # RETIRED (the loss forward and backward of the replay, and again if an overflow recovery replays the step up to
        # `check_every` steps later), not only until `step()` returns.  Costs the copy the in-place read saved (25 MB at 1080p).
        self.copy_targets = bool(copy_targets)
        if not isinstance(optimizer, FusedAdam):
            raise TypeError("TrainStepGraph drives optim.FusedAdam (flat parameter / moment buffers)")
        if not getattr(loss_computer, "clamp_input", False) or not getattr(loss_computer, "fused", True):
            raise ValueError("TrainStepGraph needs LossComputer(fused=True, clamp_input=True)")
        if getattr(loss_computer, "model", None) is not None and getattr(model, "USE_SCALE_REGULARIZATION", False):
            raise NotImplementedError("the scale regulariser is not part of the captured step")
        self.model, self.opt, self.lc = model, optimizer, loss_computer
        self.margin, self.use_graph, self.check_every = float(margin), bool(use_graph), int(check_every)
        # fuse_adam: gs_project_bwd_adam -- the projection / SH backward applies Adam in place where each gradient is
        # formed (no gradient tensors, no separate optimizer pass); same update bit for bit
        self.fuse_adam = bool(fuse_adam)
        self.dev = model.means.device
        self.W, self.H = int(data["width"]), int(data["height"])
        self.has_mask = mask is not None
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        # The step runs on a stream of its own, fenced against the caller's current stream on both sides.  (Replays on
        # the legacy NULL stream -- torch's default -- interleaved with other work on that stream end in GPU memory
        # faults on this ROCm build; DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 also avoids them.  A dedicated stream does not
        # need the runtime knob.)
        self.stream = torch.cuda.Stream(self.dev)
        self.cap = 0
        self.cap_units = self.cap_rows = 0   # capacities of what the forward walks: work units (storage units), gradient rows
        self.cap_floor = 0   # capacity carried over a rebuild: the probe sees ONE view, the steps before saw them all
        self.walk_floor = (0, 0)
        self.seen_isects = 0  # largest intersection count the status words have shown since the last (re-)build
        self.seen_tile = 0    # ... and the longest tile list
        self.seen_units = self.seen_rows = 0   # ... and the most storage units / gradient rows a step's walk took
        self.probed_walk = (0, 0)
        self._walk_history_void = False
        self.binning = None
        self.probed = (0, 0)
        self.cap_tile = _SORT_CLASSES[0]
        self.pending: deque = deque()     # steps issued but not yet confirmed applied: (t, lrs, w2c, K, gt, mask)
        # {I, n_buckets, max tile, flags, applied} of the latest finished step, written by the device (gs_step_status)
        self.status = torch.zeros((8,), dtype=torch.int64, pin_memory=True)
        self.confirmed = 0                # steps known to be applied
        self.issued = 0
        self.stats = {"captures": 0, "overflows": 0, "replayed_steps": 0, "rebuilds": 0}
        self._pool: Dict[str, Tensor] = {}
        self._warm = set()     # (binning mode, tile-sort class) pairs that have run EAGERLY in this runner (warm-up steps)
        self._key = None
        self.confirmed_at_build = 0
        self._last_inputs = (data["w2c"], data["K"], gt_img, mask)
        self._build(data, gt_img, mask)

    # ------------------------------------------------------------------------------------------ workspace
    def _state_key(self, W: Optional[int] = None, H: Optional[int] = None):
        """What the captured graph is specialised on.  `W` / `H`: the size of the frame about to be rendered (the
        reference takes it per frame from `Frame.to_data`, e.g. multi-camera COLMAP sets); default = the built size."""
        m = self.model
        return (m.means.shape[0], self.opt.flat_param.data_ptr(), m.max_radii.data_ptr(), m.grad_norm_accum.data_ptr(),
                m.collecting_counts.data_ptr(), m.active_sh_degree, getattr(m, "tile_culling", "tight"),
                self.W if W is None else int(W), self.H if H is None else int(H))

    def _build(self, data, gt_img, mask, min_cap: int = 0, min_cap_tile: int = 0, projected=None, min_walk=(0, 0)):
        """(Re-)allocates the workspace for the model as it is now, learns the capacities from one blocking probe
        of the list sizes if needed, warms every kernel up eagerly and captures the step.
        `projected` = (I, longest tile list, storage units, gradient rows) carried over from the steps before a refinement (`step()`): no probe, no
        warm-up step -- the re-build is allocation + capture and never touches the device's queue.  (A projection that
        does not hold trips the step guard like any overflow; `_recover` then probes.)"""
        m, dev = self.model, self.dev
        L = nat.lib()
        t_build = time.perf_counter()
        self.N = N = m.means.shape[0]
        self.K = 1 + m.sh_rest.shape[1]
        W, H = self.W, self.H
        self.tw, self.th = math.ceil(W / _TILE), math.ceil(H / _TILE)
        tiles = self.tw * self.th
        f32 = dict(dtype=torch.float32, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        b = self.buf = {}
        b["viewmats"] = torch.empty((1, 4, 4), **f32)
        b["Ks"] = torch.empty((1, 3, 3), **f32)
        # {gt, mask} POINTERS of the step about to run (gs_step_inputs writes them, the loss entries read through them): the
        # target image of a step is the caller's own tensor -- kept alive by the pending entry --, never copied
        b["img_slots"] = torch.zeros((2,), dtype=torch.int64, device=dev)
        b["bg"] = m.BACKGROUND.detach().reshape(1, 3).to(dev, torch.float32).contiguous().clone()
        b["radii"] = self._take("radii", (1, N), torch.int32)
        b["means2d"] = self._take("means2d", (1, N, 2), torch.float32)
        b["depths"] = self._take("depths", (1, N), torch.float32)
        b["conics"] = self._take("conics", (1, N, 3), torch.float32)
        b["colors_post"] = self._take("colors_post", (1, N, 3), torch.float32)
        # d colour / d view direction of the visible Gaussians (gs_project_fwd -> gs_project_bwd*: no SH coefficient is read
        # by the backward); GS_SH_JAC=0 keeps the coefficient-staging backward
        b["sh_jac"] = self._take("sh_jac", (N * 9,), torch.float32) if _SH_JAC else None
        b["rec"] = self._take("rec", (N, nat.GS_REC_FLOATS), torch.float32)
        b["bbox"] = self._take("bbox", (N, 4), torch.int32)
        b["tiles_per_gauss"] = self._take("tiles_per_gauss", (1, N), torch.int32)
        b["cum_tiles"] = self._take("cum_tiles", (N,), torch.int32)
        b["isect_offsets"] = torch.empty((tiles + 1,), **i32)
        b["bucket_offsets"] = torch.empty((tiles + 1,), **i32)
        b["tile_order"] = torch.empty((tiles,), **i32)
        b["info"] = torch.zeros((8,), dtype=torch.int64, device=dev)
        b["applied"] = torch.zeros((1,), dtype=torch.int64, device=dev)
        b["render_colors"] = torch.empty((1, H, W, 3), **f32)
        b["render_alphas"] = torch.empty((1, H, W, 1), **f32)
        b["loss_ws"] = torch.empty((int(L.gs_loss_workspace_floats(H, W)),), **f32)
        b["loss3"] = torch.zeros((3,), **f32)
        b["loss_ring"] = torch.zeros((self.LOSS_RING, 3), **f32)
        b["one"] = torch.ones((), **f32)
        b["v_render"] = torch.empty((H, W, 3), **f32)
        b["qcnt"] = torch.empty((tiles * 4,), **i32)
        b["v_abs"] = self._take("v_abs", (1, N, 2), torch.float32)
        self.grads = None if self.fuse_adam else {
            "means": torch.empty((N, 3), **f32), "log_scales": torch.empty((N, 3), **f32),
            "quats": torch.empty((N, 4), **f32), "sh_0": torch.empty((N, 1, 3), **f32),
            "sh_rest": torch.empty((N, self.K - 1, 3), **f32) if self.K > 1 else None,
            "logit_opacities": torch.empty((N,), **f32)}
        b["hyper"] = torch.zeros((16,), **f32)
        self._stage_inputs(max(self.opt._step, 0) + 1, [float(grp["lr"]) for grp, _ in self.opt._plist], data["w2c"], data["K"], gt_img, mask)
        probe_walk = False
        front_before, self.front_only = self.front_only, False   # (the probes below run both rounds)
        if projected is not None:
            n_isects, max_tile, units, rows = projected
            self.cap = int(n_isects * self.margin) + 4096
            need_tile = int(max_tile * self.margin)
            self.cap_tile = next((c for c in _SORT_CLASSES if c >= need_tile), 1 << 30)
            self.cap_units, self.cap_rows = int(units * self.margin) + 512, int(rows * self.margin) + 4096
            self.stats["projected_rebuilds"] = self.stats.get("projected_rebuilds", 0) + 1
        elif self.cap == 0 or min_cap or min_cap_tile or min_walk[0] or min_walk[1]:
            n_isects, max_tile = self._probe()
            self.probed = (n_isects, max_tile)
            self.cap = max(int(max(n_isects, min_cap) * self.margin) + 4096, self.cap, self.cap_floor)
            need_tile = max(int(max(max_tile, min_cap_tile) * self.margin), self.cap_tile)   # (same head-room as the lists)
            self.cap_tile = next((c for c in _SORT_CLASSES if c >= need_tile), 1 << 30)
            probe_walk = True
        self._alloc_binning()
        self._alloc_lists()
        self._alloc_rounds()
        if probe_walk:
            # what the forward WALKS on this view: one guarded forward on first-guess capacities, repeated with what it reports
            floor = (max(min_walk[0], self.walk_floor[0]), max(min_walk[1], self.walk_floor[1]))
            self.cap_units = max(self.cap_units, int(floor[0] * self.margin) + 512, self.cap // 64 + 4 * tiles + 512)
            self.cap_rows = max(self.cap_rows, int(floor[1] * self.margin) + 4096, self.cap // 8 + 4096)
            while True:
                self._alloc_walk()
                units, rows, fl = self._probe_walk()
                if not fl:
                    break
                self.cap_units = max(self.cap_units, int(units * self.margin) + 512)
                self.cap_rows = max(self.cap_rows, int(rows * self.margin) + 4096)
            self.probed_walk = (units, rows)
            # The probe saw ONE view.  The list capacity carries the history of all of them (cap_floor); the walk is scaled to
            # it -- and after an opacity reset, when the history of the walk is void (nothing saturates any more, how deep a
            # view walks is anybody's guess: 1.1 to 2.6 M rows over the six views of tests/test_gpu_train_graph.py's soak),
            # with twice the head-room until the next re-build has steps to go by.
            scale = max(1.0, (self.cap / self.margin) / max(n_isects, 1))
            wm = self.margin * (2.0 if self._walk_history_void else 1.0)
            self.cap_units = max(int(max(units * scale, floor[0]) * wm) + 512, 256)
            self.cap_rows = max(int(max(rows * scale, floor[1]) * wm) + 4096, 4096)
            self._walk_history_void = False
            if self.rounds_mode == "auto":
                # listed against walked: a gradient row is one (intersection, quadrant) pair some pixel took, 1.5 per walked entry
                self.rounds_on = n_isects >= self.ROUNDS_MIN_LISTED and n_isects >= self.ROUNDS_MIN_RATIO * max(rows, 1)
        elif self.rounds_mode == "auto" and self.rounds_on and self._live_n > 0 and self._live_sum > self.ROUNDS_MAX_LIVE * tiles * self._live_n:
            self.rounds_on = False   # (a re-build without a probe: the steps since the last build left too many tiles to the back round)
        self.front_only = bool(front_before and not probe_walk and self.rounds_on and self.rounds_mode == "auto" and self._live_sum == 0)
        self._alloc_walk()
        self._alloc_rounds()
        if probe_walk and self.rounds_mode == "auto" and self.rounds_on:
            # ... and only where the front slab finishes (nearly) the whole frame: a back round that has work pays the passes over
            # the footprints and the fixed costs of the list stages a second time -- with 38 % of the tiles still live the two-round
            # step is 4-9 % SLOWER than one round (tools/rounds_time.py heavy2Mwin), with 7 % it breaks even
            self._probe_walk()
            live = int(b["rounds"][nat.GS_ROUND_LIVE])
            self.stats["probed_live_tiles"] = live
            if live > self.ROUNDS_MAX_LIVE * tiles:
                self.rounds_on = False
            # (a probing re-build that follows a voided speculation -- `_recover` -- keeps both rounds until the next refinement)
            self.front_only = self.rounds_on and live == 0 and self.speculate_front and not self._back_needed
            self._back_needed = False
        self._live_sum = self._live_n = 0
        self._key = self._state_key()
        self._opacity_resets = getattr(m, "opacity_resets", 0)
        # eager warm-up of the guarded pipeline (raises every kernel attribute; also a functional check before capture)
        t_cap = time.perf_counter()
        self._capture(warm_up=projected is None)
        self.stats["rebuilds"] += 1
        now = time.perf_counter()
        self.stats["build_ms"] = round(self.stats.get("build_ms", 0.0) + 1e3 * (now - t_build), 2)        # whole (re-)builds, wall clock
        self.stats["capture_ms"] = round(self.stats.get("capture_ms", 0.0) + 1e3 * (now - t_cap), 2)    # ... of which warm-up + capture

    def _take(self, name: str, shape, dtype, shrink: bool = False) -> Tensor:
        """A buffer of the workspace from the runner's pool: re-used across re-builds while it fits, re-allocated with head-room
        when it does not (a model that has been refined once will be refined again: `densify_and_prune` grows N by 20-30 % per
        call, and a re-build that has to `hipMalloc` twenty-five new buffers cost 16 ms per refinement on some boxes of the
        pool against 1.3 ms of capture -- bench.py `real_loop`).  First build: exact sizes."""
        n = 1
        for d in shape:
            n *= int(d)
        t = self._pool.get(name)
        if shrink and t is not None and t.numel() > 2 * n + 4096:
            t = None   # (a first-guess buffer of the walk probe several times what the walk turned out to need: let it go)
        if t is None or t.dtype != dtype or t.numel() < n:
            slack = 1.0 if self.stats["rebuilds"] == 0 else 1.6
            t = torch.empty((int(n * slack) + 16,), dtype=dtype, device=self.dev)
            self._pool[name] = t
            self.stats["pool_allocs"] = self.stats.get("pool_allocs", 0) + 1
        return t[:n].view(*shape)

    def _alloc_binning(self):
        """Workspace of the binning pipeline the probe chose (rendering.binning_mode): per-tile lists sorted one by one
        ("tiles") or coarse-bin lists sorted and refined ("bins"; capacities for the coarse entries and the longest bin
        list learnt from the probe, overflow flagged on the device like the list capacity)."""
        L, b, dev = nat.lib(), self.buf, self.dev
        if self.binning == "bins":
            ws = int(L.gs_bins_workspace_bytes(1, self.N, self.tw, self.th, self.bin_shift, self.cap_coarse))
            b["coarse_keys"] = self._take("coarse_keys", (self.cap_coarse,), torch.int64)
        else:
            ws = int(L.gs_bin_workspace_bytes(1, self.N, self.tw, self.th))
            b["keys_tmp"] = self._take("keys_tmp", (self.cap,), torch.int64)
            b["slot_gid"] = self._take("slot_gid", (self.cap,), torch.int32)
        self.ws_bytes = ws
        b["ws"] = self._take("ws", (ws,), torch.uint8)

    def _alloc_lists(self):
        """What is sized by the LISTED intersections: the sorted lists, a quadrant-mask byte and a row base per entry."""
        b, cap = self.buf, self.cap
        b["flatten_ids"] = self._take("flatten_ids", (cap,), torch.int32)
        b["slots"] = self._take("slots", (cap,), torch.int32)
        b["qmask"] = self._take("qmask", (cap + 16,), torch.uint8)
        b["row_base"] = self._take("row_base", (cap // 16 + 4,), torch.int32)
        b["walk_state"] = self._take("walk_state", (int(nat.lib().gs_walk_state_ints(cap)),), torch.int32)

    def _alloc_walk(self):
        """What is sized by what the forward WALKS: a checkpoint, a sublist block and a descriptor per work unit, the gradient rows."""
        b = self.buf
        cu, cr = max(int(self.cap_units), 256), max(int(self.cap_rows), 1)
        self.cap_units, self.cap_rows = cu, cr
        b["ckpt"] = self._take("ckpt", (cu, 64, 4), torch.float32, shrink=True)
        b["qlist"] = self._take("qlist", (cu, nat.GS_UNIT, 2), torch.int32, shrink=True)
        b["unit_desc"] = self._take("unit_desc", (cu, 4), torch.int32, shrink=True)
        # the backward's fill classes (gs_raster.h: gs_blend_bwd groups the units by how full they are); GS_BWD_CLASSES=0: off
        b["unit_cls"] = (self._take("unit_cls", (int(nat.lib().gs_unit_classes_ints(cu, 1, self.W, self.H)),), torch.int32, shrink=True)
                         if rendering._BWD_CLASSES else None)
        b["rows"] = self._take("rows", (cr, nat.GS_ROW_FLOATS), torch.float32, shrink=True)

    ROUNDS_MIN_LISTED = 4_000_000   # "auto": below this the list stages are too short for a second round to pay
    ROUNDS_MIN_RATIO = 4.0          # ... and so they are when the frame lists less than this many entries per gradient row
    ROUNDS_MAX_LIVE = 0.05          # ... and rounds stay off where the front slab leaves more than this share of the tiles live
    speculate_front = __import__('os').environ.get('GS_TG_FRONT_ONLY', '1') != '0'   # 0: "auto" never captures the front round alone
    _back_needed = False

    def _alloc_rounds(self):
        """Depth rounds: the round block, the tiles' liveness / pixel states / sublist records between the rounds, the footprints
        and per-Gaussian counts of the round at hand."""
        b, tiles = self.buf, self.tw * self.th
        if not self.rounds_on:
            return
        b["rounds"] = torch.zeros((nat.GS_ROUND_WORDS,), dtype=torch.int64, device=self.dev)
        b["depth_hist"] = torch.zeros((4096,), dtype=torch.int32, device=self.dev)
        b["tile_live"] = torch.zeros((tiles,), dtype=torch.uint8, device=self.dev)
        b["tile_state"] = self._take("tile_state", (tiles, 4, 64, 4), torch.float32)
        b["tile_rec"] = self._take("tile_rec", (tiles, 8), torch.int32)
        b["bbox_round"] = self._take("bbox_round", (self.N, 4), torch.int32)
        b["tpg_round"] = self._take("tpg_round", (1, self.N), torch.int32)

    def _rounds_phase(self, phase: int):
        L, b = nat.lib(), self.buf
        if phase == 0:
            nat.check(L.gs_rounds_set(None, None, None, None, 0), "gs_rounds_set")
        else:
            nat.check(L.gs_rounds_set(_p(b["rounds"]), _p(b["tile_live"]), _p(b["tile_state"]), _p(b["tile_rec"]), phase), "gs_rounds_set")

    def _tpg(self):
        """The per-Gaussian intersection counts the backward reads: the projection's, or (depth rounds) each Gaussian's in its round."""
        return _p(self.buf["tpg_round"] if self.rounds_on else self.buf["tiles_per_gauss"])

    def _probe_walk(self):
        """One guarded forward (projection .. blend) on the current capacities; returns (storage units, gradient rows, flags) of
        its walk -- the counters keep counting past the capacities.  Build time only (blocking)."""
        b = self.buf
        with torch.cuda.device(self.dev), self._on_stream():
            nat.check(nat.lib().gs_guard_set(_p(b["info"]), self.cap, self.cap_tile), "gs_guard_set")
            try:
                self._stage_no = 0
                self._enqueue_forward()
            except TrainStepGraph._Stop:
                pass
            finally:
                nat.lib().gs_guard_set(None, 0, 0)
                nat.lib().gs_rounds_set(None, None, None, None, 0)
        self.stream.synchronize()
        info = b["info"].tolist()
        w = b["walk_state"][:8].tolist()
        b["info"].zero_()
        if int(info[3]) & ~48:
            raise RuntimeError(f"TrainStepGraph: the probed list capacities do not hold their own view {info}")
        return int(w[1]), int(w[3]), int(info[3]) & 48

    def _protect_pending(self, static: Tensor):
        """`static` (one of the runner's input buffers) is about to be overwritten: steps still pending that were issued
        with "same as the buffer holds" get a snapshot of it first (enqueued ahead of the overwrite on the same stream)."""
        snap = None
        for e in self.pending:
            for i in range(2, 6):
                if isinstance(e[i], Tensor) and e[i].data_ptr() == static.data_ptr():
                    if snap is None:
                        snap = static.clone()
                    e[i] = snap[0] if (e[i].dim() < static.dim()) else snap

    def _image(self, t: Tensor, shape) -> Tensor:
        """The caller's image as the loss kernels read it: float32, contiguous, on the runner's device (a tensor that already is
        comes back as it is -- the usual case; anything else is converted once)."""
        if t.device != self.dev or t.dtype != torch.float32 or not t.is_contiguous() or tuple(t.shape) != tuple(shape):
            if tuple(t.shape) != tuple(shape):
                raise ValueError(f"TrainStepGraph: image of shape {tuple(t.shape)}, expected {tuple(shape)}")
            t = t.to(device=self.dev, dtype=torch.float32).contiguous()
        return t

    def _stage_inputs(self, t: int, lrs, w2c: Tensor, K: Tensor, gt: Tensor, mask: Optional[Tensor]):
        """Everything that differs from the previous step, in ONE launch on the current stream (gs_step_inputs): Adam's bias
        corrections / learning rates, the camera into the static buffers, and the POINTERS to the target image and mask.
        Returns the (gt, mask) tensors whose pointers went in -- the caller keeps them alive until the step is applied."""
        L, b, opt = nat.lib(), self.buf, self.opt
        ok = lambda x, n: x.device == self.dev and x.dtype == torch.float32 and x.is_contiguous() and x.numel() == n   # noqa: E731
        vm_src = k_src = None
        if w2c.data_ptr() != b["viewmats"].data_ptr():   # (the static buffer itself = "same as last step")
            self._protect_pending(b["viewmats"])
            if ok(w2c, 16):
                vm_src = w2c
            else:
                b["viewmats"][0].copy_(w2c, non_blocking=True)
        if K.data_ptr() != b["Ks"].data_ptr():
            self._protect_pending(b["Ks"])
            if ok(K, 9):
                k_src = K
            else:
                b["Ks"][0].copy_(K, non_blocking=True)
        gt = self._image(gt, (self.H, self.W, 3))
        if self.has_mask:
            if mask is None:
                raise ValueError("this runner was built with a mask; pass one every step")
            mask = self._image(mask, (self.H, self.W))
        else:
            mask = None
        ns = len(opt._plist)
        b1, b2 = opt.defaults["betas"]
        nat.check(L.gs_step_inputs(self._st(), ns, (ct.c_float * ns)(*lrs), float(b1), float(b2), int(t), _p(b["hyper"]), _p(vm_src),
                                   _p(b["viewmats"]), _p(k_src), _p(b["Ks"]), _p(gt), _p(mask), _p(b["img_slots"])), "gs_step_inputs")
        self._cur_images = (gt, mask)
        return gt, mask

    def _st(self) -> int:
        return torch.cuda.current_stream(self.dev).cuda_stream

    class _OnStepStream:
        """`with runner._on_stream():` -- everything inside is enqueued on the runner's stream, ordered after the
        caller's current stream on entry and (`hand_back`) before it on exit."""

        def __init__(self, runner, wait_outer: bool = True, hand_back: bool = True):
            self.r = runner
            self.wait_outer = wait_outer
            self.hand_back = hand_back

        def __enter__(self):
            r = self.r
            self.outer = torch.cuda.current_stream(r.dev)
            if self.wait_outer:
                r.stream.wait_stream(self.outer)
            self.ctx = torch.cuda.stream(r.stream)
            self.ctx.__enter__()

        def __exit__(self, *exc):
            self.ctx.__exit__(*exc)
            if self.hand_back:
                self.outer.wait_stream(self.r.stream)
            return False

    def _on_stream(self, wait_outer: bool = True, hand_back: bool = True):
        return TrainStepGraph._OnStepStream(self, wait_outer, hand_back)

    def fence(self):
        """Orders the caller's current stream behind every step issued so far (what `handback="eager"` does on return from
        each `step()`): call it before reading parameters / statistics / outputs on the caller's stream in "lazy" mode."""
        torch.cuda.current_stream(self.dev).wait_stream(self.stream)

    class _StepOutputs(Mapping):
        """The runner's static output tensors; in `handback="lazy"` mode the first access orders the caller's current
        stream behind the step that produced them.  A read-only Mapping, not a dict subclass: `dict(out)`, `{**out}`, `.get`,
        `.values()`, `.items()` all reach the tensors through `__getitem__` -- a dict subclass is read by the C-level fast
        paths behind the fence's back."""

        def __init__(self, runner, items, lazy: bool):
            self._items, self._runner, self._pending = dict(items), runner, lazy

        def _touch(self):
            if self._pending:
                self._pending = False
                self._runner.fence()

        def __getitem__(self, k):
            self._touch()
            return self._items[k]

        def __iter__(self):
            return iter(self._items)

        def __len__(self):
            return len(self._items)

    stop_after = int(__import__('os').environ.get('GS_TG_STOP_AFTER', '0'))
    project_rebuilds = __import__('os').environ.get('GS_TG_PROJECT_REBUILDS', '1') != '0'   # 0: every re-build probes and warms up
    debug_sync = False   # set True to synchronise after every stage (locates a faulting kernel; never under capture)

    class _Stop(Exception):
        pass

    def _ck(self, rc: int, what: str) -> None:
        nat.check(rc, what)
        self._stage_no = getattr(self, "_stage_no", 0) + 1
        if self.stop_after and self._stage_no >= self.stop_after:   # debugging aid: truncate the pipeline
            raise TrainStepGraph._Stop()
        if self.debug_sync and not torch.cuda.is_current_stream_capturing():
            self.stream.synchronize()

    def _project(self):
        L, b, m = nat.lib(), self.buf, self.model
        # (the runner hands out no list arrays: "gsplat" -- exact arrays, built when read -- renders from the short lists
        #  like "tight"; "gsplat_eager" walks gsplat's own lists)
        culling = {"gsplat": 1, "tight": 1, "gsplat_eager": 0}[getattr(m, "tile_culling", "tight")]
        self._ck(L.gs_project_fwd(self._st(), 1, self.N, self.K, int(m.active_sh_degree), _p(m.means), _p(m.quats), _p(m.log_scales),
                                   _p(m.logit_opacities), _p(m.sh_0), _p(m.sh_rest) if self.K > 1 else None, 0,
                                   _p(b["viewmats"]), _p(b["Ks"]), self.W, self.H, 0.3, 0.01, 1e10, 0.0, culling, 0, 1,
                                   _p(b["radii"]), _p(b["means2d"]), _p(b["depths"]), _p(b["conics"]), _p(b["colors_post"]),
                                   _p(b["rec"]), _p(b["bbox"]), _p(b["tiles_per_gauss"]), None, _p(b["sh_jac"])), "gs_project_fwd")

    def _count(self, bbox: str = "bbox"):
        L, b = nat.lib(), self.buf
        if self.binning == "bins":
            self._ck(L.gs_bins_count(self._st(), 1, self.N, self.tw, self.th, self.bin_shift, _p(b[bbox]), _p(b["depths"]), _p(b["ws"]),
                                      self.ws_bytes, _p(b["coarse_keys"]), self.cap_coarse, self.cap_coarse_list, _p(b["cum_tiles"]),
                                      _p(b["isect_offsets"]), _p(b["bucket_offsets"]), _p(b["tile_order"]), _p(b["info"]), None),
                     "gs_bins_count")
        else:
            self._ck(L.gs_bin_count(self._st(), 1, self.N, self.tw, self.th, _p(b[bbox]), _p(b["ws"]), self.ws_bytes,
                                     _p(b["isect_offsets"]), _p(b["bucket_offsets"]), _p(b["tile_order"]), _p(b["info"]), None),
                     "gs_bin_count")

    def _probe(self):
        """Blocking reads of the list sizes for the current inputs (build time only): {I, longest tile list} from the
        per-tile count, then -- if the footprints call for the two-level binning -- {coarse entries, longest bin list}."""
        from .rendering import MAX_TILES_PER_TILE_PIPELINE, bin_shift_for, binning_choice
        L, b, dev = nat.lib(), self.buf, self.dev
        tiles = self.tw * self.th
        n_isects = max_tile = 0
        footprint = None
        if tiles <= MAX_TILES_PER_TILE_PIPELINE:
            self.binning = "tiles"
            self.ws_bytes = int(L.gs_bin_workspace_bytes(1, self.N, self.tw, self.th))
            b["ws"] = torch.empty((self.ws_bytes,), dtype=torch.uint8, device=dev)
            with torch.cuda.device(dev), self._on_stream():
                self._project()
                self._count()
            info = b["info"].tolist()
            n_isects, max_tile = int(info[0]), int(info[2])
            footprint = n_isects / max(1, self.N)
        else:
            with torch.cuda.device(dev), self._on_stream():
                self._project()
        self.binning = binning_choice(footprint, tiles)
        if self.binning == "bins":
            self.bin_shift = bin_shift_for(footprint) or 2
            self.cap_coarse, self.cap_coarse_list = 2 * self.N + 1024, 0
            while True:
                self._alloc_binning()
                with torch.cuda.device(dev), self._on_stream():
                    self._count()
                info = b["info"].tolist()
                if not int(info[3]) & 12:
                    break
                self.cap_coarse = int(int(info[4]) * self.margin) + 4096
            self.cap_coarse = int(int(info[4]) * self.margin) + 4096
            self.cap_coarse_list = int(int(info[5]) * 1.5) + 64
            self.probed_coarse = (int(info[4]), int(info[5]))   # coarse-bin entries I', longest bin list
            n_isects, max_tile = int(info[0]), int(info[2])
        b["info"].zero_()
        return n_isects, max_tile

    def _enqueue_forward(self):
        """Projection, tile lists and the training blend on the current stream (the caller holds the step guard).  With depth
        rounds: the list stages and the blend twice, on the footprints of the front and of the back round; the rounds context is
        left at "behind both" for the backward (the caller clears it)."""
        L, b = nat.lib(), self.buf
        self._project()
        if not self.rounds_on:
            self._lists_and_blend("bbox")
            return
        self._ck(L.gs_round_split(self._st(), self.N, _p(b["depths"]), _p(b["tiles_per_gauss"]), self.round_fraction, _p(b["depth_hist"]),
                                   _p(b["rounds"])), "gs_round_split")
        for phase in ((4,) if self.front_only else (1, 2)):
            self._rounds_phase(phase)
            self._ck(L.gs_round_footprints(self._st(), self.N, self.tw, self.th, _p(b["bbox"]), _p(b["depths"]), _p(b["bbox_round"]),
                                            _p(b["tpg_round"])), "gs_round_footprints")
            self._lists_and_blend("bbox_round")
        self._rounds_phase(3)

    def _lists_and_blend(self, bbox: str):
        L, b = nat.lib(), self.buf
        st = self._st()
        N, W, H = self.N, self.W, self.H
        self._count(bbox)
        if self.binning == "bins":
            self._ck(L.gs_bins_lists(st, 1, N, self.tw, self.th, self.bin_shift, _p(b[bbox]), _p(b["ws"]), self.ws_bytes,
                                      _p(b["coarse_keys"]), self.cap_coarse, _p(b["cum_tiles"]), _p(b["isect_offsets"]),
                                      None, _p(b["flatten_ids"]), _p(b["slots"]), _p(b["info"])), "gs_bins_lists")
        else:
            self._ck(L.gs_bin_emit_sort(st, 1, N, self.tw, self.th, _p(b[bbox]), _p(b["depths"]), _p(b["ws"]), self.ws_bytes,
                                         _p(b["isect_offsets"]), self.cap, self.cap_tile, _p(b["keys_tmp"]), _p(b["slot_gid"]),
                                         _p(b["cum_tiles"]), None, _p(b["flatten_ids"]), _p(b["slots"])), "gs_bin_emit_sort")
        self._ck(L.gs_blend_fwd(st, 1, W, H, _p(b["rec"]), _p(b["bg"]), _p(b["isect_offsets"]),
                                 _p(b["tile_order"]), _p(b["flatten_ids"]), _p(b["slots"]), self.cap, _p(b["render_colors"]),
                                 _p(b["render_alphas"]), _p(b["ckpt"]), _p(b["qlist"]), _p(b["qcnt"]), _p(b["qmask"]),
                                 _p(b["unit_desc"]), self.cap_units, _p(b["row_base"]), self.cap_rows, _p(b["walk_state"])),
                 "gs_blend_fwd")

    def _enqueue_step(self):
        """The whole step on the current stream, guarded; nothing here allocates or synchronises."""
        L, b, m, opt = nat.lib(), self.buf, self.model, self.opt
        st = self._st()
        N, W, H = self.N, self.W, self.H
        nat.check(L.gs_guard_set(_p(b["info"]), self.cap, self.cap_tile), "gs_guard_set")
        self._stage_no = 0
        try:
            self._enqueue_forward()
            lam = float(self.lc.lambda_ssim)
            self._ck(L.gs_l1_ssim_fwd_slots(st, H, W, lam, _p(b["render_colors"]), _p(b["img_slots"]), int(self.has_mask), 1, _p(b["loss_ws"]),
                                             _p(b["loss3"])), "gs_l1_ssim_fwd_slots")
            self._ck(L.gs_l1_ssim_bwd_slots(st, H, W, lam, _p(b["render_colors"]), _p(b["img_slots"]), 1, _p(b["loss_ws"]),
                                             _p(b["one"]), _p(b["v_render"])), "gs_l1_ssim_bwd_slots")
            self._ck(L.gs_blend_bwd(st, 1, W, H, _p(b["rec"]), _p(b["qlist"]), _p(b["qcnt"]), _p(b["unit_desc"]), self.cap_units,
                                     _p(b["ckpt"]), _p(b["qmask"]), _p(b["row_base"]), _p(b["walk_state"]),
                                     _p(b["render_colors"]), _p(b["render_alphas"]), _p(b["v_render"]), None, _p(b["rows"]), _p(b["unit_cls"])),
                      "gs_blend_bwd")
            b1, b2 = opt.defaults["betas"]
            if self.fuse_adam:
                offs = (ct.c_int64 * 6)(*opt._offs)
                self._ck(L.gs_project_bwd_adam(st, N, self.K, int(m.active_sh_degree), _p(opt.flat_param), _p(opt.exp_avg),
                                               _p(opt.exp_avg_sq), offs, _p(b["viewmats"]), _p(b["Ks"]), W, H, 0.3, 0.01, 1e10,
                                               _p(b["radii"]), _p(b["colors_post"]), self._tpg(), _p(b["cum_tiles"]),
                                               _p(b["rows"]), _p(b["row_base"]), _p(b["qmask"]), _p(b["v_abs"]), float(b1), float(b2),
                                               float(opt.defaults["eps"]), _p(b["hyper"]), _p(b["applied"]), _p(m.max_radii),
                                               _p(m.grad_norm_accum), _p(m.collecting_counts), _p(b["sh_jac"])), "gs_project_bwd_adam")
            else:
                g = self.grads
                self._ck(L.gs_project_bwd(st, 1, N, self.K, int(m.active_sh_degree), _p(m.means), _p(m.quats), _p(m.log_scales),
                                          _p(m.sh_0), _p(m.sh_rest) if self.K > 1 else None, 0, _p(b["viewmats"]), _p(b["Ks"]), W, H,
                                          0.3, 0.01, 1e10, _p(b["radii"]), _p(b["colors_post"]), self._tpg(),
                                          _p(b["cum_tiles"]), _p(b["rows"]), _p(b["row_base"]), _p(b["qmask"]), _p(g["means"]), _p(g["quats"]),
                                          _p(g["log_scales"]), _p(g["logit_opacities"]), _p(g["sh_0"]), _p(g["sh_rest"]), _p(b["v_abs"]),
                                          None, None, None, None, _p(m.logit_opacities), 1, _p(b["sh_jac"]), None, None, None), "gs_project_bwd")
                self._ck(L.gs_update_statistics(st, N, float(max(H, W)), _p(b["radii"]), _p(b["v_abs"]), _p(m.max_radii),
                                                _p(m.grad_norm_accum), _p(m.collecting_counts)), "gs_update_statistics")
                ns = len(opt._plist)
                ends = (ct.c_int64 * ns)(*opt._ends)
                lens = (ct.c_int64 * ns)(*opt._lens)
                gptr = (ct.c_void_p * ns)(*[_p(g[grp["name"]]) for grp, _ in opt._plist])
                self._ck(L.gs_adam_step_dev(st, opt.flat_param.numel(), _p(opt.flat_param), _p(opt.exp_avg), _p(opt.exp_avg_sq), ns,
                                            ends, lens, gptr, float(b1), float(b2), float(opt.defaults["eps"]), 1.0, _p(b["hyper"]),
                                            _p(b["applied"])), "gs_adam_step_dev")
            self._ck(L.gs_step_status(st, _p(b["info"]), _p(b["applied"]), self.status.data_ptr(), _p(b["loss3"]), _p(b["loss_ring"]),
                                      self.LOSS_RING, _p(b["walk_state"])), "gs_step_status")
        except TrainStepGraph._Stop:
            pass
        finally:
            L.gs_guard_set(None, 0, 0)
            L.gs_rounds_set(None, None, None, None, 0)

    def _tail(self, t: int, lrs):
        """What a step enqueues behind its graph (nothing: the single-GPU step is the graph)."""

    def _hyper(self, t: int, lrs):
        L, opt = nat.lib(), self.opt
        ns = len(opt._plist)
        b1, b2 = opt.defaults["betas"]
        nat.check(L.gs_adam_hyper(self._st(), ns, (ct.c_float * ns)(*lrs), float(b1), float(b2), int(t), _p(self.buf["hyper"])),
                  "gs_adam_hyper")

    def _capture(self, warm_up: bool = True):
        """Warm-up outside capture is NOT possible without applying a step, so the first launch of every kernel
        happens on a throw-away copy of the optimizer state: parameters, moments and statistics are saved, one eager
        guarded step runs (raising kernel attributes, validating capacities), and the state is restored.
        `warm_up=False` (a re-build on projected capacities: every kernel has run before in this process): capture only."""
        opt, m = self.opt, self.model
        names = ("max_radii", "grad_norm_accum", "collecting_counts")
        with torch.cuda.device(self.dev), self._on_stream():
            if warm_up:
                self._warm.add((self.binning, self.cap_tile))
                saved = [opt.flat_param.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone()] + [getattr(m, n).clone() for n in names]
                self._hyper(max(opt._step, 0) + 1, [float(grp["lr"]) for grp, _ in opt._plist])
                self._enqueue_step()
                self.stream.synchronize()
                info = self.buf["info"].tolist()
                with torch.no_grad():
                    opt.flat_param.copy_(saved[0]); opt.exp_avg.copy_(saved[1]); opt.exp_avg_sq.copy_(saved[2])
                    for n, s in zip(names, saved[3:]):
                        getattr(m, n).copy_(s)
                self.buf["applied"].zero_()
                if info[3] != 0:   # the capacities learnt from the probe do not hold (cannot happen unless inputs changed in between)
                    self.buf["info"].zero_()
                    raise RuntimeError(f"TrainStepGraph: warm-up step overflowed its own capacities {info}")
            self.graph = None
            if self.use_graph:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=self.stream):
                    self._enqueue_step()
                self.graph = g
                self.stats["captures"] += 1
            if warm_up:
                self.stream.synchronize()
        self.status.zero_()   # (nothing is in flight: the warm-up step's status words are stale)
        self.confirmed_at_build = self.confirmed

    # ------------------------------------------------------------------------------------------ stepping
    def _issue(self, entry):
        t, lrs, w2c, K, gt, mask, ready = entry[:7]
        conv = entry[7] if len(entry) > 7 else None
        after = entry[8] if len(entry) > 8 else None
        # The step waits for the caller's stream on entry: that orders it behind pending WRITERS of the inputs handed in and
        # behind pending READERS of what the replay overwrites -- the static outputs the previous `step()` returned, the
        # parameters and moments (fused Adam), the statistics.  (Round 3 skipped the wait for steps that take no input from
        # the caller; a loss accumulation, a PSNR kernel or an eval render queued on the caller's stream after step i could
        # then run concurrently with replay i+1: ADVICE r3.)  Only `inputs_ready=True` -- the caller's promise that its
        # stream holds neither -- skips it.  The wait is free when the caller's stream is idle, i.e. with handback="lazy" in a
        # loop that reads nothing between steps; behind an eager hand-back it costs two cross-queue signal hops.
        with torch.cuda.device(self.dev), self._on_stream(wait_outer=not ready, hand_back=self.handback == "eager"):
            if conv is not None:
                # the runner's OWN conversion / copy of a target was enqueued on the caller's stream (`step`): the promise
                # `inputs_ready=True` covers the caller's writers, not this one
                self.stream.wait_event(conv)
                entry[7] = None
            if after is not None:   # (`step(ready_event=...)`: the upload of this step's inputs on the caller's copy stream(s))
                for ev in (after if isinstance(after, (list, tuple)) else (after,)):
                    self.stream.wait_event(ev)
                entry[8] = None
            entry[4], entry[5] = self._stage_inputs(t, lrs, w2c, K, gt, mask)   # (the tensors the step reads: kept by the entry)
            if self.graph is not None:
                self.graph.replay()
            else:
                self._enqueue_step()
            self._tail(t, lrs)   # (ViewParallelGraphStep: the collectives and what applies the step; nothing here)
        self.pending.append(entry)
        self.issued += 1

    def step(self, data: Optional[Dict[str, Any]] = None, gt_img: Optional[Tensor] = None, mask: Optional[Tensor] = None,
             inputs_ready: bool = False, ready_event: Optional["torch.cuda.Event"] = None, host_src=None):
        """One training iteration.  `data` / `gt_img` / `mask` default to the previous step's (the camera's static buffers are
        re-used as they are; the target image and the mask are the previous step's own tensors, read in place).  Returns the
        runner's static output tensors (valid until the next `step()` is CALLED: reads
        enqueued on the caller's stream before that call are ordered in front of the replay that overwrites them).
        `inputs_ready=True` is the caller's promise that no work still pending on its stream (a) writes the tensors handed
        in, (b) reads the outputs of an earlier step, the model's parameters or its statistics: the step then does not wait
        for the caller's stream (see `_issue`).  Without the promise every step does.
        LIFETIME of the inputs: the target image and the mask are read IN PLACE, by the loss kernels of the replay and -- after a
        capacity overflow -- again by the replay of the skipped step, up to `check_every` steps later: a tensor handed in must
        not be written (nor its storage recycled by the loader) until the step is retired, i.e. until `finish()` or
        `check_every` further steps.  A loader that refills one staging buffer builds the runner with `copy_targets=True`.
        `ready_event`: an event the step's stream waits for before it reads anything (the upload of the inputs on a copy stream:
        `HostFeed`).  `host_src = {"w2c", "K", "image"[, "mask"]}` host tensors the device inputs were uploaded from: a feeder that
        recycles its device slots hands them in, and an overflow recovery replays the step from a FRESH upload of those instead
        of from slots that have been refilled since."""
        W, H = (self.W, self.H) if data is None else (int(data["width"]), int(data["height"]))
        if self._state_key(W, H) != self._key:
            self.finish()
            size_changed = (W, H) != (self.W, self.H)
            if size_changed:
                # another frame size: the static image buffers are re-allocated, so the frame must bring its own target
                if gt_img is None or (self.has_mask and mask is None):
                    raise ValueError("TrainStepGraph.step: a change of image size needs gt_img (and mask) of the new size")
                cur = (data["w2c"], data["K"], gt_img, mask)
                self.W, self.H = W, H
            else:
                cur = self._last_inputs if data is None else (data["w2c"], data["K"], gt_img if gt_img is not None else self._last_inputs[2],
                                                              mask if mask is not None else self._last_inputs[3])
                # (old static buffers named here stay alive through `cur` until the new ones have been filled from them)
            # the largest list the status words showed since the last build, scaled to the new model size: a refinement
            # re-captures on whichever view comes next, and a capacity learnt from that one view alone overflows on the first
            # wider one (tools/train_soak.py: 153 of 700 steps were replayed before this).  Not the old CAPACITY scaled: that
            # compounds margin on margin, and every kernel whose grid is sized by the capacity pays for the empty workgroups
            n_new = self.model.means.shape[0]
            growth = max(1.0, n_new / max(self.N, 1))
            self.cap_floor = 0 if size_changed else min(int(self.seen_isects * growth * self.margin), (1 << 31) - (1 << 20))
            self.walk_floor = (0, 0) if size_changed else (int(self.seen_units * growth), int(self.seen_rows * growth))
            # ... and when that history exists (same frame size, per-tile binning, a model that grew by less than 2 x) the
            # re-build needs no probe at all: the host never reads the device between the refinement and the next replay
            # (bench.py `real_loop`: a probing re-build cost 5.5 ms + a throw-away warm-up step per refinement, more than the
            # 100 captured steps in between had saved against the eager loop)
            # (an opacity reset changes how deep the blend walks -- nothing saturates any more -- though not the lists: the walk
            #  capacities then come from a fresh probe, the list capacities keep their floor)
            reset = getattr(self.model, "opacity_resets", 0) != self._opacity_resets
            if reset:
                self.walk_floor = (0, 0)
                self._walk_history_void = True
            projected = carried = None
            if (not size_changed and not reset and self.binning == "tiles" and self.seen_isects > 0 and growth <= 2.0 and self.project_rebuilds
                    and self._key is not None and self._state_key(W, H)[5:] == self._key[5:]):
                projected = (min(int(self.seen_isects * growth), 1 << 30), max(int(self.seen_tile * min(growth, 1.25)), 64),
                             int(self.seen_units * growth), int(self.seen_rows * growth))
                # ... but only into a tile-sort class this runner has already run EAGERLY: a class used for the first time is
                # another kernel / another LDS size, whose attributes `ensure_lds` would raise -- and which would launch for the
                # first time -- inside the stream capture (gs_binning.hip: the warm-up exists to avoid exactly that)
                cls = next((c for c in _SORT_CLASSES if c >= int(projected[1] * self.margin)), 1 << 30)
                if (self.binning, cls) not in self._warm:
                    # (the probing re-build still takes the projection as a floor: its probe sees ONE view, the history all)
                    carried, projected = projected, None
            self.seen_isects = self.seen_tile = self.seen_units = self.seen_rows = 0
            self.cap = self.cap_units = self.cap_rows = 0
            if projected is None and carried is not None:
                self._build({"w2c": cur[0], "K": cur[1]}, cur[2], cur[3] if self.has_mask else None, min_cap=carried[0], min_cap_tile=carried[1],
                            min_walk=carried[2:4])
            else:
                self._build({"w2c": cur[0], "K": cur[1]}, cur[2], cur[3] if self.has_mask else None, projected=projected)
        b = self.buf
        w2c = b["viewmats"][0] if data is None else data["w2c"]
        K = b["Ks"][0] if data is None else data["K"]
        gt = self._last_inputs[2] if gt_img is None else gt_img   # (the previous step's own tensors, by reference)
        mk = (self._last_inputs[3] if mask is None else mask) if self.has_mask else None
        gt_in, mk_in = gt, mk
        gt = self._image(gt, (self.H, self.W, 3))   # (validated / converted BEFORE the step is counted)
        if self.has_mask:
            if mk is None:
                raise ValueError("this runner was built with a mask; pass one every step")
            mk = self._image(mk, (self.H, self.W))
        if self.copy_targets and gt_img is not None:   # (a step that re-uses the previous step's tensors re-uses its private copies)
            gt = gt.clone() if gt is gt_in else gt
            mk = (mk.clone() if mk is mk_in else mk) if mk is not None else None
        conv = None
        if gt is not gt_in or mk is not mk_in:
            # a conversion (dtype / device / layout) or a private copy ran on the CALLER's stream just now: the runner's stream
            # waits for exactly that, whatever `inputs_ready` promises about the caller's own work
            conv = torch.cuda.Event()
            conv.record(torch.cuda.current_stream(self.dev))
        self._last_inputs = (w2c, K, gt, mk)
        opt = self.opt
        opt._step += 1
        # The queued entry is what an overflow recovery replays.  Inputs are kept by reference: a caller must not write
        # into a camera / target / mask tensor it has handed in before `finish()` (or `check_every` further steps) -- the
        # target and the mask are not even copied: the step reads them where they lie (`_stage_inputs`).  Entries that alias the
        # runner's OWN static camera buffers (data=None steps) are snapshotted right before a later step overwrites those
        # buffers (`_protect_pending`), so a skipped step is always replayed with its own inputs.
        self._issue([opt._step, [float(grp["lr"]) for grp, _ in opt._plist], w2c, K, gt, mk, bool(inputs_ready), conv, ready_event, host_src])
        if self.issued % self.check_every == 0:
            self._poll(block=False)
        return TrainStepGraph._StepOutputs(self, {"render_img": b["render_colors"][0], "loss3": b["loss3"], "batch_radii": b["radii"],
                                                  "absgrad": b["v_abs"]}, lazy=self.handback == "lazy")

    def _poll(self, block: bool):
        """Reads the device-written status words (plain host memory) and retires the steps known to be applied."""
        if block:
            self.stream.synchronize()
        n_isects, _, max_tile, flags, applied, units, rows, live = (int(v) for v in self.status[:8].tolist())
        if live >= 0 and applied > 0:
            self._live_sum, self._live_n = self._live_sum + live, self._live_n + 1
        self.seen_isects = max(self.seen_isects, n_isects)
        self.seen_tile = max(self.seen_tile, max_tile)
        self.seen_units, self.seen_rows = max(self.seen_units, units), max(self.seen_rows, rows)
        done = min(self.confirmed_at_build + applied - self.confirmed, len(self.pending))
        for _ in range(max(done, 0)):
            self.pending.popleft()
        self.confirmed += max(done, 0)
        if flags != 0:
            self._recover(n_isects, max_tile, (units, rows) if flags & 48 else (0, 0))

    def _recover(self, n_isects: int, max_tile: int, walk=(0, 0)):
        """A step did not fit: everything issued after the last applied step was skipped on the device.  Grow, re-capture,
        replay the skipped steps in order."""
        self.stream.synchronize()
        applied = int(self.buf["applied"].item())
        done = self.confirmed_at_build + applied - self.confirmed
        for _ in range(done):
            self.pending.popleft()
        self.confirmed += done
        redo = list(self.pending)
        self.pending.clear()
        self.issued -= len(redo)
        self.stats["overflows"] += 1
        self.stats["replayed_steps"] += len(redo)
        # (what did not fit, and into what: flags 1 = listed intersections, 2 = longest tile list, 4 / 8 = coarse bins, 16 = work units, 32 = rows)
        if int(self.status[3]) & 64:   # GS_FLAG_BACK: a frame needed the back round the captured step does not hold
            self._back_needed = True
            self.stats["back_round_needed"] = self.stats.get("back_round_needed", 0) + 1
        self.stats.setdefault("overflow_log", []).append({
            "step": self.confirmed + 1, "flags": int(self.status[3]), "isects": n_isects, "longest_list": max_tile, "work_units": walk[0], "rows": walk[1],
            "capacities": [self.cap, self.cap_tile, self.cap_units, self.cap_rows]})
        for e in redo:   # steps fed from recycled device slots (HostFeed): upload their own inputs again
            src = e[9] if len(e) > 9 else None
            if src is not None:
                e[2], e[3] = src["w2c"].to(self.dev, torch.float32), src["K"].to(self.dev, torch.float32)
                e[4] = HostFeed.to_target(src["image"], self.dev)
                e[5] = src["mask"].to(self.dev, torch.float32) if (self.has_mask and src.get("mask") is not None) else e[5]
        first = redo[0]
        # (min_cap >= 1 forces a fresh probe even when only a coarse-bin capacity was exceeded and the list sizes read 0)
        self._build({"w2c": first[2], "K": first[3]}, first[4], first[5], min_cap=max(n_isects, 1), min_cap_tile=max_tile, min_walk=walk)
        for e in redo:
            self._issue(e)

    def finish(self):
        """Blocks until every issued step is known to be applied (replaying overflowed ones)."""
        while self.pending:
            self._poll(block=True)

    LOSS_RING = 4096   # steps of loss history kept on the device

    def loss_history(self, last_n: int) -> Tensor:
        """[last_n, 3] = {l1, 1-ssim, total} of the last `last_n` APPLIED steps since the latest (re-)build, oldest first
        (one device read; call after `finish()`).  The `loss3` tensor `step()` returns is only the latest launch's value and
        is meaningless for a step the guard skipped -- this log is written by applied steps only."""
        applied = int(self.buf["applied"].item())
        n = max(0, min(int(last_n), applied, self.LOSS_RING))
        idx = (torch.arange(applied - n, applied, device=self.dev)) % self.LOSS_RING
        return self.buf["loss_ring"][idx]

    def report(self) -> Dict[str, Any]:
        return dict(self.stats, binning=self.binning, probed_isects=self.probed[0], probed_longest_list=self.probed[1],
                    probed_coarse_entries=getattr(self, "probed_coarse", (0, 0))[0] if self.binning == "bins" else 0,
                    capacity_isects=self.cap, capacity_tile_list=self.cap_tile, capacity_work_units=self.cap_units,
                    capacity_rows=self.cap_rows, probed_work_units=self.probed_walk[0], probed_rows=self.probed_walk[1],
                    seen_work_units=self.seen_units, seen_rows=self.seen_rows, steps=self.confirmed,
                    graph=self.graph is not None, rounds=bool(self.rounds_on), round_fraction=self.round_fraction if self.rounds_on else None,
                    front_round_alone=bool(self.rounds_on and self.front_only))


class ViewParallelGraphStep(TrainStepGraph):
    """The view-parallel step (one view per rank, `distributed.ViewParallelStep`'s two collectives) with everything in front of
    the first collective as ONE replayable hipGraph:

        [graph: projection .. blend forward .. loss .. blend backward .. row sums + this rank's all-gather record + guard words]
        -> all-gather (async) || projection backward from the sums, gradients + statistics into the bucket
        -> all-reduce (async) || SH half of Adam from the gathered records  ->  geometry half of Adam + statistics  ->  status

    The four launches behind the graph and the two collectives are enqueued per step (they take the step's learning rates and
    step count as launch arguments).  Capacities, step guard and overflow recovery are `TrainStepGraph`'s -- with one addition:
    a rank that skips a step on the device must take every replica with it, and every replica must notice at the same step.
    So each rank's guard flag travels inside both collectives (`gs_guard_flag_out` -> one word of the record, one of the
    bucket) and is ORed into every rank's guard in front of the kernels that apply the step (`gs_guard_merge`); and the status
    words are polled BLOCKING every `check_every` steps, so that all ranks re-size, re-capture and replay the same steps in
    the same order (a rank that found the flag sixteen steps before its peers would issue other collectives than they).
    `vp`: a `ViewParallelStep(model, optimizer, force_exchange=..., guard_words=True)`; the model must hand raw parameters to the
    rasterizer (`fuse_activations`, the default)."""

    def __init__(self, model, optimizer, loss_computer, data, gt_img, mask=None, vp=None, **kw):
        if vp is None or not vp.exchange or not vp.native or not vp.guard_words:
            raise ValueError("ViewParallelGraphStep needs a native ViewParallelStep(..., guard_words=True) with the exchange on")
        if not vp._raw_parameters():
            raise ValueError("ViewParallelGraphStep: the model must pass its raw parameters (fuse_activations)")
        self.vp = vp
        kw["fuse_adam"] = False
        super().__init__(model, optimizer, loss_computer, data, gt_img, mask, **kw)

    def _alloc_lists(self):
        super()._alloc_lists()
        self.buf["row_sums"] = self._take("row_sums", (self.N, 12), torch.float32)
        self.vp._records(); self.vp._bucket()   # (the record and the bucket the graph writes into: made before the capture)

    def _enqueue_step(self):
        """The captured part: everything up to this rank's all-gather record."""
        L, b, m, vp = nat.lib(), self.buf, self.model, self.vp
        st = self._st()
        N, W, H = self.N, self.W, self.H
        send, _ = vp._records()
        _, offs, flat = vp._bucket()
        nat.check(L.gs_guard_set(_p(b["info"]), self.cap, self.cap_tile), "gs_guard_set")
        self._stage_no = 0
        try:
            self._enqueue_forward()
            lam = float(self.lc.lambda_ssim)
            self._ck(L.gs_l1_ssim_fwd_slots(st, H, W, lam, _p(b["render_colors"]), _p(b["img_slots"]), int(self.has_mask), 1, _p(b["loss_ws"]),
                                             _p(b["loss3"])), "gs_l1_ssim_fwd_slots")
            self._ck(L.gs_l1_ssim_bwd_slots(st, H, W, lam, _p(b["render_colors"]), _p(b["img_slots"]), 1, _p(b["loss_ws"]),
                                             _p(b["one"]), _p(b["v_render"])), "gs_l1_ssim_bwd_slots")
            self._ck(L.gs_blend_bwd(st, 1, W, H, _p(b["rec"]), _p(b["qlist"]), _p(b["qcnt"]), _p(b["unit_desc"]), self.cap_units,
                                     _p(b["ckpt"]), _p(b["qmask"]), _p(b["row_base"]), _p(b["walk_state"]),
                                     _p(b["render_colors"]), _p(b["render_alphas"]), _p(b["v_render"]), None, _p(b["rows"]), _p(b["unit_cls"])), "gs_blend_bwd")
            sp = send.data_ptr()
            self._ck(L.gs_row_sums(st, 1, N, _p(b["radii"]), _p(b["colors_post"]), self._tpg(), _p(b["cum_tiles"]), _p(b["rows"]),
                                    _p(b["row_base"]), _p(b["qmask"]), _p(b["row_sums"]), sp, sp + 4 * 3 * N, float(max(H, W)), _p(b["viewmats"]),
                                    sp + 4 * 4 * N), "gs_row_sums")
            # this rank's guard flag: into its record and into the bucket (both collectives carry it)
            self._ck(L.gs_guard_flag_out(st, _p(b["info"]), sp + 4 * (4 * N + 16), flat.data_ptr() + 4 * vp._flag_off), "gs_guard_flag_out")
        except TrainStepGraph._Stop:
            pass
        finally:
            L.gs_guard_set(None, 0, 0)
            L.gs_rounds_set(None, None, None, None, 0)

    def _tail(self, t: int, lrs):
        """Behind the graph: the two collectives and the four launches that apply the step (on the runner's stream)."""
        import torch.distributed as dist
        L, b, m, vp, opt = nat.lib(), self.buf, self.model, self.vp, self.opt
        st = self._st()
        N, W, H, world = self.N, self.W, self.H, vp.world
        P = vp.record_len()
        send, recv = vp._records()
        _, offs, flat = vp._bucket()
        seg = lambda i, n: flat.data_ptr() + 4 * offs[i]   # noqa: E731
        w_gather = dist.all_gather_into_tensor(recv, send, group=vp.group, async_op=True)
        vp.collectives += 1
        nat.check(L.gs_guard_set(_p(b["info"]), self.cap, self.cap_tile), "gs_guard_set")
        try:
            nat.check(L.gs_project_bwd(st, 1, N, self.K, int(m.active_sh_degree), _p(m.means), _p(m.quats), _p(m.log_scales), _p(m.sh_0),
                                       _p(m.sh_rest) if self.K > 1 else None, 0, _p(b["viewmats"]), _p(b["Ks"]), W, H, 0.3, 0.01, 1e10, _p(b["radii"]),
                                       _p(b["colors_post"]), self._tpg(), _p(b["cum_tiles"]), _p(b["rows"]), _p(b["row_base"]), _p(b["qmask"]),
                                       seg(0, 3 * N), seg(2, 4 * N), seg(1, 3 * N), seg(3, N), None, None, _p(b["v_abs"]), None, None, None, None,
                                       _p(m.logit_opacities), 1, _p(b["sh_jac"]), _p(b["row_sums"]), seg(4, N), seg(5, N)), "gs_project_bwd")
            w_sum = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=vp.group, async_op=True)
            vp.collectives += 1
            w_gather.wait()
            nat.check(L.gs_guard_merge(st, _p(b["info"]), recv.data_ptr() + 4 * (4 * N + 16), world, P), "gs_guard_merge")
            lr = dict(zip([g["name"] for g, _ in opt._plist], lrs))
            m0, v0 = opt.moments_of(m.sh_0)
            mr, vr = opt.moments_of(m.sh_rest) if self.K > 1 else (None, None)
            b1, b2 = opt.defaults["betas"]
            nat.check(L.gs_sh_adam_views(st, world, N, self.K, int(m.active_sh_degree), _p(m.means), recv.data_ptr(), P, _p(m.sh_0), _p(m0), _p(v0),
                                         _p(m.sh_rest) if self.K > 1 else None, _p(mr), _p(vr), float(lr["sh_0"]), float(lr["sh_rest"]), float(b1),
                                         float(b2), float(opt.defaults["eps"]), int(t), 1.0 / world, _p(m.max_radii)), "gs_sh_adam_views")
            w_sum.wait()
            nat.check(L.gs_guard_merge(st, _p(b["info"]), flat.data_ptr() + 4 * vp._flag_off, 1, 0), "gs_guard_merge")
            ns = len(opt._plist)
            grads = {"means": seg(0, 3 * N), "log_scales": seg(1, 3 * N), "quats": seg(2, 4 * N), "logit_opacities": seg(3, N)}
            gptr = (ct.c_void_p * ns)(*[grads.get(g["name"]) for g, _ in opt._plist])
            nat.check(L.gs_adam_step_stats(st, opt.flat_param.numel(), _p(opt.flat_param), _p(opt.exp_avg), _p(opt.exp_avg_sq), ns,
                                           (ct.c_int64 * ns)(*opt._ends), (ct.c_int64 * ns)(*opt._lens), gptr, (ct.c_float * ns)(*lrs), float(b1),
                                           float(b2), float(opt.defaults["eps"]), int(t), 1.0 / world, N, seg(4, N), seg(5, N),
                                           _p(m.grad_norm_accum), _p(m.collecting_counts)), "gs_adam_step_stats")
            nat.check(L.gs_step_applied(st, _p(b["info"]), _p(b["applied"])), "gs_step_applied")
            nat.check(L.gs_step_status(st, _p(b["info"]), _p(b["applied"]), self.status.data_ptr(), _p(b["loss3"]), _p(b["loss_ring"]),
                                       self.LOSS_RING, _p(b["walk_state"])), "gs_step_status")
        finally:
            L.gs_guard_set(None, 0, 0)
            L.gs_rounds_set(None, None, None, None, 0)

    def _poll(self, block: bool):
        # every rank must find an overflow at the SAME step: the status words are read behind a synchronisation, never early
        super()._poll(block=True)


class HostFeed:
    """The loop the way the reference feeds it (/root/reference/train.py:36-43 `DataLoader(pin_memory=True)`, :97
    `data_to_device`, scene/data_class.py:158-162): every step's camera, target image and mask come from PAGE-LOCKED HOST
    memory -- 33 MB per step at 1080p, 23 GB/s at 700 it/s.  The reference uploads them on the compute stream in front of the
    forward; here they cross PCIe on a copy stream into one of `n_slots` device slots while the previous step computes, and the
    step waits for exactly that upload (an event), not for the caller's stream:

        feed = HostFeed(runner)                      # runner: TrainStepGraph
        for batch in loader:                         # batch: {"w2c", "K", "width", "height", "image"[, "mask"]} pinned host tensors
            feed.step(batch)

    A slot is refilled once the step that read it last has run -- the copy stream waits for that (an event on the runner's
    stream), and so does the HOST: a feeder is never more than `n_slots` steps ahead of the device, a DataLoader's prefetch depth.
    (Unthrottled, the enqueueing thread ran hundreds of uploads ahead and stalled for 8-15 ms at a time inside the runtime's copy
    path: 645 it/s instead of 695 at the bench workload.)  A step the device SKIPPED behind a capacity overflow is replayed from
    a fresh upload of its own host tensors (`step(host_src=...)`), so recycling the slots never feeds a replay another step's
    image; the host tensors themselves must stay untouched until the step is retired (a DataLoader's pinned batches are: each is
    a new allocation).
    `image` may be uint8 [H, W, 3] (8 MB instead of 25 MB across the link): it is widened on the copy stream with the
    reference's own arithmetic, `float32(x) / 255` (a true division: bit-identical to `Frame.to_data`'s numpy expression).
    Measured and not kept (round 6, tools/host_feed_probe.py; 33 MB per step, the link 55 GB/s on an idle GPU): the batch pulled
    across by a KERNEL of 8 / 32 / 128 workgroups instead of the DMA engine -- 17-19 GB/s whatever its size, 1.70 / 1.94 / 2.00 ms
    per step; the image split over 2 / 4 / 8 copy streams -- 656 / 637 / 602 it/s against 680 on one.  A concurrent GEMM slows the
    same DMA batch from 0.64 to 7.1 ms: how fast the link is under load is the platform's business."""

    def __init__(self, runner: "TrainStepGraph", n_slots: int = 2, inputs_ready: bool = False):
        """`inputs_ready`: the caller's promise of `TrainStepGraph.step` (its stream holds no pending reader of the outputs, the
        parameters or the statistics) -- the uploads themselves are ordered by events either way."""
        self.r, self.n = runner, max(2, int(n_slots))
        self.inputs_ready = bool(inputs_ready)
        dev = runner.dev
        self.copy_stream = torch.cuda.Stream(dev)
        H, W = runner.H, runner.W
        f32 = dict(dtype=torch.float32, device=dev)
        self.gt = [torch.empty((H, W, 3), **f32) for _ in range(self.n)]
        self.u8 = [None] * self.n
        self.mask = [torch.empty((H, W), **f32) if runner.has_mask else None for _ in range(self.n)]
        self.w2c = [torch.empty((4, 4), **f32) for _ in range(self.n)]
        self.K = [torch.empty((3, 3), **f32) for _ in range(self.n)]
        self.done = [None] * self.n
        self._t255 = torch.full((), 255.0, **f32)
        self.i = 0

    @staticmethod
    def to_target(image: Tensor, dev) -> Tensor:
        """A host image as the loss kernels read it: float32 [H, W, 3] on the device (uint8: / 255 as the reference's loader)."""
        x = image.to(dev, non_blocking=False)
        if x.dtype == torch.uint8:
            return torch.div(x.to(torch.float32), torch.full((), 255.0, dtype=torch.float32, device=dev))
        return x.to(torch.float32)

    def step(self, batch: Dict[str, Any]):
        r, j = self.r, self.i % self.n
        self.i += 1
        if (int(batch["width"]), int(batch["height"])) != (r.W, r.H):
            raise ValueError("HostFeed: one frame size per feed (build another for another size)")
        cs = self.copy_stream
        if self.done[j] is not None:
            self.done[j].synchronize()    # (the host: never more than n_slots steps ahead)
            cs.wait_event(self.done[j])   # the step that read this slot last has run
        img = batch["image"]
        with torch.cuda.device(r.dev), torch.cuda.stream(cs):
            self.w2c[j].copy_(batch["w2c"], non_blocking=True)
            self.K[j].copy_(batch["K"], non_blocking=True)
            if img.dtype == torch.uint8:
                if self.u8[j] is None:
                    self.u8[j] = torch.empty(tuple(img.shape), dtype=torch.uint8, device=r.dev)
                self.u8[j].copy_(img, non_blocking=True)
                torch.div(self.u8[j].to(torch.float32), self._t255, out=self.gt[j])
            else:
                self.gt[j].copy_(img, non_blocking=True)
            if r.has_mask:
                self.mask[j].copy_(batch["mask"], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(cs)
        out = r.step({"w2c": self.w2c[j], "K": self.K[j], "width": r.W, "height": r.H}, self.gt[j], self.mask[j] if r.has_mask else None,
                     inputs_ready=self.inputs_ready, ready_event=ev, host_src=batch)
        self.done[j] = torch.cuda.Event()
        self.done[j].record(r.stream)
        return out
