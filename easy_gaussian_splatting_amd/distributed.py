"""One-view-per-GPU data parallelism for the rasterization path.

The reference is single-GPU, one view per iteration (/root/reference/train.py:36-43); this is the
build's extension: every rank holds a full parameter replica, renders its own view, and the ranks
exchange ONE thing per step -- the sum of parameter gradients (plus the densification statistics
so `update_statistics` sees every view).  `backend="nccl"` is RCCL over xGMI on ROCm; the same
code runs on `gloo` for the CPU tests.

Two exchange schemes:

* `ViewParallelStep` (default of bench.py for N > 1) -- xGMI is point-to-point and per-link bound, so
  the step is built around moving fewer bytes.  The SH coefficients are 48 of the 59 parameters
  of a Gaussian, but their gradient is an outer product  v_sh[k] = Y_k(dir) * v_colour_pre  of a
  basis every rank can evaluate itself (it has the means and, after a 64-byte exchange, every
  camera) and 3 numbers per view.  So ranks all-gather ONE record per view -- `colors_pre_grad` (12 B per Gaussian), the
  normalised radii (4 B per Gaussian: `update_statistics`' MAX) and the 64-byte camera matrix -- and apply the SH half of
  Adam straight from the gathered records (`gs_sh_adam_views`, views summed in rank order -> bitwise identical replicas;
  the dense SH gradient is never written); only the 11 geometry gradients + 2 statistics per Gaussian go
  through an all-reduce.  Per rank and step at SH3, 1M Gaussians, 8 ranks: 16 MB into an
  all-gather + 52 MB all-reduce instead of a 236 MB all-reduce + three statistics collectives -- two collectives per
  step --, and the SH half of the Adam step runs while the all-reduce is still in flight.
* `GradBucket` / `all_reduce_param_grads` -- the plain scheme (all six gradients all-reduced),
  kept for optimizers other than `optim.FusedAdam`.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
import torch.distributed as dist
from torch import Tensor


def is_distributed() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def group_ready() -> bool:
    """A process group exists (possibly of one rank)."""
    return dist.is_available() and dist.is_initialized()


class GradBucket:
    """Persistent flat gradient buffer: parameters' `.grad` become views into it, so autograd
    accumulates straight into the bucket and the all-reduce needs no pack/unpack copies."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        total = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(total, dtype=ref.dtype, device=ref.device)
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            off += n

    def zero_(self):
        self.flat.zero_()

    def all_reduce_mean(self, group=None, async_op: bool = False, force: bool = False):
        """`force`: issue the collective even in a one-rank group (exercises the transport: RCCL smoke test)."""
        if not (is_distributed() or (force and group_ready())):
            return None
        world = dist.get_world_size(group)
        work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        if async_op:
            return work, world
        self.flat.div_(world)
        return None


def all_reduce_param_grads(params: Iterable[torch.nn.Parameter], group=None, force: bool = False) -> None:
    """Mean of `.grad` over ranks without a staging bucket: one async all-reduce per parameter
    tensor, largest first (sh_rest is 76 % of the bytes at SH3), then one wait and the 1/world scale.
    Used with optim.FusedAdam, whose gradients are the rasterizer's own output tensors.  `force`: also in a one-rank group."""
    if not (is_distributed() or (force and group_ready())):
        return
    world = dist.get_world_size(group)
    grads = sorted((p.grad for p in params if p.requires_grad and p.grad is not None), key=lambda g: -g.numel())
    works = [dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group, async_op=True) for g in grads]
    for w in works:
        w.wait()
    for g in grads:
        g.div_(world)


class ViewParallelStep:
    """Exchange + parameter update of one training step with one view per rank.

        vp = ViewParallelStep(model, optimizer)          # optimizer: optim.FusedAdam
        vp.begin_step(data); out = model(data); vp.after_forward(data, out)     # (both hooks optional)
        loss.backward(); vp.step(data, out)

    replaces `model.update_statistics(data, out); optimizer.step(); optimizer.zero_grad()` of the
    reference loop (/root/reference/train.py:36-43, 57-58).  The update equals the single-process
    step on the batch of all ranks' views with a mean-over-views loss.  With one rank it is exactly
    the reference sequence -- unless `force_exchange` is set: then a one-rank group still goes through every
    collective of the scheme, which is how the RCCL path is exercised on a one-GPU box.

    TWO collectives per step (round 5; rounds 2-4: four):
      (A) all-gather of one record per view, [3N pre-clamp colour gradients | N radii / max(H, W) | 16 floats of w2c]
          -- everything `update_statistics`' MAX and the SH-gradient rebuild need from the other views.  On the GPU the
          record is filled by ONE launch right after the blend backward (`gs_row_sums`, via `_view_payload`) and the
          all-gather starts from inside `backward()`, in front of the projection backward it overlaps;
      (B) all-reduce SUM of [means 3N | log_scales 3N | quats 4N | logit_opacities N | grad_norm N | count N] -- on the GPU
          written in place by the projection backward (`_grad_out`), statistics segments included: no pack pass.
    Then the SH half of Adam straight from the gathered records (`gs_sh_adam_views`: the dense SH gradient is never written)
    while (B) is in flight, the geometry half of Adam and the two additive statistics after it."""

    SH = ("sh_0", "sh_rest")
    GEOMETRY = ("means", "log_scales", "quats", "logit_opacities")

    def __init__(self, model, optimizer, group=None, sh_grad_fn=None, force_exchange: bool = False, guard_words: bool = False):
        """`guard_words`: the all-gather record and the all-reduce bucket each carry one more word (16 bytes with its pad) -- the
        rank's step-guard flag, for the captured form of the step (train_graph.ViewParallelGraphStep): a rank that skips a step on
        the device must take every replica with it."""
        self.model, self.opt, self.group = model, optimizer, group
        self.guard_words = bool(guard_words)
        self.world = dist.get_world_size(group) if group_ready() else 1
        if force_exchange and not group_ready():
            raise RuntimeError("ViewParallelStep(force_exchange=True) needs an initialised process group")
        self.exchange = self.world > 1 or bool(force_exchange)
        if self.exchange and not hasattr(optimizer, "moments_of"):
            raise TypeError("ViewParallelStep drives optim.FusedAdam (partial steps, folded 1/world scale)")
        # native: the fused GPU path (payload filled by gs_row_sums, bucket written in place, gs_sh_adam_views).  A caller that
        # brings its own `sh_grad_fn` (the CPU tests: float64 tensors, the torch oracle as renderer) gets the same exchange
        # -- same records, same bucket, same two collectives -- spelled in torch ops.
        self.native = bool(self.exchange and sh_grad_fn is None and model.means.is_cuda and model.means.dtype == torch.float32
                           and hasattr(optimizer, "flat_param"))
        if sh_grad_fn is None:
            from .rendering import sh_grad_views as sh_grad_fn
        self.sh_grad_fn = sh_grad_fn
        self.collectives = 0   # collectives issued so far (diagnostics / tests)
        self._send = self._recv = self._flat = self._offs = self._go = None
        self._gather = None    # (recv buffer, work) of the all-gather in flight
        self._cam = self._rad = None
        if self.exchange:   # started from inside backward(), as soon as the colour gradient exists
            model.on_colors_pre = self._on_colors_pre
            if self.native:
                model.grad_out = self._grad_views          # geometry gradients + statistics straight into the bucket
                model.view_payload = self._payload_view    # the all-gather record straight into the send buffer
        model.sh_grads = "colors_pre" if self.exchange else "dense"

    # ------------------------------------------------------------------------------------------ buffers
    def _layout(self):
        """Segments of the SUM all-reduce bucket [means 3N | log_scales 3N | quats 4N | logit_opacities N | grad_norm N | count N],
        each padded to 16 bytes (the layout gs_pack_view_step fills)."""
        m = self.model
        N = m.means.shape[0]
        sizes = [3 * N, 3 * N, 4 * N, N, N, N]
        offs, off = [], 0
        for n_el in sizes:
            offs.append(off)
            off = (off + n_el + 3) // 4 * 4
        self._flag_off = off   # (guard_words: the bucket's last 4 floats; [0] = the sum of the ranks' guard flags)
        return N, offs, off + (4 if self.guard_words else 0)

    def _bucket(self):
        m = self.model
        N, offs, total = self._layout()
        if self._flat is None or self._flat.numel() != total or self._flat.device != m.means.device or self._flat.dtype != m.means.dtype:
            self._flat = torch.zeros(total, dtype=m.means.dtype, device=m.means.device)   # (pads stay zero for good)
        self._offs = offs
        return N, offs, self._flat

    def _records(self):
        """(send [P], recv [world * P]), P = 4 N + 16 (+ 4 with guard_words: [4 N + 16] = this rank's guard flag): one view's record
        of the all-gather (re-made when N changed)."""
        m = self.model
        P = self.record_len()
        if self._send is None or self._send.numel() != P or self._send.device != m.means.device or self._send.dtype != m.means.dtype:
            self._send = torch.zeros(P, dtype=m.means.dtype, device=m.means.device)
            self._recv = torch.empty(self.world * P, dtype=m.means.dtype, device=m.means.device)
        return self._send, self._recv

    def record_len(self) -> int:
        return 4 * self.model.means.shape[0] + 16 + (4 if self.guard_words else 0)

    def _raw_parameters(self) -> bool:
        """The model hands its RAW parameters to the rasterizer (exp / sigmoid inside the kernels): only then are the
        rasterizer's gradients the parameters' gradients, and only then may it write them into the bucket itself (with
        activated inputs autograd still has exp / sigmoid to go through: ADVICE r4)."""
        m = self.model
        return bool(m.means.is_cuda and getattr(m, "fuse_activations", True))

    def _grad_views(self):
        """Called by the model's forward: the bucket's segments as the rasterizer's `_grad_out` tensors (None when the model
        does not pass raw parameters).  The rasterizer's backward marks the dict `_written` when it has filled them."""
        self._go = None
        if not self._raw_parameters():
            return None
        N, offs, f = self._bucket()
        self._go = {"means": f[offs[0]:offs[0] + 3 * N].view(N, 3), "scales": f[offs[1]:offs[1] + 3 * N].view(N, 3),
                    "quats": f[offs[2]:offs[2] + 4 * N].view(N, 4), "opacities": f[offs[3]:offs[3] + N],
                    "grad_norm": f[offs[4]:offs[4] + N], "count": f[offs[5]:offs[5] + N]}
        return self._go

    def _payload_view(self):
        """Called by the model's forward: this rank's all-gather record as the rasterizer's `_view_payload`."""
        return self._records()[0]

    # ------------------------------------------------------------------------------------------ hooks
    # Optional hooks (every rank must make the same calls in the same order).  Since round 5 neither issues a collective:
    # the camera and the radii travel inside the one all-gather.  They let the torch-op path start that all-gather from
    # inside backward() (the GPU path gets both from the rasterizer's own launch and needs neither).
    def begin_step(self, data) -> None:
        """Before the forward: note this view's camera."""
        if self.exchange:
            self._cam = data["w2c"]

    def after_forward(self, data, out) -> None:
        """After the forward: note this view's normalised radii (torch-op path)."""
        if self.exchange and not self.native:
            self._rad = self._radii_norm(data, out)

    def _radii_norm(self, data, out) -> Tensor:
        dt = self.model.means.dtype
        radii = out["batch_radii"][0]
        return torch.where(radii > 0, radii.to(dt) / float(max(data["height"], data["width"])), 0.0)

    def _on_colors_pre(self, colors_pre_grad: Tensor) -> None:
        """Called by the rasterizer's backward between the row sums and the projection backward, so that the transfer
        overlaps the latter; `step` calls `_start_gather` otherwise."""
        send, _ = self._records()
        N = self.model.means.shape[0]
        if colors_pre_grad.data_ptr() == send.data_ptr():
            self._start_gather()                       # the rasterizer filled the whole record (`_view_payload`)
        elif self._cam is not None and self._rad is not None:
            self._fill_record(colors_pre_grad, self._rad, self._cam)
            self._start_gather()
        # (else: radii / camera not known yet -- `step` has them)

    def _fill_record(self, colors_pre_grad: Tensor, rad: Tensor, w2c: Tensor) -> None:
        send, _ = self._records()
        N = self.model.means.shape[0]
        send[:3 * N].copy_(colors_pre_grad[0].reshape(-1))
        send[3 * N:4 * N].copy_(rad)
        send[4 * N:4 * N + 16].copy_(w2c.to(send.dtype).reshape(-1))

    def _start_gather(self) -> None:
        if self._gather is not None:
            return
        send, recv = self._records()
        work = dist.all_gather_into_tensor(recv, send, group=self.group, async_op=True)
        self._gather = (recv, work)
        self.collectives += 1

    # ------------------------------------------------------------------------------------------ the step
    def step(self, data, out) -> None:
        m, opt = self.model, self.opt
        if not self.exchange:
            m.update_statistics(data, out)
            opt.step()
            opt.zero_grad()
            return
        world, group = self.world, self.group
        N = m.means.shape[0]
        P = self.record_len()
        xys = out["batch_xys"]
        dt = m.means.dtype   # float32 in the product; the CPU tests drive this class in float64
        max_hw = float(max(data["height"], data["width"]))
        # (A) the all-gather of the per-view records, unless backward() started it already
        if self._gather is None:
            self._fill_record(xys.colors_pre_grad, self._rad if self._rad is not None else self._radii_norm(data, out), data["w2c"])
            self._start_gather()
        recv, w_gather = self._gather
        self._gather = self._cam = self._rad = None
        # (B) all-reduce SUM: geometry gradients + the two additive statistics of this view
        #     (/root/reference/model/gaussian.py:188-197), segments padded to 16 bytes
        geo = [getattr(m, name) for name in self.GEOMETRY]
        _, offs, flat = self._bucket()
        go, self._go = self._go, None
        if go is not None and go.pop("_written", False):
            # the projection backward wrote the four gradients and the two statistics segments into the bucket itself; whatever
            # autograd holds besides (a regulariser on the raw parameters: `use_scale_regularization`) is added on top
            for p, o in zip(geo, offs):
                if p.grad is not None:
                    flat[o:o + p.numel()].add_(p.grad.reshape(-1))
                    p.grad = None
        elif dt == torch.float32 and m.means.is_cuda:
            # (a model whose forward did not take the bucket -- activated inputs, or no `grad_out` hook: pack what autograd holds)
            from . import _native as nat
            g = [p.grad.contiguous() for p in geo]
            with torch.cuda.device(m.means.device):
                nat.check(nat.lib().gs_pack_view_step(
                    torch.cuda.current_stream(m.means.device).cuda_stream, N, max_hw, *[t.data_ptr() for t in g],
                    out["batch_radii"][0].contiguous().data_ptr(), xys.absgrad[0].contiguous().data_ptr(),
                    flat.data_ptr()), "gs_pack_view_step")
        else:   # the CPU tests drive this class with float64 tensors
            visible = out["batch_radii"][0] > 0
            pieces = [p.grad for p in geo]
            pieces.append(torch.where(visible, torch.linalg.vector_norm(xys.absgrad[0], dim=-1) * max_hw, 0.0))
            pieces.append(visible.to(dt))
            for t, o in zip(pieces, offs):
                flat[o:o + t.numel()].copy_(t.reshape(-1))
        w_sum = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True)
        self.collectives += 1
        # SH half while (B) is in flight: the SH gradient of all views from the gathered records, views summed in rank order
        w_gather.wait()
        rec = recv.view(world, P)
        if self.native:
            self._sh_adam_native(recv, P)
        else:
            pre_all, rad_all, cams = rec[:, :3 * N].reshape(world, N, 3), rec[:, 3 * N:4 * N], rec[:, 4 * N:4 * N + 16].reshape(world, 4, 4)
            v0, vr = self.sh_grad_fn(m.means, cams.contiguous(), pre_all.contiguous(), m.active_sh_degree, 1 + m.sh_rest.shape[1])
            m.sh_0.grad, m.sh_rest.grad = v0, vr
            opt.step(only=self.SH, grad_scale=1.0 / world)
            torch.maximum(m.max_radii, rad_all.max(dim=0).values, out=m.max_radii)
        # geometry half
        w_sum.wait()
        for p, o in zip(geo, offs):
            p.grad = flat[o:o + p.numel()].view_as(p)
        gn, cn = flat[offs[4]:offs[4] + N], flat[offs[5]:offs[5] + N]
        if self.native and m.collecting_counts.dtype == dt and m.grad_norm_accum.is_contiguous() and m.collecting_counts.is_contiguous():
            # the geometry half of Adam and the two additive statistics in ONE launch (gs_adam_step_stats)
            opt.step(only=self.GEOMETRY, grad_scale=1.0 / world, advance=False, stats=(gn, cn, m.grad_norm_accum, m.collecting_counts))
            opt.zero_grad()
            return
        opt.step(only=self.GEOMETRY, grad_scale=1.0 / world, advance=False)
        if m.collecting_counts.dtype == dt:   # (one launch for both additive statistics)
            torch._foreach_add_([m.grad_norm_accum, m.collecting_counts], [flat[offs[4]:offs[4] + N], flat[offs[5]:offs[5] + N]])
        else:
            m.grad_norm_accum.add_(flat[offs[4]:offs[4] + N])
            m.collecting_counts.add_(flat[offs[5]:offs[5] + N].to(m.collecting_counts.dtype))
        opt.zero_grad()

    def _sh_adam_native(self, recv: Tensor, P: int) -> None:
        """`gs_sh_adam_views`: sh_0 / sh_rest and their moments updated in place from the gathered records (== `gs_sh_grad_views`
        + `opt.step(only=SH, grad_scale=1/world)` bit for bit), `max_radii` folded in; advances the optimizer's step count."""
        from . import _native as nat
        m, opt = self.model, self.opt
        opt._check_views()
        opt._step += 1
        lr = {g.get("name"): float(g["lr"]) for g in opt.param_groups}
        m0, v0 = opt.moments_of(m.sh_0)
        K = 1 + m.sh_rest.shape[1]
        mr, vr = opt.moments_of(m.sh_rest) if K > 1 else (None, None)
        b1, b2 = opt.defaults["betas"]
        ptr = lambda t: None if t is None else t.data_ptr()   # noqa: E731
        with torch.cuda.device(m.means.device):
            nat.check(nat.lib().gs_sh_adam_views(
                torch.cuda.current_stream(m.means.device).cuda_stream, self.world, m.means.shape[0], K, int(m.active_sh_degree),
                m.means.data_ptr(), recv.data_ptr(), P, m.sh_0.data_ptr(), m0.data_ptr(), v0.data_ptr(),
                ptr(m.sh_rest) if K > 1 else None, ptr(mr), ptr(vr), lr["sh_0"], lr["sh_rest"], float(b1), float(b2),
                float(opt.defaults["eps"]), int(opt._step), 1.0 / self.world, m.max_radii.data_ptr()), "gs_sh_adam_views")


def all_reduce_statistics(grad_norm: Tensor, counts: Tensor, max_radii: Tensor, group=None) -> None:
    """Sum / sum / max over ranks of the three per-Gaussian statistics of
    /root/reference/model/gaussian.py:188-197 so every replica takes identical densification
    decisions."""
    if not is_distributed():
        return
    packed = torch.stack([grad_norm, counts])
    dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
    grad_norm.copy_(packed[0])
    counts.copy_(packed[1])
    dist.all_reduce(max_radii, op=dist.ReduceOp.MAX, group=group)


def shard_views(n_views: int, rank: Optional[int] = None, world: Optional[int] = None) -> List[int]:
    """Views handled by this rank: round-robin, one view per GPU per step."""
    if rank is None:
        rank = dist.get_rank() if is_distributed() else 0
    if world is None:
        world = dist.get_world_size() if is_distributed() else 1
    return list(range(rank, n_views, world))


def assert_replicas_identical(t: Tensor, what: str = "tensor", group=None) -> None:
    """Cheap replica-consistency check (one 3-number all-reduce): every rank must hold the same shape and
    the same checksum of `t`; raises on all ranks otherwise.  Used after densify_and_prune, where a
    rank-dependent RNG draw would silently make the replicas diverge."""
    if not is_distributed():
        return
    x = t.detach().double()
    mine = torch.stack([torch.tensor(float(t.numel()), dtype=torch.float64, device=t.device), x.sum(), (x * x).sum()])
    lo, hi = mine.clone(), mine.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    if not torch.equal(lo, hi):
        raise RuntimeError(f"replicas diverged: {what} differs between ranks (numel/sum/sumsq min {lo.tolist()} max {hi.tolist()})")
