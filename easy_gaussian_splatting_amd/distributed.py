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
  camera) and 3 numbers per view.  So ranks all-gather `colors_pre_grad` (12 B per Gaussian and
  view, plus the 64-byte camera matrix) and rebuild the dense SH gradient locally (`gs_sh_grad_views`, views summed in rank order
  -> bitwise identical replicas); only the 11 geometry gradients + 2 statistics per Gaussian go
  through an all-reduce.  Per rank and step at SH3, 1M Gaussians, 8 ranks: 12 MB into an
  all-gather + 52 MB all-reduce (+ 4 MB MAX for the radii) instead of a 236 MB all-reduce, and
  the SH half of the Adam step runs while the all-reduce is still in flight.
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
    collective of the scheme (camera all-gather, radii MAX, colour-gradient all-gather from inside `backward()`,
    geometry all-reduce, split Adam), which is how the RCCL path is exercised on a one-GPU box."""

    SH = ("sh_0", "sh_rest")
    GEOMETRY = ("means", "log_scales", "quats", "logit_opacities")

    def __init__(self, model, optimizer, group=None, sh_grad_fn=None, force_exchange: bool = False):
        self.model, self.opt, self.group = model, optimizer, group
        self.world = dist.get_world_size(group) if group_ready() else 1
        if force_exchange and not group_ready():
            raise RuntimeError("ViewParallelStep(force_exchange=True) needs an initialised process group")
        self.exchange = self.world > 1 or bool(force_exchange)
        if self.exchange and not hasattr(optimizer, "moments_of"):
            raise TypeError("ViewParallelStep drives optim.FusedAdam (partial steps, folded 1/world scale)")
        if sh_grad_fn is None:
            from .rendering import sh_grad_views as sh_grad_fn
        self.sh_grad_fn = sh_grad_fn
        self._cams = self._rad = self._pre = None
        self.collectives = 0   # collectives issued so far (diagnostics / tests)
        self._flat = self._offs = None
        if self.exchange:   # started from inside backward(), as soon as the colour gradient exists
            model.on_colors_pre = self._gather_colors_pre
            if model.means.is_cuda and model.means.dtype == torch.float32:
                # the projection backward writes the four geometry gradients straight into the all-reduce bucket
                model.grad_out = self._grad_views
        model.sh_grads = "colors_pre" if self.exchange else "dense"

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
        return N, offs, off

    def _grad_views(self):
        """Called by the model's forward: the bucket's gradient segments as the rasterizer's `_grad_out` tensors (re-made
        when N changed: densify_and_prune)."""
        m = self.model
        N, offs, total = self._layout()
        if self._flat is None or self._flat.numel() != total or self._flat.device != m.means.device:
            self._flat = torch.zeros(total, dtype=torch.float32, device=m.means.device)   # (pads stay zero for good)
            self._offs = offs
        f = self._flat
        return {"means": f[offs[0]:offs[0] + 3 * N].view(N, 3), "scales": f[offs[1]:offs[1] + 3 * N].view(N, 3),
                "quats": f[offs[2]:offs[2] + 4 * N].view(N, 4), "opacities": f[offs[3]:offs[3] + N]}

    # Optional hooks that move the two small collectives off the end of the step (every rank must make
    # the same calls in the same order; `step` issues whatever was not issued before).
    def begin_step(self, data) -> None:
        """Before the forward: exchange the cameras (64 B per rank)."""
        if not self.exchange or self._cams is not None:
            return
        dt, dev = self.model.means.dtype, self.model.means.device
        cams = torch.empty(self.world * 16, dtype=dt, device=dev)
        work = dist.all_gather_into_tensor(cams, data["w2c"].to(dt).reshape(-1).contiguous(), group=self.group, async_op=True)
        self._cams = (cams, work)
        self.collectives += 1

    def after_forward(self, data, out) -> None:
        """After the forward: MAX all-reduce of the normalised radii, overlapped with loss + backward."""
        if not self.exchange or self._rad is not None:
            return
        dt = self.model.means.dtype
        max_hw = float(max(data["height"], data["width"]))
        radii = out["batch_radii"][0]
        visible = radii > 0
        rad = torch.where(visible, radii.to(dt) / max_hw, 0.0)
        work = dist.all_reduce(rad, op=dist.ReduceOp.MAX, group=self.group, async_op=True)
        self._rad = (rad, visible, work)
        self.collectives += 1

    def _gather_colors_pre(self, colors_pre_grad: Tensor) -> None:
        """All-gather of this view's pre-clamp colour gradient [1,N,3] (flat 1-D buffers: the layout
        every backend accepts).  Called by the rasterizer's backward between the blend backward and
        the projection backward, so the transfer overlaps the latter; `step` calls it otherwise."""
        if self._pre is not None:
            return
        mine = colors_pre_grad[0].reshape(-1).contiguous()
        pre_all = torch.empty(self.world * mine.numel(), dtype=mine.dtype, device=mine.device)
        work = dist.all_gather_into_tensor(pre_all, mine, group=self.group, async_op=True)
        self._pre = (pre_all, work)
        self.collectives += 1

    def step(self, data, out) -> None:
        m, opt = self.model, self.opt
        if not self.exchange:
            m.update_statistics(data, out)
            opt.step()
            opt.zero_grad()
            return
        world, group = self.world, self.group
        N = m.means.shape[0]
        xys = out["batch_xys"]
        dt = m.means.dtype   # float32 in the product; the CPU tests drive this class in float64
        f32 = dict(dtype=dt, device=m.means.device)
        max_hw = float(max(data["height"], data["width"]))
        # (0) the small collectives, unless the hooks issued them already
        self.begin_step(data)
        self.after_forward(data, out)
        (cams, w_cams), (rad, visible, w_max) = self._cams, self._rad
        self._cams = self._rad = None
        # (1) all-gather of every view's pre-clamp colour gradient (normally already in flight)
        self._gather_colors_pre(xys.colors_pre_grad)
        pre_all, w_gather = self._pre
        self._pre = None
        # (2) all-reduce SUM: geometry gradients + the two additive statistics of this view
        #     (/root/reference/model/gaussian.py:188-197), segments padded to 16 bytes
        geo = [getattr(m, name) for name in self.GEOMETRY]
        _, offs, off = self._layout()
        if dt == torch.float32 and m.means.is_cuda:
            from . import _native as nat
            in_place = self._flat is not None and self._flat.numel() == off and all(p.grad is None for p in geo)
            if in_place:
                # the projection backward wrote the four gradients into the bucket itself (`_grad_out`): one small pass
                # derives the two statistics segments
                flat, g = self._flat, [None] * 4
            else:   # (a model whose forward did not take the bucket: pack the gradients autograd holds)
                flat = torch.empty(off, **f32)
                g = [p.grad.contiguous() for p in geo]
            with torch.cuda.device(m.means.device):
                nat.check(nat.lib().gs_pack_view_step(
                    torch.cuda.current_stream(m.means.device).cuda_stream, N, max_hw, *[None if t is None else t.data_ptr() for t in g],
                    out["batch_radii"][0].contiguous().data_ptr(), xys.absgrad[0].contiguous().data_ptr(),
                    flat.data_ptr()), "gs_pack_view_step")
        else:   # the CPU tests drive this class with float64 tensors
            pieces = [p.grad for p in geo]
            pieces.append(torch.where(visible, torch.linalg.vector_norm(xys.absgrad[0], dim=-1) * max_hw, 0.0))
            pieces.append(visible.to(dt))
            flat = torch.zeros(off, **f32)
            for t, o in zip(pieces, offs):
                flat[o:o + t.numel()].copy_(t.reshape(-1))
        w_sum = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True)
        self.collectives += 1
        # SH half: rebuild the dense SH gradient of all views, update while (2) is in flight
        w_cams.wait()
        w_gather.wait()
        v0, vr = self.sh_grad_fn(m.means, cams.view(world, 4, 4), pre_all.view(world, N, 3), m.active_sh_degree,
                                 1 + m.sh_rest.shape[1])
        m.sh_0.grad, m.sh_rest.grad = v0, vr
        opt.step(only=self.SH, grad_scale=1.0 / world)
        # geometry half
        w_sum.wait()
        for p, o in zip(geo, offs):
            p.grad = flat[o:o + p.numel()].view_as(p)
        opt.step(only=self.GEOMETRY, grad_scale=1.0 / world, advance=False)
        if m.collecting_counts.dtype == dt:   # (one launch for both additive statistics)
            torch._foreach_add_([m.grad_norm_accum, m.collecting_counts], [flat[offs[4]:offs[4] + N], flat[offs[5]:offs[5] + N]])
        else:
            m.grad_norm_accum.add_(flat[offs[4]:offs[4] + N])
            m.collecting_counts.add_(flat[offs[5]:offs[5] + N].to(m.collecting_counts.dtype))
        w_max.wait()
        torch.maximum(m.max_radii, rad, out=m.max_radii)
        opt.zero_grad()


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
