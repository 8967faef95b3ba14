"""One-view-per-GPU data parallelism for the rasterization path.

The reference is single-GPU, one view per iteration (/root/reference/train.py:36-43); this is the
build's extension: every rank holds a full parameter replica, renders its own view, and the ranks
exchange ONE thing per step -- the sum of parameter gradients (plus the densification statistics
so `update_statistics` sees every view).  `backend="nccl"` is RCCL over xGMI on ROCm; the same
code runs on `gloo` for the CPU tests.

Bucket layout: all six gradients are flattened into ONE fp32 buffer (59 floats per Gaussian at
SH3) so a step issues a single large all-reduce instead of six small ones -- xGMI is
point-to-point and per-link bound, large messages are what saturates it.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
import torch.distributed as dist
from torch import Tensor


def is_distributed() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


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

    @classmethod
    def from_flat(cls, flat: Tensor, params: Iterable[torch.nn.Parameter]) -> "GradBucket":
        """Wraps a gradient buffer that already backs the parameters' `.grad` views
        (optim.FusedAdam owns it)."""
        self = cls.__new__(cls)
        self.params = list(params)
        self.flat = flat
        return self

    def zero_(self):
        self.flat.zero_()

    def all_reduce_mean(self, group=None, async_op: bool = False):
        if not is_distributed():
            return None
        world = dist.get_world_size(group)
        work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        if async_op:
            return work, world
        self.flat.div_(world)
        return None


def all_reduce_param_grads(params: Iterable[torch.nn.Parameter], group=None) -> None:
    """Mean of `.grad` over ranks without a staging bucket: one async all-reduce per parameter
    tensor, largest first (sh_rest is 76 % of the bytes at SH3), then one wait and the 1/world scale.
    Used with optim.FusedAdam, whose gradients are the rasterizer's own output tensors."""
    if not is_distributed():
        return
    world = dist.get_world_size(group)
    grads = sorted((p.grad for p in params if p.requires_grad and p.grad is not None), key=lambda g: -g.numel())
    works = [dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group, async_op=True) for g in grads]
    for w in works:
        w.wait()
    for g in grads:
        g.div_(world)


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
