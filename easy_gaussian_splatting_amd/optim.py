"""Fused Adam for the reference's six parameter groups ("next" row).

`/root/reference/model/gaussian.py:389-412` builds ONE `torch.optim.Adam` with six named groups
(`means, log_scales, quats, sh_0, sh_rest, logit_opacities`), default betas/eps.  `FusedAdam`
keeps that interface (`param_groups[i]["name"]`, `["lr"]` -- what `update_learning_rate`,
`model/gaussian.py:121-128`, edits) but stores parameters, gradients and both moments in four flat
fp32 buffers, so one step is ONE HBM-streaming HIP kernel (csrc/gs_adam.hip) that can also clear
the gradients, and the gradient buffer doubles as the RCCL all-reduce bucket of `distributed.py`.
"""
from __future__ import annotations

import ctypes as ct
from typing import Dict, List

import torch

from .distributed import GradBucket


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params: List[Dict], betas=(0.9, 0.999), eps: float = 1e-8, zero_grad_in_step: bool = True):
        defaults = dict(lr=1e-3, betas=betas, eps=eps)
        super().__init__(params, defaults)
        self.zero_grad_in_step = zero_grad_in_step
        plist = [p for g in self.param_groups for p in g["params"]]
        if not plist:
            raise ValueError("no parameters")
        dev, dt = plist[0].device, plist[0].dtype
        if dt != torch.float32:
            raise TypeError("FusedAdam handles float32 parameters")
        # layout: groups back to back, each padded to a multiple of 4 elements (16-byte quads)
        self._ends: List[int] = []
        offs, off = [], 0
        for g in self.param_groups:
            for p in g["params"]:
                offs.append(off)
                off += p.numel()
            off = (off + 3) // 4 * 4
            self._ends.append(off)
        total = off
        self.flat_param = torch.zeros(total, dtype=dt, device=dev)
        self.flat_grad = torch.zeros(total, dtype=dt, device=dev)
        self.exp_avg = torch.zeros(total, dtype=dt, device=dev)
        self.exp_avg_sq = torch.zeros(total, dtype=dt, device=dev)
        with torch.no_grad():
            for p, o in zip(plist, offs):
                n = p.numel()
                self.flat_param[o:o + n].copy_(p.detach().reshape(-1))
                p.data = self.flat_param[o:o + n].view_as(p)       # parameters become views
                p.grad = self.flat_grad[o:o + n].view_as(p)        # autograd accumulates into the bucket
        self.bucket = GradBucket.from_flat(self.flat_grad, plist)
        self._step = 0

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError("closures are not supported")
        from . import _native as nat
        L = nat.lib()
        self._step += 1
        ng = len(self.param_groups)
        ends = (ct.c_int64 * ng)(*self._ends)
        lrs = (ct.c_float * ng)(*[float(g["lr"]) for g in self.param_groups])
        b1, b2 = self.defaults["betas"]
        st = torch.cuda.current_stream(self.flat_param.device).cuda_stream
        with torch.cuda.device(self.flat_param.device):
            nat.check(L.gs_adam_step(st, self.flat_param.numel(), self.flat_param.data_ptr(), self.flat_grad.data_ptr(),
                                     self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), ng, ends, lrs,
                                     float(b1), float(b2), float(self.defaults["eps"]), self._step,
                                     1 if self.zero_grad_in_step else 0), "gs_adam_step")

    def zero_grad(self, set_to_none: bool = False):
        # gradients are views into the flat bucket and must stay attached; the step already cleared
        # them when zero_grad_in_step is on
        if not self.zero_grad_in_step:
            self.flat_grad.zero_()
