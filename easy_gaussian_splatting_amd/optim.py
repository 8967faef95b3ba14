"""Fused Adam for the reference's six parameter groups ("next" row).

`/root/reference/model/gaussian.py:389-412` builds ONE `torch.optim.Adam` with six named groups
(`means, log_scales, quats, sh_0, sh_rest, logit_opacities`), default betas/eps.  `FusedAdam`
keeps that interface (`param_groups[i]["name"]`, `["lr"]` -- what `update_learning_rate`,
`model/gaussian.py:121-128`, edits) but stores parameters and both moments in three flat fp32
buffers, so one step is ONE HBM-streaming HIP kernel (csrc/gs_adam.hip).  Gradients stay exactly
where autograd put them: with `.grad` left `None` before `backward()` the rasterizer's own output
tensors become the `.grad`s without an accumulate pass, and `zero_grad()` just drops them.
"""
from __future__ import annotations

import ctypes as ct
from typing import Dict, List

import torch


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params: List[Dict], betas=(0.9, 0.999), eps: float = 1e-8):
        # torch.optim.Adam's hyper-parameter keys at their defaults, so a state_dict() from here loads into
        # torch.optim.Adam and vice versa (none of them is implemented: they must stay at these values)
        defaults = dict(lr=1e-3, betas=betas, eps=eps, weight_decay=0, amsgrad=False, maximize=False, foreach=None,
                        capturable=False, differentiable=False, fused=None, decoupled_weight_decay=False)
        super().__init__(params, defaults)
        for g in self.param_groups:   # one launch, one set of hyper-parameters (the reference uses Adam's defaults in every group)
            if tuple(g["betas"]) != tuple(betas) or g["eps"] != eps:
                raise NotImplementedError("FusedAdam: per-group betas / eps are not supported (only per-group lr)")
            if g["weight_decay"] or g["amsgrad"] or g["maximize"]:
                raise NotImplementedError("FusedAdam: weight_decay / amsgrad / maximize are not implemented")
        self._build()

    def _build(self, moments=None):
        self._plist = [(g, p) for g in self.param_groups for p in g["params"]]
        if not self._plist:
            raise ValueError("no parameters")
        if len(self._plist) > 8:
            raise NotImplementedError("FusedAdam handles up to 8 parameter tensors (the reference has 6)")
        dev, dt = self._plist[0][1].device, self._plist[0][1].dtype
        if dt != torch.float32:
            raise TypeError("FusedAdam handles float32 parameters")
        # layout: one segment per parameter tensor, each padded to a multiple of 4 elements
        self._ends: List[int] = []
        self._lens: List[int] = []
        offs, off = [], 0
        for _, p in self._plist:
            offs.append(off)
            self._lens.append(p.numel())
            off = (off + p.numel() + 3) // 4 * 4
            self._ends.append(off)
        self.flat_param = torch.zeros(off, dtype=dt, device=dev)
        self.exp_avg = torch.zeros(off, dtype=dt, device=dev)
        self.exp_avg_sq = torch.zeros(off, dtype=dt, device=dev)
        self._offs = offs
        with torch.no_grad():
            for i, ((_, p), o) in enumerate(zip(self._plist, offs)):
                n = p.numel()
                self.flat_param[o:o + n].copy_(p.detach().reshape(-1))
                if moments is not None:
                    self.exp_avg[o:o + n].copy_(moments[i][0].reshape(-1))
                    self.exp_avg_sq[o:o + n].copy_(moments[i][1].reshape(-1))
                p.data = self.flat_param[o:o + n].view_as(p)   # parameters become views of the flat buffer
                p.grad = None
        if not hasattr(self, "_step"):
            self._step = 0

    @staticmethod
    def flat_layout(numels):
        """(offsets, padded ends, total) of the flat buffers for tensors of `numels` elements: back to back, each
        padded to a multiple of 4 floats (what gs_adam_step expects)."""
        offs, ends, off = [], [], 0
        for n in numels:
            offs.append(off)
            off = (off + n + 3) // 4 * 4
            ends.append(off)
        return offs, ends, off

    def adopt_flat(self, flat_param, exp_avg, exp_avg_sq, new_params):
        """Takes over flat buffers that were filled elsewhere (model.densify_and_prune's device path, gs_refine_apply):
        `new_params` are the parameters, one per group in group order, already views of `flat_param` in
        `flat_layout` order.  The step count is kept, as the reference keeps Adam's `step` through its state surgery."""
        assert len(new_params) == len(self.param_groups)
        for group, p in zip(self.param_groups, new_params):
            group["params"] = [p]
        self._plist = [(g, p) for g in self.param_groups for p in g["params"]]
        self._lens = [p.numel() for _, p in self._plist]
        self._offs, self._ends, total = self.flat_layout(self._lens)
        assert flat_param.numel() == total == exp_avg.numel() == exp_avg_sq.numel()
        self.flat_param, self.exp_avg, self.exp_avg_sq = flat_param, exp_avg, exp_avg_sq
        for _, p in self._plist:
            p.grad = None
        self._check_views()

    def moments_of(self, param):
        """(exp_avg, exp_avg_sq) views shaped like `param`."""
        for i, (_, p) in enumerate(self._plist):
            if p is param:
                o, n = self._offs[i], p.numel()
                return self.exp_avg[o:o + n].view_as(p), self.exp_avg_sq[o:o + n].view_as(p)
        raise KeyError("parameter not owned by this optimizer")

    def replace_parameters(self, triples):
        """Densify / prune / opacity reset: new parameter tensors (one per group, in group order) with
        their moments; the flat buffers are rebuilt, the step count (bias correction) is kept, as
        the reference keeps Adam's `step` when it swaps a group's tensor."""
        assert len(triples) == len(self.param_groups)
        moments = []
        for group, (p, m, v) in zip(self.param_groups, triples):
            group["params"] = [p]
            moments.append((m.detach().clone(), v.detach().clone()))
        self._build(moments)

    # ---- checkpointing: torch.optim.Adam's layout (state[p] = {step, exp_avg, exp_avg_sq}), so a checkpoint
    # written by the reference's `optimizer.state_dict()` (/root/reference/utils.py:48-87) loads here and back
    def state_dict(self):
        for i, (_, p) in enumerate(self._plist):
            m, v = self.moments_of(p)
            self.state[p] = {"step": torch.tensor(float(self._step)), "exp_avg": m.clone(), "exp_avg_sq": v.clone()}
        try:
            return super().state_dict()
        finally:
            self.state.clear()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)   # validates the group structure, restores lr etc., fills self.state
        steps = set()
        with torch.no_grad():
            for _, p in self._plist:
                st = self.state.get(p)
                if not st:
                    continue
                m, v = self.moments_of(p)
                m.copy_(st["exp_avg"].reshape(m.shape))
                v.copy_(st["exp_avg_sq"].reshape(v.shape))
                steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError(f"FusedAdam keeps one step count; the checkpoint holds {sorted(steps)}")
        if steps:
            self._step = steps.pop()
        self.state.clear()

    def _check_views(self):
        """Parameters must still be the views of `flat_param` that `_build` made them (a later
        `model.to(...)` / `p.data = ...` silently detaches them from what `step` updates)."""
        base = self.flat_param.data_ptr()
        for (_, p), o in zip(self._plist, self._offs):
            if p.numel() and p.data_ptr() != base + 4 * o:   # (an empty tensor, e.g. sh_rest at SH degree 0, has no address)
                raise RuntimeError("FusedAdam: a parameter no longer aliases the flat buffer (moved or re-assigned "
                                   "after the optimizer was built); rebuild the optimizer")

    @torch.no_grad()
    def step(self, closure=None, *, only=None, grad_scale: float = 1.0, advance: bool = True, stats=None):
        """One Adam step.  `only`: iterable of group names -- update just those groups (the others are
        skipped like `grad is None`); with `advance=False` the step count is not incremented, so a
        step can be issued in two launches (`step(only=A)`, then `step(only=B, advance=False)`).
        `grad_scale` multiplies every gradient first (1/world: mean over ranks of summed gradients).
        `stats` = (src0, src1, dst0, dst1), float32 tensors of one length: `dst0 += src0; dst1 += src1` in the same launch
        (distributed.ViewParallelStep: the all-reduced statistics of `update_statistics`)."""
        if closure is not None:
            raise NotImplementedError("closures are not supported")
        from . import _native as nat
        L = nat.lib()
        self._check_views()
        if advance:
            self._step += 1
        ns = len(self._plist)
        grads = []
        for grp, p in self._plist:
            g = p.grad
            if only is not None and grp.get("name") not in only:
                g = None
            if g is not None and (not g.is_contiguous() or g.dtype != torch.float32):
                g = g.contiguous().float()
            grads.append(g)   # keep alive until the launch is queued
        ends = (ct.c_int64 * ns)(*self._ends)
        lens = (ct.c_int64 * ns)(*self._lens)
        gptr = (ct.c_void_p * ns)(*[None if g is None else g.data_ptr() for g in grads])
        lrs = (ct.c_float * ns)(*[float(grp["lr"]) for grp, _ in self._plist])
        b1, b2 = self.defaults["betas"]
        dev = self.flat_param.device
        st = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            if stats is not None:
                s0, s1, d0, d1 = stats
                if not all(t.is_contiguous() and t.dtype == torch.float32 and t.numel() == s0.numel() and t.device == dev for t in stats):
                    raise ValueError("stats: four contiguous float32 tensors of one length on the optimizer's device")
                nat.check(L.gs_adam_step_stats(st, self.flat_param.numel(), self.flat_param.data_ptr(), self.exp_avg.data_ptr(),
                                               self.exp_avg_sq.data_ptr(), ns, ends, lens, gptr, lrs, float(b1), float(b2),
                                               float(self.defaults["eps"]), self._step, float(grad_scale), s0.numel(), s0.data_ptr(),
                                               s1.data_ptr(), d0.data_ptr(), d1.data_ptr()), "gs_adam_step_stats")
            else:
                nat.check(L.gs_adam_step(st, self.flat_param.numel(), self.flat_param.data_ptr(), self.exp_avg.data_ptr(),
                                         self.exp_avg_sq.data_ptr(), ns, ends, lens, gptr, lrs, float(b1), float(b2),
                                         float(self.defaults["eps"]), self._step, float(grad_scale)), "gs_adam_step")

    def zero_grad(self, set_to_none: bool = True):
        for _, p in self._plist:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()
