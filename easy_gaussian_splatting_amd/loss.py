"""L1 + (1 - SSIM) as the reference's train step uses it
(/root/reference/model/gaussian.py:415-453, lambda_ssim = 0.2 in configs/*.yaml:46), plus the optional
scale-ratio regulariser (/root/reference/model/gaussian.py:376-386, 437-440).

On the GPU (default) the loss is the fused HIP path of csrc/gs_loss.hip (`_FusedL1SSIM`: two stencil
kernels forward, one backward -- SURVEY.md section 8f-1).  The reference takes SSIM from torchmetrics
(`StructuralSimilarityIndexMeasure(data_range=1.0)`), which is not installed here; `ssim()` restates that
metric in plain torch (11x11 Gaussian window, sigma 1.5, reflect padding, K1=0.01, K2=0.03, mean over the
un-padded interior) and is what the CPU tests and `LossComputer(fused=False)` use.
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F
from torch import Tensor


def _gaussian_window(size: int, sigma: float, device, dtype) -> Tensor:
    x = torch.arange(size, device=device, dtype=dtype) - (size - 1) / 2.0
    g = torch.exp(-(x / sigma) ** 2 / 2)
    return g / g.sum()


def ssim(preds: Tensor, target: Tensor, data_range: float = 1.0, kernel_size: int = 11,
         sigma: float = 1.5, k1: float = 0.01, k2: float = 0.03) -> Tensor:
    """preds/target: [B, C, H, W] -> scalar mean SSIM."""
    c1, c2 = (k1 * data_range) ** 2, (k2 * data_range) ** 2
    ch = preds.shape[1]
    pad = (kernel_size - 1) // 2
    g = _gaussian_window(kernel_size, sigma, preds.device, preds.dtype)
    kern_h = g.view(1, 1, -1, 1).expand(ch * 5, 1, -1, 1).contiguous()
    kern_w = g.view(1, 1, 1, -1).expand(ch * 5, 1, 1, -1).contiguous()
    p = F.pad(preds, (pad, pad, pad, pad), mode="reflect")
    t = F.pad(target, (pad, pad, pad, pad), mode="reflect")
    stack = torch.cat([p, t, p * p, t * t, p * t], dim=1)  # [B, 5C, H+2p, W+2p]
    out = F.conv2d(F.conv2d(stack, kern_h, groups=ch * 5), kern_w, groups=ch * 5)
    mu_p, mu_t, e_pp, e_tt, e_pt = out.split(ch, dim=1)
    s_pp = e_pp - mu_p * mu_p
    s_tt = e_tt - mu_t * mu_t
    s_pt = e_pt - mu_p * mu_t
    num = (2 * mu_p * mu_t + c1) * (2 * s_pt + c2)
    den = (mu_p * mu_p + mu_t * mu_t + c1) * (s_pp + s_tt + c2)
    full = num / den
    full = full[..., pad:-pad, pad:-pad]
    return full.mean()


class _FusedL1SSIM(torch.autograd.Function):
    """HIP path (csrc/gs_loss.hip): two stencil kernels forward, one backward, no atomics."""

    @staticmethod
    def forward(ctx, render_img: Tensor, gt_img: Tensor, mask, lambda_ssim: float, clamp_input: bool = False):
        from . import _native as nat
        L = nat.lib()
        H, W = render_img.shape[:2]
        dev = render_img.device
        ctx.set_materialize_grads(False)   # no zero-filled gradients for the two value-only outputs
        r, g = render_img.contiguous(), gt_img.contiguous()
        m = None if mask is None else mask.contiguous()
        ws = torch.empty((int(L.gs_loss_workspace_floats(H, W)),), dtype=torch.float32, device=dev)
        out = torch.empty((3,), dtype=torch.float32, device=dev)
        st = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            nat.check(L.gs_l1_ssim_fwd(st, H, W, float(lambda_ssim), r.data_ptr(), g.data_ptr(),
                                       None if m is None else m.data_ptr(), int(clamp_input), ws.data_ptr(), out.data_ptr()),
                      "gs_l1_ssim_fwd")
        ctx.clamp_input = bool(clamp_input)
        ctx.save_for_backward(r, g, ws) if m is None else ctx.save_for_backward(r, g, ws, m)
        ctx.lam = float(lambda_ssim)
        total, l1, ssim_loss = out[2], out[0], out[1]
        ctx.mark_non_differentiable(l1, ssim_loss)
        return total, l1, ssim_loss

    @staticmethod
    def backward(ctx, v_total, _v_l1, _v_ssim):
        from . import _native as nat
        L = nat.lib()
        saved = ctx.saved_tensors
        if v_total is None:
            return None, None, None, None, None
        r, g, ws = saved[:3]
        m = saved[3] if len(saved) > 3 else None
        H, W = r.shape[:2]
        v_render = torch.empty_like(r)
        vt = v_total.contiguous().float()
        st = torch.cuda.current_stream(r.device).cuda_stream
        with torch.cuda.device(r.device):
            nat.check(L.gs_l1_ssim_bwd(st, H, W, ctx.lam, r.data_ptr(), g.data_ptr(), None if m is None else m.data_ptr(),
                                       int(ctx.clamp_input), ws.data_ptr(), vt.data_ptr(), v_render.data_ptr()), "gs_l1_ssim_bwd")
        return v_render, None, None, None, None


class LossComputer:
    """`clamp_input=True`: `render_img` is the model's UN-clamped image (`GaussianModel.forward(data, clamp=False)`)
    and `torch.clamp(., 0, 1)` of /root/reference/model/gaussian.py:368 happens inside the loss (both directions);
    the result equals clamp-then-loss, two full-resolution kernels fewer per step."""

    def __init__(self, lambda_ssim: float = 0.2, fused: bool = True, clamp_input: bool = False,
                 model=None, lambda_scale: float = 0.0):
        """`model` + `lambda_scale`: the reference's `LossComputer(model, lambda_ssim, lambda_scale)`
        (/root/reference/model/gaussian.py:415-419) -- when the model's `USE_SCALE_REGULARIZATION` is set,
        `lambda_scale * model.get_regularization_dict()["scale_reg"]` is added to the total (:437-440)."""
        self.lambda_ssim = lambda_ssim
        self.fused = fused
        self.clamp_input = clamp_input
        self.model = model
        self.lambda_scale = lambda_scale

    def _add_regularization(self, d: Dict[str, Tensor]) -> Dict[str, Tensor]:
        if self.model is not None:
            reg = self.model.get_regularization_dict()
            if "scale_reg" in reg:
                d["scale_reg"] = reg["scale_reg"]
                d["total"] = d["total"] + self.lambda_scale * reg["scale_reg"]
        return d

    def get_loss_dict(self, render_img: Tensor, gt_img: Tensor, mask: Tensor = None) -> Dict[str, Tensor]:
        if self.fused and render_img.is_cuda and render_img.dtype == torch.float32:
            total, l1, ssim_loss = _FusedL1SSIM.apply(render_img, gt_img, mask, self.lambda_ssim, self.clamp_input)
            return self._add_regularization({"l1": l1, "ssim": ssim_loss, "total": total})
        if self.clamp_input:
            render_img = torch.clamp(render_img, min=0.0, max=1.0)
        if mask is not None:
            m = mask.unsqueeze(2)
            render_img = m * gt_img + (1.0 - m) * render_img
        l1 = F.l1_loss(render_img, gt_img)
        r = render_img.permute(2, 0, 1)[None]
        g = gt_img.permute(2, 0, 1)[None]
        ssim_loss = 1.0 - ssim(g, r)
        total = (1.0 - self.lambda_ssim) * l1 + self.lambda_ssim * ssim_loss
        return self._add_regularization({"l1": l1, "ssim": ssim_loss, "total": total})
