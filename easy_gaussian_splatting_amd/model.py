"""Host-side mirror of the reference's model harness around the rasterization seam.

Only what the hot path needs (SURVEY.md section 8a rows a-1, a-2, a-3 and the optimizer the train
step drives); names and argument meaning follow /root/reference/model/gaussian.py:
  GaussianModel.scales / .opacities / .shs   <- :97-107   (exp / sigmoid / cat activations)
  GaussianModel.forward(data)                <- :351-374  (C=1 batching, clamp to [0,1])
  GaussianModel.update_statistics(...)       <- :188-197  (consumer of .absgrad and radii)
  build_optimizers(...)                      <- :389-412  (one Adam, six named groups)
Densification, pruning, checkpoint IO, loaders and the viewer are out of scope.
"""
from __future__ import annotations

from typing import Any, Dict, Optional

import torch
import torch.nn as nn
from torch import Tensor

from .distributed import all_reduce_statistics, is_distributed
from .rendering import rasterization


class GaussianModel(nn.Module):
    def __init__(self, means: Tensor, log_scales: Tensor, quats: Tensor, sh_0: Tensor, sh_rest: Tensor,
                 logit_opacities: Tensor, sh_degree: int, sh_degree_interval: int = 0,
                 white_background: bool = False, fuse_sh_cat: bool = True):
        super().__init__()
        self.means = nn.Parameter(means.float())  # [N, 3]
        self.log_scales = nn.Parameter(log_scales.float())  # [N, 3]
        self.quats = nn.Parameter(quats.float())  # [N, 4] wxyz
        self.sh_0 = nn.Parameter(sh_0.float())  # [N, 1, 3]
        self.sh_rest = nn.Parameter(sh_rest.float())  # [N, K-1, 3]
        self.logit_opacities = nn.Parameter(logit_opacities.float())  # [N]
        n = means.shape[0]
        self.register_buffer("grad_norm_accum", torch.zeros(n), persistent=False)
        self.register_buffer("collecting_counts", torch.zeros(n), persistent=False)
        self.register_buffer("max_radii", torch.zeros(n), persistent=False)
        self.optimizer: Optional[torch.optim.Optimizer] = None
        self.MAX_SH_DEGREE = sh_degree
        # hand sh_0 / sh_rest to the rasterizer separately instead of materialising `self.shs`
        # (saves the [N,K,3] cat and the split of its gradient every step; same values)
        self.fuse_sh_cat = fuse_sh_cat
        self.active_sh_degree = 0 if sh_degree_interval != 0 else sh_degree
        self.BACKGROUND = nn.Parameter(torch.full((3,), 1.0 if white_background else 0.0), requires_grad=False)

    @property
    def nbr_gaussians(self) -> int:
        return self.means.shape[0]

    @property
    def scales(self) -> Tensor:
        return torch.exp(self.log_scales)

    @property
    def opacities(self) -> Tensor:
        return torch.sigmoid(self.logit_opacities)

    @property
    def shs(self) -> Tensor:
        return torch.cat([self.sh_0, self.sh_rest], dim=1)

    @property
    def param_names(self):
        return ["means", "log_scales", "quats", "sh_0", "sh_rest", "logit_opacities"]

    def register_optimizer(self, optimizer: torch.optim.Optimizer):
        if self.optimizer is not None:
            raise RuntimeError("optimizer has been registered")
        self.optimizer = optimizer

    def up_sh_degree(self):
        self.active_sh_degree = min(self.active_sh_degree + 1, self.MAX_SH_DEGREE)

    def forward(self, data: Dict[str, Any]) -> Dict[str, Optional[Tensor]]:
        w2c = data["w2c"]
        batch_render_imgs, _, meta = rasterization(
            means=self.means,
            quats=self.quats,
            scales=self.scales,
            opacities=self.opacities,
            colors=(self.sh_0, self.sh_rest) if self.fuse_sh_cat else self.shs,
            sh_degree=self.active_sh_degree,
            viewmats=w2c[None],
            Ks=data["K"][None],
            width=data["width"],
            height=data["height"],
            backgrounds=self.BACKGROUND[None],
            absgrad=True,
            packed=False,
        )
        render_img = torch.clamp(batch_render_imgs[0], min=0.0, max=1.0)
        return {
            "render_img": render_img,  # [H, W, 3]
            "batch_xys": meta["means2d"],  # [1, N, 2]
            "batch_radii": meta["radii"],  # [1, N]
        }

    @torch.no_grad()
    def update_statistics(self, data: Dict[str, Any], model_output: Dict[str, Tensor]):
        max_hw = max(data["height"], data["width"])
        radii = model_output["batch_radii"].detach()[0] / max_hw
        xys_absgrad = model_output["batch_xys"].absgrad.detach()[0]
        visible = radii > 0.0
        # masked forms of the reference's three boolean-index updates (same values, no host sync)
        zero = torch.zeros_like(radii)
        step_radii = torch.where(visible, radii, zero)
        step_grads = torch.where(visible, torch.norm(xys_absgrad, dim=-1) * max_hw, zero)
        step_counts = visible.to(self.collecting_counts.dtype)
        if is_distributed():  # one view per rank: every replica must see every view's statistics
            all_reduce_statistics(step_grads, step_counts, step_radii)
        self.max_radii.copy_(torch.maximum(self.max_radii, step_radii))
        self.grad_norm_accum.add_(step_grads)
        self.collecting_counts.add_(step_counts)


def build_optimizers(model: GaussianModel, means_lr: float, log_scales_lr: float, quats_lr: float,
                     sh_0_lr: float, sh_rest_lr: float, logit_opacities_lr: float,
                     fused=None) -> torch.optim.Optimizer:
    """One Adam, six named groups (reference signature).  `fused`: None/False/True go to
    `torch.optim.Adam(fused=...)`; "hip" selects optim.FusedAdam (one HIP kernel per step over flat
    buffers, gradients cleared in the same pass)."""
    params = [
        {"params": [model.means], "lr": means_lr, "name": "means"},
        {"params": [model.log_scales], "lr": log_scales_lr, "name": "log_scales"},
        {"params": [model.quats], "lr": quats_lr, "name": "quats"},
        {"params": [model.sh_0], "lr": sh_0_lr, "name": "sh_0"},
        {"params": [model.sh_rest], "lr": sh_rest_lr, "name": "sh_rest"},
        {"params": [model.logit_opacities], "lr": logit_opacities_lr, "name": "logit_opacities"},
    ]
    if fused == "hip":
        from .optim import FusedAdam
        optimizer = FusedAdam(params)
    else:
        kw = {} if fused is None else {"fused": fused}
        optimizer = torch.optim.Adam(params, **kw)
    model.register_optimizer(optimizer)
    return optimizer
