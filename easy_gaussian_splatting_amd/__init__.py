"""MI355X-native Gaussian-splat rasterizer: the hot path behind
`gsplat.rendering.rasterization()` as called by li199603/easy_gaussian_splatting
(`model/gaussian.py:353-367`), written as hand-made gfx950 HIP kernels behind a C ABI.

Public surface (mirrors the reference's names for this path):
  rendering.rasterization   -- gsplat-signature drop-in (forward + autograd backward)
  model.GaussianModel       -- the reference's `forward(data)` / `update_statistics` harness
  loss.LossComputer         -- L1 + (1 - SSIM) as the reference's train step uses
  distributed               -- one-view-per-GPU gradient all-reduce over RCCL
"""
from .rendering import rasterization  # noqa: F401

__all__ = ["rasterization"]
__version__ = "0.1.0"
