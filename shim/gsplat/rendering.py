"""`gsplat.rendering` stand-in exposing only what the reference imports."""
from easy_gaussian_splatting_amd.rendering import rasterization  # noqa: F401

__all__ = ["rasterization"]
