"""Import shim: lets the unmodified reference (`from gsplat.rendering import rasterization`,
/root/reference/model/gaussian.py:8) resolve to the MI355X rasterizer.  Put `shim/` first on PYTHONPATH."""
