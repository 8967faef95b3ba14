#!/usr/bin/env python3
"""Binning time (gs_bin_count + gs_bin_emit_sort stages) of the per-tile pipeline against the two-level one (2x2- and
4x4-tile bins) over scenes of growing mean footprint: where rendering.BINS_FROM_FOOTPRINT and bin_shift_for come from."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
from easy_gaussian_splatting_amd import rendering
from easy_gaussian_splatting_amd.synthetic import make_scene
dev = torch.device("cuda:0")
W, H = 1920, 1080
print(f"{'scene':34s} {'I/N':>7s} {'I':>10s} | {'tiles':>8s} {'bins 2x2':>9s} {'bins 4x4':>9s}   (ms, count + lists)")
for n, lo, hi in ((1_000_000, 0.003, 0.02), (1_000_000, 0.005, 0.04), (600_000, 0.005, 0.08), (400_000, 0.01, 0.12),
                  (300_000, 0.01, 0.2), (200_000, 0.02, 0.3), (2_000_000, 0.003, 0.03), (2_000_000, 0.005, 0.08)):
    sc = make_scene(n, W, H, sh_degree=0, seed=5, extent=(4, 2.25, 4), scale_range=(lo, hi), dist=8.0)
    t = {k: torch.from_numpy(v).to(dev) for k, v in sc.items() if isinstance(v, np.ndarray)}
    res = {}
    for mode, shift in (("tiles", ""), ("bins", "1"), ("bins", "2")):
        os.environ["GS_BINNING"], os.environ["GS_BINS_SHIFT"] = mode, shift
        with torch.no_grad():
            for it in range(7):
                if it == 2:
                    rendering.profile_stages(True)
                _, _, meta = rendering.rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["shs"], t["viewmats"], t["Ks"], W, H,
                                                     sh_degree=0, packed=False, backgrounds=t["backgrounds"], _tile_culling="tight")
        st = rendering.profile_stages(False)
        res[(mode, shift)] = float(np.mean(st["gs_bin_count"]) + np.mean(st["gs_bin_emit_sort"]))
    I = meta["flatten_ids"].numel()
    print(f"{n:>9d} scales {lo}-{hi:<12} {I / n:7.1f} {I:10d} | {res[('tiles', '')]:8.3f} {res[('bins', '1')]:9.3f} {res[('bins', '2')]:9.3f}", flush=True)
    del t
    torch.cuda.empty_cache()
