#!/usr/bin/env python3
"""Device-side densify_and_prune at 1 M Gaussians: wall time, and device time of its kernels (rocprofv3 via tools/prof_cmd.sh)."""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
from easy_gaussian_splatting_amd.model import GaussianModel, build_optimizers
from easy_gaussian_splatting_amd.synthetic import config_bench_1m
dev = torch.device("cuda:0")
sc = config_bench_1m(); T = torch.from_numpy
op = np.clip(sc["opacities"], 1e-3, 1 - 1e-3)
for rep in range(3):
    shs = T(sc["shs"])
    m = GaussianModel(means=T(sc["means"]), log_scales=torch.log(T(sc["scales"])), quats=T(sc["quats"]), sh_0=shs[:, :1].contiguous(),
                      sh_rest=shs[:, 1:].contiguous(), logit_opacities=T(np.log(op / (1 - op)).astype(np.float32)), sh_degree=3,
                      white_background=True).to(dev)
    opt = build_optimizers(m, 1.6e-4, 5e-3, 1e-3, 2.5e-3, 1.25e-4, 5e-2, fused="hip")
    N = m.nbr_gaussians
    g = torch.Generator(device=dev).manual_seed(1)
    m.grad_norm_accum.copy_(torch.rand(N, device=dev, generator=g) * 4e-4)   # ~half above the 2e-4 threshold
    m.collecting_counts.fill_(1.0)
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.time(); e0.record()
    m.densify_and_prune(generator=torch.Generator(device=dev).manual_seed(2))
    e1.record(); torch.cuda.synchronize()
    print(f"rep {rep}: {N} -> {m.nbr_gaussians}  wall {(time.time() - t0) * 1e3:.2f} ms, stream {e0.elapsed_time(e1):.2f} ms", flush=True)
