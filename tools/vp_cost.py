#!/usr/bin/env python3
"""Single-GPU cost of the non-communication parts of distributed.ViewParallelStep at the bench size
(1M Gaussians, SH3, 8 views): gs_sh_grad_views, the pack copies, the two partial Adam launches."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
from easy_gaussian_splatting_amd.rendering import sh_grad_views
from easy_gaussian_splatting_amd.optim import FusedAdam

dev = torch.device("cuda:0")
N, R = 1_000_000, 8
g = torch.Generator(device="cuda").manual_seed(0)
means = torch.randn((N, 3), device=dev, generator=g)
cams = torch.eye(4, device=dev).repeat(R, 1, 1); cams[:, :3, 3] = torch.randn((R, 3), device=dev, generator=g) * 5
pre = torch.randn((R, N, 3), device=dev, generator=g)
shapes = {"means": (N, 3), "log_scales": (N, 3), "quats": (N, 4), "sh_0": (N, 1, 3), "sh_rest": (N, 15, 3), "logit_opacities": (N,)}
ps = {k: torch.nn.Parameter(torch.randn(s, device=dev, generator=g)) for k, s in shapes.items()}
opt = FusedAdam([{"params": [p], "lr": 1e-3, "name": k} for k, p in ps.items()])
grads = {k: torch.randn(s, device=dev, generator=g) for k, s in shapes.items()}


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def set_grads():
    for k, p in ps.items(): p.grad = grads[k]

set_grads()
print("sh_grad_views R=8      %.3f ms" % timeit(lambda: sh_grad_views(means, cams, pre, 3, 16)))
print("adam full              %.3f ms" % timeit(lambda: opt.step()))
print("adam SH only           %.3f ms" % timeit(lambda: opt.step(only=("sh_0", "sh_rest"), grad_scale=0.125)))
print("adam geometry only     %.3f ms" % timeit(lambda: opt.step(only=("means", "log_scales", "quats", "logit_opacities"), grad_scale=0.125, advance=False)))
geo = [grads[k] for k in ("means", "log_scales", "quats", "logit_opacities")] + [torch.randn(N, device=dev), torch.randn(N, device=dev)]
def pack():
    flat = torch.zeros(13 * N, device=dev)
    o = 0
    for t in geo:
        flat[o:o + t.numel()].copy_(t.reshape(-1)); o += t.numel()
    return flat
print("pack 13N floats        %.3f ms" % timeit(pack))

# the early colour-gradient kernel and the projection backward without its SH write-out, at the bench workload
from easy_gaussian_splatting_amd import rendering
from easy_gaussian_splatting_amd.rendering import rasterization
from easy_gaussian_splatting_amd.synthetic import config_bench_1m
import numpy as np
sc = config_bench_1m()
t = {k: torch.from_numpy(v).to(dev) for k, v in sc.items() if isinstance(v, np.ndarray)}
ins = [t[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities")]
sh0 = t["shs"][:, :1].contiguous().requires_grad_(True); shr = t["shs"][:, 1:].contiguous().requires_grad_(True)
for mode in ("dense", "colors_pre"):
    def it():
        img, _, meta = rasterization(*ins, (sh0, shr), t["viewmats"][:1], t["Ks"][:1], sc["width"], sc["height"], sh_degree=3,
                                     packed=False, backgrounds=t["backgrounds"][:1], absgrad=True, _sh_grads=mode)
        img.sum().backward()
    for _ in range(3): it()
    rendering.profile_stages(True)
    for _ in range(5): it()
    st = rendering.profile_stages(False)
    print(mode, ", ".join(f"{k[3:]}={np.mean(v):.3f}" for k, v in sorted(st.items()) if "bwd" in k or "pre" in k))
