#!/usr/bin/env python3
"""Can a memory-bound kernel (fused Adam over the flat buffers, 1.65 GB) hide under the latency-bound binning chain (or the
VALU-bound blend kernels) when the two run on different streams?  Wall time of A alone, B alone, A || B, eager and inside one
captured hipGraph with a forked branch.   tools/overlap_probe.py"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from easy_gaussian_splatting_amd import rendering
from easy_gaussian_splatting_amd.model import build_optimizers
dev = torch.device("cuda:0")
sc, model = bench.build_workload(1_000_000, 1, dev)
opt = build_optimizers(model, 1.6e-4, 5e-3, 1e-3, 2.5e-3, 1.25e-4, 5e-2, fused="hip")
for p in model.parameters():
    if p.requires_grad: p.grad = torch.randn_like(p) * 1e-6
t = {k: torch.from_numpy(v).to(dev) for k, v in sc.items() if isinstance(v, np.ndarray)}
s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)

def A():   # memory-bound: one fused Adam pass over params + moments (gradients kept: zero_grad not called)
    opt.step()

def B():   # forward only (projection + binning chain + sort + blend_fwd inference): latency- and VALU-bound
    with torch.no_grad():
        rendering.rasterization(model.means, model.quats, model.log_scales, model.logit_opacities, (model.sh_0, model.sh_rest), t["viewmats"], t["Ks"],
                                1920, 1080, sh_degree=3, packed=False, backgrounds=t["backgrounds"], _tile_culling="tight", _activations="exp_sigmoid")

def timed(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n

def both():
    s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s1): B()
    with torch.cuda.stream(s2): A()
    torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)

def seq():
    B(); A()

print(f"A (Adam) alone {timed(A):.3f} ms; B (forward) alone {timed(B):.3f} ms; sequential {timed(seq):.3f} ms; two streams {timed(both):.3f} ms")
# the same inside one captured graph
g_seq, g_par = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
cs = torch.cuda.Stream(dev)
B(); A(); torch.cuda.synchronize()
try:
    with torch.cuda.graph(g_seq, stream=cs):
        A()
        opt._step -= 0
        A()
    with torch.cuda.graph(g_par, stream=cs):
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            A()
        A()
        torch.cuda.current_stream().wait_stream(side)
    print(f"graph: A;A sequential {timed(g_seq.replay):.3f} ms, A || A forked {timed(g_par.replay):.3f} ms")
except Exception as e:
    print("graph capture failed:", repr(e)[:300])
