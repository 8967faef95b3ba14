#!/usr/bin/env python3
"""Captured train step under both binning pipelines on one mid-size scene: wall time per step, GPU time per replay."""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
from easy_gaussian_splatting_amd.loss import LossComputer
from easy_gaussian_splatting_amd.model import GaussianModel, build_optimizers
from easy_gaussian_splatting_amd.synthetic import make_scene
from easy_gaussian_splatting_amd.train_graph import TrainStepGraph
dev = torch.device("cuda:0")
N, W, H = 150_000, 800, 800
sc = make_scene(N, W, H, sh_degree=3, n_views=1, seed=3, scale_range=(0.01, 0.06), dist=5.0)
T = torch.from_numpy
op = np.clip(sc["opacities"], 1e-3, 1 - 1e-3)
def model():
    shs = T(sc["shs"])
    return GaussianModel(means=T(sc["means"]), log_scales=torch.log(T(sc["scales"])), quats=T(sc["quats"]), sh_0=shs[:, :1].contiguous(),
                         sh_rest=shs[:, 1:].contiguous(), logit_opacities=T(np.log(op / (1 - op)).astype(np.float32)), sh_degree=3,
                         white_background=True).to(dev)
data = {"w2c": T(sc["viewmats"][0]).to(dev), "K": T(sc["Ks"][0]).to(dev), "width": W, "height": H}
gt = torch.rand((H, W, 3), device=dev)
for mode in ("tiles", "bins", "tiles", "bins"):
    os.environ["GS_BINNING"] = mode
    m = model(); opt = build_optimizers(m, 1.6e-4, 5e-3, 1e-3, 2.5e-3, 1.25e-4, 5e-2, fused="hip")
    torch.cuda.synchronize(); tb = time.time()
    r = TrainStepGraph(m, opt, LossComputer(0.2, clamp_input=True), data, gt, None)
    torch.cuda.synchronize(); t_build = time.time() - tb
    tb = time.time(); r._build(data, gt, None, min_cap=1); torch.cuda.synchronize(); t_rebuild = time.time() - tb
    for _ in range(20):
        r.step(data, gt)
    r.finish(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.time(); e0.record(r.stream)
    for _ in range(200):
        r.step(data, gt)
    e1.record(r.stream); r.finish(); torch.cuda.synchronize()
    wall = (time.time() - t0) / 200 * 1e3
    print(f"{mode}: build {t_build * 1e3:.1f} ms, rebuild {t_rebuild * 1e3:.1f} ms, wall {wall:.3f} ms/step, stream {e0.elapsed_time(e1) / 200:.3f} ms/step, {r.report()}", flush=True)
