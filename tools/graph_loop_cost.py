#!/usr/bin/env python3
"""Per-step cost of the captured step in a REAL loop shape (camera and target change every step) against the static-input
bench loop and the eager step, at the soak scene (GPU box):   tools/graph_loop_cost.py [n_steps]"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
from easy_gaussian_splatting_amd.loss import LossComputer
from easy_gaussian_splatting_amd.model import GaussianModel, build_optimizers
from easy_gaussian_splatting_amd.synthetic import make_scene
from easy_gaussian_splatting_amd.train_graph import TrainStepGraph

n_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda:0")
N, W, H, V = 400_000, 800, 800, 6
sc = make_scene(N, W, H, sh_degree=3, n_views=V, seed=3, scale_range=(0.01, 0.06), dist=5.0)
T = torch.from_numpy
def make():
    op = np.clip(sc["opacities"], 1e-3, 1 - 1e-3); shs = T(sc["shs"])
    m = GaussianModel(means=T(sc["means"]), log_scales=torch.log(T(sc["scales"])), quats=T(sc["quats"]), sh_0=shs[:, :1].contiguous(),
                      sh_rest=shs[:, 1:].contiguous(), logit_opacities=T(np.log(op / (1 - op)).astype(np.float32)), sh_degree=3,
                      white_background=True).to(dev)
    return m, build_optimizers(m, 1.6e-4, 5e-3, 1e-3, 2.5e-3, 1.25e-4, 5e-2, fused="hip")
datas = [{"w2c": T(sc["viewmats"][v]).to(dev), "K": T(sc["Ks"][v]).to(dev), "width": W, "height": H} for v in range(V)]
targets = [torch.rand((H, W, 3), device=dev) for _ in range(V)]
lc = LossComputer(0.2, clamp_input=True)

def timed(fn, finish=None):
    for i in range(20): fn(i)
    if finish: finish()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n_steps): fn(i)
    if finish: finish()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n_steps

res = {}
m, o = make(); r = TrainStepGraph(m, o, lc, datas[0], targets[0], None)
res["graph, static inputs"] = timed(lambda i: r.step(), r.finish)
res["graph, camera + target change every step"] = timed(lambda i: r.step(datas[i % V], targets[i % V]), r.finish)
res["graph, the same with inputs_ready=True"] = timed(lambda i: r.step(datas[i % V], targets[i % V], inputs_ready=True), r.finish)
keep = []
res["graph, changing inputs + loss3.clone() per step"] = timed(lambda i: keep.append(r.step(datas[i % V], targets[i % V])["loss3"][2].clone()), r.finish)
print("runner:", r.report()); del r
m, o = make()
def eager(i):
    d = datas[i % V]
    out = m(d); loss = lc.get_loss_dict(out["render_img"], targets[i % V])["total"]; loss.backward()
    m.update_statistics(d, out); o.step(); o.zero_grad()
res["eager model step, changing inputs"] = timed(eager)
for k, v in res.items():
    print(f"{v:7.3f} ms/step  {k}")
