#!/usr/bin/env python3
"""Soak run of the whole train loop (forward, fused loss, backward, statistics, fused Adam, densify/prune every 100
steps, opacity reset, means-LR schedule) at a mid-size synthetic scene: finite values, decreasing loss, bounded memory.
    tools/train_soak.py [eager|graph]     graph: every step through train_graph.TrainStepGraph (changing cameras,
                                          re-capture after each refinement, overflow replay if a view needs more room)"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
from easy_gaussian_splatting_amd.loss import LossComputer
from easy_gaussian_splatting_amd.model import GaussianModel, build_optimizers
from easy_gaussian_splatting_amd.synthetic import make_scene

dev = torch.device("cuda:0")
N, W, H, V = 150_000, 800, 800, 6
sc = make_scene(N, W, H, sh_degree=3, n_views=V, seed=3, scale_range=(0.01, 0.06), dist=5.0)
T = torch.from_numpy
def model_from(sc, noise, seed):
    g = torch.Generator().manual_seed(seed)
    op = np.clip(sc["opacities"], 1e-3, 1 - 1e-3); shs = T(sc["shs"])
    return GaussianModel(means=T(sc["means"]) + noise * 0.02 * torch.randn(sc["means"].shape, generator=g),
                         log_scales=torch.log(T(sc["scales"])) + noise * 0.2 * torch.randn(sc["scales"].shape, generator=g),
                         quats=T(sc["quats"]), sh_0=(shs[:, :1] + noise * 0.3 * torch.randn(shs[:, :1].shape, generator=g)).contiguous(),
                         sh_rest=(shs[:, 1:] * (1 - noise)).contiguous(), logit_opacities=T(np.log(op / (1 - op)).astype(np.float32)),
                         sh_degree=3, white_background=True).to(dev)
datas = [{"w2c": T(sc["viewmats"][v]).to(dev), "K": T(sc["Ks"][v]).to(dev), "width": W, "height": H} for v in range(V)]
with torch.no_grad():
    ref = model_from(sc, 0.0, 0)
    targets = [ref(d)["render_img"].clone() for d in datas]
del ref
model = model_from(sc, 1.0, 1)
opt = build_optimizers(model, 1.6e-4, 5e-3, 1e-3, 2.5e-3, 1.25e-4, 5e-2, fused="hip")
mode = sys.argv[1] if len(sys.argv) > 1 else "eager"
lc = LossComputer(0.2, clamp_input=(mode == "graph"))
runner = None
if mode == "graph":
    from easy_gaussian_splatting_amd.train_graph import TrainStepGraph
    runner = TrainStepGraph(model, opt, lc, datas[0], targets[0], None)
losses, t0 = [], time.time()
for it in range(1, 701):
    v = it % V
    model.update_learning_rate(it)
    if runner is not None:
        losses.append(runner.step(datas[v], targets[v])["loss3"][2].clone())
    else:
        out = model(datas[v])
        loss = lc.get_loss_dict(out["render_img"], targets[v])["total"]
        loss.backward()
        model.update_statistics(datas[v], out)
        opt.step(); opt.zero_grad()
        losses.append(loss.detach())
    if it % 100 == 0:
        if runner is not None:
            runner.finish()
        if it == 400:
            model.reset_opacities()
        else:
            model.densify_and_prune()
        torch.cuda.synchronize()
        l = (runner.loss_history(50)[:, 2] if runner is not None else torch.stack(losses[-50:])).mean().item()
        print(f"it {it}: N={model.nbr_gaussians} loss(mean last 50)={l:.5f} mem={torch.cuda.max_memory_allocated() / 2**30:.2f} GiB "
              f"{(time.time() - t0) / it * 1e3:.2f} ms/it", flush=True)
ls = torch.stack(losses).cpu().numpy()
assert np.isfinite(ls).all()
print("first/last 50:", ls[:50].mean(), ls[-50:].mean())
if runner is not None:
    runner.finish()
    print("runner:", runner.report())
