#!/usr/bin/env python3
"""Where the refinement's wall time goes inside the captured loop vs the eager loop (bench.py real_loop showed 9 ms vs 1 ms per call)."""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import bench
from easy_gaussian_splatting_amd.loss import LossComputer
from easy_gaussian_splatting_amd.model import build_optimizers
from easy_gaussian_splatting_amd.train_graph import TrainStepGraph
dev = torch.device("cuda:0")
sc, _ = bench.build_workload(1_000_000, 8, dev)
W, H = sc["width"], sc["height"]
datas = [{"w2c": torch.from_numpy(sc["viewmats"][v]).to(dev), "K": torch.from_numpy(sc["Ks"][v]).to(dev), "width": W, "height": H} for v in range(8)]
targets = [bench.smooth_target(H, W, 1234 + v, dev) for v in range(8)]
lrs = (1.6e-4, 5e-3, 1e-3, 2.5e-3, 1.25e-4, 5e-2)
for captured in (True, False, True):
    model = bench.model_from_scene(sc, dev); opt = build_optimizers(model, *lrs, fused="hip"); lc = LossComputer(0.2, clamp_input=True)
    gen = torch.Generator(device=dev).manual_seed(7)
    runner = TrainStepGraph(model, opt, lc, datas[0], targets[0], None, handback="lazy") if captured else None
    one = torch.ones((), device=dev)
    for it in range(1, 301):
        v = it % 8
        if runner is not None:
            runner.step(datas[v], targets[v])
        else:
            out = model(datas[v], clamp=False)
            lc.get_loss_dict(out["render_img"], targets[v], None)["total"].backward(gradient=one)
            model.update_statistics(datas[v], out); opt.step(); opt.zero_grad()
        if it % 100 == 0:
            if runner is not None: runner.finish()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            model.densify_and_prune(generator=gen)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            if runner is not None:
                runner.step(datas[0], targets[0]); t3 = time.perf_counter(); runner.finish(); torch.cuda.synchronize()
            else:
                t3 = t2
            t4 = time.perf_counter()
            print(f"captured={captured} it={it} N={model.nbr_gaussians} densify host {1e3*(t1-t0):.2f} ms, +sync {1e3*(t2-t1):.2f}, "
                  f"first step() (re-build) {1e3*(t3-t2):.2f}, finish {1e3*(t4-t3):.2f}; reserved {torch.cuda.memory_reserved()/2**30:.1f} GiB", flush=True)
    del runner, model, opt
    torch.cuda.empty_cache()
