#!/usr/bin/env python3
"""Per-step device time and host enqueue time of the host-fed captured loop at the bench workload (diagnostic for bench.py `host_fed`)."""
import json, os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
from config_run import LRS, MAKE, model_from_scene
from easy_gaussian_splatting_amd.loss import LossComputer
from easy_gaussian_splatting_amd.model import build_optimizers
from easy_gaussian_splatting_amd.train_graph import HostFeed, TrainStepGraph

dev = torch.device("cuda:0")
sc = MAKE["bench1M"]()
W, H = sc["width"], sc["height"]
model = model_from_scene(sc, dev)
opt = build_optimizers(model, *LRS, fused="hip")
data = {"w2c": torch.from_numpy(sc["viewmats"][0]).to(dev), "K": torch.from_numpy(sc["Ks"][0]).to(dev), "width": W, "height": H}
gt = torch.rand((H, W, 3), device=dev)
mask = torch.zeros((H, W), device=dev)
n_slots = int(sys.argv[1]) if len(sys.argv) > 1 else 2
handback = sys.argv[2] if len(sys.argv) > 2 else "eager"
runner = TrainStepGraph(model, opt, LossComputer(0.2, clamp_input=True), data, gt, mask, handback=handback)
pin = lambda t: t.detach().cpu().contiguous().pin_memory()
batches = [{"w2c": pin(data["w2c"]), "K": pin(data["K"]), "width": W, "height": H, "image": pin(torch.rand((H, W, 3))), "mask": pin(mask)} for _ in range(8)]
feed = HostFeed(runner, n_slots=n_slots)
for mode in ("fed", "resident"):
    for _ in range(20):
        feed.step(batches[0]) if mode == "fed" else runner.step(data, gt, mask)
    runner.finish(); torch.cuda.synchronize()
    n = 300
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    host = []
    evs[0].record(runner.stream)
    t_all = time.perf_counter()
    for i in range(n):
        t0 = time.perf_counter()
        feed.step(batches[i % 8]) if mode == "fed" else runner.step(data, gt, mask)
        host.append((time.perf_counter() - t0) * 1e6)
        evs[i + 1].record(runner.stream)
    runner.finish(); torch.cuda.synchronize()
    wall = time.perf_counter() - t_all
    ms = np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(n)])
    host = np.array(host)
    print(json.dumps({"mode": mode, "n_slots": n_slots, "handback": handback, "it_per_s": round(n / wall, 1), "dev_ms_median": round(float(np.median(ms)), 4),
                      "dev_ms_mean": round(float(ms.mean()), 4), "dev_ms_top8": [round(float(x), 3) for x in np.sort(ms)[-8:]],
                      "host_us_median": round(float(np.median(host)), 1), "host_us_mean": round(float(host.mean()), 1),
                      "host_us_top8": [round(float(x)) for x in np.sort(host)[-8:]]}))
