#!/usr/bin/env python3
"""Runs the captured / guarded train step at the bench size with per-step (or per-stage) synchronisation and prints
the device-side list sizes: tools/debug_graph.py [steps] [graph|eager|sync]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from easy_gaussian_splatting_amd.loss import LossComputer
from easy_gaussian_splatting_amd.model import build_optimizers
from easy_gaussian_splatting_amd.train_graph import TrainStepGraph

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
mode = sys.argv[2] if len(sys.argv) > 2 else "graph"
dev = torch.device("cuda:0")
sc, model = bench.build_workload(1_000_000, 8, dev)
W, H = sc["width"], sc["height"]
data = {"w2c": torch.from_numpy(sc["viewmats"][0]).to(dev), "K": torch.from_numpy(sc["Ks"][0]).to(dev), "width": W, "height": H}
g = torch.Generator(device="cpu").manual_seed(1234)
gt = torch.rand((H // 8, W // 8, 3), generator=g).to(dev)
gt = torch.nn.functional.interpolate(gt.permute(2, 0, 1)[None], size=(H, W), mode="bilinear", align_corners=False)[0].permute(1, 2, 0).contiguous()
opt = build_optimizers(model, 1.6e-4, 5e-3, 1e-3, 2.5e-3, 1.25e-4, 5e-2, fused="hip")
r = TrainStepGraph(model, opt, LossComputer(0.2, clamp_input=True), data, gt, torch.zeros((H, W), device=dev), use_graph=mode == "graph")
r.debug_sync = mode == "sync"
knob = os.environ.get("KNOB", "")
if "nocheck" in knob:
    r.check_every = 10 ** 9
if "keep" in knob:
    grave = []
    orig_poll = r._poll
    def keep_poll(block):
        grave.extend(list(r.checks))
        return orig_poll(block)
    r._poll = keep_poll
if "noself" in knob:
    orig_set = r._set_inputs
    def set_inputs(w2c, K, gt, mask):
        if w2c.data_ptr() == r.buf["viewmats"].data_ptr():
            return
        return orig_set(w2c, K, gt, mask)
    r._set_inputs = set_inputs
print("built", r.report(), flush=True)
every = int(sys.argv[3]) if len(sys.argv) > 3 else 1
use_events = len(sys.argv) > 4 and sys.argv[4] == "events"
evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
stream = torch.cuda.current_stream(dev)
if use_events:
    evs[0].record(stream)
for it in range(steps):
    r.step()
    if use_events:
        evs[it + 1].record(stream)
    if every > 0 and (it + 1) % every == 0:
        torch.cuda.synchronize()
        info = r.buf["info"].tolist()
        print(it, info, int(r.buf["applied"].item()), int(r.buf["unit_counter"].item()), flush=True)
torch.cuda.synchronize()
variant = sys.argv[5] if len(sys.argv) > 5 else ""
if variant:
    import gc
    print("phase 2:", variant, flush=True)
    if "nogc" in variant:
        gc.disable()
    if "gc" == variant:
        gc.collect()
    if "create" in variant or "record" in variant:
        evs2 = [torch.cuda.Event(enable_timing=True) for _ in range(201)]
    if "plain" in variant:
        evs2 = [torch.cuda.Event() for _ in range(201)]
    if "record" in variant:
        evs2[0].record(stream)
    for it in range(100):
        r.step()
        if "record" in variant:
            evs2[it + 1].record(stream)
    torch.cuda.synchronize()
    print("phase 2 ok", flush=True)
r.finish()
print("done", r.report())
