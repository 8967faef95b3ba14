#!/usr/bin/env python3
"""Pattern: build runner, one eager autograd step on a twin model, one replay, sync."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from test_gpu_train_graph import _setup, _eager_step
from easy_gaussian_splatting_amd.loss import LossComputer
from easy_gaussian_splatting_amd.train_graph import TrainStepGraph
dev, make, datas, gts = _setup()
(ma, oa), (mb, ob) = make(), make()
lc = LossComputer(0.2, clamp_input=True)
runner = TrainStepGraph(mb, ob, lc, datas[0], gts[0], None, use_graph=sys.argv[1] == "graph", check_every=2)
print("built", flush=True)
if len(sys.argv) > 2 and sys.argv[2] == "eager":
    _eager_step(ma, oa, lc, datas[0], gts[0], None)
    torch.cuda.synchronize(); print("eager ok", flush=True)
runner.step(datas[0], gts[0], None)
torch.cuda.synchronize(); print("replay ok", flush=True)
runner.step(datas[1], gts[1], None)
torch.cuda.synchronize(); print("replay 2 ok", flush=True)
