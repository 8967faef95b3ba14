import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_gpu_rounds as T
from easy_gaussian_splatting_amd.loss import LossComputer
from easy_gaussian_splatting_amd.train_graph import TrainStepGraph
TrainStepGraph.ROUNDS_MIN_LISTED = 1000
TrainStepGraph.ROUNDS_MIN_RATIO = 1.0
dev, make, datas, gts = T._setup("heavy")
for frac in (0.125, 0.25, 0.5, 0.75):
    m, o = make()
    r = TrainStepGraph(m, o, LossComputer(0.2, clamp_input=True), datas[0], gts[0], rounds="auto", round_fraction=frac)
    rep = r.report()
    print(frac, {k: rep.get(k) for k in ("rounds", "front_round_alone", "probed_live_tiles", "probed_isects", "probed_rows")}, r.tw * r.th)
print("---- speculation")
m, o = make()
r = TrainStepGraph(m, o, LossComputer(0.2, clamp_input=True), datas[0], gts[0], rounds="auto", round_fraction=0.5, check_every=4)
far = dict(datas[0]); w = far["w2c"].clone(); w[0, 3] += 2.5; far["w2c"] = w
for d in [datas[0], far, datas[0], far, datas[0], datas[0]]:
    r.step(d, gts[0])
r.finish()
rep = r.report()
print({k: rep.get(k) for k in ("rounds", "front_round_alone", "probed_live_tiles", "overflows", "replayed_steps", "back_round_needed", "steps", "binning")})
print(rep.get("overflow_log"))
