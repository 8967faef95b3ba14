import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_gpu_rounds as T
from easy_gaussian_splatting_amd.loss import LossComputer
from easy_gaussian_splatting_amd.train_graph import TrainStepGraph
from easy_gaussian_splatting_amd import _native as nat
if os.environ.get('DBG_SYNC'):
    TrainStepGraph.debug_sync = True
    _ck0 = TrainStepGraph._ck
    def _ck(self, rc, what):
        print('stage', what, flush=True)
        _ck0(self, rc, what)
    TrainStepGraph._ck = _ck
kind = sys.argv[1] if len(sys.argv) > 1 else "sparse"
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.125
os.environ["GS_BINNING"] = sys.argv[3] if len(sys.argv) > 3 else "tiles"
dev, make, datas, gts = T._setup(kind)
(ma, oa), (mb, ob) = make(), make()
lc = LossComputer(0.2, clamp_input=True)
ra = TrainStepGraph(ma, oa, lc, datas[0], gts[0], use_graph=False, fuse_adam=False, rounds="off")
rb = TrainStepGraph(mb, ob, lc, datas[0], gts[0], use_graph=False, fuse_adam=False, rounds="on", round_fraction=frac)
def report(oa_, ob_):
  a, b = oa_["render_img"], ob_["render_img"]
  d = (a - b).abs().max(-1).values
  print("diff pixels", int((d > 0).sum()), "of", d.numel(), "max", float(d.max()))
  ys, xs = torch.nonzero(d > 0, as_tuple=True)
  tiles = set(zip((ys // 16).tolist(), (xs // 16).tolist()))
  print("tiles with diffs", len(tiles), sorted(tiles)[:10])
  blk = rb.buf["rounds"].tolist(); print("rounds blk", blk, "info a", ra.buf["info"].tolist(), "info b", rb.buf["info"].tolist())
  live = rb.buf["tile_live"].cpu().numpy().reshape(rb.th, rb.tw)
  print("live tiles", int(live.sum()), "of", live.size)
  print("diff tiles live?", [int(live[t]) for t in sorted(tiles)[:20]])
  da = (ra.buf["render_alphas"] - rb.buf["render_alphas"]).abs()
  print("alpha diff px", int((da > 0).sum()), float(da.max()))
  print("qcnt equal", torch.equal(ra.buf["qcnt"], rb.buf["qcnt"]), "walk", ra.buf["walk_state"][:6].tolist(), rb.buf["walk_state"][:6].tolist())
  for k, ga in ra.grads.items():
      if ga is not None:
          gb = rb.grads[k]
          print(k, float((ga - gb).abs().max()) / float(ga.abs().max()))

for v in [int(x) for x in (sys.argv[4] if len(sys.argv) > 4 else "0")]:
  oa_ = ra.step(datas[v], gts[v]); ra.finish()
  ob_ = rb.step(datas[v], gts[v]); rb.finish()
  torch.cuda.synchronize()
  report(oa_, ob_)
