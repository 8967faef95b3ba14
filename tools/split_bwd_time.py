#!/usr/bin/env python3
"""Feasibility timing: the captured step's last stage as ONE kernel (gs_project_bwd_adam) against the view-parallel step's four
kernels (gs_row_sums -> gs_project_bwd(row_sums) -> gs_sh_adam_views(R = 1) -> gs_adam_step_stats), sequential and with the SH
half of Adam on a second stream beside the projection backward + geometry Adam.  Buffers are those of a TrainStepGraph after one
eager step at the bench workload; the updates are applied over and over to the same gradients (timing only)."""
import ctypes as ct, json, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
from easy_gaussian_splatting_amd import _native as nat, synthetic as SY
from easy_gaussian_splatting_amd.loss import LossComputer
from easy_gaussian_splatting_amd.model import build_optimizers
from easy_gaussian_splatting_amd.train_graph import TrainStepGraph, _p
sys.path.insert(0, os.path.join(ROOT, "tools"))
from config_run import model_from_scene, LRS
dev = torch.device("cuda:0")
sc = SY.config_bench_1m()
W, H = int(sc["width"]), int(sc["height"])
model = model_from_scene(sc, dev)
opt = build_optimizers(model, *LRS, fused="hip")
lc = LossComputer(0.2, clamp_input=True)
data = {"w2c": torch.from_numpy(sc["viewmats"][0]).to(dev), "K": torch.from_numpy(sc["Ks"][0]).to(dev), "width": W, "height": H}
g = torch.Generator().manual_seed(7)
gt = torch.nn.functional.interpolate(torch.rand((1, 3, H // 16 + 1, W // 16 + 1), generator=g), size=(H, W), mode="bilinear")[0].permute(1, 2, 0).contiguous().to(dev)
r = TrainStepGraph(model, opt, lc, data, gt, None, use_graph=False)
for _ in range(3): r.step(data, gt, None)
r.finish()
L, b, m = nat.lib(), r.buf, model
N, K = r.N, r.K
f32 = dict(dtype=torch.float32, device=dev)
row_sums = torch.empty((N, 12), **f32); pay = torch.empty((4 * N + 16,), **f32)
gm, gq, gs_, go = torch.empty((N, 3), **f32), torch.empty((N, 4), **f32), torch.empty((N, 3), **f32), torch.empty((N,), **f32)
gn, cn = torch.empty((N,), **f32), torch.empty((N,), **f32)
b1, b2 = opt.defaults["betas"]; eps = float(opt.defaults["eps"])
offs = (ct.c_int64 * 6)(*opt._offs)
main = torch.cuda.Stream(dev); side = torch.cuda.Stream(dev)
def fused(st):
    nat.check(L.gs_project_bwd_adam(st, N, K, int(m.active_sh_degree), _p(opt.flat_param), _p(opt.exp_avg), _p(opt.exp_avg_sq), offs,
              _p(b["viewmats"]), _p(b["Ks"]), W, H, 0.3, 0.01, 1e10, _p(b["radii"]), _p(b["colors_post"]), _p(b["tiles_per_gauss"]), _p(b["cum_tiles"]),
              _p(b["rows"]), _p(b["row_base"]), _p(b["qmask"]), _p(b["v_abs"]), float(b1), float(b2), eps, _p(b["hyper"]), _p(b["applied"]), _p(m.max_radii),
              _p(m.grad_norm_accum), _p(m.collecting_counts), _p(b["sh_jac"])), "fused")
def rowsums(st):
    nat.check(L.gs_row_sums(st, 1, N, _p(b["radii"]), _p(b["colors_post"]), _p(b["tiles_per_gauss"]), _p(b["cum_tiles"]), _p(b["rows"]), _p(b["row_base"]), _p(b["qmask"]),
              _p(row_sums), _p(pay[:3 * N]), _p(pay[3 * N:4 * N]), float(max(W, H)), _p(b["viewmats"]), _p(pay[4 * N:])), "row_sums")
def geo(st):
    nat.check(L.gs_project_bwd(st, 1, N, K, int(m.active_sh_degree), _p(m.means), _p(m.quats), _p(m.log_scales), _p(m.sh_0), _p(m.sh_rest), 0,
              _p(b["viewmats"]), _p(b["Ks"]), W, H, 0.3, 0.01, 1e10, _p(b["radii"]), _p(b["colors_post"]), _p(b["tiles_per_gauss"]), _p(b["cum_tiles"]),
              None, None, _p(gm), _p(gq), _p(gs_), _p(go), None, None, _p(b["v_abs"]), None, None, None, None, _p(m.logit_opacities), 1,
              _p(b["sh_jac"]), _p(row_sums), _p(gn), _p(cn)), "project_bwd(sums)")
    names = [grp["name"] for grp, _ in opt._plist]
    ns = len(names)
    grads = {"means": gm, "quats": gq, "log_scales": gs_, "logit_opacities": go}
    ends = (ct.c_int64 * ns)(*opt._ends); lens = (ct.c_int64 * ns)(*opt._lens)
    gptr = (ct.c_void_p * ns)(*[_p(grads.get(n)) for n in names])
    lrs = (ct.c_float * ns)(*[float(grp["lr"]) for grp, _ in opt._plist])
    nat.check(L.gs_adam_step_stats(st, opt.flat_param.numel(), _p(opt.flat_param), _p(opt.exp_avg), _p(opt.exp_avg_sq), ns, ends, lens, gptr, lrs,
              float(b1), float(b2), eps, 5, 1.0, N, _p(gn), _p(cn), _p(m.grad_norm_accum), _p(m.collecting_counts)), "adam_step_stats")
def sh(st):
    m0, v0 = opt.moments_of(m.sh_0); mr, vr = opt.moments_of(m.sh_rest)
    nat.check(L.gs_sh_adam_views(st, 1, N, K, int(m.active_sh_degree), _p(m.means), _p(pay), 4 * N + 16, _p(m.sh_0), _p(m0), _p(v0), _p(m.sh_rest),
              _p(mr), _p(vr), 2.5e-3, 1.25e-4, float(b1), float(b2), eps, 5, 1.0, _p(m.max_radii)), "sh_adam_views")
def seq():
    st = main.cuda_stream
    rowsums(st); geo(st); sh(st)
def conc():
    st = main.cuda_stream
    rowsums(st)
    e = torch.cuda.Event(); e.record(main); side.wait_event(e)
    sh(side.cuda_stream)
    geo(st)
    e2 = torch.cuda.Event(); e2.record(side); main.wait_event(e2)
def conc2():   # the SH half first in the queue order
    st = main.cuda_stream
    rowsums(st)
    e = torch.cuda.Event(); e.record(main); side.wait_event(e)
    geo(side.cuda_stream)
    sh(st)
    e2 = torch.cuda.Event(); e2.record(side); main.wait_event(e2)
res = {}
for name, fn in (("fused_one_kernel", lambda: fused(main.cuda_stream)), ("split_sequential", seq), ("split_sh_on_second_stream", conc), ("split_geometry_on_second_stream", conc2),
                 ("row_sums_alone", lambda: rowsums(main.cuda_stream)), ("sh_adam_alone", lambda: sh(main.cuda_stream)), ("geometry_alone", lambda: geo(main.cuda_stream))):
    with torch.cuda.stream(main):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main)
        for _ in range(40): fn()
        e1.record(main); torch.cuda.synchronize()
    res[name] = round(e0.elapsed_time(e1) / 40, 4)
print(json.dumps(res))
