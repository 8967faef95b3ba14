"""View-parallel (one view per rank) step on the GPU: the factorised SH-gradient exchange.

* gs_sh_grad_views rebuilds, from the per-view pre-clamp colour gradients, exactly the SH gradients
  the dense path writes for the same cameras;
* FusedAdam's partial steps + folded gradient scale equal one ordinary step on scaled gradients;
* two `gloo` ranks sharing cuda:0 (RCCL refuses two ranks on one device; the exchange code is the
  same) take the same steps as one process that back-propagates both views.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from easy_gaussian_splatting_amd.rendering import rasterization, sh_grad_views
from scenes import make_scene

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.mark.parametrize("deg,split", [(3, True), (3, False), (2, True), (1, False), (0, True)])
def test_sh_grad_views_matches_dense_path(deg, split):
    dev = torch.device("cuda:0")
    sc = make_scene(2500, 176, 112, sh_degree=3, n_views=3, seed=31 + deg, scale_range=(0.03, 0.2), dist=4.0)
    t = {k: torch.from_numpy(v).to(dev) for k, v in sc.items() if isinstance(v, np.ndarray)}
    vc = torch.randn((3, 112, 176, 3), generator=torch.Generator().manual_seed(1)).to(dev)

    def run(mode):
        ins = [t[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities")]
        sh0 = t["shs"][:, :1].contiguous().requires_grad_(True)
        shr = t["shs"][:, 1:].contiguous().requires_grad_(True)
        shs = t["shs"].clone().requires_grad_(True)
        colors = (sh0, shr) if split else shs
        img, _, meta = rasterization(*ins, colors, t["viewmats"], t["Ks"], 176, 112, sh_degree=deg, packed=False,
                                     backgrounds=t["backgrounds"], absgrad=True, _sh_grads=mode)
        (img * vc).sum().backward()
        sh_grads = (sh0.grad, shr.grad) if split else (shs.grad,)
        return [p.grad for p in ins], sh_grads, meta

    g_dense, sh_dense, _ = run("dense")
    g_fact, sh_fact, meta = run("colors_pre")
    assert all(g is None for g in sh_fact), "factorised mode must not write SH gradients"
    for a, b in zip(g_fact, g_dense):   # geometry gradients (incl. the SH -> direction -> mean term) untouched
        assert torch.equal(a, b)
    pre = meta["means2d"].colors_pre_grad
    assert pre.shape == (3, 2500, 3)
    assert float(pre[meta["radii"] <= 0].abs().max()) == 0.0
    rebuilt = sh_grad_views(t["means"], t["viewmats"], pre, deg, 16, split=split)
    rebuilt = rebuilt if split else (rebuilt,)
    for a, b in zip(rebuilt, sh_dense):
        assert a.shape == b.shape
        assert _rel(a, b) < 2e-6, _rel(a, b)
    ka = (deg + 1) ** 2
    full = torch.cat(rebuilt, dim=1) if split else rebuilt[0]
    assert float(full[:, ka:].abs().max()) == 0.0 if ka < 16 else True


def test_fused_adam_partial_steps_and_grad_scale():
    from easy_gaussian_splatting_amd.optim import FusedAdam
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    shapes = {"means": (1001, 3), "log_scales": (1001, 3), "quats": (1001, 4), "sh_0": (1001, 1, 3), "sh_rest": (1001, 15, 3),
              "logit_opacities": (1001,)}
    init = {k: torch.randn(s, generator=g) for k, s in shapes.items()}
    grads = [{k: torch.randn(s, generator=g) for k, s in shapes.items()} for _ in range(3)]

    def make():
        ps = {k: torch.nn.Parameter(v.clone().to(dev)) for k, v in init.items()}
        return ps, FusedAdam([{"params": [p], "lr": 1e-2 * (i + 1), "name": k} for i, (k, p) in enumerate(ps.items())])

    pa, oa = make()
    pb, ob = make()
    sh, geo = ("sh_0", "sh_rest"), ("means", "log_scales", "quats", "logit_opacities")
    for gs in grads:
        for k in shapes:
            pa[k].grad = (0.25 * gs[k]).to(dev)
            pb[k].grad = gs[k].to(dev)
        oa.step()
        ob.step(only=sh, grad_scale=0.25)
        ob.step(only=geo, grad_scale=0.25, advance=False)
    assert oa._step == ob._step == 3
    for k in shapes:
        assert _rel(pb[k].detach(), pa[k].detach()) < 1e-6, k
        ma, va = oa.moments_of(pa[k]); mb, vb = ob.moments_of(pb[k])
        assert _rel(mb, ma) < 1e-6 and _rel(vb, va) < 1e-6, k


# ------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make(dev):
    from easy_gaussian_splatting_amd.model import GaussianModel, build_optimizers
    sc = make_scene(3000, 160, 112, sh_degree=3, n_views=2, seed=12, scale_range=(0.03, 0.15), dist=4.0)
    T = torch.from_numpy
    op = np.clip(sc["opacities"], 1e-3, 1 - 1e-3)
    shs = T(sc["shs"])
    model = GaussianModel(means=T(sc["means"]), log_scales=torch.log(T(sc["scales"])), quats=T(sc["quats"]),
                          sh_0=shs[:, :1].contiguous(), sh_rest=shs[:, 1:].contiguous(),
                          logit_opacities=T(np.log(op / (1 - op)).astype(np.float32)), sh_degree=3, white_background=True).to(dev)
    opt = build_optimizers(model, 1.6e-3, 5e-3, 1e-3, 2.5e-2, 1.25e-3, 5e-2, fused="hip")
    datas = [{"w2c": T(sc["viewmats"][v]).to(dev), "K": T(sc["Ks"][v]).to(dev), "width": 160, "height": 112} for v in range(2)]
    targets = torch.rand((2, 112, 160, 3), generator=torch.Generator().manual_seed(9)).to(dev)
    return model, opt, datas, targets


def _snapshot(model):
    out = {k: getattr(model, k).detach().cpu().numpy() for k in model.param_names}
    out.update(gn=model.grad_norm_accum.cpu().numpy(), cnt=model.collecting_counts.cpu().numpy(), rad=model.max_radii.cpu().numpy())
    return out


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from easy_gaussian_splatting_amd.distributed import ViewParallelStep
    from easy_gaussian_splatting_amd.loss import LossComputer
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    model, opt, datas, targets = _make(dev)
    vp = ViewParallelStep(model, opt)
    lc = LossComputer(0.2)
    for it in range(3):
        if it != 1:
            vp.begin_step(datas[rank])
        out = model(datas[rank])
        if it != 1:
            vp.after_forward(datas[rank], out)
        lc.get_loss_dict(out["render_img"], targets[rank])["total"].backward()
        assert model.sh_0.grad is None and model.sh_rest.grad is None
        vp.step(datas[rank], out)
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), **_snapshot(model))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_one_process_on_both_views(tmp_path):
    from easy_gaussian_splatting_amd.loss import LossComputer
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (np.load(os.path.join(tmp_path, f"r{r}.npz")) for r in range(2))
    for k in r0.files:
        np.testing.assert_array_equal(r0[k], r1[k], err_msg=f"replicas diverged in {k}")
    # one process: both views per step, gradients averaged by hand, ordinary dense SH gradients
    dev = torch.device("cuda:0")
    model, opt, datas, targets = _make(dev)
    lc = LossComputer(0.2)
    for _ in range(3):
        acc, stats = None, []
        for v in range(2):
            out = model(datas[v])
            lc.get_loss_dict(out["render_img"], targets[v])["total"].backward()
            radii = out["batch_radii"][0]
            vis = radii > 0
            stats.append((torch.where(vis, out["batch_xys"].absgrad[0].norm(dim=-1) * 160.0, 0.0), vis.float(),
                          torch.where(vis, radii.float() / 160.0, 0.0)))
            gs = [getattr(model, k).grad.clone() for k in model.param_names]
            acc = gs if acc is None else [a + g for a, g in zip(acc, gs)]
            opt.zero_grad()
        for k, g in zip(model.param_names, acc):
            getattr(model, k).grad = g / 2
        opt.step()
        opt.zero_grad()
        model.grad_norm_accum += stats[0][0] + stats[1][0]
        model.collecting_counts += stats[0][1] + stats[1][1]
        model.max_radii = torch.maximum(model.max_radii, torch.maximum(stats[0][2], stats[1][2]))
    ref = _snapshot(model)
    for k in model.param_names:
        # Adam normalises the step, so tiny gradient differences can move a parameter by up to ~lr;
        # compare against the step size (lr <= 2.5e-2, 3 steps) on all but a sliver of entries
        d = np.abs(r0[k] - ref[k])
        assert np.mean(d > 1e-5) < 2e-3, (k, float(d.max()), float(np.mean(d > 1e-5)))
    np.testing.assert_allclose(r0["gn"], ref["gn"], rtol=1e-4, atol=1e-6)
    np.testing.assert_array_equal(r0["cnt"], ref["cnt"])
    np.testing.assert_array_equal(r0["rad"], ref["rad"])


def test_sh_grad_views_argument_checks_and_empty_input():
    dev = torch.device("cuda:0")
    means = torch.zeros((0, 3), device=dev)
    vm = torch.eye(4, device=dev)[None]
    v0, vr = sh_grad_views(means, vm, torch.zeros((1, 0, 3), device=dev), 3, 16)
    assert v0.shape == (0, 1, 3) and vr.shape == (0, 15, 3)
    means = torch.randn((5, 3), device=dev)
    with pytest.raises(ValueError):   # more views than the kernel's camera table holds
        sh_grad_views(means, torch.eye(4, device=dev).repeat(65, 1, 1), torch.zeros((65, 5, 3), device=dev), 3, 16)
    with pytest.raises(ValueError):   # K too small for the degree
        sh_grad_views(means, vm, torch.zeros((1, 5, 3), device=dev), 3, 9)
    with pytest.raises(RuntimeError):  # no CPU fallback
        sh_grad_views(means.cpu(), vm.cpu(), torch.zeros((1, 5, 3)), 3, 16)
    # a view in which nothing is visible contributes exact zeros
    out = sh_grad_views(means, vm, torch.zeros((1, 5, 3), device=dev), 2, 16, split=False)
    assert out.shape == (5, 16, 3) and float(out.abs().max()) == 0.0


def test_factorised_mode_needs_sh_colours():
    dev = torch.device("cuda:0")
    sc = make_scene(50, 64, 48, sh_degree=1, n_views=1, seed=3, dist=4.0)
    t = {k: torch.from_numpy(v).to(dev) for k, v in sc.items() if isinstance(v, np.ndarray)}
    with pytest.raises(ValueError):
        rasterization(t["means"], t["quats"], t["scales"], t["opacities"], torch.rand((50, 3), device=dev), t["viewmats"],
                      t["Ks"], 64, 48, packed=False, _sh_grads="colors_pre")
