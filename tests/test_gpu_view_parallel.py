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
    for a, b in zip(g_fact, g_dense):
        # geometry gradients (incl. the SH -> direction -> mean term): the same row sums bit for bit (gs_row_sums runs the very
        # statements of the projection backward's first phase), pushed through ANOTHER instantiation of the projection backward
        # -- under -ffp-contract=fast the two may fuse a multiply-add differently: equal to an ulp or two, not bitwise
        assert _rel(a, b) < 1e-6, _rel(a, b)
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
    # two additive statistics ride along in the geometry launch (gs_adam_step_stats): dst += src, whatever the segments do
    src = [torch.rand(1001, generator=g).to(dev) for _ in range(2)]
    dst = [torch.rand(1001, generator=g).to(dev) for _ in range(2)]
    want = [d.clone() for d in dst]
    for gs in grads:
        for k in shapes:
            pa[k].grad = (0.25 * gs[k]).to(dev)
            pb[k].grad = gs[k].to(dev)
        oa.step()
        ob.step(only=sh, grad_scale=0.25)
        ob.step(only=geo, grad_scale=0.25, advance=False, stats=(src[0], src[1], dst[0], dst[1]))
        want = [w + s_ for w, s_ in zip(want, src)]
    assert oa._step == ob._step == 3
    assert torch.equal(dst[0], want[0]) and torch.equal(dst[1], want[1])
    with pytest.raises(ValueError):
        ob.step(only=geo, advance=False, stats=(src[0], src[1], dst[0], dst[1][:5]))
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


_REFINE_KW = dict(densify_grad_thresh=2e-5, densify_scale_thresh=0.06, prune_scale_thresh=0.5, min_opacity=0.02)


def _make(dev, variant="plain"):
    from easy_gaussian_splatting_amd.model import GaussianModel, build_optimizers
    sc = make_scene(3000, 160, 112, sh_degree=3, n_views=2, seed=12, scale_range=(0.03, 0.15), dist=4.0)
    T = torch.from_numpy
    op = np.clip(sc["opacities"], 1e-3, 1 - 1e-3)
    shs = T(sc["shs"])
    model = GaussianModel(means=T(sc["means"]), log_scales=torch.log(T(sc["scales"])), quats=T(sc["quats"]),
                          sh_0=shs[:, :1].contiguous(), sh_rest=shs[:, 1:].contiguous(),
                          logit_opacities=T(np.log(op / (1 - op)).astype(np.float32)), sh_degree=3, white_background=True,
                          use_scale_regularization=variant == "regularised", max_scale_ratio=1.5,
                          **(_REFINE_KW if variant == "refine" else {})).to(dev)
    model.fuse_activations = variant != "activated"
    opt = build_optimizers(model, 1.6e-3, 5e-3, 1e-3, 2.5e-2, 1.25e-3, 5e-2, fused="hip")
    datas = [{"w2c": T(sc["viewmats"][v]).to(dev), "K": T(sc["Ks"][v]).to(dev), "width": 160, "height": 112} for v in range(2)]
    targets = torch.rand((2, 112, 160, 3), generator=torch.Generator().manual_seed(9)).to(dev)
    return model, opt, datas, targets


def _snapshot(model):
    out = {k: getattr(model, k).detach().cpu().numpy() for k in model.param_names}
    out.update(gn=model.grad_norm_accum.cpu().numpy(), cnt=model.collecting_counts.cpu().numpy(), rad=model.max_radii.cpu().numpy())
    return out


def _worker(rank, world, port, out_dir, variant="plain"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from easy_gaussian_splatting_amd.distributed import ViewParallelStep
    from easy_gaussian_splatting_amd.loss import LossComputer
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    model, opt, datas, targets = _make(dev, variant)
    vp = ViewParallelStep(model, opt)
    assert vp.native
    lc = LossComputer(0.2, model=model, lambda_scale=0.1)
    ns = [model.nbr_gaussians]

    def steps(its):
        for it in its:
            if it != 1:
                vp.begin_step(datas[rank])
            out = model(datas[rank])
            if it != 1:
                vp.after_forward(datas[rank], out)
            lc.get_loss_dict(out["render_img"], targets[rank])["total"].backward()
            assert model.sh_0.grad is None and model.sh_rest.grad is None
            if variant != "activated":   # the projection backward wrote the geometry gradients into the all-reduce bucket
                assert model.means.grad is None and model.quats.grad is None
            assert (model.log_scales.grad is not None) == (variant in ("activated", "regularised"))
            vp.step(datas[rank], out)

    steps(range(3))
    if variant == "refine":
        # a different generator state on every rank: the split noise must be rank 0's (model._split_noise broadcast)
        info = model.densify_and_prune(generator=torch.Generator(device=dev).manual_seed(100 + rank))
        assert info["train/densify"]["split"] > 0 and info["train/densify"]["clone"] > 0, info
        ns.append(model.nbr_gaussians)
        model.reset_opacities()
        steps(range(2))
    assert vp.collectives == 2 * (5 if variant == "refine" else 3)
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), ns=np.asarray(ns), **_snapshot(model))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("variant", ["plain", "regularised", "activated", "refine"])
def test_two_ranks_equal_one_process_on_both_views(tmp_path, variant):
    """"regularised": `use_scale_regularization` -- autograd holds a gradient for `log_scales` (the regulariser's) while the
    render's went into the bucket: both must reach Adam.  "activated": the model passes exp / sigmoid OUTPUTS to the
    rasterizer (`fuse_activations=False`): the bucket must then be packed from autograd's gradients w.r.t. the raw parameters,
    not written by the rasterizer.  "refine": 3 steps -> densify_and_prune -> reset_opacities -> 2 steps (VERDICT r4
    missing #1a): replicas bitwise equal, the trajectory of N that of one process on both views."""
    from easy_gaussian_splatting_amd.loss import LossComputer
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), variant), nprocs=2, join=True)
    r0, r1 = (np.load(os.path.join(tmp_path, f"r{r}.npz")) for r in range(2))
    for k in r0.files:
        np.testing.assert_array_equal(r0[k], r1[k], err_msg=f"replicas diverged in {k}")
    # one process: both views per step, gradients averaged by hand, ordinary dense SH gradients
    dev = torch.device("cuda:0")
    model, opt, datas, targets = _make(dev, variant)
    lc = LossComputer(0.2, model=model, lambda_scale=0.1)
    ns = [model.nbr_gaussians]
    n_steps = 0
    for phase in range(2 if variant == "refine" else 1):
      if phase == 1:
        info = model.densify_and_prune(generator=torch.Generator(device=dev).manual_seed(100))
        ns.append(model.nbr_gaussians)
        model.reset_opacities()
      for _ in range(3 if phase == 0 else 2):
        n_steps += 1
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
    assert list(r0["ns"]) == ns, (list(r0["ns"]), ns)
    if variant == "refine":
        assert ns[1] != ns[0]
    for k in model.param_names:
        # Adam normalises the step, so tiny gradient differences can move a parameter by up to ~lr;
        # compare against the step size (lr <= 2.5e-2, 3 steps) on all but a sliver of entries
        d = np.abs(r0[k] - ref[k])
        assert np.mean(d > 1e-5) < 2e-3, (k, float(d.max()), float(np.mean(d > 1e-5)))
    np.testing.assert_allclose(r0["gn"], ref["gn"], rtol=1e-4, atol=1e-6)
    np.testing.assert_array_equal(r0["cnt"], ref["cnt"])
    np.testing.assert_array_equal(r0["rad"], ref["rad"])


def _worker_captured(rank, world, port, out_dir, rounds="off"):
    """Both forms of the view-parallel step in one process, one after the other (the ranks issue the same collectives in the same
    order): the eager exchange on model A, `ViewParallelGraphStep` on model B -- with a capacity shortage staged on RANK 1 ONLY."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from easy_gaussian_splatting_amd.distributed import ViewParallelStep
    from easy_gaussian_splatting_amd.loss import LossComputer
    from easy_gaussian_splatting_amd.train_graph import ViewParallelGraphStep
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    n_steps = 9
    # (a) eager
    model, opt, datas, targets = _make(dev, "plain")
    vp = ViewParallelStep(model, opt)
    lc = LossComputer(0.2, clamp_input=True)
    for it in range(n_steps):
        vp.begin_step(datas[rank])
        out = model(datas[rank], clamp=False)
        vp.after_forward(datas[rank], out)
        lc.get_loss_dict(out["render_img"], targets[rank])["total"].backward()
        vp.step(datas[rank], out)
        model.update_learning_rate(it + 1)
    torch.cuda.synchronize()
    eager = _snapshot(model)
    eager.update({"m_" + k: opt.moments_of(getattr(model, k))[0].cpu().numpy() for k in model.param_names})
    # (b) captured
    model, opt, datas, targets = _make(dev, "plain")
    vp = ViewParallelStep(model, opt, guard_words=True)
    runner = ViewParallelGraphStep(model, opt, LossComputer(0.2, clamp_input=True), datas[rank], targets[rank], None, vp=vp, check_every=4,
                                   rounds=rounds)
    assert runner.report()["rounds"] == (rounds == "on")
    for it in range(n_steps):
        if it == 3 and rank == 1:
            # rank 1 alone runs short of work units from here on: BOTH ranks must skip these steps on the device, find it at the
            # same poll, re-capture and replay -- or the replicas (and the collectives) part ways
            runner.finish()
            runner.cap_units = 512
            with torch.cuda.device(dev):
                runner._alloc_walk()
                runner.buf["applied"].zero_()
                runner._capture(warm_up=False)
        elif it == 3:
            runner.finish()   # (rank 0 drains too: `finish` polls, and polls are collective in effect)
        runner.step(datas[rank], targets[rank])
        model.update_learning_rate(it + 1)
    runner.finish()
    torch.cuda.synchronize()
    rep = runner.report()
    cap = _snapshot(model)
    cap.update({"m_" + k: opt.moments_of(getattr(model, k))[0].cpu().numpy() for k in model.param_names})
    np.savez(os.path.join(out_dir, f"c{rank}.npz"), overflows=rep["overflows"], replayed=rep["replayed_steps"], steps=rep["steps"],
             collectives=vp.collectives, **{"e_" + k: v for k, v in eager.items()}, **{"c_" + k: v for k, v in cap.items()})
    dist.barrier()
    dist.destroy_process_group()


def test_captured_view_parallel_step_equals_the_eager_exchange(tmp_path):
    """everything in front of the first collective as one hipGraph (`train_graph.ViewParallelGraphStep`), two
    ranks over gloo on the one device.  Nine steps, bitwise the eager exchange on every rank -- parameters, Adam moments,
    statistics -- although rank 1 alone is made to run out of work-unit storage at step 3: its guard flag travels inside both
    collectives, rank 0 skips the same steps, both find out at the same (blocking) poll, re-size, re-capture and replay."""
    mp.spawn(_worker_captured, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r = [np.load(os.path.join(tmp_path, f"c{k}.npz")) for k in range(2)]
    for k in range(2):
        assert int(r[k]["steps"]) == 9 and int(r[k]["overflows"]) >= 1 and int(r[k]["replayed"]) >= 1, {f: r[k][f] for f in ("steps", "overflows", "replayed")}
        for f in r[k].files:
            if f.startswith("e_"):
                np.testing.assert_array_equal(r[k]["c_" + f[2:]], r[k][f], err_msg=f"rank {k}: captured != eager in {f[2:]}")
    assert int(r[0]["overflows"]) == int(r[1]["overflows"]) and int(r[0]["collectives"]) == int(r[1]["collectives"])
    for f in r[0].files:
        if f.startswith("c_"):
            np.testing.assert_array_equal(r[0][f], r[1][f], err_msg=f"replicas diverged in {f[2:]}")


def test_captured_view_parallel_step_in_depth_rounds(tmp_path):
    """The same nine steps with the list stages of the captured part in two depth rounds (`rounds="on"`: gs_row_sums reads a wave's
    rows as two ranges): the replicas bitwise equal to each other, and on the eager exchange's trajectory to the rounding of
    another summation order of the gradient rows."""
    mp.spawn(_worker_captured, args=(2, _free_port(), str(tmp_path), "on"), nprocs=2, join=True)
    r = [np.load(os.path.join(tmp_path, f"c{k}.npz")) for k in range(2)]
    for k in range(2):
        assert int(r[k]["steps"]) == 9 and int(r[k]["overflows"]) >= 1
        for f in r[k].files:
            if f.startswith("e_"):
                a, b = r[k]["c_" + f[2:]].astype(np.float64), r[k][f].astype(np.float64)
                assert np.abs(a - b).max() <= 2e-4 * max(np.abs(b).max(), 1e-30), (k, f[2:], np.abs(a - b).max(), np.abs(b).max())
    for f in r[0].files:
        if f.startswith("c_"):
            np.testing.assert_array_equal(r[0][f], r[1][f], err_msg=f"replicas diverged in {f[2:]}")


def test_row_sums_pass_on_gaussians_of_hundreds_of_slots():
    """`gs_row_sums` on a heavy-tailed scene (synthetic.config_long_lists: splats of several hundred tiles, most list tails
    abandoned by saturated tiles): the slot-per-lane path with its empty-item skip AND the whole-wave path of the > 128-slot
    Gaussians, single-buffered in this kernel -- against the projection backward's own (double-buffered) row sum of the dense
    mode: the same sums, hence the same colour gradient bit for bit and the same geometry gradients to an ulp or two."""
    from scenes import config_long_lists
    dev = torch.device("cuda:0")
    sc = config_long_lists(seed=3, n=30_000, width=1920, height=1080)
    W, H = 1920, 1080
    t = {k: torch.from_numpy(v).to(dev) for k, v in sc.items() if isinstance(v, np.ndarray)}
    vc = torch.randn((1, H, W, 3), generator=torch.Generator().manual_seed(2)).to(dev)

    def run(mode):
        ins = [t[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities")]
        shs = t["shs"].clone().requires_grad_(True)
        dbg = {}
        img, _, meta = rasterization(*ins, shs, t["viewmats"], t["Ks"], W, H, sh_degree=3, packed=False, backgrounds=t["backgrounds"],
                                     absgrad=True, _sh_grads=mode, _tile_culling="tight", _debug=dbg)
        (img * vc).sum().backward()
        return [p.grad for p in ins], shs.grad, meta, dbg

    g_dense, sh_dense, meta_d, dbg_d = run("dense")
    g_fact, sh_fact, meta_f, dbg_f = run("colors_pre")
    tiles = meta_d["tiles_per_gauss"][0]
    assert int((tiles > 128).sum()) > 50 and int(tiles.max()) > 300, ("the scene must exercise the whole-wave path", int((tiles > 128).sum()), int(tiles.max()))
    # the sums themselves: bit for bit (debug outputs of the projection backward = what it was handed / what it summed)
    for k in ("v_means2d", "v_conics", "v_colors_post"):
        assert torch.equal(dbg_d[k], dbg_f[k]), k
    assert torch.equal(meta_d["means2d"].absgrad, meta_f["means2d"].absgrad)
    for a, b in zip(g_fact, g_dense):
        assert _rel(a, b) < 1e-6, _rel(a, b)
    assert sh_fact is None
    rebuilt = sh_grad_views(t["means"], t["viewmats"], meta_f["means2d"].colors_pre_grad, 3, 16, split=False)
    assert _rel(rebuilt, sh_dense) < 2e-6


def test_row_sums_record_and_in_place_bucket():
    """`_view_payload` + `_grad_out` (what ViewParallelStep installs): ONE launch after the blend backward fills the rank's
    all-gather record [3N colour gradients | N radii / max(H, W) | w2c], the projection backward writes the geometry gradients
    and the two statistics segments where the caller wants them; all of it bit for bit what the plain backward returns."""
    dev = torch.device("cuda:0")
    W, H, N = 176, 112, 2500
    sc = make_scene(N, W, H, sh_degree=3, n_views=1, seed=77, scale_range=(0.03, 0.2), dist=4.0)
    t = {k: torch.from_numpy(v).to(dev) for k, v in sc.items() if isinstance(v, np.ndarray)}
    vc = torch.randn((1, H, W, 3), generator=torch.Generator().manual_seed(1)).to(dev)

    def run(**kw):
        ins = [t[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities")]
        sh0 = t["shs"][:, :1].contiguous().requires_grad_(True)
        shr = t["shs"][:, 1:].contiguous().requires_grad_(True)
        img, _, meta = rasterization(*ins, (sh0, shr), t["viewmats"], t["Ks"], W, H, sh_degree=3, packed=False,
                                     backgrounds=t["backgrounds"], absgrad=True, **kw)
        (img * vc).sum().backward()
        return ins, meta

    ins_a, meta_a = run(_sh_grads="colors_pre")
    pay = torch.full((4 * N + 16 + 5,), 7.0, device=dev)
    go = {"means": torch.empty((N, 3), device=dev), "quats": torch.empty((N, 4), device=dev), "scales": torch.empty((N, 3), device=dev),
          "opacities": torch.empty((N,), device=dev), "grad_norm": torch.empty((N,), device=dev), "count": torch.empty((N,), device=dev)}
    seen = []
    ins_b, meta_b = run(_sh_grads="colors_pre", _view_payload=pay, _grad_out=go, _on_colors_pre=lambda v: seen.append(v.data_ptr()))
    assert go.pop("_written") is True and seen == [pay.data_ptr()]
    assert all(p.grad is None for p in ins_b)                       # autograd received None for the caller-owned gradients
    for k, p in zip(("means", "quats", "scales", "opacities"), ins_a):
        assert torch.equal(go[k], p.grad), k
    pre = meta_a["means2d"].colors_pre_grad
    assert torch.equal(pay[:3 * N].view(1, N, 3), pre) and meta_b["means2d"].colors_pre_grad.data_ptr() == pay.data_ptr()
    radii = meta_a["radii"][0]
    assert torch.equal(pay[3 * N:4 * N], torch.where(radii > 0, radii.float() / float(max(W, H)), 0.0))
    assert torch.equal(pay[4 * N:4 * N + 16], t["viewmats"][0].reshape(-1)) and float(pay[4 * N + 16:].min()) == 7.0
    vis = radii > 0
    assert torch.equal(go["count"], vis.float())
    gn = torch.where(vis, meta_a["means2d"].absgrad[0].norm(dim=-1) * float(max(W, H)), 0.0)
    assert _rel(go["grad_norm"], gn) < 1e-6 and float(go["grad_norm"][~vis].abs().max()) == 0.0
    with pytest.raises(ValueError):
        run(_sh_grads="dense", _view_payload=pay)
    with pytest.raises(ValueError):
        run(_sh_grads="colors_pre", _view_payload=pay[:4 * N])


@pytest.mark.parametrize("deg,R,K", [(3, 1, 16), (3, 3, 16), (3, 8, 16), (1, 2, 16), (0, 2, 16), (0, 3, 1), (1, 2, 4)])
def test_sh_adam_views_equals_rebuild_plus_adam(deg, R, K):
    """`gs_sh_adam_views` (SH gradient of R views formed in LDS and applied by Adam in place, `max_radii` folded in) ==
    `gs_sh_grad_views` + `FusedAdam.step(only=SH, grad_scale=1/R)` + `torch.maximum`, bit for bit, over two steps."""
    from easy_gaussian_splatting_amd import _native as nat
    from easy_gaussian_splatting_amd.optim import FusedAdam
    dev = torch.device("cuda:0")
    N = 3001
    g = torch.Generator().manual_seed(5 + deg)
    means = (torch.rand((N, 3), generator=g) * 2 - 1).to(dev)
    cams = torch.eye(4).repeat(R, 1, 1)
    cams[:, :3, 3] = torch.randn((R, 3), generator=g) + torch.tensor([0.0, 0.0, 6.0])
    cams = cams.to(dev)

    def make():
        gg = torch.Generator().manual_seed(11)
        ps = {"sh_0": torch.nn.Parameter(torch.randn((N, 1, 3), generator=gg).to(dev)),
              "sh_rest": torch.nn.Parameter(torch.randn((N, K - 1, 3), generator=gg).to(dev)),
              "means": torch.nn.Parameter(means.clone())}
        return ps, FusedAdam([{"params": [p], "lr": 1e-2 * (i + 1), "name": k} for i, (k, p) in enumerate(ps.items())])

    pa, oa = make()
    pb, ob = make()
    rad_a, rad_b = torch.zeros(N, device=dev), torch.zeros(N, device=dev)
    P = 4 * N + 16
    for it in range(2):
        rec = torch.zeros((R, P), device=dev)
        pre = torch.randn((R, N, 3), generator=g).to(dev)
        pre[torch.rand((R, N), generator=g).to(dev) < 0.3] = 0.0          # culled in that view
        rad = torch.rand((R, N), generator=g).to(dev)
        rec[:, :3 * N] = pre.reshape(R, -1); rec[:, 3 * N:4 * N] = rad; rec[:, 4 * N:] = cams.reshape(R, 16)
        # reference: dense rebuild, then the SH half of Adam
        v0, vr = sh_grad_views(means, cams, pre, deg, K)
        pa["sh_0"].grad, pa["sh_rest"].grad = v0, (vr if K > 1 else None)
        oa.step(only=("sh_0", "sh_rest"), grad_scale=1.0 / R)
        torch.maximum(rad_a, rad.max(0).values, out=rad_a)
        # fused
        ob._step += 1
        m0, s0 = ob.moments_of(pb["sh_0"]); mr, sr = ob.moments_of(pb["sh_rest"])
        rest = (pb["sh_rest"].data_ptr(), mr.data_ptr(), sr.data_ptr()) if K > 1 else (None, None, None)   # (K = 1: no sh_rest at all)
        nat.check(nat.lib().gs_sh_adam_views(torch.cuda.current_stream().cuda_stream, R, N, K, deg, means.data_ptr(), rec.data_ptr(), P,
                                             pb["sh_0"].data_ptr(), m0.data_ptr(), s0.data_ptr(), *rest,
                                             1e-2, 2e-2, 0.9, 0.999, 1e-8, ob._step, 1.0 / R, rad_b.data_ptr()), "gs_sh_adam_views")
    assert oa._step == ob._step == 2
    for k in ("sh_0", "sh_rest"):
        assert torch.equal(pa[k].detach(), pb[k].detach()), k
        for x, y in zip(oa.moments_of(pa[k]), ob.moments_of(pb[k])):
            assert torch.equal(x, y), k
    assert torch.equal(rad_a, rad_b)
    ka = (deg + 1) ** 2
    if ka < K:   # coefficients beyond the active degree: zero gradient, Adam leaves parameter and moments where they were
        gg = torch.Generator().manual_seed(11)
        torch.randn((N, 1, 3), generator=gg)
        init_rest = torch.randn((N, K - 1, 3), generator=gg).to(dev)
        assert torch.equal(pb["sh_rest"].detach()[:, ka - 1:], init_rest[:, ka - 1:])


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
