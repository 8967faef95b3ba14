"""View data-parallel layer on CPU: world_size=2 `gloo` processes, one view per rank, the product's
GradBucket / statistics reductions, with the torch oracle standing in for the GPU rasterizer.
Checks  mean_rank(grads) == grads of the single-process C=2 batch with a mean-over-views loss."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _scene(n_views=2):
    sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
    from scenes import make_scene
    return make_scene(80, 36, 28, sh_degree=1, n_views=n_views, seed=21, scale_range=(0.05, 0.4), dist=4.0)


def _params(sc):
    dt = torch.float64
    return [torch.nn.Parameter(torch.tensor(sc[k], dtype=dt)) for k in ("means", "quats", "scales", "opacities", "shs")]


def _view_loss(params, sc, views, target):
    from oracle import torch_oracle as TO
    dt = torch.float64
    V = torch.tensor(sc["viewmats"][views], dtype=dt); K = torch.tensor(sc["Ks"][views], dtype=dt)
    bg = torch.tensor(sc["backgrounds"][views], dtype=dt)
    img, _, meta = TO.rasterization(*params, V, K, sc["width"], sc["height"], sh_degree=1, packed=False,
                                    backgrounds=bg, absgrad=True)
    return ((img - target[views]) ** 2).mean(dim=(1, 2, 3)).sum(), meta


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from easy_gaussian_splatting_amd.distributed import GradBucket, all_reduce_param_grads, all_reduce_statistics, shard_views
    sc = _scene()
    params = _params(sc)
    bucket = GradBucket(params)
    target = torch.rand((2, sc["height"], sc["width"], 3), generator=torch.Generator().manual_seed(5), dtype=torch.float64)
    views = shard_views(2)
    assert views == [rank]
    loss, meta = _view_loss(params, sc, views, target)
    loss.backward()
    assert params[0].grad.data_ptr() == bucket.flat.data_ptr()  # autograd accumulated into the bucket
    loose = [torch.nn.Parameter(p.detach().clone()) for p in params]   # bucket-free variant
    for q, p in zip(loose, params):
        q.grad = p.grad.detach().clone()
    bucket.all_reduce_mean()
    all_reduce_param_grads(loose)
    for q, p in zip(loose, params):
        assert torch.equal(q.grad, p.grad)
    radii = meta["radii"][0].double() / max(sc["width"], sc["height"])
    vis = radii > 0
    g = torch.where(vis, meta["means2d"].absgrad[0].norm(dim=-1), torch.zeros_like(radii))
    cnt = vis.double(); rad = torch.where(vis, radii, torch.zeros_like(radii))
    all_reduce_statistics(g, cnt, rad)
    if rank == 0:
        np.savez(os.path.join(out_dir, "dp.npz"), flat=bucket.flat.numpy(), g=g.numpy(), cnt=cnt.numpy(), rad=rad.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_view_dp_equals_two_camera_batch(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.load(os.path.join(tmp_path, "dp.npz"))
    sc = _scene()
    params = _params(sc)
    target = torch.rand((2, sc["height"], sc["width"], 3), generator=torch.Generator().manual_seed(5), dtype=torch.float64)
    loss, meta = _view_loss(params, sc, [0, 1], target)
    (loss / 2).backward()
    ref = torch.cat([p.grad.reshape(-1) for p in params]).numpy()
    np.testing.assert_allclose(got["flat"], ref, atol=1e-12 * max(1.0, np.abs(ref).max()))
    radii = meta["radii"].double() / max(sc["width"], sc["height"])
    vis = radii > 0
    g = torch.where(vis, meta["means2d"].absgrad.norm(dim=-1), torch.zeros_like(radii))
    # each rank back-propagated its own un-divided view loss; the batch above used loss/2
    np.testing.assert_allclose(got["g"], 2.0 * g.sum(0).numpy(), atol=1e-12)
    np.testing.assert_allclose(got["cnt"], vis.double().sum(0).numpy())
    np.testing.assert_allclose(got["rad"], torch.where(vis, radii, torch.zeros_like(radii)).max(0).values.numpy())


def test_single_process_helpers_are_noops():
    sys.path.insert(0, ROOT)
    from easy_gaussian_splatting_amd.distributed import GradBucket, is_distributed, shard_views
    assert not is_distributed()
    p = [torch.nn.Parameter(torch.ones(3, 2)), torch.nn.Parameter(torch.ones(5))]
    b = GradBucket(p)
    (p[0].sum() * 2 + p[1].sum() * 3).backward()
    assert b.flat.tolist() == [2.0] * 6 + [3.0] * 5
    assert b.all_reduce_mean() is None
    b.zero_()
    assert p[1].grad.abs().sum() == 0
    assert shard_views(8, rank=1, world=4) == [1, 5]


# ------------------------------------------------------------------------------------------------
# ViewParallelStep (factorised exchange): all-gather of the pre-clamp colour gradients + all-reduce of
# the geometry gradients must give the same parameter update and statistics as one process that
# back-propagates the mean loss over both views.  Stand-ins on CPU: the torch oracle renders, a
# torch SH-basis einsum plays gs_sh_grad_views, plain SGD plays FusedAdam's partial steps.
class _Model:
    GEOM = ("means", "log_scales", "quats", "logit_opacities")

    def __init__(self, sc):
        dt = torch.float64
        P = lambda a: torch.nn.Parameter(torch.tensor(a, dtype=dt))  # noqa: E731
        self.means, self.quats = P(sc["means"]), P(sc["quats"])
        self.log_scales = P(np.log(sc["scales"]))
        o = np.clip(sc["opacities"], 1e-4, 1 - 1e-4)
        self.logit_opacities = P(np.log(o / (1 - o)))
        self.sh_0, self.sh_rest = P(sc["shs"][:, :1]), P(sc["shs"][:, 1:])
        self.active_sh_degree = 1
        n = sc["means"].shape[0]
        self.grad_norm_accum, self.collecting_counts, self.max_radii = (torch.zeros(n, dtype=dt) for _ in range(3))

    def named(self):
        return {k: getattr(self, k) for k in ("means", "log_scales", "quats", "sh_0", "sh_rest", "logit_opacities")}

    def inputs(self):
        return self.means, self.quats, self.log_scales.exp(), torch.sigmoid(self.logit_opacities)


class _SGD:
    def __init__(self, model, lr=0.05):
        self.model, self.lr, self.calls = model, lr, []

    def moments_of(self, p):
        raise KeyError

    def step(self, only=None, grad_scale=1.0, advance=True):
        self.calls.append((None if only is None else tuple(only), advance))
        with torch.no_grad():
            for name, p in self.model.named().items():
                if (only is None or name in only) and p.grad is not None:
                    p -= self.lr * grad_scale * p.grad

    def zero_grad(self):
        for p in self.model.named().values():
            p.grad = None


def _torch_sh_grad_views(means, cams, pre_all, deg, K):
    from oracle import torch_oracle as TO
    cam_pos = torch.linalg.inv(cams)[:, :3, 3]
    d = means.detach()[None] - cam_pos[:, None, :]
    d = d / d.norm(dim=-1, keepdim=True)
    Y = TO.sh_basis(deg, d)                                    # [R,N,Ka]
    v = torch.einsum("rnk,rnc->nkc", Y, pre_all)
    v = torch.cat([v, torch.zeros((v.shape[0], K - v.shape[1], 3), dtype=v.dtype)], dim=1)
    return v[:, :1].contiguous(), v[:, 1:].contiguous()


def _vp_render(model, sc, view, target, early_gather=False):
    """One view through the oracle with the colours evaluated outside the rasterizer, so that the
    gradient w.r.t. the pre-clamp colour (what gs_project_bwd emits as v_colors_pre) can be read."""
    from oracle import torch_oracle as TO
    dt = torch.float64
    V = torch.tensor(sc["viewmats"][view:view + 1], dtype=dt); K = torch.tensor(sc["Ks"][view:view + 1], dtype=dt)
    bg = torch.tensor(sc["backgrounds"][view:view + 1], dtype=dt)
    means, quats, scales, opac = model.inputs()
    with torch.no_grad():
        radii = TO.project(means, quats, scales, V, K, sc["width"], sc["height"], 0.3, 0.01, 1e10, 0.0)[0]
    cols = TO.spherical_harmonics(1, means, V, torch.cat([model.sh_0, model.sh_rest], dim=1), radii)
    cols.retain_grad()
    img, _, meta = TO.rasterization(means, quats, scales, opac, cols, V, K, sc["width"], sc["height"], sh_degree=None,
                                    packed=False, backgrounds=bg, absgrad=True)
    ((img - target[view:view + 1]) ** 2).mean().backward()
    meta["means2d"].colors_pre_grad = (cols.grad * (cols > 0)).detach()
    if early_gather and getattr(model, "on_colors_pre", None) is not None:   # what the rasterizer's backward does
        model.on_colors_pre(meta["means2d"].colors_pre_grad)
    model.sh_0.grad = None; model.sh_rest.grad = None          # factorised mode: no local SH gradients
    return {"batch_xys": meta["means2d"], "batch_radii": meta["radii"]}


def _vp_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from easy_gaussian_splatting_amd.distributed import ViewParallelStep
    torch.set_num_threads(1)   # (eight ranks share this box's cores)
    sc = _scene(world)
    model = _Model(sc)
    opt = _SGD(model)
    vp = ViewParallelStep(model, opt, sh_grad_fn=_torch_sh_grad_views)
    assert model.sh_grads == "colors_pre" and vp.world == world
    target = torch.rand((world, sc["height"], sc["width"], 3), generator=torch.Generator().manual_seed(5), dtype=torch.float64)
    data = {"w2c": torch.tensor(sc["viewmats"][rank], dtype=torch.float64), "height": sc["height"], "width": sc["width"]}
    for it in range(2):   # first step with the two early-collective hooks, second without
        if it == 0:
            vp.begin_step(data)
        out = _vp_render(model, sc, rank, target, early_gather=(it == 0))
        if it == 0:
            vp.after_forward(data, out)
        vp.step(data, out)
    assert opt.calls[:2] == [(("sh_0", "sh_rest"), True), (("means", "log_scales", "quats", "logit_opacities"), False)]
    np.savez(os.path.join(out_dir, f"vp{rank}.npz"), gn=model.grad_norm_accum.numpy(), cnt=model.collecting_counts.numpy(),
             rad=model.max_radii.numpy(), **{k: v.detach().numpy() for k, v in model.named().items()})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_view_parallel_step_equals_two_view_batch(tmp_path, world):
    """world = 8: BASELINE.json configs[3]'s rank count -- the 8-way all_gather_into_tensor layout, the rank-order sum over eight
    views, 1/8 folded into the optimizer step -- on gloo (the GPU twin is
    tests/test_gpu_configs.py::test_config_s4_four_ranks_equal_single_process)."""
    from oracle import torch_oracle as TO
    port = _free_port()
    mp.spawn(_vp_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, *others = (np.load(os.path.join(tmp_path, f"vp{r}.npz")) for r in range(world))
    for r1 in others:
        for k in r0.files:
            np.testing.assert_array_equal(r0[k], r1[k], err_msg=f"replicas diverged in {k}")
    # single-process reference: mean loss over all views through the ordinary SH path, plain SGD
    sc = _scene(world)
    model = _Model(sc)
    dt = torch.float64
    target = torch.rand((world, sc["height"], sc["width"], 3), generator=torch.Generator().manual_seed(5), dtype=dt)
    V = torch.tensor(sc["viewmats"][:world], dtype=dt); K = torch.tensor(sc["Ks"][:world], dtype=dt)
    bg = torch.tensor(sc["backgrounds"][:world], dtype=dt)
    max_hw = max(sc["width"], sc["height"])
    for _ in range(2):
        means, quats, scales, opac = model.inputs()
        img, _, meta = TO.rasterization(means, quats, scales, opac, torch.cat([model.sh_0, model.sh_rest], dim=1), V, K,
                                        sc["width"], sc["height"], sh_degree=1, packed=False, backgrounds=bg, absgrad=True)
        (((img - target) ** 2).mean(dim=(1, 2, 3)).sum() / world).backward()
        vis = meta["radii"] > 0
        # each rank back-propagated its own un-divided view loss -> absgrad is `world` times the batch's
        model.grad_norm_accum += (torch.where(vis, float(world) * meta["means2d"].absgrad.norm(dim=-1) * max_hw, 0.0)).sum(0)
        model.collecting_counts += vis.double().sum(0)
        model.max_radii = torch.maximum(model.max_radii, torch.where(vis, meta["radii"].double() / max_hw, 0.0).max(0).values)
        with torch.no_grad():
            for p in model.named().values():
                p -= 0.05 * p.grad
                p.grad = None
    for k, p in model.named().items():
        np.testing.assert_allclose(r0[k], p.detach().numpy(), rtol=0, atol=1e-11 * max(1.0, float(p.detach().abs().max())), err_msg=k)
    np.testing.assert_allclose(r0["gn"], model.grad_norm_accum.numpy(), atol=1e-10)
    np.testing.assert_array_equal(r0["cnt"], model.collecting_counts.numpy())
    np.testing.assert_array_equal(r0["rad"], model.max_radii.numpy())


# ------------------------------------------------------------------------------------------------
# Refinement across replicas (VERDICT r4 missing #1a): steps -> densify_and_prune -> reset_opacities -> steps with one view per
# rank.  Exercises what only a multi-rank job reaches: the noise broadcast of GaussianModel._split_noise (every rank seeds its
# generator differently on purpose), assert_replicas_identical, and ViewParallelStep's buffers following a change of N.
# The product's model class on the CPU in float64; the torch oracle renders; a small torch Adam with FusedAdam's interface
# (partial steps, folded gradient scale, moments_of / replace_parameters) plays optim.FusedAdam.
class _PartialAdam:
    def __init__(self, model, lr=0.01, betas=(0.9, 0.999), eps=1e-8):
        self.model, self.lr, self.betas, self.eps = model, lr, betas, eps
        self.param_groups = [{"name": n, "lr": lr, "params": [getattr(model, n)]} for n in model.param_names]
        self.m = {n: torch.zeros_like(getattr(model, n)) for n in model.param_names}
        self.v = {n: torch.zeros_like(getattr(model, n)) for n in model.param_names}
        self._step = 0
        model.register_optimizer(self)

    def _name_of(self, p):
        return next(n for n in self.model.param_names if getattr(self.model, n) is p)

    def moments_of(self, p):
        n = self._name_of(p)
        return self.m[n], self.v[n]

    def replace_parameters(self, triples):
        for name, (p, m, v) in zip(self.model.param_names, triples):
            assert getattr(self.model, name) is p
            self.m[name], self.v[name] = m.detach().clone(), v.detach().clone()
        for g in self.param_groups:
            g["params"] = [getattr(self.model, g["name"])]

    def step(self, only=None, grad_scale=1.0, advance=True):
        if advance:
            self._step += 1
        b1, b2 = self.betas
        with torch.no_grad():
            for n in self.model.param_names:
                p = getattr(self.model, n)
                if (only is not None and n not in only) or p.grad is None:
                    continue
                g = p.grad * grad_scale
                self.m[n].mul_(b1).add_(g, alpha=1 - b1)
                self.v[n].mul_(b2).addcmul_(g, g, value=1 - b2)
                denom = self.v[n].sqrt() / (1 - b2 ** self._step) ** 0.5 + self.eps
                p -= self.lr / (1 - b1 ** self._step) * self.m[n] / denom

    def zero_grad(self):
        for n in self.model.param_names:
            getattr(self.model, n).grad = None


def _refine_model(sc):
    sys.path.insert(0, ROOT)
    from easy_gaussian_splatting_amd.model import GaussianModel
    T = torch.tensor
    o = np.clip(sc["opacities"], 1e-3, 1 - 1e-3)
    m = GaussianModel(means=T(sc["means"]), log_scales=T(np.log(sc["scales"])), quats=T(sc["quats"]), sh_0=T(sc["shs"][:, :1].copy()),
                      sh_rest=T(sc["shs"][:, 1:].copy()), logit_opacities=T(np.log(o / (1 - o))), sh_degree=1, white_background=True,
                      # thresholds at which a 3-step history on an 80-splat scene splits, clones AND prunes
                      densify_grad_thresh=2e-3, densify_scale_thresh=0.12, prune_scale_thresh=0.6, min_opacity=0.02).double()
    for name in ("grad_norm_accum", "collecting_counts", "max_radii"):
        setattr(m, name, getattr(m, name).double())
    return m


def _refine_render(model, sc, views, target, weight=1.0, on_colors_pre=None):
    """`views` through the oracle, colours evaluated outside the rasterizer so that the pre-clamp colour gradient can be read
    (one view: what the rasterizer's backward hands ViewParallelStep; several: the single-process reference, ordinary SH path)."""
    from oracle import torch_oracle as TO
    dt = torch.float64
    V = torch.tensor(sc["viewmats"][views], dtype=dt); K = torch.tensor(sc["Ks"][views], dtype=dt)
    bg = torch.tensor(sc["backgrounds"][views], dtype=dt)
    means, quats, scales, opac = model.means, model.quats, model.scales, model.opacities
    if len(views) == 1:
        with torch.no_grad():
            radii = TO.project(means, quats, scales, V, K, sc["width"], sc["height"], 0.3, 0.01, 1e10, 0.0)[0]
        cols = TO.spherical_harmonics(1, means, V, model.shs, radii)
        cols.retain_grad()
        img, _, meta = TO.rasterization(means, quats, scales, opac, cols, V, K, sc["width"], sc["height"], sh_degree=None,
                                        packed=False, backgrounds=bg, absgrad=True)
    else:
        img, _, meta = TO.rasterization(means, quats, scales, opac, model.shs, V, K, sc["width"], sc["height"], sh_degree=1,
                                        packed=False, backgrounds=bg, absgrad=True)
    (weight * ((img - target[views]) ** 2).mean(dim=(1, 2, 3)).sum()).backward()
    if len(views) == 1:
        meta["means2d"].colors_pre_grad = (cols.grad * (cols > 0)).detach()
        model.sh_0.grad = None; model.sh_rest.grad = None          # factorised mode: no local SH gradients
    return {"batch_xys": meta["means2d"], "batch_radii": meta["radii"]}, meta


def _refine_snapshot(model, ns):
    out = {k: getattr(model, k).detach().numpy() for k in model.param_names}
    out.update(gn=model.grad_norm_accum.numpy(), cnt=model.collecting_counts.numpy(), rad=model.max_radii.numpy(), ns=np.asarray(ns))
    return out


def _refine_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from easy_gaussian_splatting_amd.distributed import ViewParallelStep
    torch.set_num_threads(1)
    sc = _scene(world)
    model = _refine_model(sc)
    opt = _PartialAdam(model)
    vp = ViewParallelStep(model, opt, sh_grad_fn=_torch_sh_grad_views)
    assert not vp.native and model.sh_grads == "colors_pre"
    target = torch.rand((world, sc["height"], sc["width"], 3), generator=torch.Generator().manual_seed(5), dtype=torch.float64)
    data = {"w2c": torch.tensor(sc["viewmats"][rank], dtype=torch.float64), "height": sc["height"], "width": sc["width"]}
    ns = [model.nbr_gaussians]

    def steps(k):
        for _ in range(k):
            out, _ = _refine_render(model, sc, [rank], target)
            vp.step(data, out)

    steps(3)
    # every rank's generator is in a different state on purpose: the split noise must come from rank 0 (model._split_noise)
    info = model.densify_and_prune(generator=torch.Generator().manual_seed(100 + rank))
    assert info["train/densify"]["split"] > 0 and info["train/densify"]["clone"] > 0, info
    ns.append(model.nbr_gaussians)
    steps(1)
    model.reset_opacities()
    steps(2)
    info2 = model.densify_and_prune(generator=torch.Generator().manual_seed(200 + rank))
    ns.append(model.nbr_gaussians)
    steps(1)
    assert vp.collectives == 2 * 7
    np.savez(os.path.join(out_dir, f"rf{rank}.npz"), **_refine_snapshot(model, ns))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_view_parallel_refinement_keeps_replicas_identical_and_equals_one_process(tmp_path, world):
    mp.spawn(_refine_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0, *others = (np.load(os.path.join(tmp_path, f"rf{r}.npz")) for r in range(world))
    for r1 in others:
        for k in r0.files:
            np.testing.assert_array_equal(r0[k], r1[k], err_msg=f"replicas diverged in {k}")
    assert r0["ns"][1] != r0["ns"][0], "the refinement changed nothing: the test exercises nothing"
    # one process on all views, mean-over-views loss, the same optimizer, rank 0's noise
    sc = _scene(world)
    model = _refine_model(sc)
    opt = _PartialAdam(model)
    target = torch.rand((world, sc["height"], sc["width"], 3), generator=torch.Generator().manual_seed(5), dtype=torch.float64)
    max_hw = max(sc["width"], sc["height"])
    views = list(range(world))
    ns = [model.nbr_gaussians]

    def steps(k):
        for _ in range(k):
            _, meta = _refine_render(model, sc, views, target, weight=1.0 / world)
            vis = meta["radii"] > 0
            # each rank back-propagated its own un-divided view loss -> absgrad is `world` times the batch's
            model.grad_norm_accum += torch.where(vis, float(world) * meta["means2d"].absgrad.norm(dim=-1) * max_hw, 0.0).sum(0)
            model.collecting_counts += vis.double().sum(0)
            model.max_radii = torch.maximum(model.max_radii, torch.where(vis, meta["radii"].double() / max_hw, 0.0).max(0).values)
            opt.step()
            opt.zero_grad()

    steps(3)
    model.densify_and_prune(generator=torch.Generator().manual_seed(100))
    ns.append(model.nbr_gaussians)
    steps(1)
    model.reset_opacities()
    steps(2)
    model.densify_and_prune(generator=torch.Generator().manual_seed(200))
    ns.append(model.nbr_gaussians)
    steps(1)
    assert list(r0["ns"]) == ns, (list(r0["ns"]), ns)
    ref = _refine_snapshot(model, ns)
    for k in model.param_names:
        # (Adam divides by sqrt(v): a gradient that differs in the last bits -- the factorised SH sum runs in another order --
        #  moves an entry whose gradient is ~0 by up to one lr; everything else agrees to rounding)
        d = np.abs(r0[k] - ref[k])
        assert np.mean(d > 1e-9) < 5e-3 and d.max() <= 7 * 0.01 * 1.001, (k, float(d.max()), float(np.mean(d > 1e-9)))
    np.testing.assert_allclose(r0["gn"], ref["gn"], atol=1e-9)
    np.testing.assert_array_equal(r0["cnt"], ref["cnt"])
    np.testing.assert_allclose(r0["rad"], ref["rad"], rtol=1e-6)   # (densify_and_prune re-creates the statistics in float32)
