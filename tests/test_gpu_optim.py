"""Fused HIP Adam (csrc/gs_adam.hip) against torch.optim.Adam on the reference's six groups."""
import numpy as np
import pytest
import torch

from easy_gaussian_splatting_amd.model import GaussianModel, build_optimizers
from easy_gaussian_splatting_amd.optim import FusedAdam

pytestmark = pytest.mark.gpu


def _model(n, dev, seed):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    return GaussianModel(means=r(n, 3), log_scales=r(n, 3) * 0.1 - 2, quats=r(n, 4), sh_0=r(n, 1, 3), sh_rest=r(n, 15, 3) * 0.1,
                         logit_opacities=r(n), sh_degree=3).to(dev)


@pytest.mark.parametrize("n", [1001, 4096])
def test_fused_adam_matches_torch_adam(n):
    dev = torch.device("cuda:0")
    lrs = (1.6e-4, 5e-3, 1e-3, 2.5e-3, 1.25e-4, 5e-2)
    a, b = _model(n, dev, 3), _model(n, dev, 3)
    oa = build_optimizers(a, *lrs)                # torch reference
    ob = build_optimizers(b, *lrs, fused="hip")
    assert [g["name"] for g in ob.param_groups] == a.param_names
    g = torch.Generator().manual_seed(9)
    for it in range(5):
        if it == 3:  # the reference edits the means lr through the param group (model/gaussian.py:121-128)
            for o in (oa, ob):
                o.param_groups[0]["lr"] = 3e-5
        for name in a.param_names:
            grad = torch.randn(getattr(a, name).shape, generator=g).to(dev) * (10.0 ** (it - 2))
            pa, pb = getattr(a, name), getattr(b, name)
            pa.grad = grad.clone()
            pb.grad = grad.clone() if not (it == 4 and name == "quats") else None   # a skipped tensor
            if it == 4 and name == "quats":
                pa.grad = None
        oa.step(); ob.step(); oa.zero_grad(); ob.zero_grad()
        assert all(getattr(b, name).grad is None for name in b.param_names)
    for name in a.param_names:
        pa, pb = getattr(a, name).detach().cpu().numpy(), getattr(b, name).detach().cpu().numpy()
        assert np.abs(pa - pb).max() <= 2e-6 * max(1.0, np.abs(pa).max()), name


def test_fused_adam_state_dict_round_trip_and_torch_layout():
    """state_dict() exports torch.optim.Adam's layout (step / exp_avg / exp_avg_sq per parameter); loading it into a
    fresh FusedAdam -- or into torch.optim.Adam -- continues the same trajectory (/root/reference/utils.py:48-87
    checkpoints `optimizer.state_dict()`)."""
    dev = torch.device("cuda:0")
    lrs = (1.6e-4, 5e-3, 1e-3, 2.5e-3, 1.25e-4, 5e-2)
    g = torch.Generator().manual_seed(4)

    def grads(m):
        return {name: torch.randn(getattr(m, name).shape, generator=g).to(dev) for name in m.param_names}

    a = _model(777, dev, 5)
    oa = build_optimizers(a, *lrs, fused="hip")
    for _ in range(3):
        for name, gr in grads(a).items():
            getattr(a, name).grad = gr
        oa.step(); oa.zero_grad()
    sd = oa.state_dict()
    assert len(sd["state"]) == 6 and all(set(v) == {"step", "exp_avg", "exp_avg_sq"} for v in sd["state"].values())
    assert all(float(v["step"]) == 3.0 for v in sd["state"].values())
    assert len(oa.state) == 0   # nothing lingers in the base-class state
    # resume twice: a fresh FusedAdam and a plain torch Adam, same parameters, same next gradient
    b, c = _model(777, dev, 5), _model(777, dev, 5)
    ob, oc = build_optimizers(b, *lrs, fused="hip"), build_optimizers(c, *lrs)
    with torch.no_grad():
        for name in a.param_names:
            getattr(b, name).copy_(getattr(a, name)); getattr(c, name).copy_(getattr(a, name))
    ob.load_state_dict(sd); oc.load_state_dict(sd)
    assert ob._step == 3
    nxt = grads(a)
    for m, o in ((a, oa), (b, ob), (c, oc)):
        for name in m.param_names:
            getattr(m, name).grad = nxt[name].clone()
        o.step(); o.zero_grad()
    for name in a.param_names:
        pa, pb, pc = (getattr(m, name).detach() for m in (a, b, c))
        assert torch.equal(pa, pb), name
        assert float((pa - pc).abs().max()) <= 2e-6 * max(1.0, float(pc.abs().max())), name


def test_fused_adam_detects_detached_parameters_and_rejects_group_betas():
    dev = torch.device("cuda:0")
    m = _model(64, dev, 1)
    opt = build_optimizers(m, 1e-3, 1e-3, 1e-3, 1e-3, 1e-3, 1e-3, fused="hip")
    m.means.data = m.means.data.clone()   # what a later model.to(...) does: the parameter no longer aliases the flat buffer
    m.means.grad = torch.zeros_like(m.means)
    with pytest.raises(RuntimeError, match="no longer aliases"):
        opt.step()
    from easy_gaussian_splatting_amd.optim import FusedAdam
    p = torch.nn.Parameter(torch.zeros(8, device=dev))
    with pytest.raises(NotImplementedError):
        FusedAdam([{"params": [p], "lr": 1e-3, "betas": (0.5, 0.9), "name": "x"}])


def test_checkpoint_round_trip_of_a_fused_adam_model(tmp_path):
    """save_gaussian_model / load_gaussian_model (reference layout, /root/reference/utils.py:48-87) with parameters that
    are views of FusedAdam's flat buffer, with and without the optimizer; training resumes on the loaded model."""
    from easy_gaussian_splatting_amd import checkpoint as ckpt
    dev = torch.device("cuda:0")
    m = _model(300, dev, 8)
    opt = build_optimizers(m, 1.6e-4, 5e-3, 1e-3, 2.5e-3, 1.25e-4, 5e-2, fused="hip")
    for name in m.param_names:
        getattr(m, name).grad = torch.randn_like(getattr(m, name))
    opt.step(); opt.zero_grad()
    ckpt.save_gaussian_model(tmp_path / "checkpoints" / "iterations_10.pth", m)
    ckpt.save_gaussian_model(tmp_path / "checkpoints" / "iterations_20.pth", m, save_optimizer=True)
    assert m.optimizer is opt
    a = ckpt.load_gaussian_model(tmp_path, 10)
    b = ckpt.load_gaussian_model(tmp_path)          # latest: iterations_20, with optimizer
    assert a.optimizer is None and b.optimizer is not None and a.means.is_cuda
    for name in m.param_names:
        assert torch.equal(getattr(a, name), getattr(m, name)) and torch.equal(getattr(b, name), getattr(m, name))
    # resume on the checkpoint without optimizer: a fresh FusedAdam over the loaded parameters steps fine
    oa = build_optimizers(a, 1.6e-4, 5e-3, 1e-3, 2.5e-3, 1.25e-4, 5e-2, fused="hip")
    for name in a.param_names:
        getattr(a, name).grad = torch.randn_like(getattr(a, name))
    oa.step()
    # resume WITH the pickled optimizer state: the file holds a torch.optim.Adam in the reference's layout;
    # loaded as it is ("keep") and as FusedAdam ("hip"), the next step must equal the original optimizer's next step
    assert isinstance(b.optimizer, torch.optim.Adam)
    c = ckpt.load_gaussian_model(tmp_path, optimizer="hip")
    assert isinstance(c.optimizer, FusedAdam) and c.optimizer._step == opt._step == 1
    grads = {name: torch.randn_like(getattr(m, name)) for name in m.param_names}
    for mod in (m, b, c):
        for name in m.param_names:
            getattr(mod, name).grad = grads[name].clone()
        mod.optimizer.step()
    for name in m.param_names:
        assert torch.equal(getattr(c, name), getattr(m, name)), name                       # same kernel, same state
        assert torch.allclose(getattr(b, name), getattr(m, name), rtol=1e-5, atol=1e-5), name   # torch's Adam on the same state
    c.optimizer._check_views()


def test_fused_adam_with_an_empty_parameter_tensor():
    """SH degree 0: sh_rest is [N, 0, 3] -- an empty segment must neither trip the view check nor the kernel."""
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    r = lambda *s: torch.randn(*s, generator=g)
    m = GaussianModel(means=r(100, 3), log_scales=r(100, 3), quats=r(100, 4), sh_0=r(100, 1, 3), sh_rest=torch.zeros(100, 0, 3),
                      logit_opacities=r(100), sh_degree=0).to(dev)
    opt = build_optimizers(m, 1e-3, 1e-3, 1e-3, 1e-3, 1e-3, 1e-3, fused="hip")
    before = m.means.detach().clone()
    for name in m.param_names:
        getattr(m, name).grad = torch.ones_like(getattr(m, name))
    opt.step(); opt.zero_grad()
    assert float((m.means.detach() - before).abs().max()) > 0 and m.sh_rest.shape == (100, 0, 3)
