"""densify_and_prune / reset_opacities mirror on CPU with torch.optim.Adam:
the decisions and the optimizer-state surgery of /root/reference/model/gaussian.py:130-146, 199-349."""
import numpy as np
import torch

from easy_gaussian_splatting_amd.model import GaussianModel, build_optimizers
from easy_gaussian_splatting_amd.rendering import quat_to_rotmat_torch


def _model(n, seed=0, **kw):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    return GaussianModel(means=r(n, 3), log_scales=torch.log(torch.rand(n, 3, generator=g) * 0.03 + 0.001), quats=r(n, 4),
                         sh_0=r(n, 1, 3), sh_rest=r(n, 15, 3), logit_opacities=r(n) * 3, sh_degree=3, **kw)


def test_quat_convention_matches_reference_fixture():
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_model_utils.npz"))
    R = quat_to_rotmat_torch(torch.tensor(z["quats"], dtype=torch.float64)).numpy()
    np.testing.assert_allclose(R, z["rotmats"], atol=1e-14)


def test_densify_and_prune_semantics():
    n = 300
    m = _model(n)
    opt = build_optimizers(m, 1e-3, 1e-3, 1e-3, 1e-3, 1e-3, 1e-3)
    for p in m.parameters():
        if p.requires_grad:
            p.grad = torch.randn_like(p)
    opt.step()
    g = torch.Generator().manual_seed(1)
    m.grad_norm_accum = torch.rand(n, generator=g) * 0.0006
    m.collecting_counts = torch.ones(n)
    m.collecting_counts[:10] = 0          # never seen: avg = 0/1e-8 = 0 -> not densified
    m.grad_norm_accum[:10] = 0
    m.max_radii = torch.rand(n, generator=g) * 0.2
    old = {k: getattr(m, k).detach().clone() for k in m.param_names}
    old_m = {k: opt.state[getattr(m, k)]["exp_avg"].clone() for k in m.param_names}
    avg = m.grad_norm_accum / (m.collecting_counts + 1e-8)
    high = avg >= m.DENSIFY_GRAD_THRESH
    big = torch.exp(old["log_scales"]).amax(-1) >= m.DENSIFY_SCALE_THRESH
    split, clone = big & high, (~big) & high
    prune_old = (torch.sigmoid(old["logit_opacities"]) < m.MIN_OPACITY) | (m.max_radii > m.PRUNE_RADII_RATIO_THRESH) | \
                (torch.exp(old["log_scales"]).amax(-1) > m.PRUNE_SCALE_THRESH) | split
    info = m.densify_and_prune(generator=torch.Generator().manual_seed(2))
    ns, nc = int(split.sum()), int(clone.sum())
    assert info["train/densify"] == {"split": ns, "clone": nc}
    keep_old = ~prune_old
    n_keep = int(keep_old.sum())
    # survivors come first, in order, with their Adam moments intact
    assert torch.equal(m.means[:n_keep], old["means"][keep_old])
    assert torch.equal(opt.state[m.means]["exp_avg"][:n_keep], old_m["means"][keep_old])
    assert torch.equal(opt.state[m.sh_rest]["exp_avg"][:n_keep], old_m["sh_rest"][keep_old])
    # every new Gaussian starts with zero moments; children of split parents are 1/(0.8*2) the size
    assert float(opt.state[m.means]["exp_avg"][n_keep:].abs().max()) == 0.0
    new_scales = torch.exp(m.log_scales[n_keep:])
    parent_scales = torch.exp(old["log_scales"][split]).repeat(2, 1) / 1.6
    clone_scales = torch.exp(old["log_scales"][clone])
    expect = torch.cat([parent_scales, clone_scales])
    new_prune = (torch.sigmoid(torch.cat([old["logit_opacities"][split].repeat(2), old["logit_opacities"][clone]])) < m.MIN_OPACITY) | \
                (expect.amax(-1) > m.PRUNE_SCALE_THRESH)
    assert torch.allclose(new_scales, expect[~new_prune], rtol=1e-5)
    assert m.nbr_gaussians == n_keep + int((~new_prune).sum()) == info["train/nbr_gaussians"]
    # statistics restart at zero with the new size; the optimizer keeps stepping
    assert m.grad_norm_accum.shape == (m.nbr_gaussians,) and float(m.max_radii.abs().max()) == 0.0
    for p in m.parameters():
        if p.requires_grad:
            p.grad = torch.randn_like(p)
    opt.step()
    assert opt.state[m.means]["exp_avg"].shape == m.means.shape


def test_reset_opacities():
    m = _model(50, seed=3)
    opt = build_optimizers(m, 1e-3, 1e-3, 1e-3, 1e-3, 1e-3, 1e-3)
    for p in m.parameters():
        if p.requires_grad:
            p.grad = torch.randn_like(p)
    opt.step()
    before = m.opacities.detach().clone()
    m_means = opt.state[m.means]["exp_avg"].clone()
    m.reset_opacities()
    np.testing.assert_allclose(m.opacities.detach().numpy(), np.minimum(before.numpy() * 0.5, 2 * m.MIN_OPACITY), rtol=1e-5)
    assert float(opt.state[m.logit_opacities]["exp_avg"].abs().max()) == 0.0
    assert torch.equal(opt.state[m.means]["exp_avg"], m_means)
