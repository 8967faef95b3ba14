"""Host-side mirror pieces that run on CPU: SSIM restatement, activations, optimizer groups."""
import numpy as np
import torch

from easy_gaussian_splatting_amd.loss import LossComputer, ssim
from easy_gaussian_splatting_amd.model import GaussianModel, build_optimizers


def _ssim_naive(a, b):
    """Direct (non-separable) 11x11 evaluation for one channel pair, interior pixels only."""
    import torch.nn.functional as F
    x = torch.arange(11, dtype=torch.float64) - 5
    g = torch.exp(-(x / 1.5) ** 2 / 2); g = g / g.sum()
    w = (g[:, None] * g[None, :])[None, None]
    pa, pb = (F.pad(t[None, None], (5, 5, 5, 5), mode="reflect") for t in (a, b))
    mu_a, mu_b = F.conv2d(pa, w), F.conv2d(pb, w)
    saa = F.conv2d(pa * pa, w) - mu_a ** 2; sbb = F.conv2d(pb * pb, w) - mu_b ** 2; sab = F.conv2d(pa * pb, w) - mu_a * mu_b
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu_a * mu_b + c1) * (2 * sab + c2)) / ((mu_a ** 2 + mu_b ** 2 + c1) * (saa + sbb + c2))
    return m[..., 5:-5, 5:-5].mean()


def test_ssim_matches_direct_evaluation_and_bounds():
    g = torch.Generator().manual_seed(0)
    a, b = torch.rand(1, 3, 40, 52, generator=g, dtype=torch.float64), torch.rand(1, 3, 40, 52, generator=g, dtype=torch.float64)
    assert abs(ssim(a, a).item() - 1.0) < 1e-12
    ref = torch.stack([_ssim_naive(a[0, c], b[0, c]) for c in range(3)]).mean()
    assert abs(ssim(a, b).item() - ref.item()) < 1e-12
    assert ssim(a, b).item() < 0.2


def test_loss_composition_and_mask():
    g = torch.Generator().manual_seed(1)
    r, t = torch.rand(32, 48, 3, generator=g), torch.rand(32, 48, 3, generator=g)
    lc = LossComputer(lambda_ssim=0.2)
    d = lc.get_loss_dict(r, t, torch.zeros(32, 48))
    assert abs(d["total"].item() - (0.8 * d["l1"].item() + 0.2 * d["ssim"].item())) < 1e-6
    full = lc.get_loss_dict(r, t, torch.ones(32, 48))
    assert full["l1"].item() == 0.0 and abs(full["ssim"].item()) < 1e-6


def test_model_activations_and_optimizer_groups():
    N = 10
    m = GaussianModel(means=torch.zeros(N, 3), log_scales=torch.full((N, 3), np.log(0.1)), quats=torch.ones(N, 4),
                      sh_0=torch.zeros(N, 1, 3), sh_rest=torch.zeros(N, 15, 3), logit_opacities=torch.zeros(N),
                      sh_degree=3, sh_degree_interval=2000, white_background=True)
    assert torch.allclose(m.scales, torch.full((N, 3), 0.1)) and torch.allclose(m.opacities, torch.full((N,), 0.5))
    assert m.shs.shape == (N, 16, 3) and m.active_sh_degree == 0 and m.BACKGROUND.tolist() == [1.0, 1.0, 1.0]
    for _ in range(5):
        m.up_sh_degree()
    assert m.active_sh_degree == 3
    opt = build_optimizers(m, 1.6e-4, 5e-3, 1e-3, 2.5e-3, 1.25e-4, 5e-2)
    assert [g["name"] for g in opt.param_groups] == m.param_names
    assert [g["lr"] for g in opt.param_groups] == [1.6e-4, 5e-3, 1e-3, 2.5e-3, 1.25e-4, 5e-2]
