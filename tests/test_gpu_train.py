"""End-to-end: the reference's train step (forward -> L1+SSIM -> backward -> statistics -> Adam)
driven through the HIP path actually fits an image."""
import numpy as np
import pytest
import torch

from easy_gaussian_splatting_amd.loss import LossComputer
from easy_gaussian_splatting_amd.model import GaussianModel, build_optimizers
from scenes import make_scene

pytestmark = pytest.mark.gpu


def _model(sc, dev, noise, seed):
    g = torch.Generator().manual_seed(seed)
    T = lambda a: torch.from_numpy(a)
    op = np.clip(sc["opacities"], 1e-3, 1 - 1e-3)
    shs = T(sc["shs"])
    return GaussianModel(means=T(sc["means"]) + noise * 0.03 * torch.randn(sc["means"].shape, generator=g),
                         log_scales=torch.log(T(sc["scales"])) + noise * 0.2 * torch.randn(sc["scales"].shape, generator=g),
                         quats=T(sc["quats"]), sh_0=(shs[:, :1] + noise * 0.5 * torch.randn(shs[:, :1].shape, generator=g)).contiguous(),
                         sh_rest=shs[:, 1:].contiguous() * (1 - noise),
                         logit_opacities=T(np.log(op / (1 - op)).astype(np.float32)), sh_degree=3, white_background=True).to(dev)


@pytest.mark.parametrize("fused", ["hip", True])
def test_training_fits_target_views(fused):
    dev = torch.device("cuda:0")
    sc = make_scene(4000, 192, 128, sh_degree=3, n_views=4, seed=5, scale_range=(0.03, 0.15), dist=4.0)
    target_model = _model(sc, dev, 0.0, 0)
    datas = [{"w2c": torch.from_numpy(sc["viewmats"][v]).to(dev), "K": torch.from_numpy(sc["Ks"][v]).to(dev), "width": 192, "height": 128}
             for v in range(4)]
    with torch.no_grad():
        targets = [target_model(d)["render_img"] for d in datas]
    model = _model(sc, dev, 1.0, 1)
    opt = build_optimizers(model, 1.6e-3, 5e-3, 1e-3, 2.5e-2, 1.25e-3, 5e-2, fused=fused)
    lc = LossComputer(0.2)
    mask = torch.zeros(128, 192, device=dev)
    losses = []
    for it in range(240):
        v = it % 4
        out = model(datas[v])
        loss = lc.get_loss_dict(out["render_img"], targets[v], mask)["total"]
        loss.backward()
        model.update_statistics(datas[v], out)
        opt.step()
        opt.zero_grad()
        losses.append(loss.detach())
    losses = torch.stack(losses).cpu().numpy()
    first, last = losses[:8].mean(), losses[-8:].mean()
    assert np.isfinite(losses).all()
    assert last < 0.45 * first, (first, last)
    assert float(model.collecting_counts.max()) > 0 and float(model.grad_norm_accum.max()) > 0


@pytest.mark.parametrize("fused", ["hip", True])
def test_refinement_inside_the_train_loop(fused):
    """densify_and_prune / reset_opacities change N between steps; the rasterizer, the statistics
    and both optimizer flavours must follow."""
    dev = torch.device("cuda:0")
    sc = make_scene(3000, 160, 112, sh_degree=3, n_views=2, seed=6, scale_range=(0.02, 0.1), dist=4.0)
    target_model = _model(sc, dev, 0.0, 0)
    datas = [{"w2c": torch.from_numpy(sc["viewmats"][v]).to(dev), "K": torch.from_numpy(sc["Ks"][v]).to(dev), "width": 160, "height": 112}
             for v in range(2)]
    with torch.no_grad():
        targets = [target_model(d)["render_img"] for d in datas]
    model = _model(sc, dev, 1.0, 2)
    model.DENSIFY_GRAD_THRESH = 1e-5
    opt = build_optimizers(model, 1.6e-3, 5e-3, 1e-3, 2.5e-2, 1.25e-3, 5e-2, fused=fused)
    lc = LossComputer(0.2)
    sizes, losses = [model.nbr_gaussians], []
    for it in range(130):
        v = it % 2
        out = model(datas[v])
        loss = lc.get_loss_dict(out["render_img"], targets[v])["total"]
        loss.backward()
        model.update_statistics(datas[v], out)
        opt.step()
        opt.zero_grad()
        losses.append(float(loss.detach()))
        if it in (30, 60):
            info = model.densify_and_prune()
            sizes.append(info["train/nbr_gaussians"])
            assert model.means.shape[0] == model.sh_rest.shape[0] == model.grad_norm_accum.shape[0] == sizes[-1]
        if it == 45:
            model.reset_opacities()
    assert sizes[1] != sizes[0]
    # refinement and the opacity reset perturb the fit; it must recover and end below where it started
    assert np.isfinite(losses).all() and np.mean(losses[-6:]) < 0.9 * np.mean(losses[:6])
