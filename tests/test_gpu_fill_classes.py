"""The blend backward's fill classes (include/gs_raster.h, gs_blend_bwd's `unit_classes`): the work units grouped by how many of
their 32 entry slots are taken, every wave running eight units of one class at 1 / 2 / 3 / 4 entries per lane.  A unit writes
gradient rows of its own, so the grouping -- and the order the atomics of the grouping pass happen to produce -- must change
nothing: every gradient equals the ungrouped backward's, bit for bit."""
import numpy as np
import pytest
import torch

from easy_gaussian_splatting_amd import _native as nat
from easy_gaussian_splatting_amd import rendering
from scenes import make_scene

pytestmark = pytest.mark.gpu


def _grads(sc, on, monkeypatch, dbg=None):
    monkeypatch.setattr(rendering, "_BWD_CLASSES", on)
    dev = torch.device("cuda:0")
    t = {k: torch.from_numpy(v).to(dev) for k, v in sc.items() if isinstance(v, np.ndarray)}
    ins = [t[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "shs")]
    img, alpha, meta = rendering.rasterization(*ins, t["viewmats"], t["Ks"], int(sc["width"]), int(sc["height"]), sh_degree=int(sc["sh_degree"]),
                                               packed=False, backgrounds=t["backgrounds"], absgrad=True, _debug=dbg)
    vc = torch.randn(img.shape, generator=torch.Generator().manual_seed(0)).to(dev)
    va = torch.randn(alpha.shape, generator=torch.Generator().manual_seed(1)).to(dev)
    g = torch.autograd.grad((img * vc).sum() + (alpha * va).sum(), ins)
    torch.cuda.synchronize()
    return [x.clone() for x in g] + [meta["means2d"].absgrad.clone()]


@pytest.mark.parametrize("n,w,h,views", [(1, 16, 16, 1), (300, 64, 48, 1), (3000, 161, 97, 1), (8000, 320, 208, 1), (5000, 200, 120, 3),
                                         (60000, 640, 368, 1)])
def test_grouped_backward_equals_ungrouped_bit_for_bit(n, w, h, views, monkeypatch):
    sc = make_scene(n, w, h, sh_degree=3, n_views=views, seed=7 + n, scale_range=(0.02, 0.2), dist=4.0)
    g_off = _grads(sc, False, monkeypatch)
    g_on = _grads(sc, True, monkeypatch)
    for name, a, b in zip(("means", "quats", "scales", "opacities", "shs", "absgrad"), g_off, g_on):
        assert torch.equal(a, b), name
    assert any(float(x.abs().max()) > 0 for x in g_on) or n == 1


def test_classes_cover_every_tail_length(monkeypatch):
    """A frame whose quadrant sublists end in tails of every length 1 .. 32: all four classes are populated (checked on the host
    from the sublist lengths the forward leaves), and the grouped backward still equals the ungrouped one."""
    sc = make_scene(20000, 480, 272, sh_degree=1, n_views=1, seed=3, scale_range=(0.01, 0.15), dist=4.0)
    dbg = {}
    g_on = _grads(sc, True, monkeypatch, dbg)
    L = dbg["qcnt"].cpu().numpy().astype(np.int64)
    L = L[L > 0]
    tail = L - (np.ceil(L / nat.GS_UNIT).astype(np.int64) - 1) * nat.GS_UNIT
    assert set(np.ceil(tail / 8).astype(int).tolist()) == {1, 2, 3, 4}
    assert int(np.ceil(L / nat.GS_UNIT).sum()) == int(dbg["unit_counter"][0])
    g_off = _grads(sc, False, monkeypatch)
    for a, b in zip(g_off, g_on):
        assert torch.equal(a, b)
