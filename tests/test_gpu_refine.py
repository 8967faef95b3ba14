"""Device-side densify / prune (csrc/gs_refine.hip, row f-3) against the torch mirror of
/root/reference/model/gaussian.py:199-349 (model.densify_and_prune's host path): same survivors in the same order,
same children and clones, moments carried for survivors and zero for new Gaussians, same tb_info -- at 1 M Gaussians."""
import time

import numpy as np
import pytest
import torch

from easy_gaussian_splatting_amd.model import GaussianModel, build_optimizers

pytestmark = pytest.mark.gpu


def _away(x, thr, rel=1e-3):
    """Moves values that sit within `rel` of a decision threshold off it (both implementations then take the same
    side whatever their last-ulp differences in exp / sigmoid)."""
    return torch.where((x - thr).abs() < rel * thr, x * (1 + 4 * rel), x)


def _make(n, dev, seed, fused="hip", K=16):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    scales = torch.exp(torch.rand(n, 3, generator=g) * 4.2 - 6.4)                 # 0.0017 .. 0.11: straddles 0.01 and 0.1
    for thr in (0.01, 0.1, 0.016, 0.16):                                          # (children: scale / 1.6)
        scales = _away(scales, thr)
    opac = _away(torch.sigmoid(r(n) * 3.0), 0.005)
    m = GaussianModel(means=r(n, 3), log_scales=torch.log(scales), quats=r(n, 4), sh_0=r(n, 1, 3), sh_rest=r(n, K - 1, 3) * 0.1,
                      logit_opacities=torch.logit(opac), sh_degree=3).to(dev)
    opt = build_optimizers(m, 1.6e-4, 5e-3, 1e-3, 2.5e-3, 1.25e-4, 5e-2, fused=fused)
    for name in m.param_names:   # two Adam steps: non-trivial moments
        getattr(m, name).grad = torch.randn(getattr(m, name).shape, generator=g).to(dev)
    opt.step()
    m.grad_norm_accum = _away(torch.rand(n, generator=g) * 6e-4, 2e-4).to(dev)
    m.collecting_counts = (torch.rand(n, generator=g) > 0.1).float().to(dev)       # 10 % never seen
    m.max_radii = _away(torch.rand(n, generator=g) * 0.2, 0.15).to(dev)
    return m, opt


@pytest.mark.parametrize("n", [1, 777, 1_000_000])
def test_device_refine_equals_torch_mirror(n):
    dev = torch.device("cuda:0")
    (ma, oa), (mb, ob) = _make(n, dev, 5), _make(n, dev, 5)
    mb.device_refine = False                                                       # torch mirror (host path)
    old_means = ma.means.detach().clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ia = ma.densify_and_prune(generator=torch.Generator(device=dev).manual_seed(9))
    torch.cuda.synchronize()
    t_dev = time.perf_counter() - t0
    t0 = time.perf_counter()
    ib = mb.densify_and_prune(generator=torch.Generator(device=dev).manual_seed(9))
    torch.cuda.synchronize()
    t_host = time.perf_counter() - t0
    print(f"[refine] n={n}: device path {1e3 * t_dev:.2f} ms, torch mirror {1e3 * t_host:.2f} ms, {ia}")
    assert ia == ib
    n_new = ia["train/nbr_gaussians"]
    if n == 1_000_000:
        assert ia["train/densify"]["split"] > 10000 and ia["train/densify"]["clone"] > 10000 and sum(ia["train/prune"].values()) > 10000
    for k in ma.param_names:
        a, b = getattr(ma, k).detach(), getattr(mb, k).detach()
        assert a.shape == b.shape and a.shape[0] == n_new, k
        if n_new == 0:
            continue
        if k in ("means", "log_scales"):      # split children: mean + R s noise, log(s / 1.6) -- same formula, libm vs torch ulps
            assert float((a - b).abs().max()) <= 2e-6 * max(1.0, float(b.abs().max())), k
        else:
            assert torch.equal(a, b), k
        for x, y in zip(oa.moments_of(getattr(ma, k)), ob.moments_of(getattr(mb, k))):
            assert torch.equal(x, y), (k, "moments")
    # survivors lead, in their old order, bit for bit
    surv = int((oa.moments_of(ma.means)[1].abs().sum(-1) > 0).sum())   # (second moments are > 0 exactly for survivors)
    assert torch.equal(ma.means[:surv], mb.means[:surv])
    if surv:   # every survivor is an old Gaussian: its x coordinate occurs among the old ones
        srt = old_means[:, 0].contiguous().sort().values
        pos = torch.searchsorted(srt, ma.means[:surv, 0].contiguous()).clamp(max=n - 1)
        assert torch.equal(srt[pos], ma.means[:surv, 0])
    for buf in (ma.grad_norm_accum, ma.collecting_counts, ma.max_radii):
        assert buf.shape == (n_new,) and (n_new == 0 or float(buf.abs().max()) == 0.0)
    # the optimizer keeps stepping on the adopted buffers, identically on both sides
    g = torch.Generator().manual_seed(1)
    for name in ma.param_names:
        gr = torch.randn(getattr(ma, name).shape, generator=g).to(dev)
        getattr(ma, name).grad = gr.clone(); getattr(mb, name).grad = gr.clone()
    oa.step(); ob.step()
    for k in ("quats", "sh_rest", "logit_opacities"):
        assert torch.equal(getattr(ma, k).detach(), getattr(mb, k).detach()), k
    if n == 1_000_000:
        assert t_dev < 0.02, f"device refine took {1e3 * t_dev:.1f} ms"


def test_device_refine_nothing_to_do_and_everything_pruned():
    dev = torch.device("cuda:0")
    m, opt = _make(500, dev, 2)
    m.DENSIFY_GRAD_THRESH = 1e9          # nothing densified
    m.MIN_OPACITY, m.PRUNE_RADII_RATIO_THRESH, m.PRUNE_SCALE_THRESH = 0.0, 1e9, 1e9   # nothing pruned
    before = {k: getattr(m, k).detach().clone() for k in m.param_names}
    info = m.densify_and_prune()
    assert info["train/nbr_gaussians"] == 500 and info["train/densify"] == {"split": 0, "clone": 0}
    for k in m.param_names:
        assert torch.equal(getattr(m, k).detach(), before[k])
    m.MIN_OPACITY = 2.0                  # everything is "transparent": all pruned
    info = m.densify_and_prune()
    assert info["train/nbr_gaussians"] == 0 and m.means.shape == (0, 3) and m.sh_rest.shape == (0, 15, 3)


@pytest.mark.parametrize("n", [1, 63, 16384, 16385, 1_000_003])
def test_row_scan_equals_cumsum(n):
    """gs_scan_rows_i32 (the refinement's prefix scans): every row's inclusive scan, block and chunk boundaries included."""
    from easy_gaussian_splatting_amd import _native as nat
    L, d = nat.lib(), torch.device("cuda:0")
    x = torch.randint(0, 3, (3, n), dtype=torch.int32, device=d, generator=torch.Generator(device=d).manual_seed(n))
    out = torch.empty_like(x)
    ws = torch.empty((int(L.gs_scan_rows_workspace_ints(3, n)),), dtype=torch.int32, device=d)
    nat.check(L.gs_scan_rows_i32(torch.cuda.current_stream().cuda_stream, 3, n, x.data_ptr(), out.data_ptr(), ws.data_ptr()), "gs_scan_rows_i32")
    assert torch.equal(out, torch.cumsum(x, 1, dtype=torch.int32))
