"""CPU tests of the oracle itself: closed-form micro-cases, the two restatements against each
other, the hand-derived backward against autograd, and the committed fixtures."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import c_oracle as CO
from oracle import torch_oracle as TO
from scenes import make_scene

GOLD = os.path.join(os.path.dirname(__file__), "golden")
F = 100.0  # focal length of the micro-case camera


def _cam(W=32, H=32):
    V = np.eye(4, dtype=np.float64)[None]
    K = np.array([[F, 0, W / 2 + 0.5], [0, F, H / 2 + 0.5], [0, 0, 1]], dtype=np.float64)[None]  # axis hits a pixel centre
    return V, K


def _one(mean, scale=0.05, opac=0.7, rgb=(0.2, 0.5, 0.9), quat=(1, 0, 0, 0)):
    return (np.array([mean], np.float64), np.array([quat], np.float64), np.full((1, 3), scale, np.float64),
            np.array([opac], np.float64), np.array([rgb], np.float64))


def test_single_gaussian_centre_alpha_equals_opacity():
    W = H = 32
    V, K = _cam(W, H)
    z = 2.0
    m, q, s, o, c = _one([0.0, 0.0, z])  # on the optical axis -> pixel centre (16.5, 16.5)
    fw = CO.render(m, q, s, o, c, V, K, W, H, sh_degree=None, backgrounds=None, dtype=np.float64)
    assert fw["radii"][0, 0] > 0
    assert abs(fw["means2d"][0, 0, 0] - 16.5) < 1e-9 and abs(fw["means2d"][0, 0, 1] - 16.5) < 1e-9
    assert abs(fw["render_alphas"][0, 16, 16, 0] - 0.7) < 1e-12
    np.testing.assert_allclose(fw["render_colors"][0, 16, 16], 0.7 * np.array([0.2, 0.5, 0.9]), atol=1e-12)
    # isotropic: cov2d = (F*s/z)^2 + eps2d on the diagonal
    var = (F * 0.05 / z) ** 2 + 0.3
    np.testing.assert_allclose(fw["conics"][0, 0], [1 / var, 0, 1 / var], atol=1e-9)
    assert fw["radii"][0, 0] == math.ceil(3 * math.sqrt(var + math.sqrt(max(0.01, 0.0))))


def test_two_gaussians_depth_order_and_background():
    W = H = 32
    V, K = _cam(W, H)
    means = np.array([[0.0, 0.0, 3.0], [0.0, 0.0, 2.0]])  # second is nearer
    quats = np.tile([1.0, 0, 0, 0], (2, 1)); scales = np.full((2, 3), 0.05)
    opac = np.array([0.6, 0.5]); cols = np.array([[1.0, 0, 0], [0, 1.0, 0]])
    bg = np.array([[0.1, 0.2, 0.3]])
    fw = CO.render(means, quats, scales, opac, cols, V, K, W, H, sh_degree=None, backgrounds=bg, dtype=np.float64)
    t = fw["isect_offsets"].reshape(-1)
    ids = fw["flatten_ids"][t[1 * 2 + 1]:]  # tile (1,1) holds pixel (16,16)
    assert list(ids[:2]) == [1, 0], "nearer Gaussian must come first"
    px = fw["render_colors"][0, 16, 16]
    a1 = 0.5 * math.exp(-0.0)  # both centred on pixel centre (16.5,16.5)
    a0 = 0.6
    expect = a1 * np.array([0, 1.0, 0]) + (1 - a1) * a0 * np.array([1.0, 0, 0]) + (1 - a1) * (1 - a0) * bg[0]
    np.testing.assert_allclose(px, expect, atol=1e-9)


def test_sh_degree0_colour_constant():
    W = H = 32
    V, K = _cam(W, H)
    m, q, s, o, _ = _one([0.0, 0.0, 2.0])
    sh = np.array([[[1.0, -0.5, 0.25]]])
    fw = CO.render(m, q, s, o, sh, V, K, W, H, sh_degree=0, dtype=np.float64)
    np.testing.assert_allclose(fw["colors"][0, 0], np.maximum(0.2820947917738781 * sh[0, 0] + 0.5, 0), atol=1e-15)


@pytest.mark.parametrize("mean,visible", [([0, 0, 0.005], False), ([0, 0, 0.02], True), ([0, 0, -1.0], False),
                                          ([50.0, 0, 2.0], False), ([0.30, 0, 2.0], True), ([0.40, 0, 2.0], False)])
def test_culling(mean, visible):
    W = H = 32
    V, K = _cam(W, H)
    m, q, s, o, c = _one(mean, scale=0.001)
    fw = CO.render(m, q, s, o, c, V, K, W, H, sh_degree=None, dtype=np.float64)
    assert (fw["radii"][0, 0] > 0) == visible


def test_alpha_threshold_cap_and_early_out():
    W = H = 16
    V, K = _cam(W, H)
    z = 2.0
    xy = -8 * z / F  # centred on pixel (0,0) -> 0.5,0.5
    def stack(opacs):
        n = len(opacs)
        means = np.tile([xy, xy, z], (n, 1)) + np.arange(n)[:, None] * np.array([0, 0, 1e-3])
        means[:, :2] *= means[:, 2:3] / z
        return (means, np.tile([1.0, 0, 0, 0], (n, 1)), np.full((n, 3), 0.5), np.array(opacs), np.tile([1.0, 1.0, 1.0], (n, 1)))
    # below 1/255: no contribution
    fw = CO.render(*stack([0.0039]), V, K, W, H, sh_degree=None, dtype=np.float64)
    assert fw["render_alphas"][0, 0, 0, 0] == 0.0
    fw = CO.render(*stack([0.00393]), V, K, W, H, sh_degree=None, dtype=np.float64)
    assert fw["render_alphas"][0, 0, 0, 0] > 0.0039
    # cap at 0.999
    fw = CO.render(*stack([1.0]), V, K, W, H, sh_degree=None, dtype=np.float64)
    assert abs(fw["render_alphas"][0, 0, 0, 0] - 0.999) < 1e-6
    # early-out: T after two capped splats = 1e-6 <= 1e-4 -> the second one is NOT blended
    fw = CO.render(*stack([1.0, 1.0, 1.0]), V, K, W, H, sh_degree=None, dtype=np.float64)
    assert abs(fw["render_alphas"][0, 0, 0, 0] - 0.999) < 1e-6
    assert fw["last_ids"][0, 0, 0] == 0


@pytest.mark.parametrize("deg,C", [(0, 1), (3, 2)])
def test_c_oracle_matches_torch_oracle_and_autograd(deg, C):
    sc = make_scene(120, 40, 28, sh_degree=deg, n_views=C, seed=5, scale_range=(0.03, 0.4), dist=4.0, k_store=16)
    dt = torch.float64
    T = lambda a: torch.tensor(a, dtype=dt)
    ins = [T(sc[k]).requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "shs")]
    img, alpha, meta = TO.rasterization(*ins, T(sc["viewmats"]), T(sc["Ks"]), 40, 28, sh_degree=deg, packed=False,
                                        backgrounds=T(sc["backgrounds"]), absgrad=True)
    g = torch.Generator().manual_seed(0)
    vc, va = torch.randn(img.shape, generator=g, dtype=dt), torch.randn(alpha.shape, generator=g, dtype=dt)
    grads = torch.autograd.grad((img * vc).sum() + (alpha * va).sum(), ins)
    fw = CO.render(sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["shs"], sc["viewmats"], sc["Ks"], 40, 28,
                   sh_degree=deg, backgrounds=sc["backgrounds"], dtype=np.float64)
    for k in ("radii", "tiles_per_gauss", "flatten_ids", "isect_offsets", "isect_ids"):
        assert np.array_equal(fw[k], meta[k].numpy()), k
    np.testing.assert_allclose(fw["render_colors"], img.detach().numpy(), atol=1e-12)
    np.testing.assert_allclose(fw["render_alphas"], alpha.detach().numpy(), atol=1e-12)
    bw = CO.backward(fw, vc.numpy(), va.numpy())
    for name, t in zip(["v_means", "v_quats", "v_scales", "v_opacities", "v_colors"], grads):
        np.testing.assert_allclose(bw[name], t.numpy(), atol=1e-9 * max(1.0, float(t.abs().max())), err_msg=name)
    np.testing.assert_allclose(bw["v_means2d_abs"], meta["means2d"].absgrad.numpy(), atol=1e-10)


def test_hand_derived_blend_backward_vs_naive_autograd():
    sc = make_scene(25, 20, 18, sh_degree=1, seed=9, scale_range=(0.1, 0.5), dist=4.0)
    dt = torch.float64
    T = lambda a: torch.tensor(a, dtype=dt)
    ins = [T(sc[k]).requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "shs")]
    V, K, bg = T(sc["viewmats"]), T(sc["Ks"]), T(sc["backgrounds"])
    img, alpha, meta = TO.rasterization(*ins, V, K, 20, 18, sh_degree=1, packed=False, backgrounds=bg)
    radii, m2, dep, con = TO.project(*ins[:3], V, K, 20, 18)
    cols = TO.spherical_harmonics(1, ins[0], V, ins[4], radii)
    img2, a2 = TO.naive_blend_autograd(m2, con, cols, ins[3][None], bg, 20, 18, 16, meta["isect_offsets"], meta["flatten_ids"])
    g = torch.Generator().manual_seed(3)
    vc, va = torch.randn(img.shape, generator=g, dtype=dt), torch.randn(alpha.shape, generator=g, dtype=dt)
    g1 = torch.autograd.grad((img * vc).sum() + (alpha * va).sum(), ins)
    g2 = torch.autograd.grad((img2 * vc).sum() + (a2 * va).sum(), ins)
    assert (img - img2).abs().max() < 1e-12
    for a, b in zip(g1, g2):
        assert (a - b).abs().max() <= 1e-10 * max(1.0, float(b.abs().max()))


def test_reference_conventions_fixture():
    """Vectors captured from the reference's own model/utils.py (tests/golden/make_golden.py)."""
    z = np.load(os.path.join(GOLD, "ref_model_utils.npz"))
    R = TO.quat_to_rotmat(torch.tensor(z["quats"], dtype=torch.float64)).numpy()
    np.testing.assert_allclose(R, z["rotmats"], atol=1e-14)
    np.testing.assert_allclose((z["rgbs"] - 0.5) / TO.SH_C0, z["sh0"], atol=1e-14)
    # reference README example: quat (1,2,3,4)
    i = int(np.where((z["quats"] == np.array([1.0, 2, 3, 4])).all(1))[0][0])
    np.testing.assert_allclose(z["rotmats"][i], [[-2 / 3, 2 / 15, 11 / 15], [2 / 3, -1 / 3, 2 / 3], [1 / 3, 14 / 15, 2 / 15]], atol=1e-12)


@pytest.mark.parametrize("tag", ["a", "b", "c", "d", "e", "f"])
def test_c_oracle_reproduces_golden_scenes(tag):
    z = np.load(os.path.join(GOLD, f"oracle_scene_{tag}.npz"))
    for dtype, tol in ((np.float64, 1e-11), (np.float32, 2e-5)):
        fw = CO.render(z["means"], z["quats"], z["scales"], z["opacities"], z["shs"], z["viewmats"], z["Ks"],
                       int(z["width"]), int(z["height"]), sh_degree=int(z["sh_degree"]), backgrounds=z["backgrounds"], dtype=dtype)
        for k in ("radii", "tiles_per_gauss", "flatten_ids", "isect_offsets"):
            assert np.array_equal(fw[k], z[k]), (k, dtype)
        np.testing.assert_allclose(fw["render_colors"], z["render_colors"], atol=tol)
        np.testing.assert_allclose(fw["render_alphas"], z["render_alphas"], atol=tol)
        bw = CO.backward(fw, z["v_render_colors"], z["v_render_alphas"])
        for a, b in (("v_means", "v_means"), ("v_quats", "v_quats"), ("v_scales", "v_scales"),
                     ("v_opacities", "v_opacities"), ("v_colors", "v_shs"), ("v_means2d_abs", "absgrad")):
            scale = max(1.0, float(np.abs(z[b]).max()))
            assert np.abs(bw[a] - z[b]).max() <= (1e-9 if dtype == np.float64 else 2e-4) * scale, (a, dtype)


GSPLAT_FIXTURES = sorted(f for f in os.listdir(GOLD) if f.startswith("gsplat_") and f.endswith(".npz"))


@pytest.mark.skipif(not GSPLAT_FIXTURES, reason="no tests/golden/gsplat_*.npz: gsplat is not installable here -- run tests/golden/make_gsplat_golden.py "
                                                "where gsplat==1.0.0 exists and commit the files (the oracle stays PARITY UNPINNED until then)")
@pytest.mark.parametrize("name", GSPLAT_FIXTURES or ["-"])
def test_oracle_matches_gsplat_fixtures(name):
    """THE PIN: the C oracle against outputs of gsplat 1.0.0 itself on the same inputs (tests/golden/make_gsplat_golden.py).  Integer
    outputs bit-exact up to the isolated last-bit flips fp32 permits (radius = ceil(3 sigma) +- 1 on <= 1e-4 of the Gaussians);
    the image to 1e-4, gradients to 1e-3 of each tensor's largest entry -- north_star's bars, here between the oracle and upstream."""
    z = np.load(os.path.join(GOLD, name))
    fw = CO.render(z["means"], z["quats"], z["scales"], z["opacities"], z["shs"], z["viewmats"], z["Ks"], int(z["width"]), int(z["height"]),
                   sh_degree=int(z["sh_degree"]), backgrounds=z["backgrounds"], dtype=np.float32)
    n = z["radii"].size
    flips = int((fw["radii"] != z["radii"]).sum())
    assert flips <= max(1, int(1e-4 * n)) and (flips == 0 or np.abs(fw["radii"].astype(np.int64) - z["radii"]).max() <= 1), ("radii", flips)
    if flips == 0:
        for k in ("tiles_per_gauss", "isect_offsets", "flatten_ids"):
            assert np.array_equal(fw[k].reshape(-1), z[k].reshape(-1)), k
    vis = (z["radii"] > 0) & (fw["radii"] > 0)
    assert np.abs(fw["means2d"] - z["means2d"])[vis].max() <= 1e-3 and np.abs(fw["conics"] - z["conics"])[vis].max() <= 1e-4 * max(1.0, np.abs(z["conics"][vis]).max())
    assert np.abs(fw["render_colors"] - z["render_colors"]).max() <= 1e-4 + (1.0 if flips else 0.0) * 1e-2
    assert np.abs(fw["render_alphas"] - z["render_alphas"]).max() <= 1e-4 + (1.0 if flips else 0.0) * 1e-2
    bw = CO.backward(fw, z["v_render_colors"].astype(np.float32), z["v_render_alphas"].astype(np.float32))
    for a, b in (("v_means", "v_means"), ("v_quats", "v_quats"), ("v_scales", "v_scales"), ("v_opacities", "v_opacities"), ("v_colors", "v_shs"),
                 ("v_means2d_abs", "absgrad")):
        assert np.abs(bw[a] - z[b]).max() <= 1e-3 * np.abs(z[b]).max(), (a, float(np.abs(bw[a] - z[b]).max()), float(np.abs(z[b]).max()))
