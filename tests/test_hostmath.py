"""The per-Gaussian device math (easy_gaussian_splatting_amd/csrc/gs_math.h) compiled for the host
and checked against the oracle on CPU -- the same source the gfx950 kernels compile."""
import ctypes as ct
import os
import subprocess

import numpy as np
import pytest

from oracle import c_oracle as CO
from scenes import make_scene

HM = os.path.join(os.path.dirname(__file__), "hostmath")


@pytest.fixture(scope="module")
def hm():
    so = os.path.join(HM, "libhostmath.so")
    subprocess.run(["g++", "-O2", "-fPIC", "-shared", "-o", so, os.path.join(HM, "hostmath.cpp")], check=True)
    return ct.CDLL(so)


def _p(a):
    return a.ctypes.data_as(ct.c_void_p)


@pytest.mark.parametrize("use_jac", [0, 1])   # 1: the coefficient-free SH backward (direction Jacobian from the forward)
@pytest.mark.parametrize("deg,C,K", [(0, 1, 1), (1, 1, 4), (2, 2, 16), (3, 2, 16)])
def test_projection_sh_forward_and_vjp(hm, deg, C, K, use_jac):
    N, W, H = 1500, 96, 64
    sc = make_scene(N, W, H, sh_degree=deg, n_views=C, seed=11 + deg, k_store=K, scale_range=(0.02, 0.4), dist=3.0)
    fw = CO.render(sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["shs"], sc["viewmats"], sc["Ks"], W, H,
                   sh_degree=deg, dtype=np.float64)
    F = ct.c_float
    radii = np.zeros((C, N), np.int32); m2 = np.zeros((C, N, 2), np.float32); dep = np.zeros((C, N), np.float32)
    con = np.zeros((C, N, 3), np.float32); col = np.zeros((C, N, 3), np.float32); tpg = np.zeros((C, N), np.int32)
    hm.hm_forward(C, N, K, deg, _p(sc["means"]), _p(sc["quats"]), _p(sc["scales"]), _p(sc["shs"]), _p(sc["viewmats"]),
                  _p(sc["Ks"]), W, H, 16, F(0.3), F(0.01), F(1e10), F(0.0), _p(radii), _p(m2), _p(dep), _p(con), _p(col), _p(tpg))
    vis = fw["radii"] > 0
    assert vis.sum() > N // 4
    assert (radii != fw["radii"]).mean() < 2e-3  # ceil() may flip on fp32 rounding, rarely
    same = radii == fw["radii"]
    assert np.abs(m2 - fw["means2d"])[same].max() < 1e-3
    assert np.abs(dep - fw["depths"])[same].max() < 1e-5
    rel = np.abs(con - fw["conics"]) / (np.abs(fw["conics"]) + 1e-2)
    assert rel[same].max() < 1e-3
    assert np.abs(col - fw["colors"])[same & vis].max() < 1e-5
    assert (tpg != fw["tiles_per_gauss"])[same].mean() < 2e-3

    rng = np.random.default_rng(0)
    vm, vcn, vc = rng.standard_normal((C, N, 2)), rng.standard_normal((C, N, 3)), rng.standard_normal((C, N, 3))
    Lo = CO._lib(np.float64)
    D = ct.c_double
    a64 = {k: sc[k].astype(np.float64) for k in ("means", "quats", "scales", "shs", "viewmats", "Ks")}
    v_means = np.zeros((N, 3)); v_quats = np.zeros((N, 4)); v_scales = np.zeros((N, 3)); v_shs = np.zeros((N, K, 3))
    rad_use = np.ascontiguousarray(np.where(same, fw["radii"], 0).astype(np.int32))
    Lo.gso_sh_bwd(C, N, K, deg, _p(a64["means"]), _p(a64["viewmats"]), _p(a64["shs"]), _p(rad_use), _p(fw["colors"]),
                  _p(vc), _p(v_shs), _p(v_means))
    Lo.gso_project_bwd(C, N, _p(a64["means"]), _p(a64["quats"]), _p(a64["scales"]), _p(a64["viewmats"]), _p(a64["Ks"]),
                       W, H, D(float(np.float32(0.3))), D(float(np.float32(0.01))), D(1e10), _p(rad_use), _p(vm), None, _p(vcn), _p(v_means), _p(v_quats), _p(v_scales))
    h_means = np.zeros((N, 3), np.float32); h_quats = np.zeros((N, 4), np.float32)
    h_scales = np.zeros((N, 3), np.float32); h_shs = np.zeros((N, K, 3), np.float32)
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    hm.hm_backward(C, N, K, deg, _p(sc["means"]), _p(sc["quats"]), _p(sc["scales"]), _p(sc["shs"]), _p(sc["viewmats"]),
                   _p(sc["Ks"]), W, H, F(0.3), F(0.01), F(1e10), _p(rad_use), _p(f32(fw["colors"])), _p(f32(vm)), _p(f32(vcn)),
                   _p(f32(vc)), _p(h_means), _p(h_quats), _p(h_scales), _p(h_shs), use_jac)
    for name, a, b in (("means", h_means, v_means), ("quats", h_quats, v_quats), ("scales", h_scales, v_scales), ("shs", h_shs, v_shs)):
        assert np.abs(a - b).max() <= 1e-3 * np.abs(b).max(), name  # north_star: grads within 1e-3 rel


def test_opacity_aware_extent_is_conservative(hm):
    """Every pixel the blend could accept (alpha >= 1/255, evaluated in fp64) lies inside the
    extent box, and inside a tile the tight rectangle keeps."""
    rng = np.random.default_rng(3)
    n, W, H = 400, 160, 120
    mx, my = rng.uniform(-10, W + 10, n), rng.uniform(-10, H + 10, n)
    sx, sy = rng.uniform(0.6, 12, n), rng.uniform(0.6, 12, n)
    rho = rng.uniform(-0.95, 0.95, n)
    cxx, cyy, cxy = sx * sx, sy * sy, rho * sx * sy
    opac = np.concatenate([rng.uniform(0.0, 0.01, n // 4), rng.uniform(0.01, 1.0, n - n // 4)])
    det = cxx * cyy - cxy * cxy
    A, B, Cc = cyy / det, -cxy / det, cxx / det
    lam = 0.5 * (cxx + cyy) + np.sqrt(np.maximum(0.01, (0.5 * (cxx + cyy)) ** 2 - det))
    radius = np.ceil(3 * np.sqrt(lam)).astype(np.int32)
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    ex, ey = np.zeros(n, np.float32), np.zeros(n, np.float32)
    rg, rt = np.zeros((n, 4), np.int32), np.zeros((n, 4), np.int32)
    hm.hm_extents(n, _p(f32(opac)), _p(f32(cxx)), _p(f32(cyy)), _p(f32(mx)), _p(f32(my)), _p(radius), W, H, 16,
                  _p(ex), _p(ey), _p(rg), _p(rt))
    jj, ii = np.meshgrid(np.arange(W) + 0.5, np.arange(H) + 0.5)
    checked = 0
    for g in range(n):
        dx, dy = mx[g] - jj, my[g] - ii
        sigma = 0.5 * (A[g] * dx * dx + Cc[g] * dy * dy) + B[g] * dx * dy
        hit = opac[g] * np.exp(-sigma) >= (1 / 255.0) * (1 - 1e-6)
        # restrict to gsplat's own tile rectangle (what the reference would list at all)
        tx, ty = (jj // 16).astype(int), (ii // 16).astype(int)
        hit &= (tx >= rg[g, 0]) & (tx < rg[g, 1]) & (ty >= rg[g, 2]) & (ty < rg[g, 3])
        assert rt[g, 0] >= rg[g, 0] and rt[g, 1] <= rg[g, 1] and rt[g, 2] >= rg[g, 2] and rt[g, 3] <= rg[g, 3]
        if hit.any():
            checked += 1
            assert np.abs(dx[hit]).max() <= ex[g] and np.abs(dy[hit]).max() <= ey[g]
            assert ((tx[hit] >= rt[g, 0]) & (tx[hit] < rt[g, 1]) & (ty[hit] >= rt[g, 2]) & (ty[hit] < rt[g, 3])).all()
    assert checked > n // 3
