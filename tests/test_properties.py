"""Property tests (hypothesis) on the oracle and on the host build of the device math:
invariances the path must have regardless of size (SURVEY.md section 4 'Property tests')."""
import ctypes as ct
import os
import subprocess

import numpy as np
from hypothesis import given, settings, strategies as st

from oracle import c_oracle as CO
from scenes import make_scene

HM = os.path.join(os.path.dirname(__file__), "hostmath")


def _render(sc, **kw):
    return CO.render(sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["shs"], sc["viewmats"], sc["Ks"],
                     sc["width"], sc["height"], sh_degree=sc["sh_degree"], backgrounds=sc["backgrounds"], dtype=np.float64, **kw)


@settings(max_examples=15, deadline=None)
@given(seed=st.integers(0, 10_000), n=st.integers(1, 200), w=st.integers(17, 70), h=st.integers(17, 50))
def test_permutation_invariance_and_list_structure(seed, n, w, h):
    sc = make_scene(n, w, h, sh_degree=1, seed=seed, scale_range=(0.03, 0.4), dist=4.0)
    a = _render(sc)
    perm = np.random.default_rng(seed).permutation(n)
    sp = dict(sc)
    for k in ("means", "quats", "scales", "opacities", "shs"):
        sp[k] = np.ascontiguousarray(sc[k][perm])
    b = _render(sp)
    # identical image up to summation order inside equal-depth ties (none in random scenes)
    assert np.abs(a["render_colors"] - b["render_colors"]).max() < 1e-12
    assert a["n_isects"] == b["n_isects"] == int(a["tiles_per_gauss"].sum())
    offs = np.append(a["isect_offsets"].reshape(-1), a["n_isects"])
    assert (np.diff(offs) >= 0).all()
    assert (np.diff(a["isect_ids"]) >= 0).all()
    assert (a["render_alphas"] >= 0).all() and (a["render_alphas"] <= 1).all()


@settings(max_examples=15, deadline=None)
@given(seed=st.integers(0, 10_000), scale=st.floats(0.25, 4.0))
def test_quaternion_scale_invariance(seed, scale):
    """quats are passed un-normalised (/root/reference/model/gaussian.py:355): q and s*q render alike,
    and the gradient wrt q is orthogonal to q."""
    sc = make_scene(60, 40, 30, sh_degree=0, seed=seed, scale_range=(0.05, 0.4), dist=4.0)
    a = _render(sc)
    s2 = dict(sc); s2["quats"] = (sc["quats"] * scale).astype(np.float32)
    b = _render(s2)
    assert np.abs(a["render_colors"] - b["render_colors"]).max() < 1e-5
    vc = np.random.default_rng(seed).standard_normal(a["render_colors"].shape)
    bw = CO.backward(a, vc)
    dots = np.abs((bw["v_quats"] * sc["quats"]).sum(-1))
    assert dots.max() <= 1e-9 * max(1.0, np.abs(bw["v_quats"]).max())


@settings(max_examples=10, deadline=None)
@given(seed=st.integers(0, 10_000))
def test_backward_is_linear_in_upstream_gradient(seed):
    sc = make_scene(80, 48, 32, sh_degree=2, seed=seed, scale_range=(0.05, 0.4), dist=4.0)
    fw = _render(sc)
    rng = np.random.default_rng(seed)
    v1, v2 = rng.standard_normal(fw["render_colors"].shape), rng.standard_normal(fw["render_colors"].shape)
    g1, g2, g12 = CO.backward(fw, v1), CO.backward(fw, v2), CO.backward(fw, v1 + 2.0 * v2)
    for k in ("v_means", "v_quats", "v_scales", "v_opacities", "v_colors"):
        assert np.abs(g1[k] + 2.0 * g2[k] - g12[k]).max() <= 1e-9 * max(1.0, np.abs(g12[k]).max())


def test_tile_boundary_gaussians_host_math():
    """Means exactly on tile corners / edges: the device tile rectangle equals the oracle's."""
    so = os.path.join(HM, "libhostmath.so")
    subprocess.run(["g++", "-O2", "-fPIC", "-shared", "-o", so, os.path.join(HM, "hostmath.cpp")], check=True)
    hm = ct.CDLL(so)
    W = H = 64
    xs = np.array([0.0, 16.0, 31.999, 32.0, 48.0, 63.999, 64.0], np.float32)
    mx, my = np.meshgrid(xs, xs)
    mx, my = mx.reshape(-1), my.reshape(-1)
    n = mx.size
    for radius in (1, 15, 16, 17, 40):
        rad = np.full(n, radius, np.int32)
        ex = np.zeros(n, np.float32); ey = np.zeros(n, np.float32); rg = np.zeros((n, 4), np.int32); rt = np.zeros((n, 4), np.int32)
        p = lambda a: a.ctypes.data_as(ct.c_void_p)
        ones = np.ones(n, np.float32)
        hm.hm_extents(n, p(ones), p(ones), p(ones), p(mx), p(my), p(rad), W, H, 16, p(ex), p(ey), p(rg), p(rt))
        x0 = np.clip(np.floor((mx - radius) / 16), 0, 4); x1 = np.clip(np.ceil((mx + radius) / 16), 0, 4)
        y0 = np.clip(np.floor((my - radius) / 16), 0, 4); y1 = np.clip(np.ceil((my + radius) / 16), 0, 4)
        assert np.array_equal(rg, np.stack([x0, x1, y0, y1], -1).astype(np.int32))
