"""ctypes front-end of the scalar C oracle (oracle/c/gs_oracle.c).  TEST INFRASTRUCTURE ONLY.

`render()` runs the whole forward of the seam (/root/reference/model/gaussian.py:353-367) on
numpy arrays and returns every intermediate; `backward()` replays A.5/A.6 for given upstream
gradients.  fp32 (`libgso_f32.so`) mirrors the device arithmetic type, fp64 (`libgso_f64.so`)
is the shadow used to set tolerances.  PARITY UNPINNED -- see oracle/__init__.py.
"""
from __future__ import annotations

import ctypes as ct
import math
import os
import subprocess
from typing import Dict, Optional

import numpy as np

_HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "c")
_LIBS: Dict[str, ct.CDLL] = {}


def build(force: bool = False) -> None:
    """Compile the C oracle with gcc (recipe: oracle/c/Makefile)."""
    if force:
        subprocess.run(["make", "-C", _HERE, "clean"], check=True, capture_output=True)
    subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)


def _lib(dtype) -> ct.CDLL:
    name = "libgso_f64.so" if np.dtype(dtype) == np.float64 else "libgso_f32.so"
    if name not in _LIBS:
        path = os.path.join(_HERE, name)
        if not os.path.exists(path):
            build()
        _LIBS[name] = ct.CDLL(path)
        _LIBS[name].gso_isect_count.restype = ct.c_int64
    return _LIBS[name]


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(ct.c_void_p)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def render(means, quats, scales, opacities, colors, viewmats, Ks, width, height, sh_degree=None,
           backgrounds=None, near_plane=0.01, far_plane=1e10, radius_clip=0.0, eps2d=0.3,
           tile_size=16, dtype=np.float32) -> Dict[str, np.ndarray]:
    """Full forward; returns dict with render_colors/alphas and all meta arrays.
    The scalar parameters are rounded to float32 first: that is how they arrive at the boundary (gsplat's kernels and the
    C ABI here take `float eps2d` ...), and the fp64 build must restate the SAME inputs -- 0.3 as a double differs from
    0.3f by 1.2e-8, enough to flip one `ceil(3 sqrt(lambda))` in 36 M Gaussians (found by a 3000-configuration sweep)."""
    eps2d, near_plane, far_plane, radius_clip = (float(np.float32(v)) for v in (eps2d, near_plane, far_plane, radius_clip))
    L = _lib(dtype)
    R = ct.c_double if np.dtype(dtype) == np.float64 else ct.c_float
    means, quats, scales = _c(means, dtype), _c(quats, dtype), _c(scales, dtype)
    viewmats, Ks = _c(viewmats, dtype), _c(Ks, dtype)
    C, N = viewmats.shape[0], means.shape[0]
    radii = np.zeros((C, N), np.int32)
    means2d = np.zeros((C, N, 2), dtype); depths = np.zeros((C, N), dtype); conics = np.zeros((C, N, 3), dtype)
    L.gso_project_fwd(C, N, _p(means), _p(quats), _p(scales), _p(viewmats), _p(Ks), width, height,
                      R(eps2d), R(near_plane), R(far_plane), R(radius_clip), _p(radii), _p(means2d),
                      _p(depths), _p(conics))
    if sh_degree is None:
        cols = _c(np.broadcast_to(colors, (C, N, 3)), dtype)
        shs = None
    else:
        shs = _c(colors, dtype)
        cols = np.zeros((C, N, 3), dtype)
        rc = L.gso_sh_fwd(C, N, shs.shape[1], int(sh_degree), _p(means), _p(viewmats), _p(shs), _p(radii), _p(cols))
        assert rc == 0
    opac = _c(np.broadcast_to(np.asarray(opacities)[None, :], (C, N)), dtype)
    tw, th = math.ceil(width / tile_size), math.ceil(height / tile_size)
    tpg = np.zeros((C, N), np.int32)
    I = int(L.gso_isect_count(C, N, _p(means2d), _p(radii), tile_size, tw, th, _p(tpg)))
    isect_ids = np.zeros(max(I, 1), np.int64)[:I]
    flatten_ids = np.zeros(max(I, 1), np.int32)[:I]
    isect_offsets = np.zeros((C, th, tw), np.int32)
    rc = L.gso_isect_build(C, N, _p(means2d), _p(radii), _p(depths), tile_size, tw, th, ct.c_int64(I),
                           _p(isect_ids), _p(flatten_ids), _p(isect_offsets))
    assert rc == 0
    bg = None if backgrounds is None else _c(backgrounds, dtype)
    out = np.zeros((C, height, width, 3), dtype); alphas = np.zeros((C, height, width, 1), dtype)
    last_ids = np.zeros((C, height, width), np.int32)
    L.gso_blend_fwd(C, N, width, height, tile_size, _p(means2d), _p(conics), _p(cols), _p(opac), _p(bg),
                    _p(isect_offsets), _p(flatten_ids), ct.c_int64(I), _p(out), _p(alphas), _p(last_ids))
    return dict(render_colors=out, render_alphas=alphas, last_ids=last_ids, radii=radii, means2d=means2d,
                depths=depths, conics=conics, colors=cols, opacities=opac, tiles_per_gauss=tpg,
                isect_ids=isect_ids, flatten_ids=flatten_ids, isect_offsets=isect_offsets,
                tile_width=tw, tile_height=th, n_isects=I,
                _inputs=dict(means=means, quats=quats, scales=scales, shs=shs, viewmats=viewmats, Ks=Ks,
                             bg=bg, width=width, height=height, sh_degree=sh_degree, tile_size=tile_size,
                             eps2d=eps2d, near_plane=near_plane, far_plane=far_plane))


def blend_margin(fwd: Dict[str, np.ndarray], means2d_other=None, conics_other=None, mu_tol_ulps: float = 0.0,
                 conic_rtol: float = 0.0) -> np.ndarray:
    """[C,H,W] normalised distance of every pixel to the nearest blend discontinuity (see gso_blend_margin); values
    below 1e-4 mark pixels where an fp32 evaluation may legitimately flip a contributor.  With `means2d_other` /
    `conics_other` (the implementation under test's [C,N,2] / [C,N,3] arrays) the reachable perturbation of each
    exponent is MEASURED (twice the difference between the two implementations' sigma at that pixel, plus fp32
    evaluation rounding) instead of bounded a priori.  `mu_tol_ulps` > 0 (with `conic_rtol`): the a-priori form from the
    storage bounds the tested path is HELD to (its means2d within that many fp32 ulps of max(|coordinate|, 32 px), its conics
    within conic_rtol of their largest entry) -- independent of any measured output."""
    inp = fwd["_inputs"]
    dtype = fwd["means2d"].dtype
    L = _lib(dtype)
    C, N = fwd["radii"].shape
    W, H = inp["width"], inp["height"]
    out = np.ones((C, H, W), dtype)
    m_o = c_o = None
    if means2d_other is not None and conics_other is not None:
        m_o, c_o = _c(means2d_other, dtype), _c(conics_other, dtype)
    R = ct.c_double if np.dtype(dtype) == np.float64 else ct.c_float
    L.gso_blend_margin_tol(C, N, W, H, inp["tile_size"], _p(fwd["means2d"]), _p(fwd["conics"]), _p(fwd["opacities"]),
                           _p(fwd["isect_offsets"]), _p(fwd["flatten_ids"]), ct.c_int64(fwd["n_isects"]), _p(m_o), _p(c_o),
                           R(mu_tol_ulps), R(conic_rtol), _p(out))
    return out


def backward(fwd: Dict[str, np.ndarray], v_render_colors, v_render_alphas=None) -> Dict[str, np.ndarray]:
    """A.5 + A.6: gradients wrt means, quats, scales, opacities, colors(shs) plus the 2-D
    intermediates and absgrad, for upstream grads of the returned image / alpha."""
    inp = fwd["_inputs"]
    dtype = fwd["means2d"].dtype
    L = _lib(dtype)
    R = ct.c_double if dtype == np.float64 else ct.c_float
    C, N = fwd["radii"].shape
    W, H, tile = inp["width"], inp["height"], inp["tile_size"]
    vc = _c(v_render_colors, dtype)
    va = None if v_render_alphas is None else _c(v_render_alphas, dtype)
    v_m2 = np.zeros((C, N, 2), dtype); v_abs = np.zeros((C, N, 2), dtype); v_cn = np.zeros((C, N, 3), dtype)
    v_rgb = np.zeros((C, N, 3), dtype); v_op = np.zeros((C, N), dtype)
    I = fwd["n_isects"]
    L.gso_blend_bwd(C, N, W, H, tile, _p(fwd["means2d"]), _p(fwd["conics"]), _p(fwd["colors"]),
                    _p(fwd["opacities"]), _p(inp["bg"]), _p(fwd["isect_offsets"]), _p(fwd["flatten_ids"]),
                    ct.c_int64(I), _p(fwd["render_alphas"]), _p(fwd["last_ids"]), _p(vc), _p(va),
                    _p(v_m2), _p(v_abs), _p(v_cn), _p(v_rgb), _p(v_op))
    v_means = np.zeros((N, 3), dtype); v_quats = np.zeros((N, 4), dtype); v_scales = np.zeros((N, 3), dtype)
    out = dict(v_means2d=v_m2, v_means2d_abs=v_abs, v_conics=v_cn, v_colors_post=v_rgb, v_opacities_cn=v_op,
               v_opacities=v_op.sum(0))
    if inp["sh_degree"] is not None:
        shs = inp["shs"]
        v_shs = np.zeros_like(shs)
        rc = L.gso_sh_bwd(C, N, shs.shape[1], int(inp["sh_degree"]), _p(inp["means"]), _p(inp["viewmats"]),
                          _p(shs), _p(fwd["radii"]), _p(fwd["colors"]), _p(v_rgb), _p(v_shs), _p(v_means))
        assert rc == 0
        out["v_colors"] = v_shs
    else:
        out["v_colors"] = v_rgb
    L.gso_project_bwd(C, N, _p(inp["means"]), _p(inp["quats"]), _p(inp["scales"]), _p(inp["viewmats"]),
                      _p(inp["Ks"]), W, H, R(inp["eps2d"]), R(inp["near_plane"]), R(inp["far_plane"]),
                      _p(fwd["radii"]), _p(v_m2), None, _p(v_cn), _p(v_means), _p(v_quats), _p(v_scales))
    out.update(v_means=v_means, v_quats=v_quats, v_scales=v_scales)
    return out
