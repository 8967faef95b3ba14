"""Differentiable pure-PyTorch restatement of `gsplat.rendering.rasterization`
(gsplat 1.0.0 semantics) -- the checker for the HIP path.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED (see oracle/__init__.py): gsplat is an un-vendored dependency of the
reference; this file restates its published algorithm stage by stage and is anchored
on the reference call site:

  * call + arguments ........ /root/reference/model/gaussian.py:353-367
  * meta consumers .......... /root/reference/model/gaussian.py:371-372, 188-197
  * quaternion convention ... /root/reference/model/utils.py:31-55 (wxyz, normalise)
  * SH degree-0 constant .... /root/reference/model/utils.py:14-16

Stage map:
  project()                 <- fully_fused_projection (A.1)
  spherical_harmonics()     <- spherical_harmonics + clamp_min(rgb+0.5, 0) (A.2)
  isect_tiles()/isect_offset_encode()  <- tile lists and their order (A.3)
  rasterize_to_pixels()     <- blend forward (A.4) and the hand-derived blend
                               backward incl. absgrad (A.5) as an autograd.Function
  rasterization()           <- the seam itself

Everything runs in the dtype of `means` (fp32 or fp64).  Named constants are module
level so tests can cite them.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

ALPHA_MIN = 1.0 / 255.0
ALPHA_MAX = 0.999
T_MIN = 1e-4
FOV_CLAMP = 1.2999999523162842        # 1.3f: the float32 value, as in gsplat and in the device chain (oracle/c/gs_oracle.c)
RADIUS_SIGMA = 3.0
RADIUS_DISC_FLOOR = 0.009999999776482582   # 0.01f

SH_C0 = 0.2820947917738781
SH_C1 = 0.4886025119029199
SH_C2 = (1.0925484305920792, 0.31539156525252005, 0.5462742152960396)
SH_C3 = (0.5900435899266435, 2.890611442640554, 0.4570457994644658,
         0.3731763325901154, 1.445305721320277)


# --------------------------------------------------------------------------- A.1
def quat_to_rotmat(quats: Tensor) -> Tensor:
    """wxyz quaternion (normalised here) -> rotation matrix; same convention as
    /root/reference/model/utils.py:31-55."""
    q = quats / quats.norm(dim=-1, keepdim=True)
    w, x, y, z = q.unbind(-1)
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y),
    ], dim=-1)
    return R.reshape(quats.shape[:-1] + (3, 3))


def project(means: Tensor, quats: Tensor, scales: Tensor, viewmats: Tensor, Ks: Tensor,
            width: int, height: int, eps2d: float = 0.3, near_plane: float = 0.01,
            far_plane: float = 1e10, radius_clip: float = 0.0
            ) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """Appendix A.1.  Returns radii[C,N] int32, means2d[C,N,2], depths[C,N], conics[C,N,3].
    Culled entries are exactly zero (and carry zero gradient).  The scalar parameters are taken as the float32 values that
    reach the boundary (see oracle/c_oracle.py: render)."""
    import numpy as _np
    eps2d, near_plane, far_plane, radius_clip = (float(_np.float32(v)) for v in (eps2d, near_plane, far_plane, radius_clip))
    C = viewmats.shape[0]
    Rq = quat_to_rotmat(quats)                                   # [N,3,3]
    M = Rq * scales[:, None, :]                                  # Rq @ diag(s)
    covars = M @ M.transpose(-1, -2)                             # [N,3,3]
    Rv = viewmats[:, :3, :3]                                     # [C,3,3]
    tv = viewmats[:, :3, 3]                                      # [C,3]
    p_c = torch.einsum("cij,nj->cni", Rv, means) + tv[:, None, :]      # [C,N,3]
    cov_c = torch.einsum("cij,njk,clk->cnil", Rv, covars, Rv)          # [C,N,3,3]

    z = p_c[..., 2]
    depth_ok = (z >= near_plane) & (z <= far_plane)
    zs = torch.where(depth_ok, z, torch.ones_like(z))            # keeps the dead branch finite
    x, y = p_c[..., 0], p_c[..., 1]
    fx, fy = Ks[:, 0, 0][:, None], Ks[:, 1, 1][:, None]
    cx, cy = Ks[:, 0, 2][:, None], Ks[:, 1, 2][:, None]
    lim_x = FOV_CLAMP * (0.5 * width / fx)
    lim_y = FOV_CLAMP * (0.5 * height / fy)
    tx = zs * torch.minimum(lim_x, torch.maximum(-lim_x, x / zs))
    ty = zs * torch.minimum(lim_y, torch.maximum(-lim_y, y / zs))
    O = torch.zeros_like(zs)
    J = torch.stack([fx / zs, O, -fx * tx / (zs * zs),
                     O, fy / zs, -fy * ty / (zs * zs)], dim=-1).reshape(C, -1, 2, 3)
    cov2 = J @ cov_c @ J.transpose(-1, -2)                       # [C,N,2,2]
    means2d = torch.stack([fx * x / zs + cx, fy * y / zs + cy], dim=-1)

    a = cov2[..., 0, 0] + eps2d
    b = 0.5 * (cov2[..., 0, 1] + cov2[..., 1, 0])
    c = cov2[..., 1, 1] + eps2d
    det = a * c - b * b
    det_ok = det > 0
    dets = torch.where(det_ok, det, torch.ones_like(det))
    conics = torch.stack([c / dets, -b / dets, a / dets], dim=-1)

    with torch.no_grad():
        mid = 0.5 * (a + c)
        lam = mid + torch.sqrt(torch.clamp(mid * mid - det, min=RADIUS_DISC_FLOOR))
        radius = torch.ceil(RADIUS_SIGMA * torch.sqrt(lam))
        valid = depth_ok & det_ok & (radius > radius_clip)
        mx, my = means2d[..., 0], means2d[..., 1]
        inside = (mx + radius > 0) & (mx - radius < width) & (my + radius > 0) & (my - radius < height)
        valid = valid & inside
        radii = torch.where(valid, radius, torch.zeros_like(radius)).to(torch.int32)

    vm = valid
    means2d = torch.where(vm[..., None], means2d, torch.zeros_like(means2d))
    depths = torch.where(vm, z, torch.zeros_like(z))
    conics = torch.where(vm[..., None], conics, torch.zeros_like(conics))
    return radii, means2d, depths, conics


# --------------------------------------------------------------------------- A.2
def sh_basis(degree: int, dirs: Tensor) -> Tensor:
    """Real SH basis Y_k(d) for unit `dirs[...,3]`, k < (degree+1)^2 -> [..., K_act]."""
    x, y, z = dirs.unbind(-1)
    out = [torch.full_like(x, SH_C0)]
    if degree >= 1:
        out += [-SH_C1 * y, SH_C1 * z, -SH_C1 * x]
    if degree >= 2:
        xx, yy, zz = x * x, y * y, z * z
        out += [SH_C2[0] * x * y, -SH_C2[0] * y * z, SH_C2[1] * (2 * zz - xx - yy),
                -SH_C2[0] * x * z, SH_C2[2] * (xx - yy)]
    if degree >= 3:
        out += [-SH_C3[0] * y * (3 * xx - yy), SH_C3[1] * x * y * z,
                -SH_C3[2] * y * (4 * zz - xx - yy), SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy),
                -SH_C3[2] * x * (4 * zz - xx - yy), SH_C3[4] * z * (xx - yy),
                -SH_C3[0] * x * (xx - 3 * yy)]
    return torch.stack(out, dim=-1)


def spherical_harmonics(degree: int, means: Tensor, viewmats: Tensor, shs: Tensor,
                        radii: Tensor) -> Tensor:
    """Appendix A.2: rgb[C,N,3] = max(sum_k Y_k(dir) shs[n,k,:] + 0.5, 0), zero+0.5 for
    culled Gaussians (their value is never read by the blend)."""
    cam_pos = torch.linalg.inv(viewmats)[:, :3, 3]               # [C,3]
    d = means[None] - cam_pos[:, None, :]                        # [C,N,3]
    mask = radii > 0
    dn = d.norm(dim=-1, keepdim=True)
    d = d / torch.where(dn > 0, dn, torch.ones_like(dn))
    K_act = (degree + 1) ** 2
    Y = sh_basis(degree, d)                                      # [C,N,K_act]
    rgb = torch.einsum("cnk,nkd->cnd", Y, shs[:, :K_act, :])
    rgb = torch.where(mask[..., None], rgb, torch.zeros_like(rgb))
    return torch.clamp_min(rgb + 0.5, 0.0)


# --------------------------------------------------------------------------- A.3
@torch.no_grad()
def isect_tiles(means2d: Tensor, radii: Tensor, depths: Tensor, tile_size: int,
                tile_width: int, tile_height: int) -> Tuple[Tensor, Tensor, Tensor]:
    """Appendix A.3: tiles_per_gauss[C,N] int32, sorted isect_ids[I] int64, flatten_ids[I] int32.
    Order: (camera, tile, depth bits, flatten index) ascending == stable sort by key."""
    C, N = radii.shape
    r = radii.to(means2d.dtype) / tile_size
    tx = means2d[..., 0] / tile_size
    ty = means2d[..., 1] / tile_size
    x0 = torch.clamp(torch.floor(tx - r), 0, tile_width).to(torch.int64)
    x1 = torch.clamp(torch.ceil(tx + r), 0, tile_width).to(torch.int64)
    y0 = torch.clamp(torch.floor(ty - r), 0, tile_height).to(torch.int64)
    y1 = torch.clamp(torch.ceil(ty + r), 0, tile_height).to(torch.int64)
    vis = radii > 0
    cnt = torch.where(vis, (x1 - x0) * (y1 - y0), torch.zeros_like(x0))
    tiles_per_gauss = cnt.to(torch.int32)
    flat = torch.nonzero(cnt.reshape(-1) > 0).reshape(-1)        # flatten ids with >=1 tile
    n_tiles = tile_width * tile_height
    tile_bits = int(math.floor(math.log2(n_tiles))) + 1 if n_tiles > 0 else 1
    depth_bits = depths.to(torch.float32).contiguous().view(torch.int32).to(torch.int64).reshape(-1)
    keys, vals = [], []
    x0f, x1f, y0f, y1f = (t.reshape(-1) for t in (x0, x1, y0, y1))
    for f in flat.tolist():
        c = f // N
        ys = torch.arange(y0f[f], y1f[f], dtype=torch.int64)
        xs = torch.arange(x0f[f], x1f[f], dtype=torch.int64)
        tid = (ys[:, None] * tile_width + xs[None, :]).reshape(-1)
        keys.append((c << (32 + tile_bits)) | (tid << 32) | depth_bits[f])
        vals.append(torch.full_like(tid, f))
    if keys:
        keys_t = torch.cat(keys)
        vals_t = torch.cat(vals)
        order = torch.argsort(keys_t, stable=True)
        return tiles_per_gauss, keys_t[order], vals_t[order].to(torch.int32)
    return (tiles_per_gauss, torch.zeros(0, dtype=torch.int64), torch.zeros(0, dtype=torch.int32))


@torch.no_grad()
def isect_offset_encode(isect_ids: Tensor, C: int, tile_width: int, tile_height: int) -> Tensor:
    """isect_offsets[C,th,tw] int32: first sorted position of each (camera, tile) run."""
    n_tiles = tile_width * tile_height
    tile_bits = int(math.floor(math.log2(n_tiles))) + 1 if n_tiles > 0 else 1
    cam = isect_ids >> (32 + tile_bits)
    tid = (isect_ids >> 32) & ((1 << tile_bits) - 1)
    lin = cam * n_tiles + tid
    counts = torch.bincount(lin, minlength=C * n_tiles)[: C * n_tiles]
    offs = torch.cumsum(counts, 0) - counts
    return offs.to(torch.int32).reshape(C, tile_height, tile_width)


# --------------------------------------------------------------------------- A.4 / A.5
def _tile_terms(means2d, conics, colors, opacities, ids, px, py):
    """Per-tile pair quantities for pixels (px,py)[P] against the list `ids`[L]."""
    mu = means2d[ids]                                            # [L,2]
    con = conics[ids]
    op = opacities[ids]
    dx = mu[None, :, 0] - px[:, None]
    dy = mu[None, :, 1] - py[:, None]
    sigma = 0.5 * (con[None, :, 0] * dx * dx + con[None, :, 2] * dy * dy) + con[None, :, 1] * dx * dy
    vis = torch.exp(-sigma)
    alpha = torch.clamp_max(op[None, :] * vis, ALPHA_MAX)
    valid = (sigma >= 0) & (alpha >= ALPHA_MIN)
    a_eff = torch.where(valid, alpha, torch.zeros_like(alpha))
    one_m = 1.0 - a_eff
    T_incl = torch.cumprod(one_m, dim=1)
    T_excl = torch.cat([torch.ones_like(T_incl[:, :1]), T_incl[:, :-1]], dim=1)
    stop = valid & (T_incl <= T_MIN)
    L = ids.shape[0]
    idx = torch.arange(L)[None, :]
    first_stop = torch.where(stop, idx, torch.full_like(idx, L)).min(dim=1).values   # [P]
    contrib = valid & (idx < first_stop[:, None])
    return dx, dy, con, op, vis, alpha, contrib, T_excl


class _RasterizeToPixels(torch.autograd.Function):
    """Blend forward (A.4) and the hand-derived backward (A.5).  `means2d` is an INPUT, so the
    `.absgrad` attribute set in backward lands on the caller's tensor object."""

    @staticmethod
    def forward(ctx, means2d, conics, colors, opacities, backgrounds, width, height,
                tile_size, isect_offsets, flatten_ids, absgrad):
        C, N = means2d.shape[:2]
        dt = means2d.dtype
        th, tw = isect_offsets.shape[1:]
        I = flatten_ids.shape[0]
        m2, cn = means2d.reshape(C * N, 2), conics.reshape(C * N, 3)
        cl, op = colors.reshape(C * N, -1), opacities.reshape(C * N)
        D = cl.shape[-1]
        out = torch.zeros(C, height, width, D, dtype=dt)
        alphas = torch.zeros(C, height, width, 1, dtype=dt)
        last_ids = torch.zeros(C, height, width, dtype=torch.int32)
        offs = torch.cat([isect_offsets.reshape(-1).to(torch.int64), torch.tensor([I])])
        for c in range(C):
            for tyi in range(th):
                for txi in range(tw):
                    t = (c * th + tyi) * tw + txi
                    lo, hi = int(offs[t]), int(offs[t + 1])
                    y0, x0 = tyi * tile_size, txi * tile_size
                    y1, x1 = min(y0 + tile_size, height), min(x0 + tile_size, width)
                    if hi > lo:
                        ids = flatten_ids[lo:hi].to(torch.int64)
                        yy, xx = torch.meshgrid(torch.arange(y0, y1), torch.arange(x0, x1), indexing="ij")
                        px = xx.reshape(-1).to(dt) + 0.5
                        py = yy.reshape(-1).to(dt) + 0.5
                        _, _, _, _, _, alpha, contrib, T_excl = _tile_terms(m2, cn, cl, op, ids, px, py)
                        w = torch.where(contrib, alpha * T_excl, torch.zeros_like(alpha))   # [P,L]
                        col = w @ cl[ids]
                        Tf = torch.where(contrib, 1.0 - alpha, torch.ones_like(alpha)).prod(dim=1)
                        L = hi - lo
                        idx = torch.arange(L)[None, :].expand_as(contrib)
                        last = torch.where(contrib, idx, torch.full_like(idx, -1)).max(dim=1).values
                        last_abs = torch.where(last >= 0, last + lo, torch.zeros_like(last))
                        out[c, y0:y1, x0:x1] = col.reshape(y1 - y0, x1 - x0, D)
                        alphas[c, y0:y1, x0:x1, 0] = (1.0 - Tf).reshape(y1 - y0, x1 - x0)
                        last_ids[c, y0:y1, x0:x1] = last_abs.to(torch.int32).reshape(y1 - y0, x1 - x0)
        if backgrounds is not None:
            out = out + (1.0 - alphas) * backgrounds[:, None, None, :]
        ctx.save_for_backward(means2d, conics, colors, opacities, backgrounds, isect_offsets,
                              flatten_ids, alphas, last_ids)
        ctx.geom = (width, height, tile_size, absgrad)
        return out, alphas

    @staticmethod
    def backward(ctx, v_out, v_alphas):
        (means2d, conics, colors, opacities, backgrounds, isect_offsets, flatten_ids,
         alphas, last_ids) = ctx.saved_tensors
        width, height, tile_size, absgrad = ctx.geom
        C, N = means2d.shape[:2]
        dt = means2d.dtype
        th, tw = isect_offsets.shape[1:]
        I = flatten_ids.shape[0]
        m2, cn = means2d.reshape(C * N, 2), conics.reshape(C * N, 3)
        cl, op = colors.reshape(C * N, -1), opacities.reshape(C * N)
        v_m2 = torch.zeros_like(m2)
        v_abs = torch.zeros_like(m2)
        v_cn = torch.zeros_like(cn)
        v_cl = torch.zeros_like(cl)
        v_op = torch.zeros_like(op)
        offs = torch.cat([isect_offsets.reshape(-1).to(torch.int64), torch.tensor([I])])
        for c in range(C):
            bg = backgrounds[c] if backgrounds is not None else None
            for tyi in range(th):
                for txi in range(tw):
                    t = (c * th + tyi) * tw + txi
                    lo, hi = int(offs[t]), int(offs[t + 1])
                    if hi <= lo:
                        continue
                    y0, x0 = tyi * tile_size, txi * tile_size
                    y1, x1 = min(y0 + tile_size, height), min(x0 + tile_size, width)
                    ids = flatten_ids[lo:hi].to(torch.int64)
                    yy, xx = torch.meshgrid(torch.arange(y0, y1), torch.arange(x0, x1), indexing="ij")
                    px = xx.reshape(-1).to(dt) + 0.5
                    py = yy.reshape(-1).to(dt) + 0.5
                    dx, dy, con, opl, vis, alpha, contrib, T = _tile_terms(m2, cn, cl, op, ids, px, py)
                    vc = v_out[c, y0:y1, x0:x1].reshape(-1, v_out.shape[-1])          # [P,D]
                    va = v_alphas[c, y0:y1, x0:x1, 0].reshape(-1)                     # [P]
                    Tf = 1.0 - alphas[c, y0:y1, x0:x1, 0].reshape(-1)                 # [P]
                    zero = torch.zeros_like(alpha)
                    fac = torch.where(contrib, alpha * T, zero)                       # [P,L]
                    rgb = cl[ids]                                                     # [L,D]
                    v_rgb = fac.t() @ vc                                              # [L,D]
                    # suffix sums S_after[p,i,:] = sum_{j>i} rgb_j fac_pj, contracted with v_c
                    contribv = fac * (vc @ rgb.t())                                   # fac_pi * (rgb_i . v_c_p)
                    suffix = torch.flip(torch.cumsum(torch.flip(contribv, [1]), 1), [1]) - contribv
                    ra = 1.0 / (1.0 - alpha)
                    v_alpha = T * (vc @ rgb.t()) - suffix * ra + (Tf * va)[:, None] * ra
                    if bg is not None:
                        v_alpha = v_alpha - (Tf * (vc @ bg))[:, None] * ra
                    v_alpha = torch.where(contrib, v_alpha, zero)
                    unsat = (opl[None, :] * vis) <= ALPHA_MAX
                    v_sigma = torch.where(unsat, -opl[None, :] * vis * v_alpha, zero)
                    A, B, Cc = con[None, :, 0], con[None, :, 1], con[None, :, 2]
                    gx = v_sigma * (A * dx + B * dy)
                    gy = v_sigma * (B * dx + Cc * dy)
                    v_cn.index_add_(0, ids, torch.stack([(0.5 * v_sigma * dx * dx).sum(0),
                                                         (v_sigma * dx * dy).sum(0),
                                                         (0.5 * v_sigma * dy * dy).sum(0)], dim=-1))
                    v_m2.index_add_(0, ids, torch.stack([gx.sum(0), gy.sum(0)], dim=-1))
                    v_abs.index_add_(0, ids, torch.stack([gx.abs().sum(0), gy.abs().sum(0)], dim=-1))
                    v_op.index_add_(0, ids, torch.where(unsat, vis * v_alpha, zero).sum(0))
                    v_cl.index_add_(0, ids, v_rgb)
        if absgrad:
            means2d.absgrad = v_abs.reshape(C, N, 2)
        v_bg = None
        if backgrounds is not None and ctx.needs_input_grad[4]:
            v_bg = ((1.0 - alphas) * v_out).sum(dim=(1, 2))
        return (v_m2.reshape(C, N, 2), v_cn.reshape(C, N, 3), v_cl.reshape(colors.shape),
                v_op.reshape(C, N), v_bg, None, None, None, None, None, None)


def rasterize_to_pixels(means2d, conics, colors, opacities, width, height, tile_size,
                        isect_offsets, flatten_ids, backgrounds=None, absgrad=False):
    return _RasterizeToPixels.apply(means2d, conics, colors, opacities, backgrounds, width,
                                    height, tile_size, isect_offsets, flatten_ids, absgrad)


# --------------------------------------------------------------------------- the seam
def rasterization(means: Tensor, quats: Tensor, scales: Tensor, opacities: Tensor, colors: Tensor,
                  viewmats: Tensor, Ks: Tensor, width: int, height: int,
                  near_plane: float = 0.01, far_plane: float = 1e10, radius_clip: float = 0.0,
                  eps2d: float = 0.3, sh_degree: Optional[int] = None, packed: bool = True,
                  tile_size: int = 16, backgrounds: Optional[Tensor] = None,
                  render_mode: str = "RGB", sparse_grad: bool = False, absgrad: bool = False,
                  rasterize_mode: str = "classic", channel_chunk: int = 32
                  ) -> Tuple[Tensor, Tensor, Dict]:
    """gsplat 1.0.0 `rasterization()` for the argument subset the reference uses
    (/root/reference/model/gaussian.py:353-367: packed=False, absgrad=True, RGB, classic)."""
    assert not packed, "oracle restates packed=False only (what the reference passes)"
    assert render_mode == "RGB" and rasterize_mode == "classic"
    N, C = means.shape[0], viewmats.shape[0]
    radii, means2d, depths, conics = project(means, quats, scales, viewmats, Ks, width, height,
                                             eps2d, near_plane, far_plane, radius_clip)
    if sh_degree is None:
        cols = colors.expand(C, N, -1) if colors.dim() == 2 else colors
    else:
        cols = spherical_harmonics(sh_degree, means, viewmats, colors, radii)
    opac = opacities[None, :].expand(C, N)
    tile_width = math.ceil(width / tile_size)
    tile_height = math.ceil(height / tile_size)
    tiles_per_gauss, isect_ids, flatten_ids = isect_tiles(means2d, radii, depths, tile_size,
                                                          tile_width, tile_height)
    isect_offsets = isect_offset_encode(isect_ids, C, tile_width, tile_height)
    render_colors, render_alphas = rasterize_to_pixels(means2d, conics, cols, opac, width, height,
                                                       tile_size, isect_offsets, flatten_ids,
                                                       backgrounds, absgrad)
    meta = {"camera_ids": None, "gaussian_ids": None, "radii": radii, "means2d": means2d,
            "depths": depths, "conics": conics, "opacities": opac, "colors": cols,
            "tile_width": tile_width, "tile_height": tile_height,
            "tiles_per_gauss": tiles_per_gauss, "isect_ids": isect_ids,
            "flatten_ids": flatten_ids, "isect_offsets": isect_offsets,
            "width": width, "height": height, "tile_size": tile_size}
    return render_colors, render_alphas, meta


# A deliberately naive, loop-free-of-cleverness differentiable blend used ONLY to pin the
# hand-derived backward above against autograd (tests/test_oracle.py).
def naive_blend_autograd(means2d, conics, colors, opacities, backgrounds, width, height,
                         tile_size, isect_offsets, flatten_ids):
    C, N = means2d.shape[:2]
    th, tw = isect_offsets.shape[1:]
    I = flatten_ids.shape[0]
    offs = torch.cat([isect_offsets.reshape(-1).to(torch.int64), torch.tensor([I])])
    m2, cn = means2d.reshape(C * N, 2), conics.reshape(C * N, 3)
    cl, op = colors.reshape(C * N, -1), opacities.reshape(C * N)
    rows = []
    for c in range(C):
        for i in range(height):
            for j in range(width):
                t = (c * th + i // tile_size) * tw + j // tile_size
                T = torch.ones((), dtype=m2.dtype)
                acc = torch.zeros(cl.shape[-1], dtype=m2.dtype)
                for k in range(int(offs[t]), int(offs[t + 1])):
                    g = int(flatten_ids[k])
                    dx = m2[g, 0] - (j + 0.5)
                    dy = m2[g, 1] - (i + 0.5)
                    sigma = 0.5 * (cn[g, 0] * dx * dx + cn[g, 2] * dy * dy) + cn[g, 1] * dx * dy
                    alpha = torch.clamp_max(op[g] * torch.exp(-sigma), ALPHA_MAX)
                    if float(sigma.detach()) < 0 or float(alpha.detach()) < ALPHA_MIN:
                        continue
                    Tn = T * (1 - alpha)
                    if float(Tn.detach()) <= T_MIN:
                        break
                    acc = acc + cl[g] * alpha * T
                    T = Tn
                if backgrounds is not None:
                    acc = acc + T * backgrounds[c]
                rows.append(torch.cat([acc, (1 - T).reshape(1)]))
    out = torch.stack(rows).reshape(C, height, width, -1)
    return out[..., :-1], out[..., -1:]
