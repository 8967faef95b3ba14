"""GPU parity tests: the HIP path (through the C ABI) against the oracle on identical seeded
inputs, against the committed golden fixtures, and -- at BASELINE.json's full size -- through
size-independent properties.  Tolerances (BASELINE.json north_star): forward RGB <= 1e-4 abs,
gradients <= 1e-3 relative (to the largest reference magnitude of each tensor).

The blend has three discontinuities (alpha >= 1/255, T <= 1e-4, sigma >= 0).  A pixel whose
fp64 oracle run passes within a relative 1e-4 of one of them may legitimately flip a contributor
under fp32 arithmetic; such pixels (oracle.c_oracle.blend_margin) are excluded from the strict
1e-4 bound, must stay rare, and are still bounded by one flipped contributor's weight.
"""
import math
import os

import numpy as np
import pytest
import torch

import parity_log
from oracle import c_oracle as CO
from scenes import config_bench_1m, config_long_lists, config_s1, dense_scene, make_scene

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
FWD_ATOL = 1e-4
GRAD_RTOL = 1e-3


def dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


def to_dev(sc):
    return {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).to(dev()) for k, v in sc.items() if isinstance(v, np.ndarray) and v.dtype.kind == "f"}


def run_hip(sc, bwd=True, seed=0, use_bg=True, dbg=None, culling="gsplat", fw=None, upstream=None, max_flip_tile_frac=0.02):
    """With `fw` (the oracle's forward of the same scene) the random upstream gradients are zeroed on the pixels
    where the contributor set itself is undecided between fp32 and fp64 (loose_pixels), so that check_backward
    compares like with like: the gradient of the pixels on which both sides blend the same list."""
    from easy_gaussian_splatting_amd.rendering import rasterization
    t = to_dev(sc)
    ins = [t[k].clone().requires_grad_(bwd) for k in ("means", "quats", "scales", "opacities", "shs")]
    img, alpha, meta = rasterization(*ins, t["viewmats"], t["Ks"], int(sc["width"]), int(sc["height"]),
                                     sh_degree=int(sc["sh_degree"]), packed=False,
                                     backgrounds=t["backgrounds"] if use_bg else None, absgrad=True, _debug=dbg,
                                     _tile_culling=culling)
    out = dict(img=img, alpha=alpha, meta=meta, ins=ins)
    if bwd:
        g = torch.Generator().manual_seed(seed)
        vc, va = torch.randn(img.shape, generator=g), torch.randn(alpha.shape, generator=g)
        if upstream is not None:   # the (masked) upstream gradients of an earlier run of the same scene
            vc, va = (torch.from_numpy(x).float() for x in upstream)
        elif fw is not None:
            vc, va = mask_upstream(out, fw, vc, va, lists=culling != "tight", max_flip_tile_frac=max_flip_tile_frac)
        out["vc"], out["va"] = vc.numpy().astype(np.float64), va.numpy().astype(np.float64)
        out["grads"] = torch.autograd.grad((img * vc.to(dev())).sum() + (alpha * va.to(dev())).sum(), ins)
    torch.cuda.synchronize()
    return out


def mask_upstream(hip, fw, vc, va, lists=True, max_flip_tile_frac=0.02):
    rep = hip["report"] = forward_report(hip["meta"], fw, lists, max_flip_tile_frac=max_flip_tile_frac)
    rep["fw"] = fw
    keep = torch.from_numpy(~rep["loose"])
    return vc * keep[..., None], va * keep[..., None]


def run_oracle(sc, use_bg=True, dtype=np.float64):
    return CO.render(sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["shs"], sc["viewmats"], sc["Ks"],
                     int(sc["width"]), int(sc["height"]), sh_degree=int(sc["sh_degree"]),
                     backgrounds=sc["backgrounds"] if use_bg else None, dtype=dtype)


MAX_RAZOR_FRAC = 0.05   # scenes are built to stay off the blend thresholds (synthetic.dense_scene)


def _affected_tiles(meta, fw, sel, lists=True):
    """[C,th,tw] mask of the tiles whose list may legitimately differ from the oracle's because of the Gaussians in `sel`
    (a radius that flipped by one, a rectangle edge on a tile boundary): the tiles in which such a Gaussian is listed on ONE
    side only.  Everywhere else -- including the rest of its rectangle -- both sides list it, and the lists must agree.
    (Rounds 1-4 exempted the Gaussian's whole rectangle: harmless while footprints are a handful of tiles; one flipped splat
    of a heavy-tailed scene covers hundreds, and 24 of them exempted 15 % of a 1080p frame -- synthetic.config_heavy.)
    `lists`: both sides hold gsplat's lists -> exact membership from the lists themselves.  Otherwise (the short-list mode;
    the fp32 build of the oracle as reference): the symmetric difference of each side's own 3-sigma rectangle."""
    C, N = fw["radii"].shape
    tile = fw["_inputs"]["tile_size"]
    tw, th = fw["tile_width"], fw["tile_height"]
    tmask = np.zeros((C * th * tw,), bool)
    ids = np.nonzero(sel.reshape(-1))[0]
    if lists:
        fid_h = meta["flatten_ids"]
        off_h = meta["isect_offsets"].reshape(-1).long()
        fid_o, off_o = fw["flatten_ids"], fw["isect_offsets"].reshape(-1).astype(np.int64)
        for g in ids:
            pos_h = torch.nonzero(fid_h == int(g)).reshape(-1)
            t_h = (torch.searchsorted(off_h, pos_h, right=True) - 1).cpu().numpy()
            t_o = np.searchsorted(off_o, np.nonzero(fid_o == g)[0], side="right") - 1
            tmask[np.setxor1d(t_h, t_o)] = True
        return tmask.reshape(C, th, tw)
    m2_h, r_h = meta["means2d"].detach().cpu().numpy().astype(np.float64), meta["radii"].cpu().numpy().astype(np.float64)
    tmask = tmask.reshape(C, th, tw)
    for c, n in zip(*np.nonzero(sel)):
        cover = []
        for m2, r in ((m2_h[c, n], r_h[c, n]), (fw["means2d"][c, n].astype(np.float64), float(fw["radii"][c, n]))):
            k = np.zeros((th, tw), bool)
            if r > 0:
                x0, x1 = int(np.clip(np.floor((m2[0] - r) / tile), 0, tw)), int(np.clip(np.ceil((m2[0] + r) / tile), 0, tw))
                y0, y1 = int(np.clip(np.floor((m2[1] - r) / tile), 0, th)), int(np.clip(np.ceil((m2[1] + r) / tile), 0, th))
                k[y0:y1, x0:x1] = True
            cover.append(k)
        tmask[c] |= cover[0] ^ cover[1]
    return tmask


def _compare_lists(meta, fw, tmask):
    """Every tile's sorted run against the oracle's, except the tiles in `tmask`: same length, same ids, same order.
    The one admissible order difference: two entries whose depths round to a different fp32 order than the oracle's
    (float)(fp64 depth) -- the run must then hold the same ids and be in the order of the PATH'S OWN (depth bits,
    flatten id) keys, which is the stable-sort contract (A.3) applied to its own fp32 depths.  Returns the
    [tiles] mask of such runs."""
    ok = ~tmask.reshape(-1)
    off_h = meta["isect_offsets"].reshape(-1).cpu().numpy().astype(np.int64)
    fid_h = meta["flatten_ids"].cpu().numpy()
    cnt_h = np.diff(np.append(off_h, fid_h.size))
    off_o = fw["isect_offsets"].reshape(-1).astype(np.int64)
    cnt_o = np.diff(np.append(off_o, fw["n_isects"]))
    assert np.array_equal(cnt_h[ok], cnt_o[ok]), "tile list lengths differ in tiles no rounding flip touches"
    h, o = fid_h[np.repeat(ok, cnt_h)], fw["flatten_ids"][np.repeat(ok, cnt_o)]
    flipped = np.zeros(ok.size, bool)
    d = h != o
    if d.any():
        tiles = np.unique(np.repeat(np.nonzero(ok)[0], cnt_h[ok])[d])
        bits = meta["depths"].reshape(-1).cpu().numpy().view(np.int32).astype(np.int64)
        dep_o = fw["depths"].reshape(-1)
        for t in tiles:
            sh, so = fid_h[off_h[t]: off_h[t] + cnt_h[t]].astype(np.int64), fw["flatten_ids"][off_o[t]: off_o[t] + cnt_o[t]].astype(np.int64)
            assert np.array_equal(np.sort(sh), np.sort(so)), f"tile {t}: different ids in the list"
            assert np.all(np.diff(bits[sh] * (1 << 32) + sh) > 0), f"tile {t}: not in the order of its own (depth, id) keys"
            sw = sh != so
            assert np.abs(dep_o[sh[sw]] - dep_o[so[sw]]).max() <= 1e-6 * np.abs(dep_o[so[sw]]).max() + 1e-7, f"tile {t}: reordered entries are not depth neighbours"
        flipped[tiles] = True
        n_ids = np.unique(h[d]).size
        parity_log.record(n_depth_order_flips=n_ids)
        print(f"[parity] {n_ids} Gaussians ({int(d.sum())} list entries in {tiles.size} tiles) are ordered by an fp32 depth order that differs from the oracle's")
        assert n_ids <= max(4, 2e-4 * dep_o.size)
    return flipped


# what an fp64 projection chain rounded once to fp32 supports (the old bounds -- 2e-3 px, 2e-3, 1e-4 --
# were 10-1000x slack): rounding once is half an ulp, the bounds are one ulp (measured on MI355X: 0.5 ulp of the
# coordinate, 9e-8 of the conic, 6e-8 of the depth at 300 k / 2 M Gaussians, profiles/r03_parity_report.json)
EPS32 = 1.1920929e-7
MEANS2D_TOL_ULPS = 1.0    # of max(|coordinate|, 32 px): 1.2e-4 px at 1000 px
CONICS_RTOL = 2.4e-7      # of the conic's largest entry
DEPTHS_RTOL = 2.4e-7


def oracle_fp32(fw):
    """The fp32 build of the oracle on the same inputs (cached on `fw`): an implementation of the path in the arithmetic
    type the device blends in, independent of the HIP code: check_backward's printed fallback arbiter."""
    if "_fw32" not in fw:
        inp = fw["_inputs"]
        colors = inp["shs"] if inp["shs"] is not None else fw["colors"]
        fw["_fw32"] = CO.render(inp["means"], inp["quats"], inp["scales"], fw["opacities"][0], colors, inp["viewmats"], inp["Ks"],
                                inp["width"], inp["height"], sh_degree=inp["sh_degree"], backgrounds=inp["bg"], dtype=np.float32)
    return fw["_fw32"]


def razor_mask(fw):
    """[C,H,W] pixels within 1e-4 (normalised) of a blend discontinuity, decided from the fp64 oracle run ALONE -- neither
    the implementation under test (VERDICT r2 item 6d / ADVICE r2: a larger HIP error must not widen its own exemption)
    nor any other fp32 run enters.  gso_blend_margin_tol prices what an fp32 blend fed with fp32 inputs can move: the mean by
    MEANS2D_TOL_ULPS, the conic by CONICS_RTOL -- the very bounds forward_report ASSERTS on the path's outputs, so the
    exemption follows from limits the implementation is held to --, plus the rounding of an fp32 evaluation of the quadratic
    form and the accumulated relative error of T."""
    if "_razor" not in fw:
        fw["_razor"] = CO.blend_margin(fw, mu_tol_ulps=MEANS2D_TOL_ULPS, conic_rtol=CONICS_RTOL) < 1e-4
    return fw["_razor"]


def forward_report(meta, fw, lists=True, geom_slack=1.0, max_flip_tile_frac=0.02):
    """Integer outputs against the oracle's.  They are bit-exact except where the last-bit difference of two fp64
    evaluations crosses an integer decision: radius = ceil(3 sigma) (+-1); an edge mu +- r of the tile rectangle landing
    on a tile boundary; two depths that swap order.  Such Gaussians are counted, recorded (parity_log) and printed, must be
    rare (at most max(1, 1e-4 N) radii, max(1, 2e-4 N) rectangles), and only the tiles they touch are exempt from the
    bit-exact list comparison.  Returns exact_lists, the razor mask (razor_mask: independent of the HIP output) and
    `loose` = razor | the pixels of exempted tiles.  `geom_slack` multiplies the three storage bounds where the REFERENCE is
    not the fp64 run of the very same inputs (the fp32 build of the oracle; a model whose exp / sigmoid run inside the kernel)."""
    tile = fw["_inputs"]["tile_size"]
    radii = meta["radii"].cpu().numpy()
    mism = radii != fw["radii"]
    if mism.any():   # ceil(3 sigma) on the last bit: must be a +-1 flip of a visible Gaussian, and rare
        print(f"[parity] {int(mism.sum())} of {mism.size} radii differ from the oracle's")
        assert np.abs(radii.astype(np.int64) - fw["radii"])[mism].max() <= 1, "a radius differs by more than the ceil() flip"
    assert mism.sum() <= max(1, 1e-4 * mism.size), f"{int(mism.sum())} of {mism.size} radii differ"
    same = ~mism & (fw["radii"] > 0)   # (culled Gaussians carry no geometry on either side)
    d_mu = np.abs(meta["means2d"].detach().cpu().numpy() - fw["means2d"]).max(-1) / (EPS32 * np.maximum(np.abs(fw["means2d"]).max(-1), 32.0))
    e_mu = float(d_mu[same].max(initial=0))
    assert e_mu <= MEANS2D_TOL_ULPS * geom_slack, f"means2d differ by {e_mu} ulps of the coordinate"
    e_dep = float((np.abs(meta["depths"].cpu().numpy() - fw["depths"]) / np.maximum(np.abs(fw["depths"]), 1e-30))[same].max(initial=0))
    assert e_dep <= DEPTHS_RTOL * geom_slack, f"depths differ by {e_dep} relative"
    con = meta["conics"].cpu().numpy()
    e_con = float((np.abs(con - fw["conics"]).max(-1) / np.maximum(np.abs(fw["conics"]).max(-1), 1e-30))[same].max(initial=0))
    assert e_con <= CONICS_RTOL * geom_slack, f"conics differ by {e_con} of their largest entry"
    differ = mism.copy()
    n_edge = 0
    if lists:
        edge = (meta["tiles_per_gauss"].cpu().numpy() != fw["tiles_per_gauss"]) & ~mism
        if edge.any():   # same radius, different rectangle: an edge of mu +- r must sit on a tile boundary
            print(f"[parity] {int(edge.sum())} of {edge.size} tile rectangles differ from the oracle's (edge on a tile boundary)")
            mu, r = fw["means2d"][edge], fw["radii"][edge][:, None].astype(np.float64)
            edges = np.concatenate([mu - r, mu + r], axis=1) / tile
            assert np.abs(edges - np.round(edges)).min(axis=1).max() <= MEANS2D_TOL_ULPS * EPS32 * max(32.0, np.abs(mu).max() + r.max()) / tile, \
                "a tile rectangle differs away from any tile boundary"
            assert edge.sum() <= max(1, 2e-4 * edge.size)
            n_edge = int(edge.sum())
        differ |= edge
    if geom_slack > 1.0:
        # the reference is NOT the fp64 run of the same inputs (the fp32 build of the oracle, whose own geometry is tens of
        # ulps off): a rectangle can also differ with the same tile count, or -- in the short-list mode -- without being
        # compared at all.  Each side's 3-sigma rectangle from its own (means2d, radius); Gaussians whose rectangles differ
        # exempt the tiles they touch, and must be as rare as the flips above.
        def rects(m2, r):
            lo = np.clip(np.floor((m2 - r[..., None]) / tile), 0, [fw["tile_width"], fw["tile_height"]])
            hi = np.clip(np.ceil((m2 + r[..., None]) / tile), 0, [fw["tile_width"], fw["tile_height"]])
            return np.concatenate([lo, hi], axis=-1)
        shifted = (rects(meta["means2d"].detach().cpu().numpy().astype(np.float64), radii.astype(np.float64))
                   != rects(fw["means2d"].astype(np.float64), fw["radii"].astype(np.float64))).any(-1) & same
        assert shifted.sum() <= max(1, 2e-4 * shifted.size), f"{int(shifted.sum())} rectangles differ from the fp32 reference's"
        differ |= shifted
    tmask = _affected_tiles(meta, fw, differ, lists) if differ.any() else np.zeros((radii.shape[0], fw["tile_height"], fw["tile_width"]), bool)
    exact_lists = not differ.any()
    if lists:  # integer / index work is bit-exact wherever no rounding flip reaches
        if exact_lists:
            assert np.array_equal(meta["tiles_per_gauss"].cpu().numpy(), fw["tiles_per_gauss"])
            assert np.array_equal(meta["isect_offsets"].cpu().numpy(), fw["isect_offsets"])
        else:
            # (the flipped Gaussians themselves are bounded above -- 1e-4 N radii, 2e-4 N rectangles; this bounds the TILES they
            #  exempt: 2 % by default; 5 M Gaussians at 4K put 80 flips (1.6e-5 N) on 2.8 % of the tiles and pass 0.05)
            assert tmask.mean() <= max_flip_tile_frac or tmask.sum() <= 16, f"{tmask.mean()} of the tiles touched by rounding flips"
        swapped = _compare_lists(meta, fw, tmask)
        if swapped.any():
            exact_lists = False
            tmask = tmask | swapped.reshape(tmask.shape)
    razor = razor_mask(fw)
    H, W = razor.shape[1:]
    loose = razor | np.repeat(np.repeat(tmask, tile, axis=1), tile, axis=2)[:, :H, :W]
    parity_log.record(flip_tile_frac=float(tmask.mean()), max_flip_tile_frac_allowed=float(max_flip_tile_frac))
    parity_log.record(n_radius_flips=int(mism.sum()), n_rectangle_flips=n_edge, razor_fraction=float(razor.mean()),
                      loose_fraction=float(loose.mean()), max_means2d_err_ulps=e_mu, max_conics_rel_err=e_con, max_depths_rel_err=e_dep,
                      n_gaussians=int(mism.size), n_isects=int(fw["n_isects"]))
    return dict(exact_lists=exact_lists, razor=razor, loose=loose, lists=lists)


def check_forward(hip, fw, max_razor_frac=2e-2, lists=True, outlier_frac=0.0, geom_slack=1.0, max_flip_tile_frac=0.02):
    """Forward parity: forward_report's integer checks, then every pixel outside `loose` within 1e-4; razor pixels must
    stay rare and even they are bounded by one flipped contributor's weight."""
    assert max_razor_frac <= MAX_RAZOR_FRAC
    rep = hip.get("report")
    if rep is None or rep["lists"] != lists or rep.get("fw") is not fw:
        rep = forward_report(hip["meta"], fw, lists, geom_slack, max_flip_tile_frac)
    err = np.abs(hip["img"].detach().cpu().numpy() - fw["render_colors"]).max(-1)
    aerr = np.abs(hip["alpha"].detach().cpu().numpy() - fw["render_alphas"])[..., 0]
    razor, strict = rep["razor"], ~rep["loose"]
    assert razor.mean() <= max_razor_frac, f"razor-edge pixel fraction {razor.mean()}"
    # outlier_frac > 0 only where the reference itself is the wrong precision for a pixel-exact claim
    # (fp64 oracle at hundreds of contributors per pixel; the fp32 oracle is then checked strictly)
    assert (err[strict] > FWD_ATOL).sum() <= outlier_frac * strict.sum(), f"forward RGB err {err[strict].max()}, {(err[strict] > FWD_ATOL).sum()} pixels"
    assert (aerr[strict] > FWD_ATOL).sum() <= outlier_frac * strict.sum(), f"forward alpha err {aerr[strict].max()}"
    cmax = max(1.0, float(fw["colors"].max()))
    assert err.max(initial=0) <= 4.0 / 255.0 * cmax, "even a flipped contributor is bounded by ~its weight"
    parity_log.record(n_forward_checks=1, max_forward_err_strict=float(max(err[strict].max(initial=0), aerr[strict].max(initial=0))),
                      max_forward_err_any=float(err.max(initial=0)), exact_lists=bool(rep["exact_lists"]))
    return rep["exact_lists"]


NEEDLE_KAPPA = 100.0   # 10:1 aspect ratio of the screen-space footprint (the reference's max_scale_ratio, model/gaussian.py:84)


def needle_factor(fw):
    """[N] tolerance factor max(1, kappa / 100), kappa = condition number of the Gaussian's 2-D covariance (largest over
    the cameras that see it).  v_conic -> v_cov2d = -X v_conic X multiplies by the inverse twice and, for a needle, cancels
    down to the thin direction: the fp32 rounding of the blend's v_conic sums (~1e-5 relative, measured) comes out
    multiplied by ~kappa in the quaternion / scale gradients whatever precision the projection VJP itself runs in (it
    runs in fp64).  Up to kappa = 100 the north_star tolerance applies unchanged."""
    A, B, C = (fw["conics"][..., i].astype(np.float64) for i in range(3))
    mid, det = 0.5 * (A + C), A * C - B * B
    disc = np.sqrt(np.maximum(mid * mid - det, 0.0))
    with np.errstate(divide="ignore", invalid="ignore"):
        kappa = np.where((fw["radii"] > 0) & (det > 0), (mid + disc) / np.maximum(mid - disc, 1e-300), 1.0)
    return np.maximum(1.0, kappa.max(axis=0) / NEEDLE_KAPPA)


GRAD_L2_RTOL = 1e-4        # ||hip - ref||_2 / ||ref||_2 per tensor
ROW_FLOOR = 1e-3           # per-Gaussian criterion: |delta| <= rtol * max(|ref row|_max, ROW_FLOOR * tensor max)
ROW_BAD_MAX = 5e-3         # fraction of Gaussians (rows) allowed beyond the per-Gaussian criterion, or ROW_BAD_ABS rows if that is more ...
ROW_BAD_ABS = 3
ROW_BAD_HARD = 1e-2        # ... and, beyond that, at most what an INDEPENDENT fp32 evaluation of the same algorithm (the fp32 build of the C
                           # oracle) leaves beyond the same per-row tolerance on the same scene, never more than 1e-2 (round 4's flat cap).
                           # Settled by experiment in round 5 (tools/acc64_ab.py, profiles/r05_acc64_ab.json): the seven
                           # worst sweep configurations (GS_FUZZ_SCALE=2: 176, 1444, 104; scale 1: 730, 113, 444, 254) hold 22 such rows --
                           # all with a reference magnitude <= 4.3e-3 of their tensor's largest entry (pixel terms that cancel).  With the
                           # 11 per-entry sums of blend_bwd in DOUBLE (-DGS_BWD_ACC64) the same 22 rows remain: it is not the accumulation;
                           # the fp32 build of the oracle has 89 on the same scenes, 10 where the HIP path has its worst 8 (of 1187,
                           # v_opacities): it is the fp32 rounding of the per-pixel terms themselves (exp2, alpha, the T chain), shared by
                           # any fp32 evaluation -- not a defect of the s_vs / v_op path.  Suite scenes measure <= 5.5e-4.
UNMASKED_L2_RTOL = 4e-4    # unmasked upstream gradient against the fp32 oracle: isolated threshold flips, bounded
UNMASKED_MAX_RTOL = 6e-3   # (round 6: 5e-4 / 1e-2 before -- measured 3.6e-4 / 5.0e-3, tightened so that a regression shows)


def _grad_metrics(g, ref, rtol, relax=None):
    """Three views of one tensor's error: max-norm relative to the tensor's largest reference magnitude (the north_star
    criterion), relative L2 over the tensor, and the fraction of ROWS (Gaussians) holding an element beyond
    rtol * max(the row's own largest reference magnitude, ROW_FLOOR * tensor max) -- a Gaussian with a small gradient can no
    longer be 100 % wrong unnoticed.  `relax` [N]: the needle factor the two covariance-inverse tensors are divided by."""
    g, ref = np.asarray(g, np.float64), np.asarray(ref, np.float64)
    n = ref.shape[0] if ref.ndim else 1
    d = np.abs(g - ref).reshape(n, -1)
    r = np.abs(ref).reshape(n, -1)
    if relax is not None:
        d = d / relax[:, None]
    tmax = r.max(initial=0) + 1e-30
    row_tol = rtol * np.maximum(r.max(axis=1, initial=0), ROW_FLOOR * tmax)
    return {"max": float(d.max(initial=0) / tmax), "l2": float(np.linalg.norm(d) / (np.linalg.norm(r) + 1e-30)),
            "row_bad": float(np.mean(d.max(axis=1, initial=0) > row_tol)) if n else 0.0, "rows": int(n),
            "row_bad_tensor_max": float(np.mean(d.max(axis=1, initial=0) > rtol * tmax)) if n else 0.0}


def check_backward(hip, fw, rtol=GRAD_RTOL, ref_transform=None):
    """Gradients against the fp64 oracle, three criteria per tensor (`_grad_metrics`): within `rtol` of the tensor's
    largest reference magnitude (north_star), relative L2 <= 1e-4, and at most ROW_BAD_MAX of the Gaussians beyond `rtol` of
    their OWN row's magnitude (floored at 1e-3 of the tensor's).  The upstream gradients are zero on the `loose` pixels
    (run_hip(fw=...)), where fp32 arithmetic may legitimately blend a different contributor set; gradient flow THROUGH those
    pixels is compared by check_backward_unmasked against the fp32 build of the oracle.  Should a flip survive the mask
    (accumulated rounding of T over hundreds of contributors is not part of the razor margin), the fp32 build of the same
    oracle, which takes the path's decisions, is the arbiter -- printed, counted, and then for EVERY tensor of the call.
    Quaternion and scale gradients of needles (needle_factor) are held to rtol x kappa / 100; how many Gaussians that
    relaxes is recorded."""
    bw = CO.backward(fw, hip["vc"], hip["va"])
    names = ["v_means", "v_quats", "v_scales", "v_opacities", "v_colors"]
    relax = needle_factor(fw)
    ref_transform = ref_transform or {}   # e.g. {"v_colors": lambda x: x.sum(0)} for [N,3] colours shared by C cameras

    def errors(ref_bw):
        out = {}
        for name, g in zip(names, hip["grads"]):
            ref = ref_transform.get(name, lambda x: x)(ref_bw[name])
            # (the two tensors behind the inverse of the 2-D covariance carry the needle factor)
            out[name] = _grad_metrics(g.cpu().numpy(), ref, rtol, relax if name in ("v_quats", "v_scales") else None)
        ag, ref = hip["meta"]["means2d"].absgrad.cpu().numpy(), ref_bw["v_means2d_abs"]
        out["absgrad"] = _grad_metrics(ag.reshape(-1, ag.shape[-1]), ref.reshape(-1, ref.shape[-1]), rtol)
        return out

    err = errors(bw)
    if max(e["max"] for e in err.values()) > rtol:
        fw32 = oracle_fp32(fw)
        print(f"[parity] fp64 arbiter failed ({ {k: float('%.2e' % v['max']) for k, v in err.items()} }); fp32 oracle arbitrates all tensors")
        parity_log.record(n_fp32_arbiter_uses=1)
        err = errors(CO.backward(fw32, hip["vc"].astype(np.float32), hip["va"].astype(np.float32)))
    vis = (fw["radii"] > 0).any(axis=0)
    parity_log.record(n_backward_checks=1, max_rel_grad_err={k: v["max"] for k, v in err.items()},
                      max_rel_l2_grad_err={k: v["l2"] for k, v in err.items()},
                      max_row_bad_frac={k: v["row_bad"] for k, v in err.items()},
                      max_needle_factor=float(relax.max(initial=1.0)), n_needles_relaxed=int(((relax > 1.0) & vis).sum()),
                      n_visible_gaussians=int(vis.sum()))
    err32 = None
    for name, e in err.items():
        assert e["max"] <= rtol, f"{name}: rel err {e['max']}"
        assert e["l2"] <= GRAD_L2_RTOL * (rtol / GRAD_RTOL), f"{name}: relative L2 err {e['l2']}"
        allowed = max(ROW_BAD_MAX, (ROW_BAD_ABS + 0.5) / max(e["rows"], 1))
        if e["row_bad"] > allowed:
            # more rows off than the flat allowance: then no more than an independent fp32 evaluation of the same algorithm leaves
            # (rows whose pixel terms cancel; ROW_BAD_HARD's comment), and never more than round 4's cap
            if err32 is None:
                bw32 = CO.backward(oracle_fp32(fw), hip["vc"].astype(np.float32), hip["va"].astype(np.float32))
                err32 = {}
                for nm in names:
                    err32[nm] = _grad_metrics(ref_transform.get(nm, lambda x: x)(bw32[nm]), ref_transform.get(nm, lambda x: x)(bw[nm]), rtol,
                                              relax if nm in ("v_quats", "v_scales") else None)
                err32["absgrad"] = _grad_metrics(bw32["v_means2d_abs"].reshape(-1, 2), bw["v_means2d_abs"].reshape(-1, 2), rtol)
                parity_log.record(n_row_criterion_fp32_oracle_uses=1, row_bad_frac_fp32_oracle={k: v["row_bad"] for k, v in err32.items()})
                print(f"[parity] {name}: {e['row_bad']:.4f} of the rows beyond the per-Gaussian tolerance; the fp32 oracle leaves {err32[name]['row_bad']:.4f}")
            allowed = min(ROW_BAD_HARD, max(allowed, err32[name]["row_bad"]))
        assert e["row_bad"] <= allowed, f"{name}: {e['row_bad']} of the {e['rows']} Gaussians beyond {rtol} of their own gradient"
    return bw


def check_backward_unmasked(sc, fw, culling="gsplat", use_bg=True, seed=17):
    """ONE unmasked backward per scene: a random upstream gradient on EVERY pixel, razor pixels
    included, against the fp32 build of the oracle -- an independent implementation in the arithmetic the device blends
    in, which takes its own fp32 decisions at the thresholds.  Two fp32 implementations do not flip the same threshold
    contributors, so the criterion is the isolated-flip rule of the S3 train loop: L2 over the tensor <= 5e-4, no element
    beyond 1e-2 of the tensor's largest, and the fraction of Gaussians beyond the plain 1e-3 is recorded."""
    hip = run_hip(sc, culling=culling, use_bg=use_bg, seed=seed)
    fw32 = oracle_fp32(fw)
    bw = CO.backward(fw32, hip["vc"].astype(np.float32), hip["va"].astype(np.float32))
    names = ["v_means", "v_quats", "v_scales", "v_opacities", "v_colors"]
    relax = needle_factor(fw)
    err = {}
    for name, g in zip(names, hip["grads"]):
        err[name] = _grad_metrics(g.cpu().numpy(), bw[name], GRAD_RTOL, relax if name in ("v_quats", "v_scales") else None)
    ag, ref = hip["meta"]["means2d"].absgrad.cpu().numpy(), bw["v_means2d_abs"]
    err["absgrad"] = _grad_metrics(ag.reshape(-1, ag.shape[-1]), ref.reshape(-1, ref.shape[-1]), GRAD_RTOL)
    parity_log.record(n_unmasked_backward_checks=1, unmasked_vs_fp32_max={k: v["max"] for k, v in err.items()},
                      unmasked_vs_fp32_l2={k: v["l2"] for k, v in err.items()},
                      unmasked_vs_fp32_rows_beyond_1e3={k: v["row_bad_tensor_max"] for k, v in err.items()})
    for name, e in err.items():
        assert e["l2"] <= UNMASKED_L2_RTOL, f"unmasked {name}: relative L2 err {e['l2']}"
        assert e["max"] <= UNMASKED_MAX_RTOL, f"unmasked {name}: rel err {e['max']}"
    return err


SCENES = {
    "tiny_sh3": dict(n=50, width=40, height=24, sh_degree=3, seed=1, scale_range=(0.05, 0.4), dist=4.0),
    "ragged_sh2_2views": dict(n=2000, width=100, height=70, sh_degree=2, seed=2, k_store=16, n_views=2, scale_range=(0.02, 0.3), dist=4.0),
    "odd_n_sh1_k4": dict(n=1237, width=333, height=77, sh_degree=1, seed=3, k_store=4, scale_range=(0.02, 0.3), dist=4.0, white_bg=False),
    "sh0_k1": dict(n=4000, width=128, height=128, sh_degree=0, seed=4, scale_range=(0.01, 0.2), dist=4.0),
    "sh0_of_k16_3views": dict(n=900, width=64, height=48, sh_degree=0, seed=5, k_store=16, n_views=3, scale_range=(0.03, 0.3), dist=4.0),
}


@pytest.mark.parametrize("culling", ["gsplat", "gsplat_eager", "tight"])
@pytest.mark.parametrize("name", list(SCENES))
def test_forward_backward_parity_small(name, culling):
    sc = make_scene(**SCENES[name])
    fw = run_oracle(sc)
    hip = run_hip(sc, culling=culling, fw=fw)
    check_forward(hip, fw, lists=culling != "tight")
    check_backward(hip, fw)
    if culling == "gsplat":
        check_backward_unmasked(sc, fw, culling)


def test_means2d_is_a_graph_tensor_with_grad_like_gsplat():
    """gsplat hands `meta["means2d"]` out as a tensor of the autograd graph: `retain_grad()` before backward, `.grad` (dL/d means2d)
    and `.absgrad` after it.  Same calls here; `.grad` against the oracle's v_means2d; not there under no_grad."""
    from easy_gaussian_splatting_amd.rendering import rasterization
    sc = make_scene(**SCENES["ragged_sh2_2views"])
    fw = run_oracle(sc)
    t = to_dev(sc)
    ins = [t[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "shs")]
    img, alpha, meta = rasterization(*ins, t["viewmats"], t["Ks"], int(sc["width"]), int(sc["height"]), sh_degree=int(sc["sh_degree"]),
                                     packed=False, backgrounds=t["backgrounds"], absgrad=True)
    assert meta["means2d"].requires_grad
    meta["means2d"].retain_grad()   # (what gsplat's strategies call; harmless here)
    g = torch.Generator().manual_seed(5)
    vc, va = torch.randn(img.shape, generator=g), torch.randn(alpha.shape, generator=g)
    hip = dict(img=img, alpha=alpha, meta=meta, ins=ins)
    vc, va = mask_upstream(hip, fw, vc, va)
    ((img * vc.to(dev())).sum() + (alpha * va.to(dev())).sum()).backward()
    bw = CO.backward(fw, vc.numpy().astype(np.float64), va.numpy().astype(np.float64))
    got, ref = meta["means2d"].grad.cpu().numpy(), bw["v_means2d"]
    assert got.shape == ref.shape and np.abs(got - ref).max() <= GRAD_RTOL * np.abs(ref).max()
    assert np.abs(meta["means2d"].absgrad.cpu().numpy() - bw["v_means2d_abs"]).max() <= GRAD_RTOL * np.abs(bw["v_means2d_abs"]).max()
    with torch.no_grad():
        _, _, meta2 = rasterization(*ins, t["viewmats"], t["Ks"], int(sc["width"]), int(sc["height"]), sh_degree=int(sc["sh_degree"]),
                                    packed=False, backgrounds=t["backgrounds"])
    assert not meta2["means2d"].requires_grad and torch.equal(meta2["means2d"], meta["means2d"].detach())


def test_tight_culling_is_render_equivalent_subset():
    """Default tight tile culling: lists are a subset of gsplat's, image and gradients identical."""
    sc = make_scene(6000, 200, 150, sh_degree=1, seed=31, k_store=4, scale_range=(0.01, 0.3), dist=4.0)
    sc["opacities"] = np.random.default_rng(5).uniform(0.002, 1.0, 6000).astype(np.float32)  # some below 1/255
    a, b, e = run_hip(sc, culling="gsplat"), run_hip(sc, culling="tight"), run_hip(sc, culling="gsplat_eager")
    # "gsplat" (the default) renders from the short lists and builds gsplat's arrays when they are read: the arrays must be
    # those of the pipeline that walks gsplat's lists itself, bit for bit, and so must the image
    for k in ("tiles_per_gauss", "flatten_ids", "isect_offsets", "isect_ids", "radii", "means2d"):
        assert torch.equal(a["meta"][k], e["meta"][k]), k
    assert torch.equal(a["img"], e["img"]) and torch.equal(a["alpha"], e["alpha"])
    for ga, ge in zip(a["grads"], e["grads"]):
        assert float((ga - ge).abs().max()) <= 1e-5 * float(ge.abs().max())
    ta, tb = a["meta"]["tiles_per_gauss"], b["meta"]["tiles_per_gauss"]
    assert bool((tb <= ta).all()) and int(tb.sum()) < int(ta.sum())
    assert torch.equal(a["meta"]["radii"], b["meta"]["radii"]) and torch.equal(a["meta"]["means2d"], b["meta"]["means2d"])
    assert torch.equal(a["img"], b["img"]) and torch.equal(a["alpha"], b["alpha"]), "bitwise identical image"
    for ga, gb in zip(a["grads"], b["grads"]):
        assert float((ga - gb).abs().max()) <= 1e-5 * float(ga.abs().max())


@pytest.mark.parametrize("name", ["ragged_sh2_2views", "odd_n_sh1_k4", "sh3_bg"])
def test_backward_without_the_forward_direction_jacobian(name, monkeypatch):
    """Training with SH colours the forward leaves d colour / d view direction (gs_project_fwd: sh_jac) and the backward never
    reads a coefficient.  The coefficient-staging backward (sh_jac = NULL: what a caller of the C ABI without that buffer gets)
    must stay on the oracle too, and the two must agree: SH gradients bit for bit (Y (x) v_pre either way), v_means to rounding."""
    from easy_gaussian_splatting_amd import rendering
    cfg = SCENES.get(name) or dict(n=3000, width=160, height=96, sh_degree=3, seed=9, scale_range=(0.02, 0.3), dist=4.0)
    sc = make_scene(**cfg)
    fw = run_oracle(sc)
    hip = run_hip(sc, fw=fw)
    monkeypatch.setattr(rendering, "_SH_JAC", False)
    ref = run_hip(sc, upstream=(hip["vc"], hip["va"]))
    check_forward(ref, fw)
    check_backward(ref, fw)
    assert torch.equal(ref["img"], hip["img"])
    g_j, g_c = hip["grads"], ref["grads"]
    assert torch.equal(g_j[4], g_c[4])                      # v_shs
    for i in (1, 2, 3):                                     # quats, scales, opacities: untouched by the colour path
        assert torch.equal(g_j[i], g_c[i])
    scale = float(g_c[0].abs().max())
    assert float((g_j[0] - g_c[0]).abs().max()) <= 2e-5 * scale


@pytest.mark.parametrize("name", ["ragged_sh2_2views", "sh3_split"])
def test_one_launch_projection_equals_the_two_stage_form(name, monkeypatch):
    """The eager forward runs geometry + SH colour as ONE launch (stage 0; with the reference's split SH3 layout the SH slice is
    requested ahead of the projection chain); `GS_FWD_SPLIT=1` keeps the two launches around the tile count.  Same results,
    bit for bit, gradients included."""
    from easy_gaussian_splatting_amd import rendering
    if name == "sh3_split":
        sc = make_scene(3001, 150, 90, sh_degree=3, seed=23, k_store=16, scale_range=(0.02, 0.2), dist=4.0)
    else:
        sc = make_scene(**SCENES[name])
    t = to_dev(sc)
    W, H = int(sc["width"]), int(sc["height"])

    def run():
        base = [t[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities")]
        if name == "sh3_split":
            cols = (t["shs"][:, :1].clone().contiguous().requires_grad_(True), t["shs"][:, 1:].clone().contiguous().requires_grad_(True))
            leaves = base + list(cols)
        else:
            cols = t["shs"].clone().requires_grad_(True)
            leaves = base + [cols]
        img, alpha, meta = rendering.rasterization(*base, cols, t["viewmats"], t["Ks"], W, H, sh_degree=int(sc["sh_degree"]), packed=False,
                                                   backgrounds=t["backgrounds"], absgrad=True)
        vc = torch.randn(img.shape, generator=torch.Generator().manual_seed(3)).to(dev())
        grads = torch.autograd.grad((img * vc).sum() + alpha.sum(), leaves)
        return img, alpha, meta, grads

    one = run()
    monkeypatch.setattr(rendering, "_SPLIT_PROJECT", True)
    two = run()
    assert torch.equal(one[0], two[0]) and torch.equal(one[1], two[1])
    for k in ("radii", "means2d", "depths", "conics", "tiles_per_gauss", "flatten_ids", "isect_offsets"):
        assert torch.equal(one[2][k], two[2][k]), k
    for a, b in zip(one[3], two[3]):
        assert torch.equal(a, b)


def test_config_s1_parity():
    """BASELINE.json configs[0]: 10k random Gaussians, 256x256, SH degree 0."""
    sc = config_s1()
    fw = run_oracle(sc)
    hip = run_hip(sc, fw=fw)
    check_forward(hip, fw)
    check_backward(hip, fw)


def test_no_background_and_no_alpha_grad():
    sc = make_scene(600, 80, 60, sh_degree=3, seed=6, scale_range=(0.03, 0.3), dist=4.0)
    from easy_gaussian_splatting_amd.rendering import rasterization
    t = to_dev(sc)
    ins = [t[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "shs")]
    img, alpha, meta = rasterization(*ins, t["viewmats"], t["Ks"], 80, 60, sh_degree=3, packed=False, absgrad=True)
    vc = torch.randn(img.shape, generator=torch.Generator().manual_seed(0))
    grads = torch.autograd.grad((img * vc.to(dev())).sum(), ins)  # alpha unused -> v_alphas is None/zero
    fw = run_oracle(sc, use_bg=False)
    hip = dict(img=img, alpha=alpha, meta=meta, grads=grads, vc=vc.numpy().astype(np.float64), va=np.zeros(alpha.shape))
    check_forward(hip, fw)
    check_backward(hip, fw)


def test_post_activation_colours_path():
    """sh_degree=None: colours are used as given, [N,3] and [C,N,3]."""
    from easy_gaussian_splatting_amd.rendering import rasterization
    sc = make_scene(500, 64, 64, sh_degree=0, seed=8, n_views=2, scale_range=(0.03, 0.3), dist=4.0)
    t = to_dev(sc)
    rng = np.random.default_rng(0)
    for shape in ((500, 3), (2, 500, 3)):
        cols = rng.random(shape).astype(np.float32)
        c_t = torch.from_numpy(cols).to(dev()).requires_grad_(True)
        base = [t[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities")]
        img, alpha, meta = rasterization(*base, c_t, t["viewmats"], t["Ks"], 64, 64, sh_degree=None, packed=False,
                                         backgrounds=t["backgrounds"], absgrad=True)
        fw = CO.render(sc["means"], sc["quats"], sc["scales"], sc["opacities"], cols, sc["viewmats"], sc["Ks"], 64, 64,
                       sh_degree=None, backgrounds=sc["backgrounds"], dtype=np.float64)
        hip = dict(img=img, alpha=alpha, meta=meta)
        g = torch.Generator().manual_seed(1)
        vc, va = mask_upstream(hip, fw, torch.randn(img.shape, generator=g), torch.randn(alpha.shape, generator=g))
        grads = torch.autograd.grad((img * vc.to(dev())).sum() + (alpha * va.to(dev())).sum(), base + [c_t])
        assert grads[4].shape == c_t.shape
        hip.update(grads=grads, vc=vc.numpy().astype(np.float64), va=va.numpy().astype(np.float64))
        check_forward(hip, fw)   # the same 1e-4 / bit-exact-list bar as the SH path (was: 5e-3 on the image)
        # all five gradients + absgrad; [N,3] colours are shared by the cameras: their gradient is the sum over views
        check_backward(hip, fw, ref_transform={"v_colors": (lambda x: x.sum(0))} if len(shape) == 2 else None)


@pytest.mark.parametrize("tag", ["a", "b", "c", "d", "e", "f"])
def test_golden_fixtures(tag):
    """Committed oracle fixtures (tests/golden/make_golden.py): every integer output bit-exact,
    image and gradients within the north_star tolerances."""
    z = dict(np.load(os.path.join(GOLD, f"oracle_scene_{tag}.npz")))
    sc = {k: z[k] for k in ("means", "quats", "scales", "opacities", "shs", "viewmats", "Ks", "backgrounds")}
    sc.update(width=int(z["width"]), height=int(z["height"]), sh_degree=int(z["sh_degree"]))
    from easy_gaussian_splatting_amd.rendering import rasterization
    t = to_dev(sc)
    ins = [t[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "shs")]
    img, alpha, meta = rasterization(*ins, t["viewmats"], t["Ks"], sc["width"], sc["height"], sh_degree=sc["sh_degree"],
                                     packed=False, backgrounds=t["backgrounds"], absgrad=True, _tile_culling="gsplat")
    for k in ("radii", "tiles_per_gauss", "flatten_ids", "isect_offsets"):
        assert np.array_equal(meta[k].cpu().numpy(), z[k]), k
    assert np.array_equal(meta["isect_ids"].cpu().numpy() >> 32, z["isect_ids"] >> 32)  # camera|tile part
    assert np.abs(img.detach().cpu().numpy() - z["render_colors"]).max() <= FWD_ATOL
    assert np.abs(alpha.detach().cpu().numpy() - z["render_alphas"]).max() <= FWD_ATOL
    vc = torch.from_numpy(z["v_render_colors"]).float().to(dev()); va = torch.from_numpy(z["v_render_alphas"]).float().to(dev())
    grads = torch.autograd.grad((img * vc).sum() + (alpha * va).sum(), ins)
    for g, k in zip(grads, ("v_means", "v_quats", "v_scales", "v_opacities", "v_shs")):
        assert np.abs(g.cpu().numpy() - z[k]).max() <= GRAD_RTOL * np.abs(z[k]).max(), k
    assert np.abs(meta["means2d"].absgrad.cpu().numpy() - z["absgrad"]).max() <= GRAD_RTOL * np.abs(z["absgrad"]).max()


GSPLAT_FIXTURES = sorted(f for f in os.listdir(GOLD) if f.startswith("gsplat_") and f.endswith(".npz"))


@pytest.mark.skipif(not GSPLAT_FIXTURES, reason="no tests/golden/gsplat_*.npz (tests/golden/make_gsplat_golden.py, wherever gsplat==1.0.0 exists)")
@pytest.mark.parametrize("name", GSPLAT_FIXTURES or ["-"])
def test_hip_path_matches_gsplat_fixtures(name):
    """The HIP path against outputs of gsplat 1.0.0 itself (the GPU twin of tests/test_oracle.py::test_oracle_matches_gsplat_fixtures):
    forward RGB / alpha within 1e-4, every gradient and absgrad within 1e-3 of the tensor's largest entry, integer outputs
    bit-exact where no radius sits on a last-bit flip."""
    z = dict(np.load(os.path.join(GOLD, name)))
    sc = {k: z[k] for k in ("means", "quats", "scales", "opacities", "shs", "viewmats", "Ks", "backgrounds")}
    sc.update(width=int(z["width"]), height=int(z["height"]), sh_degree=int(z["sh_degree"]))
    from easy_gaussian_splatting_amd.rendering import rasterization
    t = to_dev(sc)
    ins = [t[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "shs")]
    img, alpha, meta = rasterization(*ins, t["viewmats"], t["Ks"], sc["width"], sc["height"], sh_degree=sc["sh_degree"],
                                     packed=False, backgrounds=t["backgrounds"], absgrad=True, _tile_culling="gsplat")
    radii = meta["radii"].cpu().numpy()
    flips = int((radii != z["radii"]).sum())
    assert flips <= max(1, int(1e-4 * radii.size)), flips
    if flips == 0:
        for k in ("tiles_per_gauss", "flatten_ids", "isect_offsets"):
            assert np.array_equal(meta[k].cpu().numpy().reshape(-1), z[k].reshape(-1)), k
        assert np.abs(img.detach().cpu().numpy() - z["render_colors"]).max() <= FWD_ATOL
        assert np.abs(alpha.detach().cpu().numpy() - z["render_alphas"]).max() <= FWD_ATOL
    vc = torch.from_numpy(z["v_render_colors"]).float().to(dev()); va = torch.from_numpy(z["v_render_alphas"]).float().to(dev())
    grads = torch.autograd.grad((img * vc).sum() + (alpha * va).sum(), ins)
    for g, k in zip(grads, ("v_means", "v_quats", "v_scales", "v_opacities", "v_shs")):
        assert np.abs(g.cpu().numpy() - z[k]).max() <= GRAD_RTOL * np.abs(z[k]).max(), k
    assert np.abs(meta["means2d"].absgrad.cpu().numpy() - z["absgrad"]).max() <= GRAD_RTOL * np.abs(z["absgrad"]).max()


def test_empty_and_invisible_inputs():
    from easy_gaussian_splatting_amd.rendering import rasterization
    d = dev()
    V = torch.eye(4, device=d)[None]; K = torch.tensor([[50.0, 0, 16], [0, 50.0, 16], [0, 0, 1]], device=d)[None]
    bg = torch.tensor([[0.25, 0.5, 0.75]], device=d)
    # N = 0
    z = lambda *s: torch.zeros(*s, device=d)
    img, alpha, meta = rasterization(z(0, 3), z(0, 4), z(0, 3), z(0), z(0, 16, 3), V, K, 32, 32, sh_degree=3, packed=False, backgrounds=bg)
    assert torch.allclose(img, bg.expand(1, 32, 32, 3).contiguous()) and float(alpha.abs().max()) == 0.0
    assert meta["flatten_ids"].numel() == 0 and meta["radii"].shape == (1, 0)
    # everything behind the camera: zero visible, gradients exist and are zero
    means = torch.tensor([[0.0, 0, -2.0], [0.1, 0, -3.0]], device=d, requires_grad=True)
    quats = torch.ones(2, 4, device=d, requires_grad=True); scales = torch.full((2, 3), 0.1, device=d, requires_grad=True)
    op = torch.full((2,), 0.5, device=d, requires_grad=True); sh = torch.zeros(2, 16, 3, device=d, requires_grad=True)
    img, alpha, meta = rasterization(means, quats, scales, op, sh, V, K, 32, 32, sh_degree=3, packed=False, backgrounds=bg, absgrad=True)
    assert int((meta["radii"] > 0).sum()) == 0 and torch.allclose(img, bg.expand(1, 32, 32, 3).contiguous())
    img.sum().backward()
    for p in (means, quats, scales, op, sh):
        assert p.grad is not None and float(p.grad.abs().max()) == 0.0
    assert float(meta["means2d"].absgrad.abs().max()) == 0.0


def test_large_tile_lists_hit_every_sort_class():
    """Tiles with > 2048 (LDS large class) and > 16384 (global-memory class) entries, and Gaussians
    that cover every tile (wave-cooperative binning and row-reduction paths)."""
    sc = dense_scene(20000, 12)   # faint / never-taken / flat mixture: lists are walked to the end
    fw = run_oracle(sc)
    hip = run_hip(sc, fw=fw)
    counts = np.diff(np.append(fw["isect_offsets"].reshape(-1), fw["n_isects"]))
    assert counts.max() > 16384 and fw["tiles_per_gauss"].max() == 12
    assert float(fw["render_alphas"].max()) < 1 - 2e-4, "no pixel may saturate: every list is walked to its end"
    assert check_forward(hip, fw, max_razor_frac=MAX_RAZOR_FRAC)  # lists must match exactly (bit-exact sort)
    check_backward(hip, fw)
    sc2 = dense_scene(3000, 13)
    fw2 = run_oracle(sc2)
    hip2 = run_hip(sc2, fw=fw2)
    c2 = np.diff(np.append(fw2["isect_offsets"].reshape(-1), fw2["n_isects"]))
    assert 2048 < c2.max() <= 16384
    assert check_forward(hip2, fw2, max_razor_frac=MAX_RAZOR_FRAC)
    check_backward(hip2, fw2)


@pytest.mark.parametrize("culling", ["gsplat", "gsplat_eager", "tight"])
def test_long_lists_heavy_tailed_footprints(culling):
    """Real-capture-like lists: footprints of hundreds of tiles, mean list > 2 000 entries, saturating pixels (the
    regime where emission + per-tile sort lead the forward): lists bit-exact, image and gradients within tolerance."""
    sc = config_long_lists()
    fw = run_oracle(sc)
    counts = np.diff(np.append(fw["isect_offsets"].reshape(-1), fw["n_isects"]))
    assert counts.mean() > 2000 and fw["tiles_per_gauss"].max() > 300
    hip = run_hip(sc, culling=culling, fw=fw)
    check_forward(hip, fw, max_razor_frac=MAX_RAZOR_FRAC, lists=culling != "tight")
    check_backward(hip, fw)


def test_depth_ties_break_by_index():
    """Equal depths in one tile: order must follow the flatten index (stable-sort contract A.3)."""
    from easy_gaussian_splatting_amd.rendering import rasterization
    d = dev()
    n = 200
    means = torch.zeros(n, 3, device=d); means[:, 2] = 2.0
    means[:, 0] = torch.linspace(-0.05, 0.05, n, device=d)
    V = torch.eye(4, device=d)[None]; K = torch.tensor([[40.0, 0, 8], [0, 40.0, 8], [0, 0, 1]], device=d)[None]
    img, alpha, meta = rasterization(means, torch.ones(n, 4, device=d), torch.full((n, 3), 0.05, device=d),
                                     torch.full((n,), 0.1, device=d), torch.rand(n, 3, device=d), V, K, 16, 16,
                                     sh_degree=None, packed=False, _tile_culling="gsplat")
    ids = meta["flatten_ids"].cpu().numpy()
    assert len(ids) == n and np.array_equal(ids, np.arange(n))


def test_full_size_properties():
    """BASELINE.json metric workload (1 M Gaussians, 1920x1080, SH3): size-independent properties."""
    from easy_gaussian_splatting_amd.rendering import rasterization
    sc = config_bench_1m()
    t = to_dev(sc)
    W, H = sc["width"], sc["height"]
    args = [t[k] for k in ("means", "quats", "scales", "opacities", "shs")]
    img, alpha, meta = rasterization(*args, t["viewmats"], t["Ks"], W, H, sh_degree=3, packed=False, backgrounds=t["backgrounds"])
    assert torch.isfinite(img).all() and float(alpha.min()) >= 0.0 and float(alpha.max()) <= 1.0
    # list structure: offsets monotone, every run sorted by depth then index, I == sum tiles_per_gauss
    I = meta["flatten_ids"].numel()
    assert I == int(meta["tiles_per_gauss"].sum())
    offs = meta["isect_offsets"].reshape(-1).long()
    assert bool((offs[1:] >= offs[:-1]).all()) and int(offs[0]) == 0
    keys = meta["isect_ids"]
    assert bool((keys[1:] >= keys[:-1]).all()), "(tile | depth) keys must be globally non-decreasing"
    same = keys[1:] == keys[:-1]
    fid = meta["flatten_ids"].long()
    assert bool((fid[1:][same] > fid[:-1][same]).all()), "ties ordered by flatten index"
    # every listed Gaussian is visible and its depth matches the key's low word
    assert bool((meta["radii"].reshape(-1)[fid] > 0).all())
    dbits = meta["depths"].reshape(-1)[fid].view(torch.int32).long()
    assert bool(((keys & 0xFFFFFFFF) == dbits).all())
    # permutation invariance of the image (sum order inside a pixel is depth order -> identical lists)
    perm = torch.randperm(args[0].shape[0], generator=torch.Generator().manual_seed(0)).to(dev())
    img_p, alpha_p, _ = rasterization(*[a[perm].contiguous() for a in args], t["viewmats"], t["Ks"], W, H, sh_degree=3,
                                      packed=False, backgrounds=t["backgrounds"])
    assert float((img - img_p).abs().max()) <= 4.0 / 255.0 * 2  # only exact depth ties can reorder
    assert float((img - img_p).abs().mean()) < 1e-6
    # linearity of the backward in the upstream gradient + determinism (no atomics anywhere)
    ins = [a.clone().requires_grad_(True) for a in args]
    img2, _, meta2 = rasterization(*ins, t["viewmats"], t["Ks"], W, H, sh_degree=3, packed=False, backgrounds=t["backgrounds"], absgrad=True)
    g = torch.Generator().manual_seed(3)
    v1, v2 = torch.randn(img2.shape, generator=g).to(dev()) / (W * H), torch.randn(img2.shape, generator=g).to(dev()) / (W * H)
    g1 = torch.autograd.grad((img2 * v1).sum(), ins, retain_graph=True)
    g2 = torch.autograd.grad((img2 * v2).sum(), ins, retain_graph=True)
    g12 = torch.autograd.grad((img2 * (v1 + v2)).sum(), ins, retain_graph=True)
    g12b = torch.autograd.grad((img2 * (v1 + v2)).sum(), ins)
    for a, b, c, cb in zip(g1, g2, g12, g12b):
        assert torch.equal(c, cb), "backward must be bitwise reproducible"
        assert float((a + b - c).abs().max()) <= 2e-3 * float(c.abs().max()) + 1e-12
    assert meta2["means2d"].absgrad.shape == (1, args[0].shape[0], 2)


def test_model_forward_and_statistics_mirror():
    """Rows a-1/a-2/a-3: GaussianModel.forward(data) / update_statistics against the oracle."""
    from easy_gaussian_splatting_amd.model import GaussianModel
    sc = make_scene(3000, 160, 96, sh_degree=3, seed=7, scale_range=(0.02, 0.2), dist=4.0)
    d = dev()
    T = lambda a: torch.from_numpy(a).to(d)
    op = np.clip(sc["opacities"], 1e-4, 1 - 1e-4)
    m = GaussianModel(means=T(sc["means"]), log_scales=torch.log(T(sc["scales"])), quats=T(sc["quats"]),
                      sh_0=T(sc["shs"])[:, :1].contiguous(), sh_rest=T(sc["shs"])[:, 1:].contiguous(),
                      logit_opacities=T(np.log(op / (1 - op)).astype(np.float32)), sh_degree=3, white_background=True).to(d)
    data = {"w2c": T(sc["viewmats"][0]), "K": T(sc["Ks"][0]), "width": 160, "height": 96}
    out = m(data)
    assert out["render_img"].shape == (96, 160, 3) and out["batch_xys"].shape == (1, 3000, 2) and out["batch_radii"].shape == (1, 3000)
    fw = CO.render(sc["means"], sc["quats"], sc["scales"], op, sc["shs"], sc["viewmats"], sc["Ks"], 160, 96, sh_degree=3,
                   backgrounds=np.ones((1, 3), np.float32), dtype=np.float64)
    # forward: the common bar (integer outputs bit-exact, every non-razor pixel within 1e-4) on the clamped image
    # (was: mean(err > 1e-4) < 1e-3)
    from easy_gaussian_splatting_amd.rendering import rasterization
    with torch.no_grad():   # (depths / conics are not part of the model's output: taken from the seam called with the same values)
        _, _, meta = rasterization(m.means, m.quats, m.scales, m.opacities, m.shs, data["w2c"][None], data["K"][None], 160, 96,
                                   sh_degree=3, packed=False, backgrounds=m.BACKGROUND[None])
    # (the model's scales are exp(log s) evaluated by torch; the oracle was fed s itself: an ulp apart, ~1e-6 on the conic)
    rep = forward_report({"radii": out["batch_radii"], "means2d": out["batch_xys"], "depths": meta["depths"], "conics": meta["conics"]},
                         fw, lists=False, geom_slack=16.0)
    strict = ~rep["loose"][0]
    ref_img = np.clip(fw["render_colors"][0], 0, 1)
    err = np.abs(out["render_img"].detach().cpu().numpy() - ref_img).max(-1)
    assert rep["razor"].mean() <= 2e-2 and err[strict].max() <= FWD_ATOL, (rep["razor"].mean(), err[strict].max())
    parity_log.record(n_forward_checks=1, max_forward_err_strict=float(err[strict].max()))
    # upstream gradient on the decided pixels, and away from the clamp's own two thresholds (0 and 1 +- 1e-4)
    inside = ((fw["render_colors"][0] > 1e-4) & (fw["render_colors"][0] < 1 - 1e-4)) | (fw["render_colors"][0] < -1e-4) | (fw["render_colors"][0] > 1 + 1e-4)
    vc = torch.randn(out["render_img"].shape, generator=torch.Generator().manual_seed(2)) * torch.from_numpy(strict[..., None] & inside)
    (out["render_img"] * vc.to(d)).sum().backward()
    m.update_statistics(data, out)
    vcl = vc.numpy().astype(np.float64) * ((fw["render_colors"][0] > 0) & (fw["render_colors"][0] < 1))   # d clamp
    bw = CO.backward(fw, vcl[None])
    vis = fw["radii"][0] > 0
    exp_g = np.where(vis, np.linalg.norm(bw["v_means2d_abs"][0], axis=-1) * 160, 0)
    assert np.abs(m.grad_norm_accum.cpu().numpy() - exp_g).max() <= GRAD_RTOL * exp_g.max()
    assert np.array_equal(m.collecting_counts.cpu().numpy() > 0, vis)
    assert np.allclose(m.max_radii.cpu().numpy(), np.where(vis, fw["radii"][0] / 160.0, 0))
    # all six leaves through exp / sigmoid / the split SH hand-over, against the oracle's gradients of the ACTIVATED
    # quantities (chain rule: d exp(x) = exp(x), d sigmoid(x) = o (1 - o)), on check_backward's bar
    for name in m.param_names:
        assert getattr(m, name).grad is not None
    o = op.astype(np.float64)
    grads = [m.means.grad, m.quats.grad, m.log_scales.grad / m.scales.detach(), m.logit_opacities.grad / torch.from_numpy(o * (1 - o)).to(d).float(),
             torch.cat([m.sh_0.grad, m.sh_rest.grad], dim=1)]
    hip = dict(grads=grads, meta={"means2d": out["batch_xys"]}, vc=vcl[None], va=np.zeros((1, 96, 160, 1)))
    check_backward(hip, fw)


@pytest.mark.parametrize("deg,K", [(3, 16), (1, 16), (2, 9), (0, 4), (0, 1)])
def test_split_sh_parameters_equal_concatenated(deg, K):
    """colors=(sh_0, sh_rest) (the reference model's parameter layout) == colors=cat(sh_0, sh_rest)."""
    from easy_gaussian_splatting_amd.rendering import rasterization
    sc = make_scene(3001, 150, 90, sh_degree=deg, seed=17, k_store=K, scale_range=(0.02, 0.2), dist=4.0)
    t = to_dev(sc)
    base = [t[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities")]
    shs = t["shs"].clone().requires_grad_(True)
    sh0 = t["shs"][:, :1].clone().contiguous().requires_grad_(True)
    shr = t["shs"][:, 1:].clone().contiguous().requires_grad_(True)
    kw = dict(sh_degree=deg, packed=False, backgrounds=t["backgrounds"], absgrad=True)
    img_a, al_a, _ = rasterization(*base, shs, t["viewmats"], t["Ks"], 150, 90, **kw)
    img_b, al_b, _ = rasterization(*base, (sh0, shr), t["viewmats"], t["Ks"], 150, 90, **kw)
    assert torch.equal(img_a, img_b) and torch.equal(al_a, al_b)
    vc = torch.randn(img_a.shape, generator=torch.Generator().manual_seed(0)).to(dev())
    ga = torch.autograd.grad((img_a * vc).sum(), base + [shs])
    gb = torch.autograd.grad((img_b * vc).sum(), base + [sh0, shr], allow_unused=(K == 1))
    for x, y in zip(ga[:4], gb[:4]):
        assert torch.equal(x, y)
    assert torch.equal(ga[4][:, :1], gb[4])
    if K > 1:
        assert torch.equal(ga[4][:, 1:], gb[5])


def test_concurrent_host_threads_on_separate_streams():
    """Boundary contract: entry points are callable from any Python thread (the
    reference viewer renders from per-client threads, /root/reference/viewer/viewer.py:23-27)."""
    import threading
    from easy_gaussian_splatting_amd.rendering import rasterization
    scenes = [make_scene(4000 + 500 * i, 200, 120, sh_degree=3, seed=40 + i, scale_range=(0.02, 0.2), dist=4.0) for i in range(4)]
    tens = [to_dev(s) for s in scenes]

    def render(i, out):
        t = tens[i]
        with torch.cuda.stream(torch.cuda.Stream()):
            for _ in range(5):
                img, _, _ = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["shs"], t["viewmats"], t["Ks"],
                                          200, 120, sh_degree=3, packed=False, backgrounds=t["backgrounds"])
            torch.cuda.current_stream().synchronize()
        out[i] = img

    ref = {}
    for i in range(4):
        render(i, ref)
    got = {}
    threads = [threading.Thread(target=render, args=(i, got)) for i in range(4)]
    [th.start() for th in threads]
    [th.join() for th in threads]
    for i in range(4):
        assert torch.equal(ref[i], got[i])


@pytest.mark.timeout(120)
def test_degenerate_inputs_neither_hang_nor_poison():
    """Zero / NaN quaternions, NaN and infinite means, zero and huge scales, opacities 0 / 1 / >1,
    Gaussians on the near plane and behind the camera: the path must finish, keep every finite
    Gaussian's contribution finite, and hand back finite gradients for the finite inputs."""
    from easy_gaussian_splatting_amd.rendering import rasterization
    sc = make_scene(512, 96, 64, sh_degree=1, seed=77, k_store=4, scale_range=(0.02, 0.3), dist=4.0)
    t = to_dev(sc)
    means, quats, scales, opac, shs = (t[k].clone() for k in ("means", "quats", "scales", "opacities", "shs"))
    quats[0] = 0.0
    quats[1] = float("nan")
    means[2] = float("nan")
    means[3, 2] = float("inf")
    scales[4] = 0.0
    scales[5] = 1e6
    scales[6] = torch.tensor([1e-12, 1e3, 1e-12], device=dev())
    opac[7] = 0.0
    opac[8] = 1.0
    opac[9] = 5.0
    means[10] = torch.tensor([0.0, 0.0, -4.0 + 0.01], device=dev())   # on the near plane of the z=+4 camera
    means[11] = torch.tensor([0.0, 0.0, -10.0], device=dev())        # behind the camera
    shs[12] = 1e6
    ins = [x.requires_grad_(True) for x in (means, quats, scales, opac, shs)]
    img, alpha, meta = rasterization(*ins, t["viewmats"], t["Ks"], 96, 64, sh_degree=1, packed=False,
                                     backgrounds=t["backgrounds"], absgrad=True)
    (img.sum() + alpha.sum()).backward()
    torch.cuda.synchronize()
    radii = meta["radii"][0]
    assert int(radii[0]) == 0 and int(radii[1]) == 0 and int(radii[2]) == 0 and int(radii[11]) == 0
    assert bool(torch.isfinite(alpha).all())
    good = torch.ones(512, dtype=torch.bool, device=dev())
    good[:13] = False
    for p in ins:
        assert bool(torch.isfinite(p.grad[good]).all())
    assert bool(torch.isfinite(meta["means2d"].absgrad[0][good]).all())


def fuzz_case(case, attempt=0):
    """The sweep's configuration `case` (scene seed number `attempt`): scene dict + (deg, W, H, use_bg, split, culling)."""
    rng = np.random.default_rng(1000 + case)
    big = int(os.environ.get("GS_FUZZ_SCALE", "1"))   # ad-hoc hunting: GS_FUZZ_SCALE=4 draws up to 48 k Gaussians at ~1000 x 800
    deg = int(rng.integers(0, 4))
    K = int(rng.choice([(deg + 1) ** 2, 16]))
    C = int(rng.integers(1, 4))
    n = int(rng.integers(1, 3000 * big * big))
    W, H = int(rng.integers(17, 260 * big)), int(rng.integers(17, 200 * big))
    smax = float(rng.choice([0.05, 0.2, 0.8]))
    dist, white = float(rng.uniform(2.5, 6.0)), bool(rng.integers(0, 2))
    use_bg, split, culling = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)) and K > 1, str(rng.choice(["tight", "gsplat"]))   # (drawn before "gsplat_eager" existed: see below)
    sc = make_scene(n, W, H, sh_degree=deg, seed=2000 + case + 1000 * attempt, k_store=K, n_views=C, scale_range=(0.01, smax),
                    dist=dist, white_bg=white)
    if culling == "gsplat" and case % 2:   # half of the reference-list cases walk gsplat's own lists in the render pipeline
        culling = "gsplat_eager"
    return sc, (deg, W, H, use_bg, split, culling)


# GS_FUZZ_CASES=N widens the sweep (tools / ad-hoc hunting; the committed suite runs 24 cases)
@pytest.mark.parametrize("case", range(int(os.environ.get("GS_FUZZ_CASES", "24"))))
def test_randomised_configurations(case, monkeypatch):
    """Seeded sweep over sizes / SH degree / stored K / cameras / background / splat scale / culling
    mode / SH layout: every combination must meet the same forward and gradient tolerances.  A drawn scene in which
    more than 5 % of the pixels sit on a blend threshold (a flat, faint splat covering the image at alpha ~ 1/255) is
    useless as a parity case: the next scene seed of the same configuration is drawn instead (at most 8, printed)."""
    from easy_gaussian_splatting_amd.rendering import rasterization
    # the binning pipeline alternates with the case: per-tile sort / two-level with 2x2- and 4x4-tile bins
    monkeypatch.setenv("GS_BINNING", ("tiles", "bins", "bins")[case % 3])
    monkeypatch.setenv("GS_BINS_SHIFT", ("", "1", "2")[case % 3])
    for attempt in range(8):
        sc, (deg, W, H, use_bg, split, culling) = fuzz_case(case, attempt)
        t = to_dev(sc)
        base = [t[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities")]
        if split:
            sh = (t["shs"][:, :1].clone().contiguous().requires_grad_(True), t["shs"][:, 1:].clone().contiguous().requires_grad_(True))
            leaves = base + list(sh)
        else:
            sh = t["shs"].clone().requires_grad_(True)
            leaves = base + [sh]
        img, alpha, meta = rasterization(*base, sh, t["viewmats"], t["Ks"], W, H, sh_degree=deg, packed=False,
                                         backgrounds=t["backgrounds"] if use_bg else None, absgrad=True, _tile_culling=culling)
        fw = run_oracle(sc, use_bg=use_bg)
        hip = dict(img=img, alpha=alpha, meta=meta)
        rep = forward_report(meta, fw, lists=culling != "tight")
        rep["fw"], hip["report"] = fw, rep
        razor = float(rep["razor"].mean())
        if razor <= MAX_RAZOR_FRAC:
            break
        parity_log.record(n_redrawn_seeds=1)
        print(f"[parity] case {case}: scene seed {2000 + case + 1000 * attempt} is a razor-edge scene ({razor:.3f}); drawing the next")
    g = torch.Generator().manual_seed(case)
    keep = torch.from_numpy(~rep["loose"])[..., None]   # upstream gradients only where the contributor sets are decided
    vc, va = torch.randn(img.shape, generator=g) * keep, torch.randn(alpha.shape, generator=g) * keep
    grads = torch.autograd.grad((img * vc.to(dev())).sum() + (alpha * va.to(dev())).sum(), leaves)
    if split:
        grads = list(grads[:4]) + [torch.cat([grads[4], grads[5]], dim=1)]
    hip.update(grads=grads, vc=vc.numpy().astype(np.float64), va=va.numpy().astype(np.float64))
    check_forward(hip, fw, max_razor_frac=MAX_RAZOR_FRAC, lists=culling != "tight")
    if fw["n_isects"] > 0:
        check_backward(hip, fw)


def test_sweep_redraw_budget():
    """re-drawn sweep seeds are counted, not silently skipped.  Whether a drawn scene is a razor-edge scene is
    a property of the oracle run alone (razor_mask), so the count is the same on every box: at most 1 % of the cases
    (at least one allowed)."""
    recs = [r for k, r in parity_log.RECORDS.items() if "test_randomised_configurations" in k]
    if not recs:
        pytest.skip("the sweep did not run in this session")
    redrawn = sum(int(r.get("n_redrawn_seeds", 0)) for r in recs)
    parity_log.record(n_sweep_cases=len(recs), sweep_redrawn_seeds=redrawn)
    assert redrawn <= max(1, math.ceil(0.01 * len(recs))), f"{redrawn} of {len(recs)} sweep cases were re-drawn"


def _binning_scenes():
    big = make_scene(2500, 320, 208, sh_degree=1, seed=61, k_store=4, n_views=2, scale_range=(0.05, 0.6), dist=5.0)   # footprints of 20+ tiles
    return {"long_lists": config_long_lists(), "dense_3000": dense_scene(3000, 13), "two_views_big": big,
            "tiny": make_scene(50, 40, 24, sh_degree=3, seed=1, scale_range=(0.05, 0.4), dist=4.0),
            "ragged": make_scene(1237, 333, 77, sh_degree=1, seed=3, k_store=4, scale_range=(0.02, 0.3), dist=4.0)}


@pytest.mark.parametrize("culling", ["gsplat", "gsplat_eager", "tight"])
@pytest.mark.parametrize("name", ["long_lists", "dense_3000", "two_views_big", "tiny", "ragged"])
def test_binning_pipelines_agree_bit_for_bit(name, culling, monkeypatch):
    """The per-tile pipeline (every tile list emitted and sorted) and the two-level one (coarse bins sorted, tiles refined
    out of them; 2x2- and 4x4-tile bins) must produce the same lists, offsets, gradient-row slots, image and gradients --
    bit for bit -- including the first call on a device, whose coarse buffer is too small and is re-run."""
    from easy_gaussian_splatting_amd import rendering
    sc = _binning_scenes()[name]
    runs = {}
    for mode, shift in (("tiles", ""), ("bins", "1"), ("bins", "2")):
        monkeypatch.setenv("GS_BINNING", mode)
        monkeypatch.setenv("GS_BINS_SHIFT", shift)
        rendering.reset_hints()
        retries = rendering.stats["coarse_retries"]
        dbg = {}
        hip = run_hip(sc, culling=culling, dbg=dbg)
        if mode == "bins" and name == "long_lists":
            assert rendering.stats["coarse_retries"] > retries, "the first call must have outgrown the default coarse buffer"
            hip2 = run_hip(sc, culling=culling)   # second call: sized from the hint, no retry
            assert torch.equal(hip2["meta"]["flatten_ids"], hip["meta"]["flatten_ids"]) and torch.equal(hip2["img"], hip["img"])
        runs[(mode, shift)] = hip
    ref = runs[("tiles", "")]
    for key, hip in runs.items():
        for k in ("flatten_ids", "isect_ids", "isect_offsets", "tiles_per_gauss", "radii"):
            assert torch.equal(hip["meta"][k], ref["meta"][k]), (key, k)
        assert torch.equal(hip["img"], ref["img"]) and torch.equal(hip["alpha"], ref["alpha"]), key
        for a, b in zip(hip["grads"], ref["grads"]):
            assert torch.equal(a, b), key
        assert torch.equal(hip["meta"]["means2d"].absgrad, ref["meta"]["means2d"].absgrad)


def test_image_beyond_the_per_tile_histogram_takes_the_two_level_binning(monkeypatch):
    """5120 x 2880 = 57 600 tiles: more than the per-tile pipeline's LDS histogram holds (40 928); rasterization() must
    take the two-level binning by itself and agree with the oracle; the per-tile entry point must refuse loudly."""
    from easy_gaussian_splatting_amd import rendering
    monkeypatch.delenv("GS_BINNING", raising=False)   # the automatic choice is what is under test
    sc = make_scene(3000, 5120, 2880, sh_degree=1, seed=91, k_store=4, scale_range=(0.01, 0.2), dist=5.0)
    fw = run_oracle(sc)
    hip = run_hip(sc, fw=fw)
    assert rendering.last_binning(dev()) == "bins"
    check_forward(hip, fw)
    check_backward(hip, fw)
    monkeypatch.setenv("GS_BINNING", "tiles")
    with pytest.raises(ValueError, match="tile grid too large"):
        run_hip(sc, bwd=False)


def test_two_level_binning_flags_and_capacities():
    """C-ABI contract of gs_bins_count: a coarse key buffer or a sort-class bound that is too small raises flags 4 / 8 in
    info[3], reports the sizes needed in info[4] / info[5], and emits nothing; the repeated call with those sizes succeeds
    and agrees with the per-tile count."""
    import ctypes as ct
    from easy_gaussian_splatting_amd import _native as nat
    d = dev()
    sc = config_long_lists()
    t = to_dev(sc)
    L = nat.lib()
    C, N = 1, sc["means"].shape[0]
    W, H = int(sc["width"]), int(sc["height"])
    tw, th = (W + 15) // 16, (H + 15) // 16
    tiles = tw * th
    i32 = dict(dtype=torch.int32, device=d); f32 = dict(dtype=torch.float32, device=d)
    radii = torch.empty((C, N), **i32); m2 = torch.empty((C, N, 2), **f32); dep = torch.empty((C, N), **f32)
    con = torch.empty((C, N, 3), **f32); col = torch.empty((C, N, 3), **f32); rec = torch.empty((C * N, 12), **f32)
    bbox = torch.empty((C * N, 4), **i32); tpg = torch.empty((C, N), **i32)
    st = torch.cuda.current_stream().cuda_stream
    P = lambda x: x.data_ptr()
    nat.check(L.gs_project_fwd(st, C, N, 16, 3, P(t["means"]), P(t["quats"]), P(t["scales"]), P(t["opacities"]), P(t["shs"]), None, 0,
                               P(t["viewmats"]), P(t["Ks"]), W, H, 0.3, 0.01, 1e10, 0.0, 0, 1, 0, P(radii), P(m2), P(dep),
                               P(con), P(col), P(rec), P(bbox), P(tpg), None, None), "gs_project_fwd")
    off = torch.empty((tiles + 1,), **i32); boff = torch.empty((tiles + 1,), **i32); order = torch.empty((tiles,), **i32)
    cum = torch.empty((C * N,), **i32)
    info = torch.zeros((8,), dtype=torch.int64, device=d)
    host = (ct.c_int64 * 8)()

    def count(cap, list_cap, shift=2):
        keys = torch.empty((max(cap, 1),), dtype=torch.int64, device=d)
        ws = torch.empty((int(L.gs_bins_workspace_bytes(C, N, tw, th, shift, cap)),), dtype=torch.uint8, device=d)
        nat.check(L.gs_bins_count(st, C, N, tw, th, shift, P(bbox), P(dep), P(ws), ws.numel(), P(keys), cap, list_cap, P(cum), P(off),
                                  P(boff), P(order), P(info), host), "gs_bins_count")
        return [int(v) for v in host]

    h = count(1000, 0)
    assert h[3] & 4 and h[4] > 1000, h
    need, longest = h[4], h[5]
    assert longest > 1024
    h = count(need, 1024)
    assert h[3] == 8 and h[5] == longest, h
    h = count(need, longest)
    assert h[3] == 0 and h[4] == need and h[0] == int(tpg.sum()), h
    assert torch.equal(torch.diff(off.long()), torch.bincount(_tile_of_entries(bbox, tw), minlength=tiles))
    # the list pass with an explicit isect_ids buffer (rasterization() passes NULL and derives the keys on demand): the
    # kernel-written keys must equal the derived ones
    from easy_gaussian_splatting_amd.rendering import _LazyMeta
    I = h[0]
    keys = torch.empty((need,), dtype=torch.int64, device=d)
    ws = torch.empty((int(L.gs_bins_workspace_bytes(C, N, tw, th, 2, need)),), dtype=torch.uint8, device=d)
    nat.check(L.gs_bins_count(st, C, N, tw, th, 2, P(bbox), P(dep), P(ws), ws.numel(), P(keys), need, 0, P(cum), P(off), P(boff), P(order),
                              P(info), host), "gs_bins_count")
    ids = torch.empty((I,), dtype=torch.int64, device=d); fid = torch.empty((I,), **i32); slots = torch.empty((I,), **i32)
    nat.check(L.gs_bins_lists(st, C, N, tw, th, 2, P(bbox), P(ws), ws.numel(), P(keys), need, P(cum), P(off), P(ids), P(fid), P(slots), P(info)),
              "gs_bins_lists")
    lazy = _LazyMeta({"n_cameras": C, "tile_width": tw, "tile_height": th, "flatten_ids": fid, "isect_offsets": off[:tiles].view(C, th, tw),
                      "depths": dep, "isect_ids": _LazyMeta.PENDING})
    assert torch.equal(ids, lazy["isect_ids"])
    assert bool((ids[1:] >= ids[:-1]).all()) and torch.equal(torch.sort(slots.long()).values, torch.arange(I, device=d))


def _tile_of_entries(bbox, tw):
    """Tile index of every (Gaussian, tile) pair of the footprints in `bbox` (rectangles; masks for <= 32 tiles)."""
    b = bbox.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    out = []
    for x, y, mask, cnt in b:
        if cnt == 0:
            continue
        x0, x1, y0, y1 = x & 0xFFFF, x >> 16, y & 0xFFFF, y >> 16
        w = x1 - x0
        for i in range(w * (y1 - y0)):
            if w * (y1 - y0) > 32 or (mask >> i) & 1:
                out.append((y0 + i // w) * tw + x0 + i % w)
    return torch.tensor(out, dtype=torch.int64, device=bbox.device)


@pytest.mark.parametrize("binning", ["tiles", "bins"])
@pytest.mark.parametrize("n,lo,hi", [(250, 64, 1024), (1500, 1024, 4096), (5000, 4096, 8192), (10000, 8192, 16384),
                                     (30000, 16384, 65536), (70000, 65536, 1 << 20)])
def test_every_sort_size_class_orders_like_a_stable_global_sort(n, lo, hi, binning, monkeypatch):
    """One scene per size class of the list sort (LDS radix <= 1024 / 4096 / 8192, 8192-key segments + rank merge <= 65536,
    global bitonic network beyond), per-tile and coarse-bin lists alike: inside every tile the ids must be ordered by
    (depth bits, flatten id), every id exactly once."""
    monkeypatch.setenv("GS_BINNING", binning)
    sc = dense_scene(n, 100 + n)
    hip = run_hip(sc, bwd=False)
    meta = hip["meta"]
    off = meta["isect_offsets"].reshape(-1).cpu().numpy().astype(np.int64)
    fid = meta["flatten_ids"].cpu().numpy().astype(np.int64)
    ids = meta["isect_ids"].cpu().numpy()
    counts = np.diff(np.append(off, fid.size))
    assert lo < counts.max() <= hi, counts.max()
    depth_bits = meta["depths"].reshape(-1).cpu().numpy().view(np.int32).astype(np.int64)
    assert fid.size == int(meta["tiles_per_gauss"].sum())
    for t in range(off.size):
        seg = fid[off[t]: off[t] + counts[t]]
        key = depth_bits[seg] * (1 << 32) + seg
        assert np.all(np.diff(key) > 0), f"tile {t} not in (depth, id) order"
        assert np.array_equal(ids[off[t]: off[t] + counts[t]] & 0xFFFFFFFF, depth_bits[seg])


def test_saturated_tiles_stop_early_and_leave_clean_masks():
    """Opaque, stacked splats: every pixel of most tiles reaches T <= 1e-4 long before the end of its list,
    so the forward abandons the rest of the list (whose quadrant masks must still read 'no rows') and the
    backward must not pick up anything from the abandoned part.  Run twice over dirty allocator memory."""
    n = 6000
    sc = make_scene(n, 96, 80, sh_degree=1, seed=41, scale_range=(0.25, 0.6), dist=4.0, extent=(0.4, 0.4, 0.6))
    sc["opacities"] = np.full(n, 0.97, dtype=np.float32)
    fw = run_oracle(sc)
    assert float((fw["render_alphas"] > 1 - 2e-4).mean()) > 0.5, "scene must saturate"
    counts = np.diff(np.append(fw["isect_offsets"].reshape(-1), fw["n_isects"]))
    assert counts.max() > 1024
    for _ in range(2):
        junk = torch.full((64 << 20,), 0x7f, dtype=torch.uint8, device="cuda:0")   # poison the caching allocator's pool
        del junk
        hip = run_hip(sc, fw=fw)
        assert check_forward(hip, fw, max_razor_frac=MAX_RAZOR_FRAC)
        check_backward(hip, fw)


def test_in_kernel_activations_equal_torch_activations():
    """_activations='exp_sigmoid': log-scales / logit opacities in, exp / sigmoid inside the projection kernels,
    gradients w.r.t. the raw parameters out -- must equal torch.exp / torch.sigmoid in front of the plain call."""
    from easy_gaussian_splatting_amd.rendering import rasterization
    sc = make_scene(4000, 208, 144, sh_degree=2, n_views=2, seed=55, scale_range=(0.02, 0.2), dist=4.0)
    t = to_dev(sc)
    op = t["opacities"].clamp(1e-3, 1 - 1e-3)
    raw_s, raw_o = torch.log(t["scales"]), torch.log(op / (1 - op))
    vc = torch.randn((2, 144, 208, 3), generator=torch.Generator().manual_seed(2)).to(raw_s.device)

    def run(fused):
        ls, lo = raw_s.clone().requires_grad_(True), raw_o.clone().requires_grad_(True)
        others = [t[k].clone().requires_grad_(True) for k in ("means", "quats", "shs")]
        s_in, o_in = (ls, lo) if fused else (torch.exp(ls), torch.sigmoid(lo))
        img, alpha, meta = rasterization(others[0], others[1], s_in, o_in, others[2], t["viewmats"], t["Ks"], 208, 144,
                                         sh_degree=2, packed=False, backgrounds=t["backgrounds"], absgrad=True,
                                         _activations="exp_sigmoid" if fused else "none")
        (img * vc).sum().backward()
        return img.detach(), meta, [ls.grad, lo.grad] + [p.grad for p in others]

    img_a, meta_a, g_a = run(True)
    img_b, meta_b, g_b = run(False)
    assert torch.equal(meta_a["radii"], meta_b["radii"])
    assert float((img_a - img_b).abs().max()) < 2e-6
    for name, a, b in zip(("log_scales", "logit_opacities", "means", "quats", "shs"), g_a, g_b):
        rel = float((a - b).abs().max() / (b.abs().max() + 1e-30))
        assert rel < 2e-5, (name, rel)
    assert float((meta_a["means2d"].absgrad - meta_b["means2d"].absgrad).abs().max()) <= 2e-5 * float(meta_b["means2d"].absgrad.abs().max())


@pytest.mark.skipif((os.cpu_count() or 1) < 32, reason="the C oracle needs many host cores to finish the 1M / 1080p workload in seconds")
def test_full_size_matches_oracle():
    """BASELINE.json metric workload (1 M Gaussians, 1920x1080, SH3) against the oracle itself.  With 400-700
    contributors per pixel fp32 and fp64 arithmetic take different threshold decisions in a few hundred pixels
    (fp32 oracle vs fp64 oracle: ~200 pixels above 1e-4), so: against the fp64 oracle at most 1e-5 of the
    non-razor pixels may exceed 1e-4 (observed: 3 of 2.06 M, mean error 1.3e-6); against the fp32 build of the
    same oracle every non-razor pixel must be within 1e-4 (observed max 8e-7); lists bit-exact; all gradients
    within 1e-3 rel (fp32 oracle arbitrating flips, as in every backward check)."""
    sc = config_bench_1m()
    fw = run_oracle(sc)
    fw32 = run_oracle(sc, dtype=np.float32)
    hip = run_hip(sc, fw=fw)
    check_forward(hip, fw, outlier_frac=1e-5)
    # (the fp32 oracle's own geometry is tens of ulps off the fp64 truth: its storage bounds are those of an fp32 chain;
    #  lists are compared bit for bit whenever every radius agrees)
    check_forward(hip, fw32, geom_slack=1e3)
    check_backward(hip, fw)
    hip_t = run_hip(sc, culling="tight", upstream=(hip["vc"], hip["va"]))      # the default list mode: same image and gradients, shorter lists
    assert hip_t["meta"]["flatten_ids"].numel() < hip["meta"]["flatten_ids"].numel()
    check_forward(hip_t, fw32, lists=False, geom_slack=1e3)
    for a, b in zip(hip_t["grads"], hip["grads"]):   # (tight == gsplat-mode gradients; the oracle check ran on the latter)
        assert float((a - b).abs().max()) <= 1e-4 * float(b.abs().max())
    del hip, hip_t
    check_backward_unmasked(sc, fw, "tight")   # gradient flow through the razor pixels too, against the fp32 oracle


def test_tile_launch_order_is_a_longest_first_permutation():
    """gs_bin_count's tile_order: a permutation of the tiles in which quarter-octave length classes never
    increase (longest lists first); the order inside a class is arbitrary."""
    import ctypes as ct
    from easy_gaussian_splatting_amd import _native as nat
    d = dev()
    sc = make_scene(20000, 640, 368, sh_degree=0, seed=77, scale_range=(0.01, 0.3), dist=4.0)
    t = to_dev(sc)
    L = nat.lib()
    C, N = 1, 20000
    tw, th = (sc["width"] + 15) // 16, (sc["height"] + 15) // 16
    tiles = tw * th
    i32 = dict(dtype=torch.int32, device=d); f32 = dict(dtype=torch.float32, device=d)
    radii = torch.empty((C, N), **i32); m2 = torch.empty((C, N, 2), **f32); dep = torch.empty((C, N), **f32)
    con = torch.empty((C, N, 3), **f32); col = torch.empty((C, N, 3), **f32); rec = torch.empty((C * N, 12), **f32)
    bbox = torch.empty((C * N, 4), **i32); tpg = torch.empty((C, N), **i32)
    st = torch.cuda.current_stream().cuda_stream
    P = lambda x: x.data_ptr()
    nat.check(L.gs_project_fwd(st, C, N, 1, 0, P(t["means"]), P(t["quats"]), P(t["scales"]), P(t["opacities"]), P(t["shs"]), None, 0,
                               P(t["viewmats"]), P(t["Ks"]), sc["width"], sc["height"], 0.3, 0.01, 1e10, 0.0, 1, 0, 0, P(radii), P(m2), P(dep),
                               P(con), P(col), P(rec), P(bbox), P(tpg), None, None), "gs_project_fwd")
    ws_bytes = int(L.gs_bin_workspace_bytes(C, N, tw, th))
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=d)
    off = torch.empty((tiles + 1,), **i32); boff = torch.empty((tiles + 1,), **i32); order = torch.full((tiles,), -1, **i32)
    info = torch.empty((4,), dtype=torch.int64, device=d)
    host = (ct.c_int64 * 4)()
    nat.check(L.gs_bin_count(st, C, N, tw, th, P(bbox), P(ws), ws_bytes, P(off), P(boff), P(order), P(info), host), "gs_bin_count")
    off, order = off.cpu().numpy().astype(np.int64), order.cpu().numpy()
    lens = np.diff(off)
    assert sorted(order.tolist()) == list(range(tiles))

    def cls(x):
        x = int(x)
        if x < 2:
            return 0
        lg = x.bit_length() - 1
        return min(63, (lg << 2) | ((x << (31 - lg)) >> 29 & 3))

    seq = np.array([cls(lens[i]) for i in order])
    assert np.all(np.diff(seq) <= 0), "length classes must not increase along the launch order"
    assert seq[0] == max(cls(v) for v in lens) and host[2] == lens.max() and host[0] == lens.sum()


def test_no_grad_takes_the_inference_path():
    """Under torch.no_grad() parameters that require grad must not make the forward build the backward's
    lists and checkpoints (same image, no training buffers kept)."""
    from easy_gaussian_splatting_amd.rendering import rasterization
    sc = make_scene(3000, 160, 112, sh_degree=2, seed=9, scale_range=(0.02, 0.2), dist=4.0)
    t = to_dev(sc)
    ins = [t[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "shs")]
    img_g, _, _ = rasterization(*ins, t["viewmats"], t["Ks"], 160, 112, sh_degree=2, packed=False, backgrounds=t["backgrounds"])
    assert img_g.requires_grad
    with torch.no_grad():
        img_n, _, _ = rasterization(*ins, t["viewmats"], t["Ks"], 160, 112, sh_degree=2, packed=False, backgrounds=t["backgrounds"])
    assert not img_n.requires_grad and torch.equal(img_n, img_g.detach())
    # the inference call's workspace layout holds none of the backward's buffers (checkpoints, quadrant sublists, rows)
    from easy_gaussian_splatting_amd import workspace as WS
    with torch.no_grad():
        out = rasterization(*ins, t["viewmats"], t["Ks"], 160, 112, sh_degree=2, packed=False, backgrounds=t["backgrounds"], _tile_culling="tight")
    lay_inf = out[2]._lease.lease.layout
    out_t = rasterization(*ins, t["viewmats"], t["Ks"], 160, 112, sh_degree=2, packed=False, backgrounds=t["backgrounds"], _tile_culling="tight")
    lay_train = out_t[0].grad_fn.state["lease"].layout
    for slot in (WS.CKPT, WS.QLIST, WS.QMASK, WS.ROW_BASE, WS.UNIT_DESC, WS.ROWS, WS.SLOTS):
        assert lay_inf.offsets[slot] == -1 and lay_train.offsets[slot] >= 0, slot
    assert lay_inf.arena_bytes[2] == 256 and lay_inf.arena_bytes[1] < lay_train.arena_bytes[1]
