#!/usr/bin/env python3
"""Reproduce one configuration of tests/test_gpu_parity.py::test_randomised_configurations outside pytest and print where
the HIP path and the oracle (fp64 and fp32 builds) disagree most: tools/debug_case.py <case> [attempt]."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from oracle import c_oracle as CO
from easy_gaussian_splatting_amd.rendering import rasterization
import test_gpu_parity as TP

case = int(sys.argv[1]) if len(sys.argv) > 1 else 7
attempt = int(sys.argv[2]) if len(sys.argv) > 2 else None
dev = torch.device("cuda:0")
if attempt is None:   # the scene seed the test settles on: the first whose razor fraction is within bounds
    for attempt in range(8):
        sc, (deg, W, H, use_bg, split, culling) = TP.fuzz_case(case, attempt)
        t = TP.to_dev(sc)
        _, _, meta = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["shs"], t["viewmats"], t["Ks"], W, H, sh_degree=deg,
                                   packed=False, _tile_culling=culling)
        fw = TP.run_oracle(sc, use_bg=use_bg)
        if (CO.blend_margin(fw) < 1e-4).mean() <= TP.MAX_RAZOR_FRAC:
            break
    print("attempt", attempt)
sc, (deg, W, H, use_bg, split, culling) = TP.fuzz_case(case, attempt)
n, C = sc["means"].shape[0], sc["viewmats"].shape[0]
print(dict(deg=deg, C=C, n=n, W=W, H=H, use_bg=use_bg, split=split, culling=culling))
t = TP.to_dev(sc)
base = [t[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "shs")]
dbg = {}
img, alpha, meta = rasterization(*base, t["viewmats"], t["Ks"], W, H, sh_degree=deg, packed=False,
                                 backgrounds=t["backgrounds"] if use_bg else None, absgrad=True, _tile_culling=culling, _debug=dbg)
fw = TP.run_oracle(sc, use_bg=use_bg)
m2 = meta["means2d"].detach().cpu().numpy()
e = np.abs(m2 - fw["means2d"]).max(-1); c, i = np.unravel_index(e.argmax(), e.shape)
print(f"means2d: worst abs err {e.max():.3e} at cam {c} gaussian {i}: hip {m2[c, i]} oracle {fw['means2d'][c, i]} depth {fw['depths'][c, i]} radius {fw['radii'][c, i]}")
try:
    rep = TP.forward_report(meta, fw, lists=culling != "tight")
except AssertionError as ex:
    print("forward_report:", str(ex).splitlines()[0]); sys.exit(0)
print("razor", rep["razor"].mean(), "loose", rep["loose"].mean(), "exact lists", rep["exact_lists"])
g = torch.Generator().manual_seed(case)
keep = torch.from_numpy(~rep["loose"])[..., None]
vc, va = torch.randn(img.shape, generator=g) * keep, torch.randn(alpha.shape, generator=g) * keep
grads = torch.autograd.grad((img * vc.to(dev)).sum() + (alpha * va.to(dev)).sum(), base)
names = ["v_means", "v_quats", "v_scales", "v_opacities", "v_colors"]
for dt in (np.float64, np.float32):
    f = fw if dt == np.float64 else TP.run_oracle(sc, use_bg=use_bg, dtype=np.float32)
    bw = CO.backward(f, vc.numpy().astype(dt), va.numpy().astype(dt))
    print("oracle", dt.__name__, "I", f["n_isects"])
    for nm, gg in zip(names, grads):
        er = np.abs(gg.cpu().numpy() - bw[nm]); idx = np.unravel_index(er.argmax(), er.shape)
        print(f"   {nm}: rel {er.max() / np.abs(bw[nm]).max():.3e} at {idx} hip {gg.cpu().numpy()[idx]:.6e} ref {bw[nm][idx]:.6e} max {np.abs(bw[nm]).max():.3e}")
    for nm in ("v_means2d", "v_conics", "v_colors_post"):
        er = np.abs(dbg[nm].cpu().numpy() - bw[nm]); idx = np.unravel_index(er.argmax(), er.shape)
        print(f"   [2D] {nm}: rel {er.max() / np.abs(bw[nm]).max():.3e} at {idx} hip {dbg[nm].cpu().numpy()[idx]:.6e} ref {bw[nm][idx]:.6e}")
    if dt == np.float64:
        bw64 = bw
for nm, gg in zip(names[:3], grads[:3]):
    er = np.abs(gg.cpu().numpy() - bw64[nm]).reshape(n, -1).max(-1); gi = int(er.argmax())
    print(f"worst gaussian for {nm}: {gi} scales {sc['scales'][gi]} quat {sc['quats'][gi]} opac {sc['opacities'][gi]:.4f} radii {fw['radii'][:, gi]} "
          f"depth {fw['depths'][:, gi]} means2d {fw['means2d'][:, gi].tolist()} conic {fw['conics'][:, gi].tolist()}")
gi = int(np.abs(grads[2].cpu().numpy() - bw64["v_scales"]).max(-1).argmax())
for c in range(C):
    for nm in ("v_conics", "v_means2d"):
        h, r = dbg[nm].cpu().numpy()[c, gi], bw64[nm][c, gi]
        print(f"gaussian {gi} cam {c} {nm}: hip {h} ref {r} rel-to-own {np.abs(h - r).max() / (np.abs(r).max() + 1e-30):.2e} tiles {fw['tiles_per_gauss'][c, gi]}")
