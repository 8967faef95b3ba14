import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, torch
from scenes import make_scene
from oracle import c_oracle as CO
from easy_gaussian_splatting_amd.rendering import rasterization
case = int(sys.argv[1]) if len(sys.argv) > 1 else 7
rng = np.random.default_rng(1000 + case)
deg = int(rng.integers(0, 4)); K = int(rng.choice([(deg + 1) ** 2, 16])); C = int(rng.integers(1, 4)); n = int(rng.integers(1, 3000))
W, H = int(rng.integers(17, 260)), int(rng.integers(17, 200)); smax = float(rng.choice([0.05, 0.2, 0.8]))
sc = make_scene(n, W, H, sh_degree=deg, seed=2000 + case, k_store=K, n_views=C, scale_range=(0.01, smax), dist=float(rng.uniform(2.5, 6.0)), white_bg=bool(rng.integers(0, 2)))
use_bg, split, culling = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)) and K > 1, str(rng.choice(["tight", "gsplat"]))
print(dict(deg=deg, K=K, C=C, n=n, W=W, H=H, smax=smax, use_bg=use_bg, split=split, culling=culling))
dev = torch.device("cuda:0")
t = {k: torch.from_numpy(v).to(dev) for k, v in sc.items() if isinstance(v, np.ndarray)}
base = [t[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "shs")]
dbg = {}
img, alpha, meta = rasterization(*base, t["viewmats"], t["Ks"], W, H, sh_degree=deg, packed=False, backgrounds=t["backgrounds"] if use_bg else None, absgrad=True, _tile_culling=culling, _debug=dbg)
g = torch.Generator().manual_seed(case)
vc, va = torch.randn(img.shape, generator=g), torch.randn(alpha.shape, generator=g)
grads = torch.autograd.grad((img * vc.to(dev)).sum() + (alpha * va.to(dev)).sum(), base)
for dt in (np.float64, np.float32):
    fw = CO.render(sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["shs"], sc["viewmats"], sc["Ks"], W, H, sh_degree=deg, backgrounds=sc["backgrounds"] if use_bg else None, dtype=dt)
    bw = CO.backward(fw, vc.numpy().astype(dt), va.numpy().astype(dt))
    print("oracle", dt.__name__, "I", fw["n_isects"])
    for nm, gg in zip(["v_means", "v_quats", "v_scales", "v_opacities", "v_colors"], grads):
        e = np.abs(gg.cpu().numpy() - bw[nm]); i = np.unravel_index(e.argmax(), e.shape)
        print("  ", nm, "rel", e.max() / np.abs(bw[nm]).max(), "at", i, "hip", gg.cpu().numpy()[i], "ref", bw[nm][i], "max", np.abs(bw[nm]).max())
    for nm, key in (("v_means2d", "v_means2d"), ("v_conics", "v_conics"), ("v_colors_post", "v_colors_post")):
        e = np.abs(dbg[nm].cpu().numpy() - bw[key]); i = np.unravel_index(e.argmax(), e.shape)
        print("  [2D]", nm, "rel", e.max() / np.abs(bw[key]).max(), "at", i, dbg[nm].cpu().numpy()[i], bw[key][i])
    if dt == np.float64:
        gi = int(np.unravel_index(np.abs(grads[0].cpu().numpy() - bw["v_means"]).argmax(), (n, 3))[0])
        print("   worst gaussian", gi, "scales", sc["scales"][gi], "mean", sc["means"][gi], "radii", fw["radii"][:, gi], "depth", fw["depths"][:, gi], "means2d", fw["means2d"][:, gi], "conic", fw["conics"][:, gi], "opac", sc["opacities"][gi])
