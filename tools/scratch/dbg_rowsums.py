import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
from easy_gaussian_splatting_amd.rendering import rasterization
from scenes import make_scene
dev = torch.device("cuda:0")
for C in (1, 3):
    sc = make_scene(2500, 176, 112, sh_degree=3, n_views=C, seed=34, scale_range=(0.03, 0.2), dist=4.0)
    t = {k: torch.from_numpy(v).to(dev) for k, v in sc.items() if isinstance(v, np.ndarray)}
    vc = torch.randn((C, 112, 176, 3), generator=torch.Generator().manual_seed(1)).to(dev)
    def run(mode):
        ins = [t[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities")]
        shs = t["shs"].clone().requires_grad_(True)
        dbg = {}
        img, _, meta = rasterization(*ins, shs, t["viewmats"], t["Ks"], 176, 112, sh_degree=3, packed=False,
                                     backgrounds=t["backgrounds"], absgrad=True, _sh_grads=mode, _debug=dbg)
        (img * vc).sum().backward()
        return [p.grad for p in ins], dbg, meta
    ga, da, ma = run("dense")
    gb, db, mb = run("colors_pre")
    for k in ("v_means2d", "v_conics", "v_colors_post"):
        print(C, k, torch.equal(da[k], db[k]), float((da[k] - db[k]).abs().max()))
    print(C, "absgrad", torch.equal(ma["means2d"].absgrad, mb["means2d"].absgrad))
    for n, a, b in zip(("means", "quats", "scales", "opac"), ga, gb):
        d = (a - b).abs()
        print(C, n, torch.equal(a, b), float(d.max()), float(d.max() / a.abs().max()), int((d > 0).sum()), a.numel())
