import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import numpy as np, torch
from oracle import c_oracle as CO
from scenes import config_bench_1m
from easy_gaussian_splatting_amd.rendering import rasterization
sc = config_bench_1m()
dev = torch.device("cuda:0")
t = {k: torch.from_numpy(v).to(dev) for k, v in sc.items() if isinstance(v, np.ndarray)}
img, alpha, meta = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["shs"], t["viewmats"], t["Ks"], sc["width"], sc["height"],
                                 sh_degree=3, packed=False, backgrounds=t["backgrounds"], _tile_culling="gsplat")
img = img.cpu().numpy()
for dt in (np.float64, np.float32):
    fw = CO.render(sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["shs"], sc["viewmats"], sc["Ks"], sc["width"], sc["height"],
                   sh_degree=3, backgrounds=sc["backgrounds"], dtype=dt)
    err = np.abs(img - fw["render_colors"]).max(-1)
    razor = CO.blend_margin(fw) < 1e-4
    e = err[~razor]
    print(dt.__name__, "razor frac", razor.mean(), "strict: max", e.max(), "p99.99", np.quantile(e, 0.9999), "mean", e.mean(),
          "count>1e-4", int((e > 1e-4).sum()), "count>2e-4", int((e > 2e-4).sum()), "of", e.size, flush=True)
    if dt is np.float64:
        ref64 = fw["render_colors"]
    else:
        print("fp32 oracle vs fp64 oracle: max", np.abs(fw["render_colors"] - ref64).max(), "count>1e-4", int((np.abs(fw["render_colors"] - ref64).max(-1) > 1e-4).sum()))
