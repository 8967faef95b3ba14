import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import numpy as np, torch
from easy_gaussian_splatting_amd.synthetic import config_bench_1m
from easy_gaussian_splatting_amd.rendering import rasterization
dev = torch.device("cuda:0")
sc = config_bench_1m()
t = {k: torch.from_numpy(v).to(dev) for k, v in sc.items() if isinstance(v, np.ndarray)}
with torch.no_grad():
    img, a, meta = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["shs"], t["viewmats"][:1], t["Ks"][:1], sc["width"], sc["height"], sh_degree=3, packed=False, backgrounds=t["backgrounds"][:1])
off = meta["isect_offsets"].reshape(-1).cpu().numpy().astype(np.int64)
I = meta["flatten_ids"].numel()
cnt = np.diff(np.append(off, I))
print("tiles", cnt.size, "mean", cnt.mean(), "max", cnt.max(), "p50/p90/p99", np.percentile(cnt, [50, 90, 99]))
alpha = a[0, :, :, 0].cpu().numpy()
print("alpha mean", alpha.mean(), "frac saturated (T<1e-4)", (alpha > 1 - 1e-4).mean())
