"""Per-stage device times of forward + backward at the bench workload WITHOUT an optimizer step (GPU box).

For diagnostic builds whose results are deliberately wrong (e.g. `-DGS_ROWSUM_DIAG=1`: the row sum of project_bwd with
coalesced instead of per-slot addresses) -- the parameters never change, so every variant sees the same lists:
    GS_LIB_PATH=easy_gaussian_splatting_amd/libgsraster_<variant>.so python tools/stage_probe.py [iters]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from easy_gaussian_splatting_amd.synthetic import config_bench_1m
from easy_gaussian_splatting_amd import rendering

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
sc = config_bench_1m()
t = {k: torch.from_numpy(v).to(dev) for k, v in sc.items() if isinstance(v, np.ndarray)}
W, H, deg = sc["width"], sc["height"], sc["sh_degree"]
ins = [t[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "shs")]
vc = None
for it in range(iters + 3):
    if it == 3:
        rendering.profile_stages(True)
    img, alpha, meta = rendering.rasterization(*ins, t["viewmats"], t["Ks"], W, H, sh_degree=deg, packed=False,
                                               backgrounds=t["backgrounds"], absgrad=True, _tile_culling="tight")
    if vc is None:
        vc = torch.rand_like(img) / (W * H)
    torch.autograd.grad((img * vc).sum(), ins)
st = rendering.profile_stages(False) or {}
torch.cuda.synchronize()
print(os.environ.get("GS_LIB_PATH", "base").split("libgsraster")[-1], {k[3:]: round(float(np.mean(v)), 4) for k, v in sorted(st.items())})
