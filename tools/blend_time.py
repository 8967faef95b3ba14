#!/usr/bin/env python3
"""Kernel times of the rasterizer's stages on a FIXED scene (no optimizer in the loop): rasterization forward + backward of the
bench workload, HIP events around every stage (rendering.profile_stages), 30 repetitions.  For A/B runs of library variants
whose backward may be numerically wrong on purpose (timing experiments):  GS_LIB_PATH=... tools/blend_time.py [n_gauss]"""
import json, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
from easy_gaussian_splatting_amd import rendering
from easy_gaussian_splatting_amd.synthetic import config_bench_1m
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = torch.device("cuda:0")
sc = config_bench_1m(n=n)
t = {k: torch.from_numpy(v).to(dev) for k, v in sc.items() if isinstance(v, np.ndarray)}
ins = [t[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities")]
sh0, shr = t["shs"][:, :1].contiguous().requires_grad_(True), t["shs"][:, 1:].contiguous().requires_grad_(True)
vc = None
def step():
    global vc
    img, _, meta = rendering.rasterization(*ins, (sh0, shr), t["viewmats"], t["Ks"], 1920, 1080, sh_degree=3, packed=False,
                                           backgrounds=t["backgrounds"], absgrad=True, _tile_culling="tight")
    if vc is None:
        vc = torch.randn_like(img) / (1920 * 1080)
    torch.autograd.grad((img * vc).sum(), ins + [sh0, shr])
    return meta
meta = step()
for _ in range(5): step()
rendering.profile_stages(True)
for _ in range(30): step()
st = rendering.profile_stages(False) or {}
print(json.dumps({"lib": os.path.basename(os.environ.get("GS_LIB_PATH", "libgsraster.so")), "n_isects": int(meta["flatten_ids"].shape[0]),
                  **{k[3:]: round(float(np.median(v)), 4) for k, v in sorted(st.items())}}))
