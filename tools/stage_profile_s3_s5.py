import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import numpy as np, torch
from easy_gaussian_splatting_amd.synthetic import make_scene
from easy_gaussian_splatting_amd import rendering
from easy_gaussian_splatting_amd.rendering import rasterization
dev = torch.device("cuda:0")
for name, kw in {"S3": dict(n=2_000_000, width=1920, height=1080, sh_degree=3, extent=(4, 2.25, 4), scale_range=(0.003, 0.03), dist=8.0, white_bg=False),
                 "S5": dict(n=5_000_000, width=3840, height=2160, sh_degree=3, extent=(4, 2.25, 4), scale_range=(0.002, 0.02), dist=8.0, white_bg=False)}.items():
    sc = make_scene(seed=42, **kw)
    t = {k: torch.from_numpy(v).to(dev) for k, v in sc.items() if isinstance(v, np.ndarray)}
    W, H = sc["width"], sc["height"]
    ins = [t[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities")]
    sh0 = t["shs"][:, :1].contiguous().requires_grad_(True); shr = t["shs"][:, 1:].contiguous().requires_grad_(True)
    fwd = lambda: rasterization(*ins, (sh0, shr), t["viewmats"], t["Ks"], W, H, sh_degree=3, packed=False, backgrounds=t["backgrounds"], absgrad=True, _tile_culling=os.environ.get("GS_CULL", "gsplat_eager"))
    img, a, meta = fwd(); vc = torch.randn_like(img) / (W * H)
    for _ in range(2):
        img, a, meta = fwd(); (img * vc).sum().backward()
    rendering.profile_stages(True)
    for _ in range(5):
        img, a, meta = fwd(); (img * vc).sum().backward()
    st = rendering.profile_stages(False)
    print(name, "I", meta["flatten_ids"].numel(), ", ".join(f"{k[3:]}={np.mean(v):.3f}" for k, v in sorted(st.items())), flush=True)
    del t, ins, sh0, shr, img, a, meta, vc; torch.cuda.empty_cache()
