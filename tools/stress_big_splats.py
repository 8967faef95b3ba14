"""Large / elongated splats (footprints of hundreds of tiles, long tile lists): timing + sanity."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from easy_gaussian_splatting_amd.synthetic import make_scene
from easy_gaussian_splatting_amd import rendering
from easy_gaussian_splatting_amd.rendering import rasterization
dev = torch.device("cuda:0")
for name, kw in {
    "200k big splats 1080p": dict(n=200_000, width=1920, height=1080, sh_degree=3, extent=(4, 2.25, 4), scale_range=(0.02, 0.3), dist=8.0, white_bg=False),
    "1M mixed 1080p": dict(n=1_000_000, width=1920, height=1080, sh_degree=3, extent=(4, 2.25, 4), scale_range=(0.003, 0.15), dist=8.0, white_bg=False),
}.items():
    sc = make_scene(seed=1, **kw)
    t = {k: torch.from_numpy(v).to(dev) for k, v in sc.items() if isinstance(v, np.ndarray)}
    W, H = sc["width"], sc["height"]
    ins = [t[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities")]
    sh0 = t["shs"][:, :1].contiguous().requires_grad_(True); shr = t["shs"][:, 1:].contiguous().requires_grad_(True)
    fwd = lambda: rasterization(*ins, (sh0, shr), t["viewmats"], t["Ks"], W, H, sh_degree=3, packed=False, backgrounds=t["backgrounds"], absgrad=True)
    img, a, meta = fwd(); vc = torch.randn_like(img) / (W * H)
    for _ in range(2):
        img, a, meta = fwd(); (img * vc).sum().backward()
    rendering.profile_stages(True)
    torch.cuda.synchronize(); t0 = time.time(); it = 5
    for _ in range(it):
        img, a, meta = fwd(); (img * vc).sum().backward()
    torch.cuda.synchronize(); fb = (time.time() - t0) / it
    st = rendering.profile_stages(False)
    tpg = meta["tiles_per_gauss"].float()
    cnt = torch.diff(torch.cat([meta["isect_offsets"].reshape(-1), torch.tensor([meta["flatten_ids"].numel()], device=dev, dtype=torch.int32)]))
    print(f"{name}: I={meta['flatten_ids'].numel()} max tiles/gauss={int(tpg.max())} max list={int(cnt.max())} mean list={float(cnt.float().mean()):.0f} "
          f"fwd+bwd {fb*1e3:.2f} ms  stages " + ", ".join(f"{k[3:]}={np.mean(v):.3f}" for k, v in sorted(st.items())), flush=True)
