"""Runs the BASELINE.json configs S2/S3/S5 (synthetic stand-ins) through the
HIP path: forward + backward timing and sanity checks.  GPU box only."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from easy_gaussian_splatting_amd.synthetic import make_scene
from easy_gaussian_splatting_amd.rendering import rasterization

dev = torch.device("cuda:0")
CFG = {
    "S2 lego-like 300k 800x800 SH3": dict(n=300_000, width=800, height=800, sh_degree=3, extent=(2, 2, 2), scale_range=(0.003, 0.03), dist=5.0, white_bg=True),
    "S3 truck-like 2M 1920x1080 SH3": dict(n=2_000_000, width=1920, height=1080, sh_degree=3, extent=(4, 2.25, 4), scale_range=(0.003, 0.03), dist=8.0, white_bg=False),
    "S5 stress 5M 3840x2160 SH3": dict(n=5_000_000, width=3840, height=2160, sh_degree=3, extent=(4, 2.25, 4), scale_range=(0.002, 0.02), dist=8.0, white_bg=False),
}
for name, kw in CFG.items():
    sc = make_scene(seed=42, **kw)
    t = {k: torch.from_numpy(v).to(dev) for k, v in sc.items() if isinstance(v, np.ndarray)}
    W, H = sc["width"], sc["height"]
    ins = [t[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities")]
    sh0 = t["shs"][:, :1].contiguous().requires_grad_(True); shr = t["shs"][:, 1:].contiguous().requires_grad_(True)
    def fwd():
        return rasterization(*ins, (sh0, shr), t["viewmats"], t["Ks"], W, H, sh_degree=3, packed=False, backgrounds=t["backgrounds"], absgrad=True)
    img, a, meta = fwd()
    vc = torch.randn_like(img) / (W * H)
    for _ in range(2):
        img, a, meta = fwd(); (img * vc).sum().backward()
    torch.cuda.synchronize(); t0 = time.time(); it = 10
    for _ in range(it):
        img, a, meta = fwd(); (img * vc).sum().backward()
    torch.cuda.synchronize(); fb = (time.time() - t0) / it
    with torch.no_grad():
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(it):
            img2, _, _ = fwd()
        torch.cuda.synchronize(); ff = (time.time() - t0) / it
    ok = bool(torch.isfinite(img).all()) and all(bool(torch.isfinite(p.grad).all()) for p in ins + [sh0, shr])
    print(f"{name}: vis={(meta['radii']>0).sum().item()} I={meta['flatten_ids'].numel()} fwd {ff*1e3:.3f} ms, fwd+bwd {fb*1e3:.3f} ms, finite={ok}, "
          f"alpha mean {a.mean().item():.3f}, peak mem {torch.cuda.max_memory_allocated()/2**30:.2f} GiB", flush=True)
    del t, ins, sh0, shr, img, a, meta, vc
    torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
