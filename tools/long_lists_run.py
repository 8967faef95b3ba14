#!/usr/bin/env python3
"""The bench's secondary workload (synthetic.config_long_lists: 200 k heavy-tailed splats at 1080p, mean tile list > 2 000)
forward + backward in a loop, for kernel-level profiles: tools/prof_cmd.sh longlists tools/long_lists_run.py [gsplat_eager|gsplat|tight] [iters]"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
from easy_gaussian_splatting_amd.synthetic import config_long_lists
from easy_gaussian_splatting_amd.rendering import rasterization
mode = sys.argv[1] if len(sys.argv) > 1 else "tight"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
sc = config_long_lists(n=200_000, width=1920, height=1080)
t = {k: torch.from_numpy(v).to(dev) for k, v in sc.items() if isinstance(v, np.ndarray)}
W, H = int(sc["width"]), int(sc["height"])
ins = [t[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "shs")]
from easy_gaussian_splatting_amd import rendering
for it in range(iters + 2):
    if it == 2:
        rendering.profile_stages(True)
    img, alpha, meta = rasterization(*ins, t["viewmats"], t["Ks"], W, H, sh_degree=int(sc["sh_degree"]), packed=False,
                                     backgrounds=t["backgrounds"], absgrad=True, _tile_culling=mode)
    img.sum().backward()
st_ms = rendering.profile_stages(False)
print("stages (ms):", {k[3:]: round(float(np.mean(v)), 4) for k, v in sorted(st_ms.items())}, "total", round(sum(float(np.mean(v)) for v in st_ms.values()), 4))
cnt = torch.diff(torch.cat([meta["isect_offsets"].reshape(-1), torch.tensor([meta["flatten_ids"].numel()], device=dev, dtype=torch.int32)]))
print(f"{mode}: I={meta['flatten_ids'].numel()} mean list {float(cnt.float().mean()):.0f} max list {int(cnt.max())} "
      f"lists >1024: {int((cnt > 1024).sum())} >4096: {int((cnt > 4096).sum())} >8192: {int((cnt > 8192).sum())} >16384: {int((cnt > 16384).sum())} of {cnt.numel()}")
dbg = {}
img, alpha, meta = rasterization(*ins, t["viewmats"], t["Ks"], W, H, sh_degree=int(sc["sh_degree"]), packed=False,
                                 backgrounds=t["backgrounds"], absgrad=True, _tile_culling=mode, _debug=dbg)
torch.cuda.synchronize()
q = int(dbg["qcnt"].sum()); u = int(dbg["unit_counter"][0])
print(f"quadrant entries kept for the backward {q} ({q / (4 * meta['flatten_ids'].numel()):.3f} of 4 I), work units {u}; saturated pixels {float((alpha > 1 - 2e-4).float().mean()):.3f}")
import time
with torch.no_grad():
    for _ in range(3):
        rasterization(*ins, t["viewmats"], t["Ks"], W, H, sh_degree=int(sc["sh_degree"]), packed=False, backgrounds=t["backgrounds"], _tile_culling=mode)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(10):
        rasterization(*ins, t["viewmats"], t["Ks"], W, H, sh_degree=int(sc["sh_degree"]), packed=False, backgrounds=t["backgrounds"], _tile_culling=mode)
    torch.cuda.synchronize(); print(f"inference forward {(time.time() - t0) * 100:.3f} ms")
