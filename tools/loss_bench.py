#!/usr/bin/env python3
"""Device time of the fused L1 + SSIM forward / backward at 1080p (HIP events, 200 launches each)."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
from easy_gaussian_splatting_amd import _native as nat
dev = torch.device("cuda:0"); H, W = 1080, 1920
L = nat.lib()
r = torch.rand((H, W, 3), device=dev); g = torch.rand((H, W, 3), device=dev)
ws = torch.empty((int(L.gs_loss_workspace_floats(H, W)),), device=dev); out3 = torch.zeros(3, device=dev); one = torch.ones((), device=dev); v = torch.empty_like(r)
st = torch.cuda.current_stream().cuda_stream
def fwd(): nat.check(L.gs_l1_ssim_fwd(st, H, W, 0.2, r.data_ptr(), g.data_ptr(), None, 1, ws.data_ptr(), out3.data_ptr()), "fwd")
def bwd(): nat.check(L.gs_l1_ssim_bwd(st, H, W, 0.2, r.data_ptr(), g.data_ptr(), None, 1, ws.data_ptr(), one.data_ptr(), v.data_ptr()), "bwd")
for name, fn in (("fwd", fwd), ("bwd", bwd)):
    for _ in range(20): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): fn()
    e1.record(); torch.cuda.synchronize()
    print(name, f"{e0.elapsed_time(e1) / 200 * 1e3:.1f} us", end="  ")
print()
