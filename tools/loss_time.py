#!/usr/bin/env python3
"""Times gs_l1_ssim_fwd / gs_l1_ssim_bwd alone (HIP events, the bench resolution) for the product library and any variants:
   tools/loss_time.py [H W] -- lib names after `--` are libgsraster_<name>.so files in build/variants/ (GS_ALLOW_VARIANT=1).
Each variant's outputs are compared with the first library's (largest absolute difference of loss3 / maps-derived gradient)."""
import ctypes as ct, json, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
from easy_gaussian_splatting_amd import _native as nat
args = sys.argv[1:]
names = ["product"]
if "--" in args:
    i = args.index("--"); names += args[i + 1:]; args = args[:i]
H, W = (int(args[0]), int(args[1])) if len(args) >= 2 else (1080, 1920)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
gt = torch.rand((H, W, 3), device=dev, generator=g)
render = (gt + 0.1 * torch.randn((H, W, 3), device=dev, generator=g)).contiguous()
# the train step reads a DIFFERENT view's ground truth every step (cold in the 256 MB last-level cache): the timed loops rotate
# over enough copies of the inputs to overflow it; the first copy is the one the outputs are compared on
NSETS = int(os.environ.get("GS_LOSS_SETS", "24"))
gts = [gt] + [gt.clone() for _ in range(NSETS - 1)]
renders = [render] + [render.clone() for _ in range(NSETS - 1)]
one = torch.ones((1,), device=dev)
# GS_LOSS_MASK=1: with a camera mask, as the train step passes one (the reference's loss always composites with it)
mask = (torch.rand((H, W), device=dev, generator=g) < 0.05).float() if os.environ.get("GS_LOSS_MASK", "1") == "1" else None
mp = None if mask is None else mask.data_ptr()
def load(name):
    path = nat.LIB_PATH if name == "product" else os.path.join(nat.VARIANT_DIR, f"libgsraster_{name}.so")
    L = ct.CDLL(path)
    for fn in ("gs_loss_workspace_floats", "gs_l1_ssim_fwd", "gs_l1_ssim_bwd"):
        f = getattr(L, fn); f.restype, f.argtypes = nat.SIGNATURES[fn]
    return L
ref = None
for name in names:
    L = load(name)
    ws = torch.zeros((int(L.gs_loss_workspace_floats(H, W)),), device=dev)
    out3 = torch.zeros((3,), device=dev); v = torch.zeros((H, W, 3), device=dev)
    st = torch.cuda.current_stream().cuda_stream
    fwd = lambda i=0: L.gs_l1_ssim_fwd(st, H, W, 0.2, renders[i].data_ptr(), gts[i].data_ptr(), mp, 1, ws.data_ptr(), out3.data_ptr())
    bwd = lambda i=0: L.gs_l1_ssim_bwd(st, H, W, 0.2, renders[i].data_ptr(), gts[i].data_ptr(), mp, 1, ws.data_ptr(), one.data_ptr(), v.data_ptr())
    res = {"lib": name}
    for tag, fn in (("fwd_us", fwd), ("bwd_us", bwd)):
        for i in range(20): fn(i % NSETS)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(300): fn(i % NSETS)
        e1.record(); torch.cuda.synchronize()
        res[tag] = round(e0.elapsed_time(e1) / 300 * 1e3, 2)   # (the forward entry includes the one-block reduction)
    assert fwd() == 0 and bwd() == 0
    cur = (out3.clone(), v.clone())
    if ref is None: ref = cur
    res["loss3"] = [float(x) for x in cur[0]]
    res["max_abs_diff_vs_first"] = [float((cur[0] - ref[0]).abs().max()), float((cur[1] - ref[1]).abs().max())]
    res["v_absmax"] = float(cur[1].abs().max())
    print(json.dumps(res), flush=True)
