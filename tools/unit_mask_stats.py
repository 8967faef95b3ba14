"""Per-work-unit pixel coverage of blend_bwd at the bench workload (instrumented variant library: every (entry, quadrant)
pair the forward lists ORs its ballot of taking pixels into its unit's 64-bit mask).  Prints what a backward that streams
only the union of its unit's pixels (or only the pixel rows it touches) could save, including the lock-step of the eight
pipelines of a wave (the trip count is the maximum over the eight units)."""
import ctypes as ct, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from easy_gaussian_splatting_amd import _native as nat
from easy_gaussian_splatting_amd import workspace as WS
from easy_gaussian_splatting_amd.rendering import rasterization
from easy_gaussian_splatting_amd.synthetic import config_bench_1m
dev = torch.device("cuda:0")
sc = config_bench_1m()
t = {k: torch.from_numpy(v).to(dev) for k, v in sc.items() if isinstance(v, np.ndarray)}
L = nat.lib(); L.gs_debug_unit_masks.argtypes = [ct.c_void_p, ct.c_int, ct.c_int]
ins = [t[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "shs")]
for it in range(2):
    assert L.gs_debug_unit_masks(None, 1 << 22, 1) == 0
    dbg = {}
    img, alpha, meta = rasterization(*ins, t["viewmats"], t["Ks"], 1920, 1080, sh_degree=3, packed=False, backgrounds=t["backgrounds"],
                                     absgrad=True, _tile_culling="tight", _debug=dbg)
    torch.cuda.synchronize()
n_units = int(dbg["unit_counter"].item())
masks = np.zeros(1 << 22, np.uint64)
assert L.gs_debug_unit_masks(masks.ctypes.data, 1 << 22, 0) == 0
# unit descriptors live in the lease the autograd node holds: read them through the lease of the pool's busy lease
lease = img.grad_fn.state["lease"] if hasattr(img.grad_fn, "state") else None
ud = lease.view(WS.UNIT_DESC, 4 * n_units).view(-1, 4).cpu().numpy()
rows = ud[:, 3].astype(np.int64)
m = masks[rows]
pc = np.array([bin(int(x)).count("1") for x in m])
rowmask = np.zeros(len(m), np.int64)
for r in range(8):
    rowmask |= (((m >> np.uint64(8 * r)) & np.uint64(0xFF)) != 0).astype(np.int64) << r
nrows = np.array([bin(int(x)).count("1") for x in rowmask])
span = np.array([(int(x).bit_length() - (int(x) & -int(x)).bit_length() + 1) if x else 0 for x in rowmask])
n_in = ud[:, 1] & 0xFF
print("units", n_units, "mean entries/unit %.1f" % n_in.mean())
print("pixels in the union: mean %.1f of 64; rows touched: mean %.2f of 8; contiguous row span: mean %.2f" % (pc.mean(), nrows.mean(), span.mean()))
g = (n_units // 8) * 8
mx = pc[:g].reshape(-1, 8).max(1)
mxs = span[:g].reshape(-1, 8).max(1)
print("per wave (8 units in lock-step): max pixels mean %.1f -> steps %.1f + 7 (now 64 + 7); max row span mean %.2f -> steps %.1f + 7" % (mx.mean(), mx.mean(), mxs.mean(), 8 * mxs.mean()))
print("steps now 71; pixel-compacted %.1f (%.0f %%); row-window %.1f (%.0f %%)" % (mx.mean() + 7, 100 * (mx.mean() + 7) / 71, 8 * mxs.mean() + 7, 100 * (8 * mxs.mean() + 7) / 71))
# if the units of a wave were sorted by pixel count (upper bound on what regrouping could give)
so = np.sort(pc[:g])[::-1].reshape(-1, 8).max(1)
print("with units grouped by coverage: %.1f + 7" % so.mean())
