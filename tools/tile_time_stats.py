"""Per-tile run time of blend_fwd<train> at the bench workload (needs the instrumented variant library built by the
recipe in DESIGN.md section 8: GS_LIB_PATH=.../libgsraster_timing.so).  Prints how the kernel's duration relates to the
distribution of per-tile (= per-wave) durations: the critical path against the mean."""
import ctypes as ct, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from easy_gaussian_splatting_amd import _native as nat
from easy_gaussian_splatting_amd.rendering import rasterization
from easy_gaussian_splatting_amd.synthetic import config_bench_1m
dev = torch.device("cuda:0")
sc = config_bench_1m()
t = {k: torch.from_numpy(v).to(dev) for k, v in sc.items() if isinstance(v, np.ndarray)}
mode = sys.argv[1] if len(sys.argv) > 1 else "tight"
ins = [t[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "shs")]
for _ in range(3):
    img, alpha, meta = rasterization(*ins, t["viewmats"], t["Ks"], 1920, 1080, sh_degree=3, packed=False, backgrounds=t["backgrounds"], absgrad=True, _tile_culling=mode)
torch.cuda.synchronize()
n = 120 * 68
cyc = np.zeros(n, np.int64); bk = np.zeros(n, np.int32)
L = nat.lib(); L.gs_debug_tile_cycles.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_int]
assert L.gs_debug_tile_cycles(cyc.ctypes.data, bk.ctypes.data, n) == 0
off = meta["isect_offsets"].reshape(-1).cpu().numpy().astype(np.int64)
ln = np.diff(np.append(off, meta["flatten_ids"].numel()))
us = cyc / 2100.0   # shader-clock cycles at ~2.1 GHz
print("mode", mode, "I", int(ln.sum()), "mean len %.0f max %d" % (ln.mean(), ln.max()))
print("per-tile us: mean %.1f median %.1f p90 %.1f p99 %.1f max %.1f" % (us.mean(), np.median(us), np.percentile(us, 90), np.percentile(us, 99), us.max()))
print("buckets walked: mean %.2f of %.2f listed" % (bk.mean(), np.ceil(ln / 64).mean()))
print("corr(us, len) %.3f  corr(us, buckets walked) %.3f" % (np.corrcoef(us, ln)[0, 1], np.corrcoef(us, bk)[0, 1]))
order = np.argsort(-us)[:10]
print("slowest tiles:", [(int(i), int(ln[i]), int(bk[i]), round(float(us[i]), 1)) for i in order])
print("sum of tile time / 6144 wave slots = %.1f us (perfectly balanced kernel)" % (us.sum() / 6144))
for thr in (1.25, 1.5, 2.0):
    sel = ln > thr * ln.mean()
    if not sel.any() or sel.all():
        continue
    print(f"tiles with len > {thr} x mean: {int(sel.sum())}; their mean time {us[sel].mean():.1f} us, max {us[sel].max():.1f}; max over the others {us[~sel].max():.1f}")
