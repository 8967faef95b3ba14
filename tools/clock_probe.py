#!/usr/bin/env python3
"""Sustained shader clock under the blend kernels themselves.

Run on the GPU box with the -DGS_CLOCK_PROBE variant of the library (tools/tune_variants.sh build "clk:-DGS_CLOCK_PROBE"):
    GS_ALLOW_VARIANT=1 GS_LIB_PATH=build/variants/libgsraster_clk.so python tools/clock_probe.py [out.json]
Every wave of blend_fwd / blend_bwd adds d(s_memtime) and d(s_memrealtime) to device words; the ratio x the wall-clock rate is
the shader clock those waves ran at.  Steady state: 30 eager train steps at the bench workload before the counters are read.
Also samples `rocm-smi --showclocks` from a side thread while the loop runs (what the driver reports), if it is readable."""
import ctypes as ct, json, os, subprocess, sys, threading, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import bench
from easy_gaussian_splatting_amd import _native as nat
from easy_gaussian_splatting_amd.loss import LossComputer
from easy_gaussian_splatting_amd.model import build_optimizers

dev = torch.device("cuda:0")
L = nat.lib()
try:
    L.gs_debug_clock_probe
except AttributeError:
    raise SystemExit("this library was not built with -DGS_CLOCK_PROBE")
L.gs_debug_clock_probe.argtypes = [ct.c_void_p, ct.c_int]
L.gs_debug_clock_probe.restype = ct.c_int
sc, model = bench.build_workload(1_000_000, 8, dev)
W, H = sc["width"], sc["height"]
datas = [{"w2c": torch.from_numpy(sc["viewmats"][v]).to(dev), "K": torch.from_numpy(sc["Ks"][v]).to(dev), "width": W, "height": H} for v in range(8)]
targets = [bench.smooth_target(H, W, 1234 + v, dev) for v in range(8)]
opt = build_optimizers(model, 1.6e-4, 5e-3, 1e-3, 2.5e-3, 1.25e-4, 5e-2, fused="hip")
lc = LossComputer(0.2, clamp_input=True)
one = torch.ones((), device=dev)

def step(i):
    out = model(datas[i % 8], clamp=False)
    lc.get_loss_dict(out["render_img"], targets[i % 8], None)["total"].backward(gradient=one)
    model.update_statistics(datas[i % 8], out)
    opt.step(); opt.zero_grad()

smi, stop = [], threading.Event()
def poll():
    while not stop.is_set():
        try:
            o = subprocess.run(["rocm-smi", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout
            j = json.loads(o)
            for card, v in j.items():
                s = v.get("sclk clock speed:") or v.get("sclk clock level:") or ""
                smi.append(str(s))
        except Exception as e:
            smi.append("unreadable: " + repr(e)[:80]); return
        time.sleep(0.2)
th = threading.Thread(target=poll, daemon=True); th.start()
for i in range(20): step(i)
torch.cuda.synchronize()
L.gs_debug_clock_probe(None, 1)
n = 60
t0 = time.perf_counter()
for i in range(n): step(i)
torch.cuda.synchronize()
wall = time.perf_counter() - t0
stop.set()
acc = (ct.c_int64 * 4)()
L.gs_debug_clock_probe(acc, 0)
# wall clock rate of s_memrealtime: hipDeviceAttributeWallClockRate (tools/micro/clock_probe prints it; 100 MHz on gfx9)
wall_khz = int(os.environ.get("GS_WALL_CLOCK_KHZ", "100000"))
res = {"workload": "1 M Gaussians, 1920x1080, SH3, eager train steps, 8 shuffled views", "steps": n, "ms_per_step": round(1e3 * wall / n, 4),
       "wall_clock_rate_kHz": wall_khz,
       "blend_fwd_kernel<train>": {"s_memtime_cycles": int(acc[0]), "s_memrealtime_ticks": int(acc[1]),
                                   "clock_MHz": round(acc[0] / max(acc[1], 1) * wall_khz * 1e-3, 1)},
       "blend_bwd_kernel": {"s_memtime_cycles": int(acc[2]), "s_memrealtime_ticks": int(acc[3]),
                            "clock_MHz": round(acc[2] / max(acc[3], 1) * wall_khz * 1e-3, 1)},
       "rocm_smi_sclk_samples": smi[:40]}
print(json.dumps(res))
if len(sys.argv) > 1:
    with open(sys.argv[1], "w") as f:
        json.dump(res, f, indent=1)
