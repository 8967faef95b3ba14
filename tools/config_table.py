"""BASELINE.json configs S1/S2/S3/S5 (synthetic stand-ins at the stated N / HxW) through the
HIP path: forward, forward+backward and full train step; S1 also on the host cores with the oracle
(BASELINE.md section 3 protocol).  Prints a markdown table.  GPU box only."""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from easy_gaussian_splatting_amd.synthetic import config_s1, config_s2, config_s3, config_s5
from easy_gaussian_splatting_amd.train_graph import TrainStepGraph
from easy_gaussian_splatting_amd.model import GaussianModel, build_optimizers
from easy_gaussian_splatting_amd.loss import LossComputer
from oracle import c_oracle as CO

dev = torch.device("cuda:0")
CFG = {
    "S1 10k 256x256 SH0": config_s1(),
    "S2 300k 800x800 SH3": config_s2(),
    "S3 2M 1920x1080 SH3": config_s3(),
    "S5 5M 3840x2160 SH3": config_s5(),
}

def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return statistics.median(ts)

print("| config | N_vis | I (tight) | forward ms | fwd+bwd ms | train step ms (eager) | train step ms (hipGraph) | it/s (hipGraph) | peak GiB |")
print("|---|---|---|---|---|---|---|---|---|")
for name, sc in CFG.items():
    T = lambda a: torch.from_numpy(a).to(dev)
    W, H, deg = sc["width"], sc["height"], sc["sh_degree"]
    op = np.clip(sc["opacities"], 1e-6, 1 - 1e-6)
    shs = T(sc["shs"])
    model = GaussianModel(means=T(sc["means"]), log_scales=torch.log(T(sc["scales"])), quats=T(sc["quats"]), sh_0=shs[:, :1].contiguous(),
                          sh_rest=shs[:, 1:].contiguous(), logit_opacities=T(np.log(op / (1 - op)).astype(np.float32)), sh_degree=deg,
                          white_background=bool(sc["backgrounds"][0, 0] > 0.5)).to(dev)
    data = {"w2c": T(sc["viewmats"][0]), "K": T(sc["Ks"][0]), "width": W, "height": H}
    gt = torch.rand(H, W, 3, device=dev)
    opt = build_optimizers(model, 1.6e-4, 5e-3, 1e-3, 2.5e-3, 1.25e-4, 5e-2, fused="hip")
    lc = LossComputer(0.2)
    vc = torch.randn(H, W, 3, device=dev) / (W * H)
    def fwd():
        with torch.no_grad():
            return model(data)
    def fwdbwd():
        out = model(data); (out["render_img"] * vc).sum().backward(); opt.zero_grad()
    def step():
        out = model(data); lc.get_loss_dict(out["render_img"], gt)["total"].backward(); model.update_statistics(data, out); opt.step(); opt.zero_grad()
    out = model(data)
    from easy_gaussian_splatting_amd.rendering import rasterization
    with torch.no_grad():
        _, _, meta = rasterization(model.means, model.quats, model.scales, model.opacities, (model.sh_0, model.sh_rest), data["w2c"][None], data["K"][None], W, H, sh_degree=deg, packed=False, _tile_culling="tight")
    reps = 20 if sc["means"].shape[0] <= 2_000_000 else 8
    f, fb, st = timed(fwd, reps), timed(fwdbwd, reps), timed(step, reps)
    runner = TrainStepGraph(model, opt, LossComputer(0.2, clamp_input=True), data, gt, None)
    def gstep():
        for _ in range(5):
            runner.step()
    sg = timed(gstep, max(reps // 2, 3)) / 5
    runner.finish()
    print(f"| {name} | {int((meta['radii']>0).sum())} | {meta['flatten_ids'].numel()} | {f:.3f} | {fb:.3f} | {st:.3f} | {sg:.3f} | {1e3/sg:.0f} | {torch.cuda.max_memory_allocated()/2**30:.1f} |", flush=True)
    del runner
    if name.startswith("S1"):
        cores = os.cpu_count()
        ts_f, ts_fb = [], []
        for r in range(13):
            t0 = time.perf_counter()
            fw = CO.render(sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["shs"], sc["viewmats"], sc["Ks"], W, H, sh_degree=deg, backgrounds=sc["backgrounds"], dtype=np.float32)
            t1 = time.perf_counter()
            CO.backward(fw, np.ones_like(fw["render_colors"]) / (W * H))
            t2 = time.perf_counter()
            if r >= 3:
                ts_f.append((t1 - t0) * 1e3); ts_fb.append((t2 - t0) * 1e3)
        cf, cfb = statistics.median(ts_f), statistics.median(ts_fb)
        print(f"| S1 on the host: oracle/c (C+OpenMP), {cores} cores | | | {cf:.2f} | {cfb:.2f} | | | |")
        print(f"| S1 GPU / CPU speed-up | | | {cf/f:.0f}x | {cfb/fb:.0f}x | | | |", flush=True)
    del model, opt, out, meta, gt, vc
    torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
