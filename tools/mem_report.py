#!/usr/bin/env python3
"""Where the captured step's memory goes: builds `TrainStepGraph` on one of tools/config_run.py's workloads and prints the
size of every buffer of its workspace next to torch's peak, in bytes per LISTED intersection.
    python tools/mem_report.py heavy2M gsplat_eager [steps]"""
import json, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from config_run import LRS, MAKE, model_from_scene
from easy_gaussian_splatting_amd.loss import LossComputer
from easy_gaussian_splatting_amd.model import build_optimizers
from easy_gaussian_splatting_amd.train_graph import TrainStepGraph


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "heavy2M"
    mode = sys.argv[2] if len(sys.argv) > 2 else "gsplat_eager"
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    dev = torch.device("cuda:0")
    sc = MAKE[name]()
    W, H = int(sc["width"]), int(sc["height"])
    model = model_from_scene(sc, dev)
    model.tile_culling = mode
    opt = build_optimizers(model, *LRS, fused="hip")
    data = {"w2c": torch.from_numpy(sc["viewmats"][0]).to(dev), "K": torch.from_numpy(sc["Ks"][0]).to(dev), "width": W, "height": H}
    g = torch.Generator().manual_seed(7)
    gt = torch.nn.functional.interpolate(torch.rand((1, 3, H // 16 + 1, W // 16 + 1), generator=g), size=(H, W), mode="bilinear")[0].permute(1, 2, 0).contiguous().to(dev)
    torch.cuda.synchronize()
    base = torch.cuda.memory_allocated(dev)
    torch.cuda.reset_peak_memory_stats(dev)
    runner = TrainStepGraph(model, opt, LossComputer(lambda_ssim=0.2, clamp_input=True), data, gt, None)
    after_build = torch.cuda.memory_allocated(dev)
    peak_build = torch.cuda.max_memory_allocated(dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        runner.step()
    runner.finish()
    ev0.record(runner.stream)
    for _ in range(steps):
        runner.step()
    ev1.record(runner.stream)
    runner.finish()
    torch.cuda.synchronize()
    rep = runner.report()
    listed = max(1, rep["probed_isects"])
    sizes = {k: int(t.numel() * t.element_size()) for k, t in runner._pool.items()}
    for k, t in runner.buf.items():
        if isinstance(t, torch.Tensor) and k not in sizes:
            sizes[k] = int(t.numel() * t.element_size())
    top = sorted(sizes.items(), key=lambda kv: -kv[1])
    print(json.dumps({
        "config": name, "list_mode": mode, "n_gaussians": int(sc["means"].shape[0]), "listed": listed, "capacity": rep["capacity_isects"],
        "binning": rep["binning"], "overflows": rep["overflows"], "captures": rep["captures"],
        "model_and_inputs_GiB": round(base / 2 ** 30, 2), "runner_resident_GiB": round((after_build - base) / 2 ** 30, 2),
        "peak_GiB": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 2), "peak_during_build_GiB": round(peak_build / 2 ** 30, 2),
        "peak_bytes_per_listed": round(torch.cuda.max_memory_allocated(dev) / listed, 1),
        "runner_bytes_per_listed": round((after_build - base) / listed, 1),
        "ms_per_step": round(ev0.elapsed_time(ev1) / steps, 4),
        "status_words": [int(v) for v in runner.status[:8].tolist()],
        "buffers_MiB": {k: round(v / 2 ** 20, 1) for k, v in top[:24]}}), flush=True)


if __name__ == "__main__":
    main()
