#!/usr/bin/env python3
"""The captured train step with the list stages in one round and in two depth rounds (include/gs_raster.h "Depth rounds"), on one of
the workloads of tools/config_run.py:
    python tools/rounds_time.py heavy2M [tight|gsplat_eager] [fractions, e.g. 0.0625,0.125,0.25 | a list with off / auto: exactly these] [steps]
One JSON line per variant: ms per step (HIP events around `steps` replays), listed intersections per step (both rounds), the
front round's share, live tiles behind it, and the runner's report."""
import json, os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
from config_run import LRS, MAKE, model_from_scene
from easy_gaussian_splatting_amd import _native as nat
from easy_gaussian_splatting_amd.loss import LossComputer
from easy_gaussian_splatting_amd.model import build_optimizers
from easy_gaussian_splatting_amd.train_graph import TrainStepGraph


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "heavy2M"
    mode = sys.argv[2] if len(sys.argv) > 2 else "tight"
    fracs = [x if x in ("off", "auto") else float(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "0.0625,0.125,0.25").split(",")]
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 30
    exact = any(isinstance(x, str) for x in fracs)   # ("off" / "auto" named in the list: exactly these variants)
    dev = torch.device("cuda:0")
    sc = MAKE[name]()
    W, H = int(sc["width"]), int(sc["height"])
    data = {"w2c": torch.from_numpy(sc["viewmats"][0]).to(dev), "K": torch.from_numpy(sc["Ks"][0]).to(dev), "width": W, "height": H}
    g = torch.Generator().manual_seed(7)
    gt = torch.nn.functional.interpolate(torch.rand((1, 3, H // 16 + 1, W // 16 + 1), generator=g), size=(H, W), mode="bilinear")[0].permute(1, 2, 0).contiguous().to(dev)
    for variant in (fracs if exact else ["off"] + fracs + ["auto"]):
        model = model_from_scene(sc, dev)
        model.tile_culling = mode
        opt = build_optimizers(model, *LRS, fused="hip")
        kw = {"rounds": variant} if isinstance(variant, str) else {"rounds": "on", "round_fraction": variant}
        t0 = time.perf_counter()
        runner = TrainStepGraph(model, opt, LossComputer(0.2, clamp_input=True), data, gt, None, **kw)
        build_s = time.perf_counter() - t0
        for _ in range(5):
            runner.step()
        runner.finish()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(runner.stream):
            e0.record()
        for _ in range(steps):
            runner.step()
        with torch.cuda.stream(runner.stream):
            e1.record()
        runner.finish()
        torch.cuda.synchronize()
        rep = runner.report()
        out = {"config": name, "list_mode": mode, "variant": variant, "ms_per_step": round(e0.elapsed_time(e1) / steps, 4),
               "listed_per_step": int(runner.buf["info"][0]), "probed_listed_one_round": rep["probed_isects"], "rounds": rep["rounds"],
               "binning": rep["binning"], "overflows": rep["overflows"], "gradient_rows": rep["seen_rows"], "build_s": round(build_s, 2),
               "loss": [round(float(v), 6) for v in runner.buf["loss3"].tolist()]}
        if rep["rounds"]:
            blk = runner.buf["rounds"].tolist()
            out.update(front_listed=blk[nat.GS_ROUND_BASE], live_tiles_behind_front=blk[nat.GS_ROUND_LIVE], tiles=runner.tw * runner.th,
                       front_gaussians=blk[nat.GS_ROUND_FRONT_N], round_fraction=rep["round_fraction"])
        print(json.dumps(out), flush=True)
        del runner, model, opt
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
