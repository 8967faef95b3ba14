#!/usr/bin/env python3
"""Benchmark of the rasterization hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one training iteration of the reference's loop around the seam
(/root/reference/train.py:93-157 without DataLoader / TensorBoard): activations ->
rasterization forward -> clamp -> L1 + (1-SSIM) -> backward -> update_statistics ->
[RCCL all-reduce of the flat gradient bucket when N>1] -> Adam step -> zero grads,
on the workload BASELINE.json quotes its metric on: 1 M Gaussians, 1920x1080, SH degree 3
(synthetic generator of SURVEY.md section 8d; there is no dataset in this environment).
One rank per GPU, one view per rank per step (weak scaling); `value` = view-iterations/s of the
whole job.  Forward-only render fps of the same workload is reported next to it.

Extra objects: `roofline` for the dominant blend kernel (algorithmic bytes of SURVEY.md 8d over
its HIP-event duration on the launch stream) and `cpu_baseline` (the C/OpenMP oracle timed on
the host cores on a bounded sample; rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from easy_gaussian_splatting_amd import rendering  # noqa: E402
from easy_gaussian_splatting_amd.distributed import GradBucket, ViewParallelStep, all_reduce_param_grads  # noqa: E402
from easy_gaussian_splatting_amd.loss import LossComputer  # noqa: E402
from easy_gaussian_splatting_amd.model import GaussianModel, build_optimizers  # noqa: E402
from easy_gaussian_splatting_amd.synthetic import config_bench_1m  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def build_workload(n_gauss: int, n_views: int, device):
    sc = config_bench_1m(seed=42, n=n_gauss, n_views=max(n_views, 1))
    t = lambda a: torch.from_numpy(a).to(device)
    shs = t(sc["shs"])
    op = np.clip(sc["opacities"], 1e-6, 1 - 1e-6)
    model = GaussianModel(means=t(sc["means"]), log_scales=torch.log(t(sc["scales"])), quats=t(sc["quats"]),
                          sh_0=shs[:, :1].contiguous(), sh_rest=shs[:, 1:].contiguous(),
                          logit_opacities=t(np.log(op / (1 - op)).astype(np.float32)),
                          sh_degree=sc["sh_degree"], sh_degree_interval=0, white_background=False).to(device)
    return sc, model


def pmc_traffic(kernel: str):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes of this same
    command (profiles/r01_pmc_traffic.json, produced by tools/pmc_summary.py with the gfx950
    FETCH_SIZE x2 correction of MI355X_MICROARCH.md); None when no such profile is committed."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(kernel, {}).get("traffic_bytes_per_launch")
    except (OSError, ValueError):
        return None


def cpu_baseline(sc, sample_n: int):
    """Oracle (C + OpenMP, all host cores) on the first `sample_n` Gaussians of the same workload:
    one rasterization forward + backward.  Reported baseline only."""
    from oracle import c_oracle as CO
    CO.build()
    cores = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    n = min(sample_n, sc["means"].shape[0])
    W, H = sc["width"], sc["height"]
    t0 = time.time()
    fw = CO.render(sc["means"][:n], sc["quats"][:n], sc["scales"][:n], sc["opacities"][:n], sc["shs"][:n],
                   sc["viewmats"][:1], sc["Ks"][:1], W, H, sh_degree=sc["sh_degree"],
                   backgrounds=sc["backgrounds"][:1], dtype=np.float32)
    vc = (np.random.default_rng(0).standard_normal(fw["render_colors"].shape) / (W * H)).astype(np.float32)
    CO.backward(fw, vc)
    dt = time.time() - t0
    return {"value": round(1.0 / dt, 4), "unit": "iters/s", "cores": cores, "kind": "port",
            "sample": f"oracle/c (C+OpenMP) rasterization fwd+bwd, first {n} of the workload's Gaussians "
                      f"at {W}x{H} SH{sc['sh_degree']}, I={fw['n_isects']}, 1 repetition, {dt:.2f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--cpu-sample", type=int, default=1_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--torch-adam", action="store_true", help="torch.optim.Adam(fused=True) instead of the HIP Adam")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback for the product path)")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" is RCCL on ROCm.  GS_BENCH_BACKEND=gloo only exists to exercise this code path with
        # several ranks on a 1-GPU box (RCCL refuses two ranks on one device); never used for numbers.
        backend = os.environ.get("GS_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)

    n_views = max(8, world)
    sc, model = build_workload(args.gaussians, n_views, device)
    W, H = sc["width"], sc["height"]
    view = rank % n_views
    data = {"w2c": torch.from_numpy(sc["viewmats"][view]).to(device), "K": torch.from_numpy(sc["Ks"][view]).to(device),
            "width": W, "height": H}
    g = torch.Generator(device="cpu").manual_seed(1234 + view)
    gt_img = torch.rand((H // 8, W // 8, 3), generator=g).to(device)
    gt_img = torch.nn.functional.interpolate(gt_img.permute(2, 0, 1)[None], size=(H, W), mode="bilinear",
                                             align_corners=False)[0].permute(1, 2, 0).contiguous()
    mask = torch.zeros((H, W), device=device)
    if args.torch_adam:
        optimizer = build_optimizers(model, 1.6e-4, 5e-3, 1e-3, 2.5e-3, 1.25e-4, 5e-2, fused=True)
        bucket = GradBucket(model.parameters())
    else:  # one HIP kernel per step over flat params/moments; gradients are autograd's own tensors
        optimizer = build_optimizers(model, 1.6e-4, 5e-3, 1e-3, 2.5e-3, 1.25e-4, 5e-2, fused="hip")
        bucket = None
    loss_computer = LossComputer(lambda_ssim=0.2, clamp_input=True)   # the model's clamp(0,1) is applied inside the loss kernels
    # N > 1: factorised exchange (all-gather of colour gradients + all-reduce of the geometry
    # gradients, distributed.ViewParallelStep); GS_DP_EXCHANGE=allreduce selects the plain all-reduce
    exchange = "none" if world == 1 else os.environ.get("GS_DP_EXCHANGE", "factorised")
    vp = ViewParallelStep(model, optimizer) if (exchange == "factorised" and bucket is None) else None

    one = torch.ones((), device=device)   # root gradient, allocated once (backward() would fill a new one per step)

    def train_step():
        if vp is not None:
            vp.begin_step(data)
        out = model(data, clamp=False)
        if vp is not None:
            vp.after_forward(data, out)
        loss = loss_computer.get_loss_dict(out["render_img"], gt_img, mask)["total"]
        loss.backward(gradient=one)
        if vp is not None:
            vp.step(data, out)
            return out
        model.update_statistics(data, out)
        if bucket is not None:
            bucket.all_reduce_mean()
            optimizer.step()
            bucket.zero_()
        else:
            all_reduce_param_grads(model.parameters())
            optimizer.step()
            optimizer.zero_grad()
        return out

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # ---- train iterations (the timed region)
    for _ in range(args.warmup):
        train_step()
    barrier()
    rendering.stats["sync_wait_ns"] = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = train_step()
    t_enqueued = time.perf_counter() - t0
    barrier()
    elapsed = time.perf_counter() - t0
    host_wait_ms = rendering.stats["sync_wait_ns"] * 1e-6 / args.steps
    if world > 1:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- forward-only render fps (eval path: /root/reference/eval.py:38-43, but synchronised)
    with torch.no_grad():
        for _ in range(max(3, args.warmup // 2)):
            model(data)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            model(data)
        torch.cuda.synchronize()
        fwd_elapsed = time.perf_counter() - t1

    # ---- per-stage device times (HIP events on the launch stream), outside the timed region
    rendering.profile_stages(True)
    for _ in range(min(args.steps, 20)):
        out = train_step()
    stages = rendering.profile_stages(False) or {}
    stage_ms = {k: float(np.mean(v)) for k, v in stages.items()}

    if rank == 0:
        with torch.no_grad():
            _, _, meta = rendering.rasterization(model.means, model.quats, model.scales, model.opacities, model.shs,
                                                 data["w2c"][None], data["K"][None], W, H,
                                                 sh_degree=model.active_sh_degree, packed=False,
                                                 backgrounds=model.BACKGROUND[None], absgrad=True)
            _, _, meta_ref = rendering.rasterization(model.means, model.quats, model.scales, model.opacities, model.shs,
                                                     data["w2c"][None], data["K"][None], W, H,
                                                     sh_degree=model.active_sh_degree, packed=False,
                                                     backgrounds=model.BACKGROUND[None], _tile_culling="gsplat")
        n_isects = int(meta["flatten_ids"].shape[0])
        # I of the reference's own lists (gsplat's 3-sigma rectangles): the unit SURVEY.md 8d's
        # byte model counts; the default tight culling walks a render-equivalent subset of it
        n_isects_ref = int(meta_ref["flatten_ids"].shape[0])
        n_vis = int((meta["radii"] > 0).sum().item())
        # algorithmic bytes per launch: SURVEY.md section 8d
        alg = {"gs_blend_fwd": 40 * n_isects_ref + 20 * H * W,
               "gs_blend_bwd": 40 * n_isects_ref + 24 * H * W + 88 * n_isects_ref}
        dom = max(alg, key=lambda k: stage_ms.get(k, 0.0))
        t_ms = stage_ms.get(dom, float("nan"))
        achieved = alg[dom] / (t_ms * 1e-3) / 1e9 if t_ms == t_ms and t_ms > 0 else None
        roofline = {"bound": "hbm", "kernel": {"gs_blend_fwd": "blend_fwd_kernel", "gs_blend_bwd": "blend_bwd_kernel"}[dom],
                    "achieved": None if achieved is None else round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": None if achieved is None else round(achieved / HBM_PEAK_GBS, 5),
                    "traffic": pmc_traffic(roof_kernel := {"gs_blend_fwd": "blend_fwd_kernel", "gs_blend_bwd": "blend_bwd_kernel"}[dom]),
                    "algorithmic_bytes": alg[dom], "avg_launch_ms": None if t_ms != t_ms else round(t_ms, 4)}
        result = {
            "metric": "train iters/sec + forward render fps, 1M Gaussians @ 1080p",
            "value": round(world * args.steps / elapsed, 3), "unit": "iters/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "forward_fps": round(args.steps / fwd_elapsed, 2),
            "forward_ms": round(1e3 * fwd_elapsed / args.steps, 4),
            "config": {"workload": f"{args.gaussians} Gaussians, {W}x{H}, SH degree {sc['sh_degree']}, "
                                   "1 view per GPU per step, full train step (fwd + L1/SSIM + bwd + stats + Adam)",
                       "n_visible": n_vis, "n_isects": n_isects, "n_isects_gsplat_lists": n_isects_ref,
                       "parallelism": f"view-dp{world}",
                       "exchange": exchange if vp is not None or world == 1 else "allreduce"},
            "stage_ms": {k: round(v, 4) for k, v in sorted(stage_ms.items())},
            # host diagnostics: ms/step the host spent blocked on the list-size read-back; if this is ~0
            # the Python side, not the GPU, paces the loop on this box
            "host": {"enqueue_ms_per_step": round(1e3 * t_enqueued / args.steps, 4),
                     "blocked_on_readback_ms_per_step": round(host_wait_ms, 4)},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(sc, args.cpu_sample)
        print(json.dumps(result), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
