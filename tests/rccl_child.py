"""Child process of tests/test_gpu_rccl.py (and of nothing else): one rank of a world_size-1 job on cuda:0.

    python tests/rccl_child.py <mode> <out.pt>

mode "nccl" / "gloo": initialise that backend with ONE rank and drive the FULL exchange path of
`distributed.ViewParallelStep(force_exchange=True)` -- camera all-gather, radii MAX all-reduce, the colour-gradient
all-gather issued asynchronously from inside `backward()`, the geometry SUM all-reduce, the split Adam step -- and, on a
second model, the plain scheme (`all_reduce_param_grads(force=True)` / `GradBucket.all_reduce_mean(force=True)`).
mode "plain": no process group, the reference's one-rank step.  Writes every parameter, Adam moment and statistic after
three steps to <out.pt>.  ("nccl" is RCCL on ROCm.  A fresh process per mode: a process that has initialised the GPU is
never re-executed.)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    mode, out_path = sys.argv[1], sys.argv[2]
    import numpy as np
    import torch
    import torch.distributed as dist

    from easy_gaussian_splatting_amd.distributed import GradBucket, ViewParallelStep, all_reduce_param_grads
    from easy_gaussian_splatting_amd.loss import LossComputer
    from easy_gaussian_splatting_amd.model import GaussianModel, build_optimizers
    from scenes import make_scene

    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    if mode in ("nccl", "gloo"):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        if mode == "nccl":
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=0, world_size=1)
    N, W, H = 20000, 320, 208
    sc = make_scene(N, W, H, sh_degree=3, n_views=3, seed=3, scale_range=(0.01, 0.08), dist=4.0)
    T = torch.from_numpy
    op = np.clip(sc["opacities"], 1e-3, 1 - 1e-3)
    shs = T(sc["shs"])
    lrs = (1.6e-4, 5e-3, 1e-3, 2.5e-3, 1.25e-4, 5e-2)

    def make(fused):
        m = GaussianModel(means=T(sc["means"]), log_scales=torch.log(T(sc["scales"])), quats=T(sc["quats"]),
                          sh_0=shs[:, :1].contiguous(), sh_rest=shs[:, 1:].contiguous(),
                          logit_opacities=T(np.log(op / (1 - op)).astype(np.float32)), sh_degree=3, white_background=True).to(dev)
        return m, build_optimizers(m, *lrs, fused=fused)

    datas = [{"w2c": T(sc["viewmats"][v]).to(dev), "K": T(sc["Ks"][v]).to(dev), "width": W, "height": H} for v in range(3)]
    g = torch.Generator().manual_seed(11)
    gts = [torch.rand((H, W, 3), generator=g).to(dev) for _ in range(3)]
    lc = LossComputer(0.2, clamp_input=True)
    result = {"mode": mode}

    # ---- (1) the factorised exchange of ViewParallelStep
    m, opt = make("hip")
    vp = ViewParallelStep(m, opt, force_exchange=(mode != "plain"))
    for it in range(3):
        d, gt = datas[it], gts[it]
        vp.begin_step(d)
        out = m(d, clamp=False)
        vp.after_forward(d, out)
        lc.get_loss_dict(out["render_img"], gt)["total"].backward()
        vp.step(d, out)
    torch.cuda.synchronize()
    result["vp"] = {k: getattr(m, k).detach().cpu() for k in m.param_names}
    for k in m.param_names:
        ea, es = opt.moments_of(getattr(m, k))
        result["vp"][k + ".exp_avg"], result["vp"][k + ".exp_avg_sq"] = ea.cpu().clone(), es.cpu().clone()
    for k in ("max_radii", "grad_norm_accum", "collecting_counts"):
        result["vp"][k] = getattr(m, k).cpu().clone()
    result["vp_collectives"] = vp.collectives
    result["vp_exchange"] = vp.exchange

    # ---- (2) the plain scheme: per-parameter async all-reduce (FusedAdam) and the flat GradBucket (torch Adam)
    m2, opt2 = make("hip")
    for it in range(2):
        out = m2(datas[it], clamp=False)
        lc.get_loss_dict(out["render_img"], gts[it])["total"].backward()
        m2.update_statistics(datas[it], out)
        all_reduce_param_grads(m2.parameters(), force=True)
        opt2.step(); opt2.zero_grad()
    m3, opt3 = make(True)
    bucket = GradBucket(m3.parameters())
    for it in range(2):
        out = m3(datas[it], clamp=False)
        lc.get_loss_dict(out["render_img"], gts[it])["total"].backward()
        bucket.all_reduce_mean(force=True)
        opt3.step(); bucket.zero_()
    torch.cuda.synchronize()
    result["plain_fused"] = {k: getattr(m2, k).detach().cpu() for k in m2.param_names}
    result["plain_bucket"] = {k: getattr(m3, k).detach().cpu() for k in m3.param_names}
    if mode in ("nccl", "gloo"):
        result["backend"] = dist.get_backend()
        result["world"] = dist.get_world_size()
        dist.barrier()
        dist.destroy_process_group()
    torch.save(result, out_path)
    print("rccl_child ok", mode, flush=True)


if __name__ == "__main__":
    main()
