"""BASELINE.json configs 2-5 on the GPU, each against the oracle or -- where the oracle cannot finish in
seconds -- through size-independent properties:

  configs[1]  "Lego"  ~300 k Gaussians, 800x800, SH3                 forward + backward vs the C oracle, both list modes
  configs[2]  "Truck" ~2 M Gaussians, 1920x1080, SH3, train loop     oracle fwd+bwd, then 3 train steps against
                                                                     torch.optim.Adam fed with oracle gradients
  configs[3]  8-view batch, 1 view per GPU, gradient exchange        2 ranks (gloo, sharing the one device) at 2 M / 1080p
                                                                     == one process back-propagating both views
  configs[4]  5 M Gaussians, 3840x2160, SH3, densification on        list/key/offset properties, determinism, one
                                                                     densify_and_prune + reset_opacities cycle
Real Lego / Truck captures do not exist in this environment: the configs run at their stated N / HxW with the
seeded generator of SURVEY.md section 8d (tests/scenes.py).  Settings follow /root/reference/configs/tandt_db.yaml:17-44
and nerf_synthetic.yaml:2 (white background for Lego, black for Truck; lambda_ssim 0.2; Adam learning rates).
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import c_oracle as CO
from scenes import config_heavy, config_s2, config_s3, config_s5
from test_gpu_parity import check_backward, check_backward_unmasked, check_forward, run_hip, run_oracle, to_dev

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
MANY_CORES = (os.cpu_count() or 1) >= 32
LRS = (1.6e-4, 5e-3, 1e-3, 2.5e-3, 1.25e-4, 5e-2)   # /root/reference/configs/tandt_db.yaml:24-31 (means_lr_init first)


def dev():
    return torch.device("cuda:0")


# ------------------------------------------------------------------------------------------------ configs[1]
@pytest.mark.parametrize("culling", ["gsplat", "gsplat_eager", "tight"])
def test_config_s2_lego_forward_backward(culling):
    """~300 k Gaussians, 800x800, SH3, white background: full forward + backward parity, lists bit-exact in the
    reference's list mode."""
    sc = config_s2()
    fw = run_oracle(sc)
    hip = run_hip(sc, culling=culling, fw=fw)
    exact = check_forward(hip, fw, lists=culling != "tight")
    if culling != "tight":
        assert int(hip["meta"]["flatten_ids"].numel()) == fw["n_isects"] or not exact
    else:
        assert int(hip["meta"]["flatten_ids"].numel()) < fw["n_isects"]
    check_backward(hip, fw)
    if culling == "tight":   # (the model's list mode) gradient flow through the razor pixels too, against the fp32 oracle
        del hip
        check_backward_unmasked(sc, fw, culling)


# ------------------------------------------------------------------------------------------------ configs[2]
def _model_from_scene(sc, device, sh_degree=3, **kw):
    from easy_gaussian_splatting_amd.model import GaussianModel
    T = torch.from_numpy
    op = np.clip(sc["opacities"], 1e-4, 1 - 1e-4)
    shs = T(sc["shs"])
    return GaussianModel(means=T(sc["means"]), log_scales=torch.log(T(sc["scales"])), quats=T(sc["quats"]),
                         sh_0=shs[:, :1].contiguous(), sh_rest=shs[:, 1:].contiguous(),
                         logit_opacities=T(np.log(op / (1 - op)).astype(np.float32)), sh_degree=sh_degree,
                         white_background=bool(sc["backgrounds"][0, 0] > 0.5), **kw).to(device)


def _target_image(H, W, seed):
    g = torch.Generator().manual_seed(seed)
    low = torch.rand((H // 8, W // 8, 3), generator=g, dtype=torch.float64)
    return torch.nn.functional.interpolate(low.permute(2, 0, 1)[None], size=(H, W), mode="bilinear",
                                           align_corners=False)[0].permute(1, 2, 0).contiguous()


def _oracle_param_grads(p64, sc, gt64, lambda_ssim=0.2, dtype=np.float64):
    """Gradients of the reference's training loss w.r.t. the six raw parameters, entirely on the CPU in fp64:
    C oracle forward -> clamp -> L1 + (1 - SSIM) (plain-torch restatement, autograd) -> C oracle backward ->
    exp / sigmoid / split chain rule (/root/reference/model/gaussian.py:97-107, 351-374, 421-444)."""
    from easy_gaussian_splatting_amd.loss import LossComputer
    W, H = int(sc["width"]), int(sc["height"])
    scales, op = np.exp(p64["log_scales"]), 1.0 / (1.0 + np.exp(-p64["logit_opacities"]))
    shs = np.concatenate([p64["sh_0"], p64["sh_rest"]], axis=1)
    fw = CO.render(p64["means"], p64["quats"], scales, op, shs, sc["viewmats"][:1], sc["Ks"][:1], W, H, sh_degree=3,
                   backgrounds=sc["backgrounds"][:1], dtype=dtype)
    img = torch.from_numpy(fw["render_colors"][0].astype(np.float64)).requires_grad_(True)
    loss = LossComputer(lambda_ssim, fused=False).get_loss_dict(torch.clamp(img, 0.0, 1.0), gt64, torch.zeros((H, W), dtype=torch.float64))
    loss["total"].backward()
    bw = CO.backward(fw, img.grad.numpy()[None].astype(dtype))
    g = {"means": bw["v_means"], "quats": bw["v_quats"], "log_scales": bw["v_scales"] * scales,
         "logit_opacities": bw["v_opacities"] * op * (1.0 - op), "sh_0": bw["v_colors"][:, :1], "sh_rest": bw["v_colors"][:, 1:]}
    return {k: np.ascontiguousarray(v, dtype=np.float64) for k, v in g.items()}, float(loss["total"].detach()), fw, bw


@pytest.mark.skipif(not MANY_CORES, reason="the C oracle needs many host cores to finish 2 M / 1080p in seconds")
def test_config_s3_truck_train_loop_against_oracle_gradients():
    """~2 M Gaussians, 1920x1080, SH3: (1) forward / backward of the rasterizer against the oracle at this size;
    (2) three iterations of the reference's train step (model forward -> fused L1+SSIM -> backward ->
    update_statistics -> fused Adam) against an independent fp64 trajectory: oracle gradients fed to
    torch.optim.Adam.  Adam turns a gradient into a step of ~lr whatever its size, so parameters are compared in
    units of the group's learning rate; the Adam moments (linear / quadratic in the gradient) are compared with the
    backward's 1e-3 relative tolerance."""
    from easy_gaussian_splatting_amd.loss import LossComputer
    from easy_gaussian_splatting_amd.model import build_optimizers
    sc = config_s3()
    W, H = 1920, 1080
    # (1) the seam at this size, reference list mode
    fw = run_oracle(sc)
    hip = run_hip(sc, fw=fw)
    check_forward(hip, fw, outlier_frac=1e-5)
    check_backward(hip, fw)
    del hip
    torch.cuda.empty_cache()
    # (2) the train loop
    d = dev()
    model = _model_from_scene(sc, d)
    opt = build_optimizers(model, *LRS, fused="hip")
    lc = LossComputer(lambda_ssim=0.2, clamp_input=True)
    data = {"w2c": torch.from_numpy(sc["viewmats"][0]).to(d), "K": torch.from_numpy(sc["Ks"][0]).to(d), "width": W, "height": H}
    gt64 = _target_image(H, W, 7)
    gt = gt64.float().to(d)
    mask = torch.zeros((H, W), device=d)
    names = model.param_names
    ref_p = {k: torch.from_numpy(getattr(model, k).detach().cpu().numpy().astype(np.float64)).requires_grad_(True) for k in names}
    ref_opt = torch.optim.Adam([{"params": [ref_p[k]], "lr": lr, "name": k} for k, lr in zip(names, LRS)])
    lr_of = dict(zip(names, LRS))
    for it in range(3):
        model.update_learning_rate(it)
        ref_opt.param_groups[0]["lr"] = model.means_lr_scheduler(it)
        # the fp64 trajectory restarts every step from the fp32 parameters the HIP path is about to use (errors do
        # not compound through the parameters); its Adam moments are its own, built from oracle gradients only
        with torch.no_grad():
            for k in names:
                ref_p[k].copy_(getattr(model, k).detach().cpu().double())
        out = model(data, clamp=False)
        loss = lc.get_loss_dict(out["render_img"], gt, mask)["total"]
        loss.backward()
        hip_grads = {k: getattr(model, k).grad.detach().cpu().numpy() for k in names}
        model.update_statistics(data, out)
        opt.step()
        opt.zero_grad()
        p_now = {k: v.detach().numpy() for k, v in ref_p.items()}
        g, ref_loss, fw_k, bw_k = _oracle_param_grads(p_now, sc, gt64)
        assert abs(float(loss) - ref_loss) <= 2e-5 * max(1.0, abs(ref_loss)), (it, float(loss), ref_loss)
        def errors(ref):
            mx = {k: np.abs(hip_grads[k] - ref[k]).max() / (np.abs(ref[k]).max() + 1e-30) for k in names}
            l2 = {k: np.linalg.norm((hip_grads[k] - ref[k]).ravel()) / (np.linalg.norm(ref[k].ravel()) + 1e-30) for k in names}
            # rows (Gaussians) holding an element off by more than the 1e-3 tolerance
            bad = {k: float(np.mean((np.abs(hip_grads[k] - ref[k]).reshape(ref[k].shape[0], -1).max(1)) > 1e-3 * np.abs(ref[k]).max())) for k in names}
            return mx, l2, bad

        rel, rel_l2, rel_bad = errors(g)
        if max(rel.values()) > 1e-3:
            # a contributor flipped at a blend threshold under fp32 arithmetic moves one Gaussian's gradient by more
            # than the tolerance: the fp32 build of the oracle arbitrates, for every tensor (as in check_backward)
            print(f"[parity] step {it}: fp64 arbiter failed ({ {k: float('%.2e' % v) for k, v in rel.items()} }); fp32 oracle arbitrates")
            g, _, fw_k, bw_k = _oracle_param_grads(p_now, sc, gt64, dtype=np.float32)
            rel, rel_l2, rel_bad = errors(g)
        for k in names:
            # At 2 M Gaussians two fp32 implementations do not flip the same threshold contributors either: one sharp
            # splat losing or gaining a single alpha = 1/255 pixel moves ITS gradient by a few 1e-3 of the tensor's
            # maximum.  Such flips are isolated and bounded: at most 2e-5 of the Gaussians (40 of 2 M) may be off by
            # more than the 1e-3 tolerance, none by more than 1e-2 (a contributor's weight is at most ~1/255 of the
            # pixel), and the L2 error over the whole tensor must be below 5e-4.  (1e-3 max-norm holds on every
            # fixture, at 1 M, and at 2 M in part 1 with a fixed upstream gradient.)
            assert rel_bad[k] <= 2e-5, (it, k, rel_bad[k], rel[k])
            assert rel[k] <= 1e-2, (it, k, rel[k])
            assert rel_l2[k] <= 5e-4, (it, k, rel_l2[k])
            ref_p[k].grad = torch.from_numpy(np.ascontiguousarray(g[k]))
        ref_opt.step()
        for k in names:
            p_hip = getattr(model, k).detach().cpu().numpy().astype(np.float64)
            diff = np.abs(p_hip - ref_p[k].detach().numpy())
            frac = float(np.mean(diff > 0.05 * lr_of[k]))
            assert frac < 2e-3, (it, k, frac, float(diff.max()))
            assert diff.max() <= 2.05 * lr_of[k] + 1e-6, (it, k, float(diff.max()))
            m_hip, v_hip = (x.detach().cpu().numpy() for x in opt.moments_of(getattr(model, k)))
            st = ref_opt.state[ref_p[k]]
            m_ref, v_ref = st["exp_avg"].numpy(), st["exp_avg_sq"].numpy()
            # (same rule as the gradients they are linear / quadratic in: isolated flips, bounded)
            for got, ref, tol in ((m_hip, m_ref, 2e-3), (v_hip, v_ref, 4e-3)):
                d = np.abs(got - ref).reshape(ref.shape[0], -1).max(1) / np.abs(ref).max()
                assert float(np.mean(d > tol)) <= 2e-5 and d.max() <= 2e-2, (it, k, float(np.mean(d > tol)), float(d.max()))
        if it == 0:   # statistics of the first step against the oracle's absgrad / radii
            vis = fw_k["radii"][0] > 0
            exp_g = np.where(vis, np.linalg.norm(bw_k["v_means2d_abs"][0], axis=-1) * max(H, W), 0)
            assert np.abs(model.grad_norm_accum.cpu().numpy() - exp_g).max() <= 2e-3 * exp_g.max()
            assert float(np.mean((model.collecting_counts.cpu().numpy() > 0) != vis)) <= 1e-5


# ------------------------------------------------------------------------------------------------ configs[3]
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


N_S4 = 2_000_000


def _make_s4(device):
    from easy_gaussian_splatting_amd.model import build_optimizers
    sc = config_s3(n=N_S4, n_views=8)   # the 8-view batch; ranks take views 0 and 3
    model = _model_from_scene(sc, device)
    opt = build_optimizers(model, *LRS, fused="hip")
    views = (0, 3)
    datas = [{"w2c": torch.from_numpy(sc["viewmats"][v]).to(device), "K": torch.from_numpy(sc["Ks"][v]).to(device),
              "width": 1920, "height": 1080} for v in views]
    targets = [_target_image(1080, 1920, 20 + v).float().to(device) for v in views]
    return model, opt, datas, targets


def _snapshot(model):
    out = {k: getattr(model, k).detach().cpu().numpy() for k in model.param_names}
    out.update(gn=model.grad_norm_accum.cpu().numpy(), cnt=model.collecting_counts.cpu().numpy(), rad=model.max_radii.cpu().numpy())
    return out


def _s4_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from easy_gaussian_splatting_amd.distributed import ViewParallelStep
    from easy_gaussian_splatting_amd.loss import LossComputer
    d = torch.device("cuda:0")
    torch.cuda.set_device(d)
    model, opt, datas, targets = _make_s4(d)
    vp = ViewParallelStep(model, opt)
    lc = LossComputer(0.2, clamp_input=True)
    for it in range(2):
        vp.begin_step(datas[rank])
        out = model(datas[rank], clamp=False)
        vp.after_forward(datas[rank], out)
        lc.get_loss_dict(out["render_img"], targets[rank])["total"].backward()
        vp.step(datas[rank], out)
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), **_snapshot(model))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_config_s4_sharded_views_equal_single_process(tmp_path):
    """Two ranks, one view each, at the Truck size (2 M Gaussians, 1920x1080): replicas bitwise identical, and the
    same update as one process that back-propagates both views and averages the gradients.  (`gloo` over the one
    device of this box; RCCL refuses two ranks on one GPU -- the exchange code is backend-independent.)"""
    from easy_gaussian_splatting_amd.loss import LossComputer
    mp.spawn(_s4_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (np.load(os.path.join(tmp_path, f"r{r}.npz")) for r in range(2))
    for k in r0.files:
        np.testing.assert_array_equal(r0[k], r1[k], err_msg=f"replicas diverged in {k}")
    d = dev()
    model, opt, datas, targets = _make_s4(d)
    lc = LossComputer(0.2, clamp_input=True)
    for _ in range(2):
        acc, stats = None, []
        for v in range(2):
            out = model(datas[v], clamp=False)
            lc.get_loss_dict(out["render_img"], targets[v])["total"].backward()
            radii = out["batch_radii"][0]
            vis = radii > 0
            stats.append((torch.where(vis, out["batch_xys"].absgrad[0].norm(dim=-1) * 1920.0, 0.0), vis.float(),
                          torch.where(vis, radii.float() / 1920.0, 0.0)))
            gs = [getattr(model, k).grad.clone() for k in model.param_names]
            acc = gs if acc is None else [a + g for a, g in zip(acc, gs)]
            opt.zero_grad()
        for k, g in zip(model.param_names, acc):
            getattr(model, k).grad = g / 2
        opt.step()
        opt.zero_grad()
        model.grad_norm_accum += stats[0][0] + stats[1][0]
        model.collecting_counts += stats[0][1] + stats[1][1]
        model.max_radii = torch.maximum(model.max_radii, torch.maximum(stats[0][2], stats[1][2]))
    ref = _snapshot(model)
    lr_of = dict(zip(model.param_names, LRS))
    for k in model.param_names:
        diff = np.abs(r0[k] - ref[k])
        assert np.mean(diff > 0.05 * lr_of[k]) < 2e-3, (k, float(diff.max()), float(np.mean(diff > 0.05 * lr_of[k])))
    np.testing.assert_allclose(r0["gn"], ref["gn"], rtol=1e-4, atol=1e-6 * float(ref["gn"].max()))
    np.testing.assert_array_equal(r0["cnt"], ref["cnt"])
    np.testing.assert_array_equal(r0["rad"], ref["rad"])


# ---- configs[3] beyond two ranks: FOUR ranks, one view each (VERDICT r3 item 3b asked for the stated world size, 8; a GPU box
# admits six processes on its card and the test runner is one of them, so the 8-rank exchange runs over gloo on the CPU --
# tests/test_distributed.py, world 8 -- the R = 8 kernels in one process -- tests/test_gpu_view_parallel.py -- and this test
# takes the widest world the card admits with room to spare)
N_S4X8, W_S4X8, R_S4X8 = 200_000, 800, 4


def _make_s4x8(device):
    from easy_gaussian_splatting_amd.model import build_optimizers
    from scenes import make_scene
    sc = make_scene(N_S4X8, W_S4X8, W_S4X8, sh_degree=3, n_views=R_S4X8, seed=42, extent=(3.0, 3.0, 3.0), scale_range=(0.003, 0.03),
                    dist=8.0, white_bg=False)
    model = _model_from_scene(sc, device)
    opt = build_optimizers(model, *LRS, fused="hip")
    datas = [{"w2c": torch.from_numpy(sc["viewmats"][v]).to(device), "K": torch.from_numpy(sc["Ks"][v]).to(device),
              "width": W_S4X8, "height": W_S4X8} for v in range(R_S4X8)]
    targets = [_target_image(W_S4X8, W_S4X8, 40 + v).float().to(device) for v in range(R_S4X8)]
    return model, opt, datas, targets


def _s4x8_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from easy_gaussian_splatting_amd.distributed import ViewParallelStep
    from easy_gaussian_splatting_amd.loss import LossComputer
    d = torch.device("cuda:0")
    torch.cuda.set_device(d)
    model, opt, datas, targets = _make_s4x8(d)
    vp = ViewParallelStep(model, opt)
    lc = LossComputer(0.2, clamp_input=True)
    for it in range(2):
        vp.begin_step(datas[rank])
        out = model(datas[rank], clamp=False)
        vp.after_forward(datas[rank], out)
        lc.get_loss_dict(out["render_img"], targets[rank])["total"].backward()
        assert model.sh_0.grad is None and model.sh_rest.grad is None   # the factorised exchange: no dense SH gradient exists
        vp.step(datas[rank], out)
    torch.cuda.synchronize()
    if rank in (0, 2, R_S4X8 - 1):
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), **_snapshot(model))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_config_s4_four_ranks_equal_single_process(tmp_path):
    """A sharded view batch beyond two ranks: FOUR ranks (gloo, sharing the one device of this box; RCCL refuses two ranks
    on one GPU, the exchange code is backend-independent; five processes on the card with the test runner, the box admits
    six), one view each, ~200 k Gaussians at 800x800.  Exercises what two ranks cannot: the R-way all_gather_into_tensor
    layout of the per-view colour gradients, gs_sh_grad_views with R > 2, the rank-order sum, `1/R` folded into Adam.
    Replicas bitwise identical; the update equals ONE process that back-propagates all four views and averages."""
    from easy_gaussian_splatting_amd.loss import LossComputer
    R = R_S4X8
    mp.spawn(_s4x8_worker, args=(R, _free_port(), str(tmp_path)), nprocs=R, join=True)
    r0, r5, r7 = (np.load(os.path.join(tmp_path, f"r{r}.npz")) for r in (0, 2, R - 1))
    for k in r0.files:
        np.testing.assert_array_equal(r0[k], r5[k], err_msg=f"replicas 0 / 2 diverged in {k}")
        np.testing.assert_array_equal(r0[k], r7[k], err_msg=f"replicas 0 / {R - 1} diverged in {k}")
    d = dev()
    model, opt, datas, targets = _make_s4x8(d)
    lc = LossComputer(0.2, clamp_input=True)
    hw = float(W_S4X8)
    for _ in range(2):
        acc, gn, cnt, rad = None, 0.0, 0.0, None
        for v in range(R):
            out = model(datas[v], clamp=False)
            lc.get_loss_dict(out["render_img"], targets[v])["total"].backward()
            radii = out["batch_radii"][0]
            vis = radii > 0
            gn = gn + torch.where(vis, out["batch_xys"].absgrad[0].norm(dim=-1) * hw, 0.0)
            cnt = cnt + vis.float()
            r = torch.where(vis, radii.float() / hw, 0.0)
            rad = r if rad is None else torch.maximum(rad, r)
            gs = [getattr(model, k).grad.clone() for k in model.param_names]
            acc = gs if acc is None else [a + g for a, g in zip(acc, gs)]
            opt.zero_grad()
        for k, g in zip(model.param_names, acc):
            getattr(model, k).grad = g / R
        opt.step()
        opt.zero_grad()
        model.grad_norm_accum += gn
        model.collecting_counts += cnt
        model.max_radii = torch.maximum(model.max_radii, rad)
    ref = _snapshot(model)
    lr_of = dict(zip(model.param_names, LRS))
    for k in model.param_names:
        diff = np.abs(r0[k] - ref[k])
        assert np.mean(diff > 0.05 * lr_of[k]) < 2e-3, (k, float(diff.max()), float(np.mean(diff > 0.05 * lr_of[k])))
    np.testing.assert_allclose(r0["gn"], ref["gn"], rtol=1e-4, atol=1e-6 * float(ref["gn"].max()))
    np.testing.assert_array_equal(r0["cnt"], ref["cnt"])
    np.testing.assert_array_equal(r0["rad"], ref["rad"])


# ------------------------------------------------------------------------------------------------ configs[4]
@pytest.mark.skipif(not MANY_CORES, reason="the C oracle needs many host cores to finish 5 M / 4K in a minute")
@pytest.mark.timeout(900)
def test_config_s5_4k_against_oracle():
    """5 M Gaussians, 3840x2160, SH3 against the ORACLE itself (round 3 checked this size through
    properties only): the fp64 C oracle for the integer outputs, the lists and -- with the 1e-5 outlier allowance the fp64
    reference needs at hundreds of contributors per pixel -- the image; the fp32 build of the same oracle strictly (every
    non-razor pixel within 1e-4); all five input gradients and absgrad within 1e-3 of the fp64 oracle's, plus the relative-L2
    and per-Gaussian criteria of check_backward; then the model's own list mode ("tight") on the same upstream gradient."""
    import parity_log
    import time
    sc = config_s5()
    t0 = time.time()
    fw = run_oracle(sc)
    fw32 = run_oracle(sc, dtype=np.float32)
    t_or = time.time() - t0
    hip = run_hip(sc, fw=fw, max_flip_tile_frac=0.05)
    # (razor pixels grow with the contributors per pixel: 0.3 % at S2, 1.5 % at 1 M, 1.7 % at 2 M, 2.2 % here; the suite's
    #  hard limit is 5 %)
    exact = check_forward(hip, fw, outlier_frac=1e-5, max_razor_frac=0.03, max_flip_tile_frac=0.05)
    # (the fp32 build's own geometry is tens of ulps off the fp64 truth: 80 of its 5 M radii / rectangles differ from the
    #  path's, and the tiles those touch are exempt; against the fp64 oracle above: 0 flips, lists identical)
    check_forward(hip, fw32, geom_slack=1e3, max_razor_frac=0.03, max_flip_tile_frac=0.05)
    assert int(hip["meta"]["flatten_ids"].numel()) == fw["n_isects"] or not exact
    t0 = time.time()
    check_backward(hip, fw)
    parity_log.record(oracle_forward_s=round(t_or, 1), oracle_backward_s=round(time.time() - t0, 1), n_isects_gsplat=int(fw["n_isects"]))
    hip_t = run_hip(sc, culling="tight", upstream=(hip["vc"], hip["va"]))
    assert hip_t["meta"]["flatten_ids"].numel() < hip["meta"]["flatten_ids"].numel()
    # the short lists render the SAME image bit for bit (so every forward check above holds for them) ...
    assert torch.equal(hip_t["img"], hip["img"]) and torch.equal(hip_t["alpha"], hip["alpha"])
    assert torch.equal(hip_t["meta"]["radii"], hip["meta"]["radii"]) and torch.equal(hip_t["meta"]["means2d"], hip["meta"]["means2d"])
    for a, b in zip(hip_t["grads"], hip["grads"]):   # ... and the same gradients to rounding
        assert float((a - b).abs().max()) <= 1e-4 * float(b.abs().max())


# ------------------------------------------------------------------------------------------------ the metric's N, realistic footprint
@pytest.mark.skipif(not MANY_CORES, reason="the C oracle needs many host cores to finish 30-60 M intersections in a minute")
@pytest.mark.timeout(1500)
@pytest.mark.parametrize("n", [1_000_000, 2_000_000])
def test_heavy_footprint_against_oracle(n):
    """VERDICT r4 missing #2: 1 M and 2 M Gaussians at 1080p whose gsplat lists hold ~29 entries per Gaussian
    (`synthetic.config_heavy`: heavy-tailed scales, what a trained Truck looks like to the tile lists) -- N >= 1 M together with
    I / N >= 20, where binning, sort and the 192 B / intersection row arena dominate.  Against the fp64 C oracle: integer outputs
    and the lists, the image (1e-5 outlier allowance at hundreds of contributors per pixel); against its fp32 build every
    non-razor pixel within 1e-4; all five input gradients + absgrad under check_backward's three criteria; then the model's own
    list mode ("tight") on the same upstream gradient: same image bit for bit, gradients to rounding."""
    import parity_log
    import time
    from easy_gaussian_splatting_amd import rendering
    sc = config_heavy(n=n)
    t0 = time.time()
    fw = run_oracle(sc)
    fw32 = run_oracle(sc, dtype=np.float32)
    t_or = time.time() - t0
    assert fw["n_isects"] >= 20 * n, (fw["n_isects"], n)
    hip = run_hip(sc, fw=fw, max_flip_tile_frac=0.05)
    binning_first = rendering.last_binning()     # (the first call of a shape has no footprint history: per-tile pipeline)
    exact = check_forward(hip, fw, outlier_frac=1e-5, max_razor_frac=0.05, max_flip_tile_frac=0.05)
    check_forward(hip, fw32, geom_slack=1e3, max_razor_frac=0.05, max_flip_tile_frac=0.05)
    assert int(hip["meta"]["flatten_ids"].numel()) == fw["n_isects"] or not exact
    t0 = time.time()
    check_backward(hip, fw)
    parity_log.record(oracle_forward_s=round(t_or, 1), oracle_backward_s=round(time.time() - t0, 1), n_isects_gsplat=int(fw["n_isects"]),
                      isects_per_gaussian=round(fw["n_isects"] / n, 1))
    hip_t = run_hip(sc, culling="tight", upstream=(hip["vc"], hip["va"]))
    # (footprints of this size take the two-level binning as soon as a call of the shape has reported them)
    assert rendering.binning_choice(fw["n_isects"] / n, 120 * 68) == "bins"
    parity_log.record(n_isects_tight=int(hip_t["meta"]["flatten_ids"].numel()), binning={"first_call": binning_first, "later": rendering.last_binning()})
    assert hip_t["meta"]["flatten_ids"].numel() < hip["meta"]["flatten_ids"].numel()
    assert torch.equal(hip_t["img"], hip["img"]) and torch.equal(hip_t["alpha"], hip["alpha"])
    assert torch.equal(hip_t["meta"]["radii"], hip["meta"]["radii"]) and torch.equal(hip_t["meta"]["means2d"], hip["meta"]["means2d"])
    for a, b in zip(hip_t["grads"], hip["grads"]):
        assert float((a - b).abs().max()) <= 1e-4 * float(b.abs().max())


@pytest.mark.timeout(900)
def test_config_s5_4k_properties_and_refine_cycle():
    """5 M Gaussians, 3840x2160, SH3 with densification on: too large for the CPU oracle in seconds, so the list
    contract is checked through properties (global key order, tie order, I == sum of tile counts, offsets monotone
    and int32-safe, keys carry the depth bits, determinism), then one train step + densify_and_prune +
    reset_opacities + another step at that N (/root/reference/model/gaussian.py:130-146, 259-349)."""
    from easy_gaussian_splatting_amd.loss import LossComputer
    from easy_gaussian_splatting_amd.model import build_optimizers
    from easy_gaussian_splatting_amd.rendering import rasterization
    sc = config_s5()
    W, H = 3840, 2160
    t = to_dev(sc)
    args = [t[k] for k in ("means", "quats", "scales", "opacities", "shs")]
    with torch.no_grad():
        img, alpha, meta = rasterization(*args, t["viewmats"], t["Ks"], W, H, sh_degree=3, packed=False, backgrounds=t["backgrounds"])
        assert torch.isfinite(img).all() and float(alpha.min()) >= 0.0 and float(alpha.max()) <= 1.0
        I = meta["flatten_ids"].numel()
        assert I == int(meta["tiles_per_gauss"].sum()) and 0 < I < 2 ** 31
        offs = meta["isect_offsets"].reshape(-1).long()
        assert offs.numel() == 240 * 135 and bool((offs[1:] >= offs[:-1]).all()) and int(offs[0]) == 0 and int(offs[-1]) <= I
        keys = meta["isect_ids"]
        assert bool((keys[1:] >= keys[:-1]).all()), "(tile | depth) keys must be globally non-decreasing"
        fid = meta["flatten_ids"].long()
        same = keys[1:] == keys[:-1]
        assert bool((fid[1:][same] > fid[:-1][same]).all()), "ties ordered by flatten index"
        assert bool((meta["radii"].reshape(-1)[fid] > 0).all())
        dbits = meta["depths"].reshape(-1)[fid].view(torch.int32).long()
        assert bool(((keys & 0xFFFFFFFF) == dbits).all())
        tile_of_key = (keys >> 32)
        starts = torch.searchsorted(tile_of_key, torch.arange(240 * 135, device=keys.device))
        assert torch.equal(starts, offs), "isect_offsets must be the first position of every tile's run"
        img2, alpha2, meta2 = rasterization(*args, t["viewmats"], t["Ks"], W, H, sh_degree=3, packed=False, backgrounds=t["backgrounds"])
        assert torch.equal(img, img2) and torch.equal(meta["flatten_ids"], meta2["flatten_ids"]), "deterministic"
        img_t, _, meta_t = rasterization(*args, t["viewmats"], t["Ks"], W, H, sh_degree=3, packed=False, backgrounds=t["backgrounds"],
                                         _tile_culling="tight")
        assert torch.equal(img, img_t) and meta_t["flatten_ids"].numel() < I
    del img, img2, img_t, alpha, alpha2, meta, meta2, meta_t, keys, fid, dbits, tile_of_key, same
    torch.cuda.empty_cache()
    # densification on
    d = dev()
    model = _model_from_scene(sc, d, densify_grad_thresh=2e-6)   # (threshold scaled so this synthetic scene splits and clones)
    opt = build_optimizers(model, *LRS, fused="hip")
    lc = LossComputer(0.2, clamp_input=True)
    data = {"w2c": t["viewmats"][0], "K": t["Ks"][0], "width": W, "height": H}
    gt = _target_image(H, W, 5).float().to(d)

    def step():
        out = model(data, clamp=False)
        loss = lc.get_loss_dict(out["render_img"], gt)["total"]
        loss.backward()
        model.update_statistics(data, out)
        opt.step()
        opt.zero_grad()
        return float(loss)

    l0 = step()
    n0 = model.nbr_gaussians
    # this synthetic scene's gradient scale is arbitrary: put the threshold at the 90th percentile of the visible
    # Gaussians' average so the cycle really splits and clones
    avg = model.grad_norm_accum / (model.collecting_counts + 1e-8)
    model.DENSIFY_GRAD_THRESH = float(torch.quantile(avg[model.collecting_counts > 0][:1_000_000], 0.9))
    info = model.densify_and_prune(generator=torch.Generator(device=d).manual_seed(1))
    n1 = model.nbr_gaussians
    ns, nc = info["train/densify"]["split"], info["train/densify"]["clone"]
    pruned_other = sum(info["train/prune"].values())
    assert ns > 0 and nc > 0 and n1 == info["train/nbr_gaussians"] and info["n_before"] == n0
    grown = n0 - ns + ns * model.NUM_SPLITS + nc      # split parents always go
    assert grown - pruned_other <= n1 <= grown
    for buf in (model.grad_norm_accum, model.collecting_counts, model.max_radii):
        assert buf.shape == (n1,) and float(buf.abs().max()) == 0.0
    for k in model.param_names:
        m, v = opt.moments_of(getattr(model, k))
        assert getattr(model, k).shape[0] == n1 and m.shape == getattr(model, k).shape and v.shape == m.shape
    model.reset_opacities()
    assert float(model.opacities.max()) <= 2 * model.MIN_OPACITY * (1 + 1e-5)
    mo, vo = opt.moments_of(model.logit_opacities)
    assert float(mo.abs().max()) == 0.0 and float(vo.abs().max()) == 0.0
    l1 = step()
    assert np.isfinite(l0) and np.isfinite(l1)
