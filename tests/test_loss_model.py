"""Host-side mirror pieces that run on CPU: SSIM restatement, activations, optimizer groups."""
import numpy as np
import torch

from easy_gaussian_splatting_amd.loss import LossComputer, ssim
from easy_gaussian_splatting_amd.model import GaussianModel, build_optimizers


def _ssim_naive(a, b):
    """Direct (non-separable) 11x11 evaluation for one channel pair, interior pixels only."""
    import torch.nn.functional as F
    x = torch.arange(11, dtype=torch.float64) - 5
    g = torch.exp(-(x / 1.5) ** 2 / 2); g = g / g.sum()
    w = (g[:, None] * g[None, :])[None, None]
    pa, pb = (F.pad(t[None, None], (5, 5, 5, 5), mode="reflect") for t in (a, b))
    mu_a, mu_b = F.conv2d(pa, w), F.conv2d(pb, w)
    saa = F.conv2d(pa * pa, w) - mu_a ** 2; sbb = F.conv2d(pb * pb, w) - mu_b ** 2; sab = F.conv2d(pa * pb, w) - mu_a * mu_b
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu_a * mu_b + c1) * (2 * sab + c2)) / ((mu_a ** 2 + mu_b ** 2 + c1) * (saa + sbb + c2))
    return m[..., 5:-5, 5:-5].mean()


def test_ssim_matches_direct_evaluation_and_bounds():
    g = torch.Generator().manual_seed(0)
    a, b = torch.rand(1, 3, 40, 52, generator=g, dtype=torch.float64), torch.rand(1, 3, 40, 52, generator=g, dtype=torch.float64)
    assert abs(ssim(a, a).item() - 1.0) < 1e-12
    ref = torch.stack([_ssim_naive(a[0, c], b[0, c]) for c in range(3)]).mean()
    assert abs(ssim(a, b).item() - ref.item()) < 1e-12
    assert ssim(a, b).item() < 0.2


def test_loss_composition_and_mask():
    g = torch.Generator().manual_seed(1)
    r, t = torch.rand(32, 48, 3, generator=g), torch.rand(32, 48, 3, generator=g)
    lc = LossComputer(lambda_ssim=0.2)
    d = lc.get_loss_dict(r, t, torch.zeros(32, 48))
    assert abs(d["total"].item() - (0.8 * d["l1"].item() + 0.2 * d["ssim"].item())) < 1e-6
    full = lc.get_loss_dict(r, t, torch.ones(32, 48))
    assert full["l1"].item() == 0.0 and abs(full["ssim"].item()) < 1e-6


def test_model_activations_and_optimizer_groups():
    N = 10
    m = GaussianModel(means=torch.zeros(N, 3), log_scales=torch.full((N, 3), np.log(0.1)), quats=torch.ones(N, 4),
                      sh_0=torch.zeros(N, 1, 3), sh_rest=torch.zeros(N, 15, 3), logit_opacities=torch.zeros(N),
                      sh_degree=3, sh_degree_interval=2000, white_background=True)
    assert torch.allclose(m.scales, torch.full((N, 3), 0.1)) and torch.allclose(m.opacities, torch.full((N,), 0.5))
    assert m.shs.shape == (N, 16, 3) and m.active_sh_degree == 0 and m.BACKGROUND.tolist() == [1.0, 1.0, 1.0]
    for _ in range(5):
        m.up_sh_degree()
    assert m.active_sh_degree == 3
    opt = build_optimizers(m, 1.6e-4, 5e-3, 1e-3, 2.5e-3, 1.25e-4, 5e-2)
    assert [g["name"] for g in opt.param_groups] == m.param_names
    assert [g["lr"] for g in opt.param_groups] == [1.6e-4, 5e-3, 1e-3, 2.5e-3, 1.25e-4, 5e-2]


def test_means_lr_schedule_matches_reference_fixture():
    """`update_learning_rate` / `LR_Scheduler` against values captured from the reference's own
    model/utils.py:19-28 (tests/golden/make_golden.py: `lrs`)."""
    import os
    from easy_gaussian_splatting_amd.model import LR_Scheduler
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_model_utils.npz"))
    sched = LR_Scheduler(float(z["lr_init"]), float(z["lr_final"]), int(z["lr_max_steps"]))
    got = np.array([sched(int(s)) for s in z["lr_steps"]])
    assert np.allclose(got, z["lrs"], rtol=1e-12, atol=0)
    N = 6
    m = GaussianModel(means=torch.zeros(N, 3), log_scales=torch.zeros(N, 3), quats=torch.ones(N, 4), sh_0=torch.zeros(N, 1, 3),
                      sh_rest=torch.zeros(N, 15, 3), logit_opacities=torch.zeros(N), sh_degree=3,
                      means_lr_init=float(z["lr_init"]), means_lr_final=float(z["lr_final"]), means_lr_schedule_max_steps=int(z["lr_max_steps"]))
    try:
        m.update_learning_rate(10)
        raise AssertionError("must refuse without an optimizer")
    except RuntimeError:
        pass
    opt = build_optimizers(m, 1.6e-4, 5e-3, 1e-3, 2.5e-3, 1.25e-4, 5e-2)
    for s, lr in zip(z["lr_steps"], z["lrs"]):
        m.update_learning_rate(int(s))
        lrs = {g["name"]: g["lr"] for g in opt.param_groups}
        assert abs(lrs["means"] - lr) <= 1e-12 * lr
        assert lrs["log_scales"] == 5e-3 and lrs["logit_opacities"] == 5e-2   # only the means group is scheduled


def test_scale_ratio_regulariser_and_total():
    """mean(max(s_max / s_min, R) - R), added as lambda_scale * reg to the total; gradient reaches log_scales
    (/root/reference/model/gaussian.py:376-386, 437-440)."""
    g = torch.Generator().manual_seed(3)
    N = 50
    ls = torch.randn(N, 3, generator=g) * 1.5
    m = GaussianModel(means=torch.zeros(N, 3), log_scales=ls, quats=torch.ones(N, 4), sh_0=torch.zeros(N, 1, 3),
                      sh_rest=torch.zeros(N, 0, 3), logit_opacities=torch.zeros(N), sh_degree=0,
                      use_scale_regularization=True, max_scale_ratio=4.0)
    reg = m.get_regularization_dict()["scale_reg"]
    s = np.exp(ls.double().numpy())
    ratio = s.max(1) / s.min(1)
    ref = np.mean(np.maximum(ratio, 4.0) - 4.0)
    assert ref > 0 and abs(reg.item() - ref) < 1e-5 * ref
    r, t = torch.rand(32, 48, 3, generator=g), torch.rand(32, 48, 3, generator=g)
    plain = LossComputer(lambda_ssim=0.2).get_loss_dict(r, t, torch.zeros(32, 48))
    d = LossComputer(lambda_ssim=0.2, model=m, lambda_scale=0.5).get_loss_dict(r, t, torch.zeros(32, 48))
    assert abs(d["total"].item() - (plain["total"].item() + 0.5 * reg.item())) < 1e-6 and "scale_reg" in d
    d["total"].backward()
    assert m.log_scales.grad is not None and float(m.log_scales.grad.abs().max()) > 0
    # d reg / d log_s = ratio/N * (+1 at argmax, -1 at argmin) where ratio > R
    exp_g = np.zeros((N, 3))
    for i in range(N):
        if ratio[i] > 4.0:
            exp_g[i, s[i].argmax()] += 0.5 * ratio[i] / N
            exp_g[i, s[i].argmin()] -= 0.5 * ratio[i] / N
    assert np.abs(m.log_scales.grad.numpy() - exp_g).max() < 1e-5 * np.abs(exp_g).max()
    m.USE_SCALE_REGULARIZATION = False
    assert m.get_regularization_dict() == {}


def test_lease_pack_hook_does_not_tie_a_node_to_its_own_outputs():
    """ADVICE r4 (CPU form of tests/test_gpu_workspace.py::test_forward_without_backward_...): the pack hook of
    rendering.rasterization attaches the workspace lease to the node's saved INPUTS and saves its OUTPUTS detached."""
    import gc
    import weakref
    import torch
    from easy_gaussian_splatting_amd.rendering import _Holder, _lease_pack_hook

    class Lease:
        pass

    class Double(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, holder):
            y = x * 2
            holder.out_ptrs = (y.data_ptr(),)
            ctx.save_for_backward(x, y)
            return y

        @staticmethod
        def backward(ctx, g):
            x, y = ctx.saved_tensors
            assert torch.equal(y, 2 * x)
            return g * 2, None

    def forward(x):
        lease, holder = Lease(), _Holder(False)
        holder.lease_ref = lease
        with torch.autograd.graph.saved_tensors_hooks(_lease_pack_hook(holder), lambda p: p[0]):
            y = Double.apply(x, holder)
        holder.lease_ref = None
        return y, weakref.ref(lease)

    x = torch.ones(3, requires_grad=True)
    y, w_lease = forward(x)
    assert w_lease() is not None          # held by the saved input
    w_y = weakref.ref(y)
    del y
    gc.collect()
    assert w_y() is None and w_lease() is None, "forward without backward leaked"
    y, w_lease = forward(x)
    y.sum().backward()
    assert w_lease() is None and torch.equal(x.grad, torch.full((3,), 2.0))   # released at the end of backward, output alive


def test_checkpoint_does_not_pickle_the_training_drivers_hooks(tmp_path):
    """ViewParallelStep installs bound methods on the model (`on_colors_pre`, `grad_out`, `view_payload`); a
    checkpoint written during a multi-rank run must not carry the step object, its optimizer and its exchange buffers, nor
    name a class the reference cannot import."""
    import pickle
    import socket
    import torch
    import torch.distributed as dist
    from easy_gaussian_splatting_amd import checkpoint as CK
    from easy_gaussian_splatting_amd.distributed import ViewParallelStep
    from easy_gaussian_splatting_amd.model import GaussianModel

    n = 17
    model = GaussianModel(means=torch.randn(n, 3), log_scales=torch.randn(n, 3), quats=torch.randn(n, 4), sh_0=torch.randn(n, 1, 3),
                          sh_rest=torch.randn(n, 3, 3), logit_opacities=torch.randn(n), sh_degree=1)

    class _Opt:   # (FusedAdam's interface; the real one steps on the GPU only)
        param_groups = []

        def moments_of(self, p):
            raise KeyError

    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        vp = ViewParallelStep(model, _Opt(), sh_grad_fn=lambda *a: None, force_exchange=True)
        assert model.__dict__["on_colors_pre"].__self__ is vp and model.sh_grads == "colors_pre"
        blob = pickle.dumps(model)
        assert b"ViewParallelStep" not in blob and b"distributed" not in blob
        CK.save_gaussian_model(tmp_path / "checkpoints" / "iterations_7.pth", model)
        back = CK.load_gaussian_model(tmp_path, device="cpu")
        for k in GaussianModel._RUNTIME_HOOKS:
            assert k not in back.__dict__, k
        assert back.on_colors_pre is None and back.grad_out is None and back.view_payload is None and back.sh_grads == "dense"
        assert torch.equal(back.means, model.means)
        assert model.__dict__["on_colors_pre"].__self__ is vp     # the live model keeps its hooks
    finally:
        dist.destroy_process_group()
