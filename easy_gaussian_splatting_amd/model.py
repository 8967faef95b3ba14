"""Host-side mirror of the reference's model harness around the rasterization seam.

Only what the hot path needs (SURVEY.md section 8a rows a-1, a-2, a-3 and the optimizer the train
step drives); names and argument meaning follow /root/reference/model/gaussian.py:
  GaussianModel.scales / .opacities / .shs   <- :97-107   (exp / sigmoid / cat activations)
  GaussianModel.forward(data)                <- :351-374  (C=1 batching, clamp to [0,1])
  GaussianModel.update_statistics(...)       <- :188-197  (consumer of .absgrad and radii)
  build_optimizers(...)                      <- :389-412  (one Adam, six named groups)
  GaussianModel.densify_and_prune / reset_opacities  <- :130-146, 259-349  ("next" row f-3: same
      decisions, but one gather per tensor and three host reads instead of ~20 boolean-index
      copies and 8 host syncs; works on torch.optim.Adam and on optim.FusedAdam state)
Checkpoint IO and loaders live in checkpoint.py / scene.py; the viewer is out of scope.

Lines that mirror the reference verbatim, and why: the accessors `nbr_gaussians / scales / opacities / shs / param_names`,
`register_optimizer`, `up_sh_degree`, `update_learning_rate` (error strings included) and the six-group parameter list of
`build_optimizers` follow /root/reference/model/gaussian.py:97-128, 389-412 line for line -- rows a-1 / a-2 of SURVEY.md 8
require the same names, call site and optimizer group order (checkpoints are pickled under the reference's class paths and
must load into either class), so there is no second way to write them.  Everything else in this file (device-side
refinement, fused statistics, in-kernel activations, the split SH hand-over) is this package's own.
"""
from __future__ import annotations

import math
from typing import Any, Dict, Optional

import torch
import torch.nn as nn
from torch import Tensor

from .distributed import all_reduce_statistics, is_distributed
from .rendering import rasterization


class _Clamp01(torch.autograd.Function):
    """`torch.clamp(x, 0, 1)` (/root/reference/model/gaussian.py:368) with a one-pass backward
    (aten's clamp backward is four elementwise kernels on a full-resolution image)."""

    @staticmethod
    def forward(ctx, x: Tensor) -> Tensor:
        from . import _native as nat
        x = x.contiguous()
        out = torch.empty_like(x)
        with torch.cuda.device(x.device):
            nat.check(nat.lib().gs_clamp01(torch.cuda.current_stream(x.device).cuda_stream, x.numel(), x.data_ptr(), None,
                                           out.data_ptr()), "gs_clamp01")
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, v_out: Tensor):
        from . import _native as nat
        (x,) = ctx.saved_tensors
        v_out = v_out.contiguous()
        v_in = torch.empty_like(x)
        with torch.cuda.device(x.device):
            nat.check(nat.lib().gs_clamp01(torch.cuda.current_stream(x.device).cuda_stream, x.numel(), x.data_ptr(),
                                           v_out.data_ptr(), v_in.data_ptr()), "gs_clamp01")
        return v_in


def clamp01(x: Tensor) -> Tensor:
    if x.is_cuda and x.dtype == torch.float32:
        return _Clamp01.apply(x)
    return torch.clamp(x, min=0.0, max=1.0)


class LR_Scheduler:
    """Log-linear interpolation lr_init -> lr_final over max_steps, constant afterwards
    (/root/reference/model/utils.py:19-28; pinned by tests/golden/ref_model_utils.npz `lrs`)."""

    def __init__(self, lr_init: float, lr_final: float, max_steps: int) -> None:
        self.lr_init, self.lr_final, self.max_steps = lr_init, lr_final, max_steps

    def __call__(self, cur_step: int) -> float:
        t = min(1.0, cur_step / self.max_steps)
        return float(math.exp(math.log(self.lr_init) * (1.0 - t) + math.log(self.lr_final) * t))


class GaussianModel(nn.Module):
    # Defaults of this package's own attributes, at class level: an object unpickled from a file the REFERENCE wrote
    # (checkpoint.load_gaussian_model) has none of them in its __dict__ and takes these.
    fuse_sh_cat = True          # hand (sh_0, sh_rest) to the rasterizer, no per-step torch.cat
    fuse_activations = True     # exp / sigmoid inside the projection kernels (GPU)
    tile_culling = "tight"      # render-equivalent shorter lists ("gsplat": meta's list arrays bit-exact, built when read)
    sh_grads = "dense"
    on_colors_pre = None
    grad_out = None             # callable -> {name: tensor}: caller-owned geometry-gradient tensors (distributed.ViewParallelStep)
    view_payload = None         # callable -> tensor: this rank's all-gather record, filled by the rasterizer's backward (same)
    device_refine = True
    # hooks a training driver installs on a live model (bound methods of distributed.ViewParallelStep): never pickled -- a
    # checkpoint must not drag the step object, its optimizer and its exchange buffers along, nor name classes the reference
    # cannot import; an unpickled model falls back to the class defaults above
    _RUNTIME_HOOKS = ("on_colors_pre", "grad_out", "view_payload", "sh_grads")

    def __getstate__(self):
        state = self.__dict__.copy()
        for k in self._RUNTIME_HOOKS:
            state.pop(k, None)
        return state

    def __init__(self, means: Tensor, log_scales: Tensor, quats: Tensor, sh_0: Tensor, sh_rest: Tensor,
                 logit_opacities: Tensor, sh_degree: int, sh_degree_interval: int = 0,
                 white_background: bool = False, fuse_sh_cat: bool = True,
                 densify_grad_thresh: float = 0.0002, densify_scale_thresh: float = 0.01, num_splits: int = 2,
                 prune_radii_ratio_thresh: float = 0.15, prune_scale_thresh: float = 0.1, min_opacity: float = 0.005,
                 means_lr_init: float = 1.6e-4, means_lr_final: float = 1.6e-6, means_lr_schedule_max_steps: int = 30000,
                 use_scale_regularization: bool = False, max_scale_ratio: float = 10.0):
        super().__init__()
        # (defaults: /root/reference/configs/tandt_db.yaml:17-44)
        self.means_lr_scheduler = LR_Scheduler(means_lr_init, means_lr_final, means_lr_schedule_max_steps)
        self.USE_SCALE_REGULARIZATION, self.MAX_SCALE_RATIO = use_scale_regularization, float(max_scale_ratio)
        self.DENSIFY_GRAD_THRESH, self.DENSIFY_SCALE_THRESH, self.NUM_SPLITS = densify_grad_thresh, densify_scale_thresh, num_splits
        self.PRUNE_RADII_RATIO_THRESH, self.PRUNE_SCALE_THRESH, self.MIN_OPACITY = prune_radii_ratio_thresh, prune_scale_thresh, min_opacity
        self.means = nn.Parameter(means.float())  # [N, 3]
        self.log_scales = nn.Parameter(log_scales.float())  # [N, 3]
        self.quats = nn.Parameter(quats.float())  # [N, 4] wxyz
        self.sh_0 = nn.Parameter(sh_0.float())  # [N, 1, 3]
        self.sh_rest = nn.Parameter(sh_rest.float())  # [N, K-1, 3]
        self.logit_opacities = nn.Parameter(logit_opacities.float())  # [N]
        n = means.shape[0]
        self.register_buffer("grad_norm_accum", torch.zeros(n), persistent=False)
        self.register_buffer("collecting_counts", torch.zeros(n), persistent=False)
        self.register_buffer("max_radii", torch.zeros(n), persistent=False)
        self.optimizer: Optional[torch.optim.Optimizer] = None
        self.MAX_SH_DEGREE = sh_degree
        # hand sh_0 / sh_rest to the rasterizer separately instead of materialising `self.shs`
        # (saves the [N,K,3] cat and the split of its gradient every step; same values)
        self.fuse_sh_cat = fuse_sh_cat
        self.active_sh_degree = 0 if sh_degree_interval != 0 else sh_degree
        self.BACKGROUND = nn.Parameter(torch.full((3,), 1.0 if white_background else 0.0), requires_grad=False)

    @classmethod
    def from_pointcloud(cls, pc, sh_degree: int, sh_degree_interval: int = 0, **kwargs) -> "GaussianModel":
        """The reference's constructor (/root/reference/model/gaussian.py:14-95): means = the SfM / random points,
        isotropic scales = half the mean distance to the 3 nearest neighbours, identity rotations, SH band 0 from the
        point colours (model/utils.py:14-16), higher bands zero, opacity 0.8.  `pc`: scene.Pointcloud; the remaining
        keyword arguments are this class's (= the reference constructor's) hyper-parameters."""
        import numpy as np
        from sklearn.neighbors import NearestNeighbors   # type: ignore  (the reference's own dependency, model/utils.py:2)
        xyzs = np.asarray(pc.xyzs)
        dists, _ = NearestNeighbors(n_neighbors=4, metric="euclidean").fit(xyzs).kneighbors(xyzs)
        avg_dist = np.repeat(np.mean(dists[:, 1:].astype(np.float32), axis=1, keepdims=True), repeats=3, axis=1)
        n = xyzs.shape[0]
        quats = torch.zeros((n, 4), dtype=torch.float32)
        quats[:, 0] = 1.0
        shs = torch.zeros((n, (sh_degree + 1) ** 2, 3), dtype=torch.float32)
        shs[:, 0] = torch.tensor((np.asarray(pc.rgbs) / 255.0 - 0.5) / 0.28209479177387814, dtype=torch.float32)
        return cls(means=torch.tensor(xyzs, dtype=torch.float32), log_scales=torch.log(torch.tensor(avg_dist, dtype=torch.float32) / 2.0),
                   quats=quats, sh_0=shs[:, 0:1].contiguous(), sh_rest=shs[:, 1:].contiguous(),
                   logit_opacities=torch.logit(0.8 * torch.ones((n,), dtype=torch.float32)), sh_degree=sh_degree,
                   sh_degree_interval=sh_degree_interval, **kwargs)

    @property
    def nbr_gaussians(self) -> int:
        return self.means.shape[0]

    @property
    def scales(self) -> Tensor:
        return torch.exp(self.log_scales)

    @property
    def opacities(self) -> Tensor:
        return torch.sigmoid(self.logit_opacities)

    @property
    def shs(self) -> Tensor:
        return torch.cat([self.sh_0, self.sh_rest], dim=1)

    @property
    def param_names(self):
        return ["means", "log_scales", "quats", "sh_0", "sh_rest", "logit_opacities"]

    def register_optimizer(self, optimizer: torch.optim.Optimizer):
        if self.optimizer is not None:
            raise RuntimeError("optimizer has been registered")
        self.optimizer = optimizer

    def up_sh_degree(self):
        self.active_sh_degree = min(self.active_sh_degree + 1, self.MAX_SH_DEGREE)

    def update_learning_rate(self, step: int):
        """Only the `means` group follows a schedule (/root/reference/model/gaussian.py:121-128)."""
        if self.optimizer is None:
            raise RuntimeError("optimizer has not been registered")
        for param_group in self.optimizer.param_groups:
            if param_group["name"] == "means":
                param_group["lr"] = self.means_lr_scheduler(step)
                return
        raise RuntimeError("the param_group 'means' isn't in the optimizer")

    def get_regularization_dict(self) -> Dict[str, Tensor]:
        """mean(max(max_k s_k / min_k s_k, MAX_SCALE_RATIO) - MAX_SCALE_RATIO) when enabled
        (/root/reference/model/gaussian.py:376-386; off in both shipped configs)."""
        reg: Dict[str, Tensor] = {}
        if self.USE_SCALE_REGULARIZATION:
            scales = self.scales
            ratio = scales.amax(dim=1) / scales.amin(dim=1)
            max_ratio = self.MAX_SCALE_RATIO   # (a Python float: checkpoint.load_gaussian_model converts the reference's 0-d tensor once)
            if isinstance(max_ratio, Tensor):   # (an object unpickled by hand: convert once, not per step)
                max_ratio = self.MAX_SCALE_RATIO = float(max_ratio.detach().cpu())
            reg["scale_reg"] = torch.mean(torch.clamp(ratio, min=max_ratio) - max_ratio)
        return reg

    # ---------------------------------------------------------------- refinement (row f-3)
    def _moments(self, name: str):
        """(exp_avg, exp_avg_sq) of parameter `name` from either optimizer flavour (zeros if unset)."""
        p = getattr(self, name)
        opt = self.optimizer
        if opt is None:
            raise RuntimeError("optimizer has not been registered")
        if hasattr(opt, "moments_of"):
            return opt.moments_of(p)
        st = opt.state.get(p, {})
        if "exp_avg" in st:
            return st["exp_avg"], st["exp_avg_sq"]
        return torch.zeros_like(p), torch.zeros_like(p)

    def _replace_parameters(self, new: Dict[str, Tensor], new_m: Dict[str, Tensor], new_v: Dict[str, Tensor]):
        """Swaps in resized parameter tensors and their Adam moments (reference: the per-group
        state surgery of model/gaussian.py:199-257)."""
        opt = self.optimizer
        old = {name: getattr(self, name) for name in self.param_names}
        for name in self.param_names:
            setattr(self, name, nn.Parameter(new[name].contiguous()))
        if hasattr(opt, "replace_parameters"):
            opt.replace_parameters([(getattr(self, n), new_m[n], new_v[n]) for n in self.param_names])
            return
        for group in opt.param_groups:
            name = group["name"]
            state = opt.state.pop(old[name], {})
            group["params"][0] = getattr(self, name)
            if "step" in state:   # a parameter Adam has never stepped keeps its (lazily created) empty state
                state["exp_avg"], state["exp_avg_sq"] = new_m[name].contiguous(), new_v[name].contiguous()
                opt.state[group["params"][0]] = state

    def _split_noise(self, generator: Optional[torch.Generator]) -> Tensor:
        """Standard-normal samples for the split children, [NUM_SPLITS, N, 3]: drawn for EVERY Gaussian so that the
        draw needs no count from the device (the reference draws exactly the split parents' samples after a host
        sync, /root/reference/model/gaussian.py:164-165; same distribution)."""
        noise = torch.randn((self.NUM_SPLITS, self.nbr_gaussians, 3), device=self.means.device, generator=generator)
        if is_distributed():
            # replicas must stay bitwise identical (distributed.ViewParallelStep): every rank splits with
            # rank 0's noise, whatever state its own generator is in
            torch.distributed.broadcast(noise, src=0)
        return noise

    @torch.no_grad()
    def _densify_and_prune_device(self, generator: Optional[torch.Generator]) -> Dict[str, Any]:
        """Row f-3 on the device (csrc/gs_refine.hip): decisions, one prefix scan, one gather of all six parameters
        and both Adam moments into fresh flat buffers; ONE host read (new sizes + tb_info counters)."""
        import ctypes as ct
        from . import _native as nat
        from .optim import FusedAdam
        L, opt, dev = nat.lib(), self.optimizer, self.means.device
        n_old, S, K = self.nbr_gaussians, int(self.NUM_SPLITS), 1 + self.sh_rest.shape[1]
        st = torch.cuda.current_stream(dev).cuda_stream
        noise = self._split_noise(generator)
        flags = torch.empty((3, n_old), dtype=torch.int32, device=dev)
        counters = torch.empty((5,), dtype=torch.int64, device=dev)
        with torch.cuda.device(dev):
            nat.check(L.gs_refine_flags(st, n_old, S, float(self.DENSIFY_GRAD_THRESH), float(self.DENSIFY_SCALE_THRESH),
                                        float(self.PRUNE_RADII_RATIO_THRESH), float(self.PRUNE_SCALE_THRESH), float(self.MIN_OPACITY),
                                        self.grad_norm_accum.data_ptr(), self.collecting_counts.data_ptr(), self.max_radii.data_ptr(),
                                        self.log_scales.data_ptr(), self.logit_opacities.data_ptr(), flags.data_ptr(),
                                        counters.data_ptr()), "gs_refine_flags")
            incl = torch.empty_like(flags)
            scan_ws = torch.empty((int(L.gs_scan_rows_workspace_ints(3, n_old)),), dtype=torch.int32, device=dev)
            nat.check(L.gs_scan_rows_i32(st, 3, n_old, flags.data_ptr(), incl.data_ptr(), scan_ws.data_ptr()), "gs_scan_rows_i32")
            # the one host read: three totals (they size the new buffers) + the five tb_info counters
            host = torch.cat([incl[:, -1].to(torch.int64) if n_old else torch.zeros(3, dtype=torch.int64, device=dev), counters]).tolist()
            tot_old, tot_child, tot_clone, ns, nc, c0, c1, c2 = (int(v) for v in host)
            n_new = tot_old + S * tot_child + tot_clone
            widths = [3, 3, 4, 3, 3 * (K - 1), 1]
            old_offs = list(opt._offs)
            new_offs, new_ends, total = FusedAdam.flat_layout([n_new * w for w in widths])
            f32 = dict(dtype=torch.float32, device=dev)
            new_p, new_m, new_v = torch.empty(total, **f32), torch.empty(total, **f32), torch.empty(total, **f32)
            for o, w, e in zip(new_offs, widths, new_ends):   # the (<= 3 float) pads between segments
                if e > o + n_new * w:
                    for buf in (new_p, new_m, new_v):
                        buf[o + n_new * w:e].zero_()
            src = torch.empty((max(n_new, 1),), dtype=torch.int32, device=dev)
            tag = torch.empty((max(n_new, 1),), dtype=torch.int8, device=dev)
            nat.check(L.gs_refine_apply(st, n_old, S, K, flags.data_ptr(), incl.data_ptr(), tot_old, tot_child, tot_clone,
                                        noise.data_ptr(), opt.flat_param.data_ptr(), opt.exp_avg.data_ptr(), opt.exp_avg_sq.data_ptr(),
                                        (ct.c_int64 * 6)(*old_offs), new_p.data_ptr(), new_m.data_ptr(), new_v.data_ptr(),
                                        (ct.c_int64 * 6)(*new_offs), src.data_ptr(), tag.data_ptr()), "gs_refine_apply")
        shapes = {"means": (n_new, 3), "log_scales": (n_new, 3), "quats": (n_new, 4), "sh_0": (n_new, 1, 3),
                  "sh_rest": (n_new, K - 1, 3), "logit_opacities": (n_new,)}
        new_params = []
        for name, o, w in zip(self.param_names, new_offs, widths):
            setattr(self, name, nn.Parameter(new_p[o:o + n_new * w].view(shapes[name])))
            new_params.append(getattr(self, name))
        opt.adopt_flat(new_p, new_m, new_v, new_params)
        self.grad_norm_accum = torch.zeros((n_new,), device=dev)
        self.collecting_counts = torch.zeros((n_new,), device=dev)
        self.max_radii = torch.zeros((n_new,), device=dev)
        if is_distributed():
            from .distributed import assert_replicas_identical
            assert_replicas_identical(self.means, "means after densify_and_prune")
        return {"train/densify": {"split": ns, "clone": nc},
                "train/prune": {"low_opacity": c0, "large_radii": c1 - c0, "large_scale": c2 - c1},
                "train/nbr_gaussians": n_new, "n_before": n_old}

    @torch.no_grad()
    def densify_and_prune(self, generator: Optional[torch.Generator] = None) -> Dict[str, Any]:
        """Clone / split high-gradient Gaussians, prune transparent / huge ones, reset statistics
        (/root/reference/model/gaussian.py:259-349).  With optim.FusedAdam on the GPU the whole refinement runs on the
        device (`_densify_and_prune_device`: one host read); otherwise -- torch.optim.Adam, CPU -- the torch path below:
        decisions on the device, three host reads (split, clone, survivors) because tensor shapes need them."""
        from .rendering import quat_to_rotmat_torch
        if self.means.is_cuda and hasattr(self.optimizer, "adopt_flat") and getattr(self, "device_refine", True):
            return self._densify_and_prune_device(generator)
        n_old = self.nbr_gaussians
        avg = self.grad_norm_accum / (self.collecting_counts + 1e-8)
        avg = torch.where(torch.isnan(avg), torch.zeros_like(avg), avg)
        high = avg >= self.DENSIFY_GRAD_THRESH
        big = self.scales.amax(dim=-1) >= self.DENSIFY_SCALE_THRESH
        split_mask, clone_mask = big & high, (~big) & high
        split_idx = torch.nonzero(split_mask).squeeze(1)       # host read 1
        clone_idx = torch.nonzero(clone_mask).squeeze(1)       # host read 2
        ns, nc = split_idx.numel(), clone_idx.numel()
        params = {name: getattr(self, name).detach() for name in self.param_names}
        new_parts = {name: [] for name in self.param_names}
        if ns:
            rep = split_idx.repeat(self.NUM_SPLITS)            # [parents..., parents...] like .repeat(NUM_SPLITS, 1)
            scales = torch.exp(params["log_scales"][rep])
            noise = self._split_noise(generator)[:, split_idx, :].reshape(-1, 3)   # copy-major, like `rep`
            R = quat_to_rotmat_torch(params["quats"][rep])
            offs = torch.bmm(R, (scales * noise).unsqueeze(-1)).squeeze(-1)
            for name in self.param_names:
                v = params[name][rep]
                if name == "means":
                    v = v + offs
                elif name == "log_scales":
                    v = torch.log(scales / (0.8 * self.NUM_SPLITS))
                new_parts[name].append(v)
        if nc:
            for name in self.param_names:
                new_parts[name].append(params[name][clone_idx])
        n_new = ns * self.NUM_SPLITS + nc
        # prune mask over [old | new]; new entries start with max_radii = 0 and are never "split parents"
        cat = {name: (torch.cat([params[name]] + new_parts[name], dim=0) if n_new else params[name]) for name in self.param_names}
        zeros_new = torch.zeros((n_new,), device=self.max_radii.device)
        max_radii = torch.cat([self.max_radii, zeros_new])
        was_split = torch.cat([split_mask, zeros_new.bool()])
        low_op = torch.sigmoid(cat["logit_opacities"]) < self.MIN_OPACITY
        big_r = max_radii > self.PRUNE_RADII_RATIO_THRESH
        big_s = torch.exp(cat["log_scales"]).amax(dim=-1) > self.PRUNE_SCALE_THRESH
        prune = low_op | big_r | big_s | was_split
        keep_idx = torch.nonzero(~prune).squeeze(1)            # host read 3
        counts = torch.stack([low_op.sum(), (low_op | big_r).sum(), (low_op | big_r | big_s).sum()])
        new, new_m, new_v = {}, {}, {}
        for name in self.param_names:
            m, v = self._moments(name)
            pad = [torch.zeros_like(p) for p in new_parts[name]]
            new[name] = cat[name][keep_idx]
            new_m[name] = (torch.cat([m] + pad, dim=0) if n_new else m)[keep_idx]
            new_v[name] = (torch.cat([v] + pad, dim=0) if n_new else v)[keep_idx]
        self._replace_parameters(new, new_m, new_v)
        n = self.nbr_gaussians
        dev = self.means.device
        self.grad_norm_accum = torch.zeros((n,), device=dev)
        self.collecting_counts = torch.zeros((n,), device=dev)
        self.max_radii = torch.zeros((n,), device=dev)
        c0, c1, c2 = (int(x) for x in counts.tolist())
        if is_distributed():
            from .distributed import assert_replicas_identical
            assert_replicas_identical(self.means, "means after densify_and_prune")
        return {"train/densify": {"split": ns, "clone": nc},
                "train/prune": {"low_opacity": c0, "large_radii": c1 - c0, "large_scale": c2 - c1},
                "train/nbr_gaussians": n, "n_before": n_old}

    @torch.no_grad()
    def reset_opacities(self):
        """opacity <- min(opacity / 2, 2 * MIN_OPACITY); the group's Adam moments restart at zero
        (/root/reference/model/gaussian.py:130-146)."""
        target = torch.minimum(self.opacities * 0.5, torch.full_like(self.logit_opacities, self.MIN_OPACITY * 2.0))
        new = {name: getattr(self, name).detach() for name in self.param_names}
        new["logit_opacities"] = torch.logit(target)
        new_m, new_v = {}, {}
        for name in self.param_names:
            m, v = self._moments(name)
            new_m[name], new_v[name] = (torch.zeros_like(m), torch.zeros_like(v)) if name == "logit_opacities" else (m, v)
        self._replace_parameters(new, new_m, new_v)
        # (how far the blend walks into its lists jumps with this call -- nothing saturates any more --: a captured step runner
        #  sizes what the walk leaves from a fresh probe instead of from the steps before, train_graph.TrainStepGraph.step)
        self.opacity_resets = getattr(self, "opacity_resets", 0) + 1

    def forward(self, data: Dict[str, Any], clamp: bool = True) -> Dict[str, Optional[Tensor]]:
        """`clamp=False` returns the un-clamped image for `LossComputer(clamp_input=True)` (the clamp of
        /root/reference/model/gaussian.py:368 then happens inside the loss kernels)."""
        w2c = data["w2c"]
        # on the GPU the raw parameters go in and exp / sigmoid happen inside the projection kernels
        raw = self.means.is_cuda and getattr(self, "fuse_activations", True)
        batch_render_imgs, _, meta = rasterization(
            means=self.means,
            quats=self.quats,
            scales=self.log_scales if raw else self.scales,
            opacities=self.logit_opacities if raw else self.opacities,
            colors=(self.sh_0, self.sh_rest) if self.fuse_sh_cat else self.shs,
            sh_degree=self.active_sh_degree,
            viewmats=w2c[None],
            Ks=data["K"][None],
            width=data["width"],
            height=data["height"],
            backgrounds=self.BACKGROUND[None],
            absgrad=True,
            packed=False,
            # the model consumes only radii / means2d / the image (:188-197, 371-372), which are identical in
            # both list modes: take the shorter, render-equivalent lists (set `tile_culling = "gsplat"` for
            # bit-exact gsplat list arrays in `meta`)
            _tile_culling=getattr(self, "tile_culling", "tight"),
            _sh_grads=getattr(self, "sh_grads", "dense"),
            _activations="exp_sigmoid" if raw else "none",
            _on_colors_pre=getattr(self, "on_colors_pre", None),
            _grad_out=self.grad_out() if callable(getattr(self, "grad_out", None)) else None,
            _view_payload=self.view_payload() if callable(getattr(self, "view_payload", None)) else None,
        )
        render_img = batch_render_imgs.squeeze(0)   # (a view both ways: `[0]` would cost a zero-fill + copy in backward)
        if clamp:
            render_img = clamp01(render_img)
        return {
            "render_img": render_img,  # [H, W, 3]
            "batch_xys": meta["means2d"],  # [1, N, 2]
            "batch_radii": meta["radii"],  # [1, N]
        }

    @torch.no_grad()
    def update_statistics(self, data: Dict[str, Any], model_output: Dict[str, Tensor]):
        max_hw = max(data["height"], data["width"])
        raw_radii, absgrad = model_output["batch_radii"].detach(), model_output["batch_xys"].absgrad.detach()
        if raw_radii.is_cuda and not is_distributed() and raw_radii.shape[0] == 1 and self.max_radii.is_contiguous():
            # single view, single process: one HIP launch instead of ~10 masked torch kernels
            from . import _native as nat
            st = torch.cuda.current_stream(raw_radii.device).cuda_stream
            with torch.cuda.device(raw_radii.device):
                nat.check(nat.lib().gs_update_statistics(
                    st, raw_radii.shape[1], float(max_hw), raw_radii.contiguous().data_ptr(), absgrad.contiguous().data_ptr(),
                    self.max_radii.data_ptr(), self.grad_norm_accum.data_ptr(), self.collecting_counts.data_ptr()),
                    "gs_update_statistics")
            return
        radii = raw_radii[0] / max_hw
        xys_absgrad = absgrad[0]
        visible = radii > 0.0
        # masked forms of the reference's three boolean-index updates (same values, no host sync)
        zero = torch.zeros_like(radii)
        step_radii = torch.where(visible, radii, zero)
        step_grads = torch.where(visible, torch.norm(xys_absgrad, dim=-1) * max_hw, zero)
        step_counts = visible.to(self.collecting_counts.dtype)
        if is_distributed():  # one view per rank: every replica must see every view's statistics
            all_reduce_statistics(step_grads, step_counts, step_radii)
        self.max_radii.copy_(torch.maximum(self.max_radii, step_radii))
        self.grad_norm_accum.add_(step_grads)
        self.collecting_counts.add_(step_counts)


def build_optimizers(model: GaussianModel, means_lr: float, log_scales_lr: float, quats_lr: float,
                     sh_0_lr: float, sh_rest_lr: float, logit_opacities_lr: float,
                     fused=None) -> torch.optim.Optimizer:
    """One Adam, six named groups (reference signature).  `fused`: None/False/True go to
    `torch.optim.Adam(fused=...)`; "hip" selects optim.FusedAdam (one HIP kernel per step over flat
    buffers, gradients cleared in the same pass)."""
    params = [
        {"params": [model.means], "lr": means_lr, "name": "means"},
        {"params": [model.log_scales], "lr": log_scales_lr, "name": "log_scales"},
        {"params": [model.quats], "lr": quats_lr, "name": "quats"},
        {"params": [model.sh_0], "lr": sh_0_lr, "name": "sh_0"},
        {"params": [model.sh_rest], "lr": sh_rest_lr, "name": "sh_rest"},
        {"params": [model.logit_opacities], "lr": logit_opacities_lr, "name": "logit_opacities"},
    ]
    if fused == "hip":
        from .optim import FusedAdam
        optimizer = FusedAdam(params)
    else:
        kw = {} if fused is None else {"fused": fused}
        optimizer = torch.optim.Adam(params, **kw)
    model.register_optimizer(optimizer)
    return optimizer
