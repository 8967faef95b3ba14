"""Checkpoint layout and camera export of the reference:

    <output>/checkpoints/iterations_<N>.pth      torch.save of the whole GaussianModel module (optimizer detached
                                                 unless save_optimizer=True)        /root/reference/utils.py:78-87
    <output>/cameras.json                        list of Frame.to_json dicts        /root/reference/scene/scene.py:91-94

`load_gaussian_model(path, iterations=None)` picks the requested or the latest `iterations_*.pth`
(/root/reference/utils.py:48-75).  The reference pickles the module by class reference
(`model.gaussian.GaussianModel`, with a `model.utils.LR_Scheduler` inside); this package's GaussianModel carries the
same attribute names, so files written here under those class paths unpickle in the reference, and files written by
the reference unpickle here -- `reference_class_paths()` installs / removes the two alias modules around the
(un)pickling, nothing of the reference is imported.
"""
from __future__ import annotations

import contextlib
import json
import sys
import types
from pathlib import Path
from typing import List, Optional

import numpy as np
import torch

from . import model as _model


@contextlib.contextmanager
def reference_class_paths():
    """Temporarily exposes this package's GaussianModel / LR_Scheduler as `model.gaussian.GaussianModel` and
    `model.utils.LR_Scheduler`, the paths the reference's pickles name (/root/reference/model/gaussian.py:14,
    model/utils.py:19)."""
    names = ("model", "model.gaussian", "model.utils")
    saved = {n: sys.modules.get(n) for n in names}
    pkg = types.ModuleType("model")
    pkg.__path__ = []   # a package
    gaussian = types.ModuleType("model.gaussian")
    utils = types.ModuleType("model.utils")
    gaussian.GaussianModel = _model.GaussianModel
    utils.LR_Scheduler = _model.LR_Scheduler
    pkg.gaussian, pkg.utils = gaussian, utils
    old_names = (_model.GaussianModel.__module__, _model.LR_Scheduler.__module__)
    sys.modules.update({"model": pkg, "model.gaussian": gaussian, "model.utils": utils})
    _model.GaussianModel.__module__, _model.LR_Scheduler.__module__ = "model.gaussian", "model.utils"
    try:
        yield
    finally:
        _model.GaussianModel.__module__, _model.LR_Scheduler.__module__ = old_names
        for n in names:
            if saved[n] is None:
                sys.modules.pop(n, None)
            else:
                sys.modules[n] = saved[n]


def as_torch_adam(optimizer) -> torch.optim.Adam:
    """A `torch.optim.Adam` over the same parameter objects with the same groups, step count and moments (copies) --
    what the reference's pickles hold (/root/reference/model/gaussian.py:389-412).  `optim.FusedAdam.state_dict()` is
    already in Adam's layout, so this is a plain load."""
    if isinstance(optimizer, torch.optim.Adam):
        return optimizer
    groups = [{"params": list(g["params"]), "lr": g["lr"], "name": g.get("name")} for g in optimizer.param_groups]
    adam = torch.optim.Adam(groups, betas=tuple(optimizer.defaults["betas"]), eps=optimizer.defaults["eps"])
    adam.load_state_dict(optimizer.state_dict())
    return adam


def as_fused_adam(optimizer):
    """The reverse: an `optim.FusedAdam` over the parameters of a `torch.optim.Adam` (they become views of its flat
    buffer, wherever they live now), moments and step count restored."""
    from .optim import FusedAdam
    groups = [{"params": list(g["params"]), "lr": g["lr"], "name": g.get("name")} for g in optimizer.param_groups]
    state = optimizer.state_dict()
    fused = FusedAdam(groups, betas=tuple(optimizer.defaults["betas"]), eps=optimizer.defaults["eps"])
    fused.load_state_dict(state)
    return fused


def save_gaussian_model(path: Path, gaussian_model: torch.nn.Module, save_optimizer: bool = False,
                        reference_compatible: bool = True):
    """`torch.save(gaussian_model, path)` with the optimizer detached unless asked for (reference behaviour,
    /root/reference/utils.py:78-87).  With `reference_compatible` the file names the reference's class paths and holds
    only what the reference's classes know: `MAX_SCALE_RATIO` as a tensor (the reference calls
    `torch.max(ratio, self.MAX_SCALE_RATIO)`, model/gaussian.py:380-382) and, with `save_optimizer`, a
    `torch.optim.Adam` carrying this optimizer's state (an `optim.FusedAdam` object would name a class the reference
    cannot import)."""
    path = Path(path)
    path.parent.mkdir(parents=True, exist_ok=True)
    live_optimizer = gaussian_model.optimizer
    live_ratio = gaussian_model.__dict__.get("MAX_SCALE_RATIO")
    try:
        if not save_optimizer:
            gaussian_model.optimizer = None   # type: ignore
        elif reference_compatible and live_optimizer is not None:
            gaussian_model.optimizer = as_torch_adam(live_optimizer)
        if reference_compatible and live_ratio is not None and not isinstance(live_ratio, torch.Tensor):
            gaussian_model.MAX_SCALE_RATIO = torch.tensor(float(live_ratio), dtype=torch.float32, device=gaussian_model.means.device)
        with (reference_class_paths() if reference_compatible else contextlib.nullcontext()):
            torch.save(gaussian_model, path)
    finally:
        gaussian_model.optimizer = live_optimizer
        if live_ratio is not None:
            gaussian_model.MAX_SCALE_RATIO = live_ratio


def find_checkpoint(path: Path, iterations: Optional[int] = None) -> Path:
    cpt_lst = list((Path(path) / "checkpoints").glob("*.pth"))
    if iterations is not None:
        for cpt in cpt_lst:
            if cpt.stem == f"iterations_{iterations}":
                return cpt
        raise ValueError(f"cannot find checkpoint for iteration {iterations}")
    best, best_it = None, 0
    for cpt in cpt_lst:
        it = int(cpt.stem.split("_")[1])
        if it > best_it:
            best, best_it = cpt, it
    if best is None:
        raise ValueError("no checkpoint found")
    return best


def load_gaussian_model(path: Path, iterations: Optional[int] = None, device: Optional[str] = None,
                        optimizer: str = "keep") -> torch.nn.Module:
    """The reference loads to the CPU and then calls `.cuda()` (/root/reference/utils.py:48-75); `device=None` does that
    when a GPU is present.  Works on files written here and on files written by the reference (whose objects lack this
    package's extra attributes: they fall back to the class defaults of `model.GaussianModel`; their statistics and
    `MAX_SCALE_RATIO` are plain tensor attributes that `Module.to` does not move).
    A pickled optimizer (`save_optimizer=True`) is made steppable on `device`: `optimizer="keep"` leaves a
    `torch.optim.Adam` an Adam (state moved to the device), `"hip"` turns it into `optim.FusedAdam`; an
    `optim.FusedAdam` object found in an older file of this package is always rebuilt over the moved parameters."""
    if optimizer not in ("keep", "hip"):
        raise ValueError("optimizer: 'keep' or 'hip'")
    target = find_checkpoint(path, iterations)
    with reference_class_paths():
        gaussian_model = torch.load(target, map_location="cpu", weights_only=False)
    if device is None:
        device = "cuda" if torch.cuda.is_available() else "cpu"
    opt = gaussian_model.__dict__.get("optimizer")
    opt_state = None if opt is None else opt.state_dict()   # (before .to(): FusedAdam reads its flat CPU buffers here)
    gaussian_model = gaussian_model.to(device)
    for name in ("grad_norm_accum", "collecting_counts", "max_radii"):   # plain attributes in the reference's pickles
        t = gaussian_model.__dict__.get(name)
        if isinstance(t, torch.Tensor):
            gaussian_model.__dict__[name] = t.to(device)
    # MAX_SCALE_RATIO is a 0-d tensor in files the reference wrote (or `reference_compatible=True` here): a Python float from
    # now on -- read ONCE, on the CPU copy -- so that get_regularization_dict() never costs a device-to-host sync per
    # training step; save_gaussian_model(reference_compatible=True) re-tensorises it at save time
    ratio = gaussian_model.__dict__.get("MAX_SCALE_RATIO")
    if isinstance(ratio, torch.Tensor):
        gaussian_model.__dict__["MAX_SCALE_RATIO"] = float(ratio.detach().cpu())
    if opt is not None:
        from .optim import FusedAdam
        groups = [{"params": list(g["params"]), "lr": g["lr"], "name": g.get("name")} for g in opt.param_groups]
        betas, eps = tuple(opt.defaults["betas"]), opt.defaults["eps"]
        want_fused = optimizer == "hip" or isinstance(opt, FusedAdam)
        if want_fused and torch.device(device).type != "cuda":
            want_fused = False   # (FusedAdam steps on the GPU only; on the CPU the state stays a torch.optim.Adam)
        new = FusedAdam(groups, betas=betas, eps=eps) if want_fused else torch.optim.Adam(groups, betas=betas, eps=eps)
        new.load_state_dict(opt_state)   # (torch casts the moments to each parameter's device)
        gaussian_model.optimizer = new
    return gaussian_model


class CameraState:
    """What the reference's viewer / eval code gets from cameras.json (/root/reference/utils.py:28-46): w2c [4,4],
    K [3,3] (principal point at the image centre), width, height."""

    def __init__(self, w2c: np.ndarray, K: np.ndarray, width: int, height: int):
        self.w2c, self.K, self.width, self.height = w2c, K, width, height


def load_camera_states(path: Path) -> List[CameraState]:
    camera_states = []
    with open(Path(path) / "cameras.json", "r") as f:
        for cam in json.load(f):
            c2w = np.eye(4)
            c2w[:3, :3] = np.array(cam["rotation"])
            c2w[:3, 3] = np.array(cam["position"])
            K = np.array([[cam["fx"], 0, cam["width"] / 2], [0, cam["fy"], cam["height"] / 2], [0, 0, 1]], dtype=np.float32)
            camera_states.append(CameraState(np.linalg.inv(c2w), K, cam["width"], cam["height"]))
    return camera_states
