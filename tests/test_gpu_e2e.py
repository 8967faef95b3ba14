"""The reference's acceptance test, end to end (VERDICT r5 missing #3; /root/reference/README.md:5-9, train.py:93-157,
eval.py:120-133, configs/nerf_synthetic.yaml): a dataset ON DISK in nerf_synthetic's layout -> `Scene` -> `generate_pointcloud`
-> `GaussianModel.from_pointcloud` -> the reference's schedule (densify / prune every 100 steps in (100, 1500], opacity reset
every 1000, SH degree + 1 every 500, means-LR schedule) on `TrainStepGraph` -> PSNR on held-out views.  No real capture exists
in this container: the dataset is rendered by the HIP forward from ground-truth Gaussians on surfaces (tools/e2e_train.py)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_train_from_a_loaded_dataset_to_held_out_psnr(tmp_path):
    import e2e_train as E
    res = E.run(3000, ("captured", "eager"), out_dir=str(tmp_path / "dome"))
    cap, eag = res["runs"]["captured"], res["runs"]["eager"]
    import parity_log
    parity_log.record(e2e={"dataset": res["dataset"], **{m: {k: r[k] for k in ("psnr", "ssim", "train_iters_per_s", "n_gaussians_initial",
                                                                               "n_gaussians_final", "active_sh_degree", "schedule", "runner")}
                                                         for m, r in res["runs"].items()}})
    # the loaders saw what was written
    assert cap["train_views"] == 40 and cap["held_out_views"] == 8 and cap["n_gaussians_initial"] == 100_000
    # the schedule ran: 14 refinements (N moved both ways), one reset, SH degree 0 -> 3
    assert len(cap["n_gaussians"]) == 15 and len(set(cap["n_gaussians"])) > 10 and cap["active_sh_degree"] == 3
    assert cap["runner"]["rebuilds"] >= 15 and cap["runner"]["overflows"] == 0
    # quality: the reference's README quotes 30+ dB on nerf_synthetic after 30 k steps; this scene has to get there in 3 k
    assert cap["psnr"] >= 30.0 and cap["ssim"] >= 0.93, cap
    # the captured runner IS the eager loop: the same refinement decisions, the same picture
    assert cap["n_gaussians"] == eag["n_gaussians"]
    assert abs(cap["psnr"] - eag["psnr"]) <= 0.1, (cap["psnr"], eag["psnr"])
