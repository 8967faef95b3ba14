#!/usr/bin/env python3
"""Eager train steps (model mirror + HIP loss + fused Adam) of one of the BASELINE.json / realistic-footprint workloads, for
kernel-level profiles of the configurations beside the headline (VERDICT r4 missing #4):
    tools/prof_cmd.sh r05_s3 tools/config_run.py S3 6          (rocprofv3 --kernel-trace --stats)
    tools/prof_pmc.sh r05_s3 tools/config_run.py S3 3          (FETCH_SIZE / WRITE_SIZE passes)
names: bench1M, S3 (2 M / 1080p), S5 (5 M / 4K), heavy1M, heavy2M (synthetic.config_heavy), longlists (200 k heavy-tailed splats).
Prints one JSON line: stage times (HIP events), the list sizes, and what the blend kernels actually WALKED -- the unit their
roofline is priced on (a saturated tile abandons the rest of its list, so `128 * I_listed / t` would exceed the HBM peak)."""
import json, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
from easy_gaussian_splatting_amd import rendering, synthetic as SY
from easy_gaussian_splatting_amd.loss import LossComputer
from easy_gaussian_splatting_amd.model import GaussianModel, build_optimizers

MAKE = {"bench1M": SY.config_bench_1m, "S3": SY.config_s3, "S5": SY.config_s5, "heavy1M": lambda: SY.config_heavy(n=1_000_000),
        "heavy2M": lambda: SY.config_heavy(n=2_000_000), "longlists": lambda: SY.config_long_lists(n=200_000, width=1920, height=1080)}
LRS = (1.6e-4, 5e-3, 1e-3, 2.5e-3, 1.25e-4, 5e-2)


def _heavy_window(n):
    """config_heavy with nothing in front of depth 8 in the left 40 % of the image: the front slab of a depth split finishes
    the right part of the frame and leaves the left part live -- a frame that needs BOTH depth rounds (tools/rounds_time.py)."""
    sc = SY.config_heavy(n=n)
    zc = sc["means"][:, 2] + 8.0
    keep = ~((sc["means"][:, 0] / zc < -0.12) & (zc < 8.0))
    for k in ("means", "quats", "scales", "opacities", "shs"):
        sc[k] = np.ascontiguousarray(sc[k][keep])
    return sc


MAKE["heavy2Mwin"] = lambda: _heavy_window(2_000_000)
MAKE["heavy1Mwin"] = lambda: _heavy_window(1_000_000)


def model_from_scene(sc, dev):
    T = torch.from_numpy
    op = np.clip(sc["opacities"], 1e-4, 1 - 1e-4)
    shs = T(sc["shs"])
    return GaussianModel(means=T(sc["means"]), log_scales=torch.log(T(sc["scales"])), quats=T(sc["quats"]), sh_0=shs[:, :1].contiguous(),
                         sh_rest=shs[:, 1:].contiguous(), logit_opacities=T(np.log(op / (1 - op)).astype(np.float32)),
                         sh_degree=int(sc["sh_degree"]), white_background=bool(sc["backgrounds"][0, 0] > 0.5)).to(dev)


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "S3"
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    mode = sys.argv[3] if len(sys.argv) > 3 else "tight"
    dev = torch.device("cuda:0")
    sc = MAKE[name]()
    W, H = int(sc["width"]), int(sc["height"])
    model = model_from_scene(sc, dev)
    model.tile_culling = mode
    opt = build_optimizers(model, *LRS, fused="hip")
    lc = LossComputer(0.2, clamp_input=True)
    data = {"w2c": torch.from_numpy(sc["viewmats"][0]).to(dev), "K": torch.from_numpy(sc["Ks"][0]).to(dev), "width": W, "height": H}
    g = torch.Generator().manual_seed(7)
    gt = torch.nn.functional.interpolate(torch.rand((1, 3, H // 16 + 1, W // 16 + 1), generator=g), size=(H, W), mode="bilinear")[0].permute(1, 2, 0).contiguous().to(dev)
    one = torch.ones((), device=dev)

    def step():
        out = model(data, clamp=False)
        lc.get_loss_dict(out["render_img"], gt, None)["total"].backward(gradient=one)
        model.update_statistics(data, out)
        opt.step(); opt.zero_grad()

    for it in range(iters + 2):
        if it == 2:
            rendering.profile_stages(True)
        step()
    st = rendering.profile_stages(False) or {}
    dbg = {}
    ins = [p.detach().clone().requires_grad_(True) for p in (model.means, model.quats, model.log_scales, model.logit_opacities)]
    _, _, meta = rendering.rasterization(ins[0], ins[1], ins[2], ins[3], (model.sh_0, model.sh_rest), data["w2c"][None], data["K"][None], W, H,
                                         sh_degree=model.active_sh_degree, packed=False, backgrounds=model.BACKGROUND[None], absgrad=True,
                                         _tile_culling=mode, _activations="exp_sigmoid", _debug=dbg)
    torch.cuda.synchronize()
    n_isects = int(meta["flatten_ids"].numel())
    walked, pairs = int(dbg["walked_isects"]), int(dbg["qcnt"].sum())
    t_bwd = float(np.mean(st["gs_blend_bwd"])) if "gs_blend_bwd" in st else None
    alg = 128 * walked + 24 * H * W
    print(json.dumps({"config": name, "list_mode": mode, "n_gaussians": int(sc["means"].shape[0]), "image": f"{W}x{H}", "n_isects_listed": n_isects,
                      "binning": rendering.last_binning(dev), "walked_isects": walked, "walked_quadrant_pairs": pairs, "work_units": int(dbg["unit_counter"][0]),
                      "stage_ms": {k[3:]: round(float(np.mean(v)), 4) for k, v in sorted(st.items())},
                      "blend_bwd_roofline": None if not t_bwd else {"unit": "walked intersections (those with gradient rows)", "algorithmic_bytes": alg,
                                                                    "achieved_GBps": round(alg / t_bwd / 1e6, 1), "frac_of_8TBps": round(alg / t_bwd / 1e6 / 8000, 4)}}), flush=True)


if __name__ == "__main__":
    main()
