#!/usr/bin/env python3
"""Pins the oracle to gsplat ITSELF -- the one thing this container cannot do (gsplat is not vendored under /root/reference, not
installed, not installable: /root/reference/requirements.txt:1, README.md:16).  Run this wherever `gsplat==1.0.0` and a CUDA
(or HIP-enabled) torch exist -- never on the GPU box of this project, which has neither gsplat nor network:

    pip install gsplat==1.0.0
    python tests/golden/make_gsplat_golden.py            # writes tests/golden/gsplat_<tag>.npz, a few 100 kB each
    git add tests/golden/gsplat_*.npz

It calls EXACTLY what the reference calls (/root/reference/model/gaussian.py:353-367: `rasterization(..., packed=False,
absgrad=True, sh_degree=..., backgrounds=...)`) on the INPUTS of the committed oracle fixtures (oracle_scene_a..f: same seeds,
same v_render_*), on S1 (configs[0]: 10 k Gaussians, 256x256, SH0) and on one heavy-tailed scene, and stores gsplat's

    render_colors, render_alphas, radii, means2d, depths, conics, tiles_per_gauss, isect_offsets, flatten_ids,
    v_means, v_quats, v_scales, v_opacities, v_shs, absgrad        (for the stored v_render_colors / v_render_alphas)

Consumers (they skip while no gsplat_*.npz exists, and run the day the files are committed):
    tests/test_oracle.py::test_oracle_matches_gsplat_fixtures            (CPU: the C oracle, fp32 and fp64, against gsplat)
    tests/test_gpu_parity.py::test_hip_path_matches_gsplat_fixtures      (-m gpu: the HIP path through the C ABI against gsplat)
Only data is stored: inputs and gsplat's outputs as arrays.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def scenes():
    """tag -> dict of float32 inputs + width / height / sh_degree + v_render_colors / v_render_alphas."""
    from easy_gaussian_splatting_amd.synthetic import config_long_lists, config_s1
    out = {}
    for tag in "abcdef":   # the committed oracle fixtures: identical inputs and upstream gradients
        z = np.load(os.path.join(HERE, f"oracle_scene_{tag}.npz"))
        out[tag] = {k: z[k] for k in ("means", "quats", "scales", "opacities", "shs", "viewmats", "Ks", "backgrounds", "v_render_colors",
                                      "v_render_alphas")}
        out[tag].update(width=int(z["width"]), height=int(z["height"]), sh_degree=int(z["sh_degree"]))
    for tag, sc in (("s1", config_s1()), ("heavy", config_long_lists(seed=1, n=6000, width=320, height=192))):
        rng = np.random.default_rng(7)
        C, H, W = sc["viewmats"].shape[0], sc["height"], sc["width"]
        sc = dict(sc)
        sc["v_render_colors"] = rng.standard_normal((C, H, W, 3)) / (H * W)
        sc["v_render_alphas"] = rng.standard_normal((C, H, W, 1)) / (H * W)
        out[tag] = sc
    return out


def main():
    import torch
    import gsplat
    from gsplat.rendering import rasterization
    if not gsplat.__version__.startswith("1.0."):
        print(f"warning: gsplat {gsplat.__version__} -- the reference pins 1.0.0 (README.md:16); meta['radii'] must be [C, N]", file=sys.stderr)
    dev = torch.device("cuda")
    for tag, sc in scenes().items():
        T = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device=dev)
        ins = [T(sc[k]).requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "shs")]
        img, alpha, meta = rasterization(ins[0], ins[1], ins[2], ins[3], ins[4], T(sc["viewmats"]), T(sc["Ks"]), int(sc["width"]), int(sc["height"]),
                                         sh_degree=int(sc["sh_degree"]), packed=False, backgrounds=T(sc["backgrounds"]), absgrad=True)
        meta["means2d"].retain_grad()
        vc, va = T(sc["v_render_colors"]), T(sc["v_render_alphas"])
        grads = torch.autograd.grad((img * vc).sum() + (alpha * va).sum(), ins, retain_graph=False)
        npy = lambda t: t.detach().cpu().numpy()
        out = dict(gsplat_version=gsplat.__version__, torch_version=torch.__version__, device_name=torch.cuda.get_device_name(0),
                   width=int(sc["width"]), height=int(sc["height"]), sh_degree=int(sc["sh_degree"]),
                   **{k: np.asarray(sc[k], dtype=np.float32) for k in ("means", "quats", "scales", "opacities", "shs", "viewmats", "Ks", "backgrounds")},
                   v_render_colors=np.asarray(sc["v_render_colors"]), v_render_alphas=np.asarray(sc["v_render_alphas"]),
                   render_colors=npy(img), render_alphas=npy(alpha), radii=npy(meta["radii"]), means2d=npy(meta["means2d"]),
                   depths=npy(meta["depths"]), conics=npy(meta["conics"]), tiles_per_gauss=npy(meta["tiles_per_gauss"]),
                   isect_offsets=npy(meta["isect_offsets"]), flatten_ids=npy(meta["flatten_ids"]),
                   v_means=npy(grads[0]), v_quats=npy(grads[1]), v_scales=npy(grads[2]), v_opacities=npy(grads[3]), v_shs=npy(grads[4]),
                   absgrad=npy(meta["means2d"].absgrad))
        path = os.path.join(HERE, f"gsplat_{tag}.npz")
        np.savez_compressed(path, **out)
        print(path, os.path.getsize(path), "bytes; I =", out["flatten_ids"].shape[0])


if __name__ == "__main__":
    main()
