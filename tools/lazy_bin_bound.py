#!/usr/bin/env python3
"""Bound of a refine-on-demand (lazy) binning pipeline, measured without building it.

On realistic footprints the list stages are the largest part of the step and 97-98 % of what they sort is never read: a tile
saturates a few hundred entries into a list of thousands.  A lazy pipeline would partition every bin's entries by depth into
slabs (ONE pass over the coarse keys), sort + refine only the front slab, blend, and come back for the next slab only where a
tile has not saturated.  What the front slab would cost is measured here on the product's own kernels: the same scene cut down
to the nearest 1/4, 1/8, 1/16 of its Gaussians (a depth slab of the whole scene), through the same two-level binning -- plus a
modelled partition pass (16 bytes per coarse key at a measured stream rate).  Also reported: how much of the image the front
slab alone already finishes (pixels whose final transmittance equals the full render's: those tiles never come back).

    python tools/lazy_bin_bound.py [heavy2M|heavy1M|S3] [tight|gsplat_eager]   ->  one JSON line
"""
import json, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
from config_run import MAKE
from easy_gaussian_splatting_amd import rendering

STREAM_GBPS = 5000.0   # what the library's plain streaming passes reach on MI355X (adam_step_kernel: 6.4 TB/s; a key partition writes scattered runs)


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "heavy2M"
    mode = sys.argv[2] if len(sys.argv) > 2 else "tight"
    dev = torch.device("cuda:0")
    sc = MAKE[name]()
    W, H = int(sc["width"]), int(sc["height"])
    t = {k: torch.from_numpy(v).to(dev) for k, v in sc.items() if isinstance(v, np.ndarray)}
    depth = (t["means"] @ t["viewmats"][0, :3, :3].T + t["viewmats"][0, :3, 3])[:, 2]
    order = torch.argsort(depth)
    rows, full_alpha = [], None
    for frac in (1.0, 0.25, 0.125, 0.0625):
        keep = order[: int(frac * order.numel())]
        ins = [t[k][keep].contiguous() for k in ("means", "quats", "scales", "opacities", "shs")]
        render = lambda: rendering.rasterization(*ins, t["viewmats"], t["Ks"], W, H, sh_degree=int(sc["sh_degree"]), packed=False,
                                                 backgrounds=t["backgrounds"], _tile_culling=mode)
        with torch.no_grad():
            for _ in range(3):
                img, alpha, meta = render()
            rendering.profile_stages(True)
            for _ in range(8):
                render()
            st = rendering.profile_stages(False)
            n_listed = int(meta["flatten_ids"].numel())
        ms = {k[3:]: round(float(np.mean(v)), 4) for k, v in st.items()}
        if full_alpha is None:
            full_alpha = alpha.clone()
        done = float((alpha >= full_alpha - 1e-6).float().mean())   # pixels the slab alone brings to the full render's opacity
        rows.append({"nearest_fraction": frac, "n_gaussians": int(keep.numel()), "n_isects_listed": n_listed, "binning": rendering.last_binning(dev),
                     "bin_count_ms": ms.get("bin_count"), "bin_emit_sort_ms": ms.get("bin_emit_sort"), "blend_fwd_ms": ms.get("blend_fwd"),
                     "list_stages_ms": round(ms.get("bin_count", 0) + ms.get("bin_emit_sort", 0), 4), "pixels_finished_by_this_slab": round(done, 4)})
        del ins
        rendering.reset_hints()
        torch.cuda.empty_cache()
    full = rows[0]
    coarse_keys = None   # (not exposed by the eager seam: the partition pass is priced on the LISTED entries / 4 as a floor and on all of them as a ceiling)
    out = {"config": name, "list_mode": mode, "image": f"{W}x{H}", "rows": rows, "partition_pass_model_ms": {
        "floor": round(16 * full["n_isects_listed"] / 4 / (STREAM_GBPS * 1e6), 4), "ceiling": round(16 * full["n_isects_listed"] / (STREAM_GBPS * 1e6), 4),
        "note": "one pass over the coarse keys (16 B each, between I / 4 and I of them) at 5 TB/s"}}
    for r in rows[1:]:
        lo = r["list_stages_ms"] + out["partition_pass_model_ms"]["floor"]
        hi = r["list_stages_ms"] + out["partition_pass_model_ms"]["ceiling"]
        r["lazy_first_round_ms"] = [round(lo, 4), round(hi, 4)]
        r["saving_vs_full_ms"] = [round(full["list_stages_ms"] - hi, 4), round(full["list_stages_ms"] - lo, 4)]
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
