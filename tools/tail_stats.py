#!/usr/bin/env python3
"""How full the backward's work units are (a work unit = 32 entries of one quadrant sublist; the last unit of a sublist -- its
tail -- is part-filled, and `blend_bwd_kernel` passes over all 32 slots of every unit):  tools/tail_stats.py [bench1M|heavy1M|heavy2M|S3]
Prints, per workload: sublists, work units, the share of empty entry slots, the tails by fill class (<= 8 / 16 / 24 / 32 entries),
and the pair work a backward with 1 / 2 / 3 / 4 entries per lane for units of those classes would save (an upper bound: the
per-step overhead of a unit does not shrink).  Bound first, build second (HISTORY.md)."""
import json, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
from easy_gaussian_splatting_amd import rendering, synthetic
from easy_gaussian_splatting_amd import _native as nat

name = sys.argv[1] if len(sys.argv) > 1 else "bench1M"
mode = sys.argv[2] if len(sys.argv) > 2 else "tight"
dev = torch.device("cuda:0")
if name == "bench1M":
    sc = synthetic.config_bench_1m(seed=42, n=1_000_000, n_views=1)
elif name == "heavy1M":
    sc = synthetic.config_heavy(n=1_000_000)
elif name == "heavy2M":
    sc = synthetic.config_heavy(n=2_000_000)
else:
    sc = synthetic.config_s3()
t = {k: torch.from_numpy(v).to(dev) for k, v in sc.items() if isinstance(v, np.ndarray)}
W, H = int(sc["width"]), int(sc["height"])
ins = [t[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "shs")]
dbg = {}
img, alpha, meta = rendering.rasterization(*ins, t["viewmats"][:1], t["Ks"][:1], W, H, sh_degree=int(sc["sh_degree"]), packed=False,
                                           backgrounds=t["backgrounds"][:1], absgrad=True, _tile_culling=mode, _debug=dbg)
torch.cuda.synchronize()
U = int(nat.GS_UNIT)
L = dbg["qcnt"].cpu().numpy().astype(np.int64)
L = L[L > 0]
units = int(np.ceil(L / U).sum())
tail = L - (np.ceil(L / U).astype(np.int64) - 1) * U          # entries in the last unit of each sublist, 1 .. U
cls = np.ceil(tail / (U // 4)).astype(np.int64)                # 1 .. 4 entries per lane would do
out = {"config": name, "list_mode": mode, "n_isects": int(meta["flatten_ids"].numel()), "sublists": int(L.size), "rows": int(L.sum()),
       "work_units": units, "work_units_counter": int(dbg["unit_counter"][0]), "mean_sublist": round(float(L.mean()), 1),
       "empty_slot_share": round(1.0 - float(L.sum()) / (units * U), 4),
       "tails_by_class": {f"<= {8 * c}": int((cls == c).sum()) for c in (1, 2, 3, 4)},
       "pair_work_saved_2_classes": round(float((cls <= 2).sum() * 0.5) / units, 4),
       "pair_work_saved_4_classes": round(float(((4 - cls) / 4.0).sum()) / units, 4)}
print(json.dumps(out), flush=True)
