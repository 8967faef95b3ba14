#!/usr/bin/env python3
"""Which tiles of the S5 workload (5 M / 4K) hold a different list length than the oracle's, and which Gaussians cause it."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from test_gpu_parity import run_hip, run_oracle, _affected_tiles
from scenes import config_s5
sc = config_s5()
fw = run_oracle(sc)
hip = run_hip(sc, bwd=False)
meta = hip["meta"]
radii = meta["radii"].cpu().numpy()
mism = radii != fw["radii"]
tpg_h, tpg_o = meta["tiles_per_gauss"].cpu().numpy(), fw["tiles_per_gauss"]
edge = (tpg_h != tpg_o) & ~mism
print("radius flips", int(mism.sum()), "rect flips", int(edge.sum()))
differ = mism | edge
tmask = _affected_tiles(meta, fw, differ)
off_h = meta["isect_offsets"].reshape(-1).cpu().numpy().astype(np.int64)
fid_h = meta["flatten_ids"].cpu().numpy()
cnt_h = np.diff(np.append(off_h, fid_h.size))
off_o = fw["isect_offsets"].reshape(-1).astype(np.int64)
cnt_o = np.diff(np.append(off_o, fw["n_isects"]))
bad = np.nonzero((cnt_h != cnt_o) & ~tmask.reshape(-1))[0]
print("tiles with different length outside the exempt set:", bad.size, bad[:20], "I hip/oracle", fid_h.size, fw["n_isects"])
tw = fw["tile_width"]
for t in bad[:10]:
    sh = set(fid_h[off_h[t]: off_h[t] + cnt_h[t]].tolist()); so = set(fw["flatten_ids"][off_o[t]: off_o[t] + cnt_o[t]].tolist())
    print("tile", t, "(x, y)", t % tw, t // tw, "len", cnt_h[t], cnt_o[t], "only hip", sorted(sh - so)[:5], "only oracle", sorted(so - sh)[:5])
    for g in list(sh ^ so)[:5]:
        m_h, m_o = meta["means2d"][0, g].cpu().numpy(), fw["means2d"][0, g]
        print("   g", g, "radius", radii[0, g], fw["radii"][0, g], "tiles", tpg_h[0, g], tpg_o[0, g], "mean hip", m_h, "oracle", m_o,
              "edges/16 oracle", (m_o - fw["radii"][0, g]) / 16, (m_o + fw["radii"][0, g]) / 16, "in differ", bool(differ[0, g]))
