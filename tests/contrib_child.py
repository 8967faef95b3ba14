"""Child process of tests/test_gpu_contributors.py: loads the -DGS_BWD_CHECK build of the library (GS_LIB_PATH, set by the
parent before this process imports the package) and compares, per pixel, what the forward and the backward of the same call
saw: final transmittance and number of contributors.  Prints one JSON line per scene."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)

from easy_gaussian_splatting_amd import _native as nat   # noqa: E402
from easy_gaussian_splatting_amd.rendering import rasterization   # noqa: E402
from scenes import config_long_lists, dense_scene, make_scene   # noqa: E402


def check(name, sc, culling):
    dev = torch.device("cuda:0")
    L = nat.lib()
    set_fn = L.gs_debug_bwd_check_set          # only the check build exports it (AttributeError otherwise: wrong library)
    set_fn.restype = int
    import ctypes as ct
    set_fn.argtypes = [ct.c_void_p] * 4
    t = {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).to(dev) for k, v in sc.items() if isinstance(v, np.ndarray)}
    W, H, C = int(sc["width"]), int(sc["height"]), t["viewmats"].shape[0]
    tw, th = (W + 15) // 16, (H + 15) // 16
    ins = [t[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "shs")]
    kw = dict(sh_degree=int(sc["sh_degree"]), packed=False, backgrounds=t["backgrounds"], absgrad=True, _tile_culling=culling)
    with torch.no_grad():   # sizes first
        _, _, meta0 = rasterization(*[x.detach() for x in ins], t["viewmats"], t["Ks"], W, H, **kw)
        n_isects = int(meta0["flatten_ids"].numel())
    max_units = 8 * (n_isects // 64 + C * tw * th + 1)
    fwd_T = torch.full((C, H, W), -7.0, device=dev)
    fwd_cnt = torch.full((C, H, W), -7, dtype=torch.int32, device=dev)
    unit_out = torch.full((max_units, 64, 2), -9.0, device=dev)
    unit_hdr = torch.full((max_units, 2), -1, dtype=torch.int32, device=dev)
    set_fn(fwd_T.data_ptr(), fwd_cnt.data_ptr(), unit_out.data_ptr(), unit_hdr.data_ptr())
    dbg = {}
    img, alpha, meta = rasterization(*ins, t["viewmats"], t["Ks"], W, H, _debug=dbg, **kw)
    vc = torch.randn(img.shape, generator=torch.Generator().manual_seed(0)).to(dev)
    torch.autograd.grad((img * vc).sum(), ins)
    torch.cuda.synchronize()
    set_fn(None, None, None, None)
    n_units = int(dbg["unit_counter"].item())
    hdr = unit_hdr[:n_units].cpu().numpy().astype(np.int64)
    out = unit_out[:n_units].cpu().numpy()
    assert n_units <= max_units and (hdr[:, 0] >= 0).all(), "a work unit left no header"
    order = np.lexsort((hdr[:, 1], hdr[:, 0]))           # by sublist (tile * 4 + quadrant), then position in the sublist
    fT, fC = fwd_T.cpu().numpy(), fwd_cnt.cpu().numpy()
    # chain the units of every sublist: final T = the last value that is not the "finished before this unit" marker; counts add
    bT = np.ones((C, H, W), np.float32)
    bC = np.zeros((C, H, W), np.int64)
    tiles = tw * th
    for u in order:
        tq = hdr[u, 0]
        tile, q = tq // 4, tq % 4
        cam, tt = tile // tiles, tile % tiles
        y0, x0 = (tt // tw) * 16 + 8 * (q >> 1), (tt % tw) * 16 + 8 * (q & 1)
        blkT = out[u, :, 0].reshape(8, 8); blkC = out[u, :, 1].reshape(8, 8)
        ys, xs = min(8, H - y0), min(8, W - x0)
        if ys <= 0 or xs <= 0:
            continue
        written = blkT[:ys, :xs] != -9.0
        assert written.all(), "a pixel inside the image left no record"
        live = blkT[:ys, :xs] >= 0.0
        sub = bT[cam, y0:y0 + ys, x0:x0 + xs]
        sub[live] = blkT[:ys, :xs][live]
        bC[cam, y0:y0 + ys, x0:x0 + xs] += blkC[:ys, :xs].astype(np.int64)
    same_T = fT.view(np.uint32) == bT.view(np.uint32)
    same_C = fC.astype(np.int64) == bC
    res = {"scene": name, "culling": culling, "pixels": int(fT.size), "n_isects": n_isects, "work_units": n_units,
           "T_mismatches": int((~same_T).sum()), "count_mismatches": int((~same_C).sum()),
           "mean_contributors": float(bC.mean()), "max_contributors": int(bC.max()),
           "saturated_fraction": float((fT < 1e-2).mean()),   # (a stopped pixel keeps the T in FRONT of the entry that would take it to 1e-4)
           "alpha_consistent": bool(np.array_equal((1.0 - fT).astype(np.float32), alpha.detach().cpu().numpy()[..., 0]))}
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    check("sh3_two_views", make_scene(20000, 320, 208, sh_degree=3, n_views=2, seed=5, scale_range=(0.01, 0.08), dist=4.0), "gsplat_eager")
    check("sh3_two_views", make_scene(20000, 320, 208, sh_degree=3, n_views=2, seed=5, scale_range=(0.01, 0.08), dist=4.0), "tight")
    check("ragged_faint_long_lists", dense_scene(6000, seed=3, width=77, height=53), "gsplat_eager")
    check("saturated_heavy_tailed", config_long_lists(seed=1, n=45_000, width=640, height=368), "tight")
