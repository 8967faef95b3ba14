import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["GS_FUZZ_SCALE"] = "2"
import numpy as np, torch
import test_gpu_parity as TP
from easy_gaussian_splatting_amd.rendering import rasterization
case = 2084
sc, (deg, W, H, use_bg, split, culling) = TP.fuzz_case(case, 0)
t = TP.to_dev(sc)
_, _, meta = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["shs"], t["viewmats"], t["Ks"], W, H, sh_degree=deg, packed=False, _tile_culling=culling)
fw = TP.run_oracle(sc, use_bg=use_bg)
r = meta["radii"].cpu().numpy(); ro = fw["radii"]
idx = np.argwhere(r != ro)
print("flips", idx.tolist(), "W,H", W, H)
for c, i in idx:
    print("hip radius", r[c, i], "oracle", ro[c, i], "means2d", meta["means2d"][c, i].tolist(), fw["means2d"][c, i], "depth", fw["depths"][c, i], "conic", fw["conics"][c, i], "scale", sc["scales"][i], "quat", sc["quats"][i])
    # recompute 3*sqrt(lambda) in fp64 from the oracle's own conic
    A, B, Cc = [float(x) for x in fw["conics"][c, i]]
    det = A * Cc - B * B; a, b, cc = Cc / det, -B / det, A / det
    mid = 0.5 * (a + cc); lam = mid + np.sqrt(max(0.01, mid * mid - (a * cc - b * b)))
    print("3 sqrt(lam) =", repr(3 * np.sqrt(lam)))
