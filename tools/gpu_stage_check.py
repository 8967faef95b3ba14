"""Stage-by-stage comparison of the HIP path against the C oracle on the GPU box (debug aid)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from scenes import make_scene, config_s1, config_bench_1m
from oracle import c_oracle as CO
from easy_gaussian_splatting_amd.rendering import rasterization

dev = torch.device("cuda:0")

def run(sc, name, bwd=True, oracle=True):
    t = {k: torch.from_numpy(v).to(dev) for k, v in sc.items() if isinstance(v, np.ndarray)}
    W, H, deg = sc["width"], sc["height"], sc["sh_degree"]
    ins = [t[k].clone().requires_grad_(bwd) for k in ("means", "quats", "scales", "opacities", "shs")]
    dbg = {}
    img, alpha, meta = rasterization(*ins, t["viewmats"], t["Ks"], W, H, sh_degree=deg, packed=False,
                                     backgrounds=t["backgrounds"], absgrad=True, _debug=dbg)
    torch.cuda.synchronize()
    print(f"== {name}: N={ins[0].shape[0]} {W}x{H} deg={deg} C={t['viewmats'].shape[0]} I={meta['flatten_ids'].shape[0]} vis={(meta['radii']>0).sum().item()}")
    if oracle:
        fw = CO.render(sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["shs"], sc["viewmats"], sc["Ks"], W, H,
                       sh_degree=deg, backgrounds=sc["backgrounds"], dtype=np.float64)
        d = lambda a, b: float(np.abs(a.detach().cpu().numpy().astype(np.float64) - b).max()) if a.numel() else 0.0
        rad_ok = (meta["radii"].cpu().numpy() == fw["radii"]).all()
        print("  radii equal:", rad_ok, " n mismatch:", int((meta["radii"].cpu().numpy() != fw["radii"]).sum()))
        print("  means2d", d(meta["means2d"], fw["means2d"]), "depths", d(meta["depths"], fw["depths"]),
              "conics", d(meta["conics"], fw["conics"]), "tpg", d(meta["tiles_per_gauss"], fw["tiles_per_gauss"]))
        print("  I oracle", fw["n_isects"])
        if meta["flatten_ids"].shape[0] == fw["n_isects"]:
            print("  offsets eq", bool((meta["isect_offsets"].cpu().numpy() == fw["isect_offsets"]).all()),
                  "flatten eq", bool((meta["flatten_ids"].cpu().numpy() == fw["flatten_ids"]).all()),
                  "isect_ids eq", bool((meta["isect_ids"].cpu().numpy() == fw["isect_ids"]).all()))
        print("  img", d(img, fw["render_colors"]), "alpha", d(alpha, fw["render_alphas"]))
    if bwd:
        g = torch.Generator(device="cpu").manual_seed(1)
        vc = torch.randn(img.shape, generator=g).to(dev); va = torch.randn(alpha.shape, generator=g).to(dev)
        grads = torch.autograd.grad((img * vc).sum() + (alpha * va).sum(), ins)
        torch.cuda.synchronize()
        if oracle:
            bw = CO.backward(fw, vc.cpu().numpy().astype(np.float64), va.cpu().numpy().astype(np.float64))
            rel = lambda a, b: float(np.abs(a.detach().cpu().numpy().astype(np.float64) - b).max() / (np.abs(b).max() + 1e-30))
            print("  [2D] v_means2d", rel(dbg["v_means2d"], bw["v_means2d"]), "v_conics", rel(dbg["v_conics"], bw["v_conics"]),
                  "v_colors_post", rel(dbg["v_colors_post"], bw["v_colors_post"]), "absgrad", rel(meta["means2d"].absgrad, bw["v_means2d_abs"]))
            for nme, gg in zip(["v_means", "v_quats", "v_scales", "v_opacities", "v_colors"], grads):
                print("  ", nme, "rel-max", rel(gg, bw[nme]), "max", float(np.abs(bw[nme]).max()))
    return img

def timeit(sc, name, iters=20):
    t = {k: torch.from_numpy(v).to(dev) for k, v in sc.items() if isinstance(v, np.ndarray)}
    W, H, deg = sc["width"], sc["height"], sc["sh_degree"]
    ins = [t[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "shs")]
    with torch.no_grad():
        for _ in range(3):
            img, a, meta = rasterization(*ins, t["viewmats"], t["Ks"], W, H, sh_degree=deg, packed=False, backgrounds=t["backgrounds"], absgrad=True)
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(iters):
            img, a, meta = rasterization(*ins, t["viewmats"], t["Ks"], W, H, sh_degree=deg, packed=False, backgrounds=t["backgrounds"], absgrad=True)
        torch.cuda.synchronize(); fwd = (time.time() - t0) / iters
    vc = torch.randn_like(img) / (W * H)
    for _ in range(3):
        img, a, meta = rasterization(*ins, t["viewmats"], t["Ks"], W, H, sh_degree=deg, packed=False, backgrounds=t["backgrounds"], absgrad=True)
        (img * vc).sum().backward()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(iters):
        img, a, meta = rasterization(*ins, t["viewmats"], t["Ks"], W, H, sh_degree=deg, packed=False, backgrounds=t["backgrounds"], absgrad=True)
        (img * vc).sum().backward()
    torch.cuda.synchronize(); fb = (time.time() - t0) / iters
    print(f"## {name}: I={meta['flatten_ids'].shape[0]} vis={(meta['radii']>0).sum().item()} fwd {fwd*1e3:.3f} ms ({1/fwd:.0f} fps)  fwd+bwd {fb*1e3:.3f} ms ({1/fb:.0f} it/s)")

if __name__ == "__main__":
    print(torch.cuda.get_device_name(0))
    run(make_scene(50, 40, 24, sh_degree=3, seed=1, scale_range=(0.05, 0.4), dist=4.0), "tiny")
    run(make_scene(2000, 100, 70, sh_degree=2, seed=2, k_store=16, n_views=2, scale_range=(0.02, 0.3), dist=4.0), "2view-deg2")
    run(config_s1(), "S1")
    run(make_scene(30000, 320, 200, sh_degree=1, seed=3, k_store=4, scale_range=(0.01, 0.6), dist=4.0), "big-splats")
    for nm, sc in [("1M-1080p", config_bench_1m())]:
        timeit(sc, nm)
