"""is the per-Gaussian gradient criterion's tail ROUNDING or a defect?   (GPU box)
    tools/acc64_ab.py <out.json> [scale:case ...]
For each sweep configuration (tests/test_gpu_parity.py::fuzz_case at GS_FUZZ_SCALE = scale) the same backward is taken three ways
and compared, row by row (Gaussian by Gaussian), with the fp64 oracle:
  hip_fp32   the product library (fp32 sums in blend_bwd),
  hip_acc64  a -DGS_BWD_ACC64 build (the 11 per-entry sums in double; everything else unchanged),
  hip_exact  a -DGS_BWD_ACC64 -DGS_EXACT_MATH build (also exp2 and 1/x correctly rounded, in both blend kernels, instead of
             v_exp_f32 / v_rcp_f32),
  oracle_fp32  the fp32 build of the C oracle -- another fp32 evaluation order of the same algorithm.
A row is "bad" when one of its elements is beyond 1e-3 of max(the row's own largest reference magnitude, 1e-3 of the tensor's).
Rounding predicts: the bad rows of hip_fp32 have small reference magnitude relative to the tensor (cancellation), shrink or
vanish under acc64 AND/OR show up at the same order in oracle_fp32; a defect in the s_vs / v_op path would survive acc64 and be
absent from oracle_fp32.  Each library is loaded by a child process (GS_LIB_PATH)."""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
NAMES = ["v_means", "v_quats", "v_scales", "v_opacities", "v_colors"]


def child(scale, case):
    os.environ["GS_FUZZ_SCALE"] = str(scale)
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import torch
    import test_gpu_parity as P
    from easy_gaussian_splatting_amd.rendering import rasterization
    os.environ["GS_BINNING"] = ("tiles", "bins", "bins")[case % 3]
    os.environ["GS_BINS_SHIFT"] = ("", "1", "2")[case % 3]
    sc, (deg, W, H, use_bg, split, culling) = P.fuzz_case(case, 0)
    t = P.to_dev(sc)
    base = [t[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities")]
    sh = t["shs"].clone().requires_grad_(True)
    img, alpha, meta = rasterization(*base, sh, t["viewmats"], t["Ks"], W, H, sh_degree=deg, packed=False,
                                     backgrounds=t["backgrounds"] if use_bg else None, absgrad=True, _tile_culling=culling)
    fw = P.run_oracle(sc, use_bg=use_bg)
    rep = P.forward_report(meta, fw, lists=culling != "tight")
    g = torch.Generator().manual_seed(case)
    keep = torch.from_numpy(~rep["loose"])[..., None]
    vc, va = torch.randn(img.shape, generator=g) * keep, torch.randn(alpha.shape, generator=g) * keep
    grads = torch.autograd.grad((img * vc.to(P.dev())).sum() + (alpha * va.to(P.dev())).sum(), base + [sh])
    vc64, va64 = vc.numpy().astype(np.float64), va.numpy().astype(np.float64)
    bw = P.CO.backward(fw, vc64, va64)
    fw32 = P.oracle_fp32(fw)
    bw32 = P.CO.backward(fw32, vc64.astype(np.float32), va64.astype(np.float32))
    relax = P.needle_factor(fw)
    out = {"n": int(sc["means"].shape[0]), "razor": float(rep["razor"].mean()), "tensors": {}}
    for name, gt in zip(NAMES, grads):
        ref = np.asarray(bw[name], np.float64)
        n = ref.shape[0]
        r = np.abs(ref).reshape(n, -1)
        tmax = r.max(initial=0) + 1e-30
        tol = 1e-3 * np.maximum(r.max(axis=1, initial=0), 1e-3 * tmax)
        rl = relax[:, None] if name in ("v_quats", "v_scales") else 1.0
        d_hip = (np.abs(gt.cpu().numpy().astype(np.float64) - ref).reshape(n, -1) / rl).max(axis=1, initial=0)
        d_o32 = (np.abs(np.asarray(bw32[name], np.float64) - ref).reshape(n, -1) / rl).max(axis=1, initial=0)
        out["tensors"][name] = {"rows": n, "bad_hip": np.nonzero(d_hip > tol)[0].tolist(), "bad_oracle_fp32": np.nonzero(d_o32 > tol)[0].tolist(),
                                "err_over_tol_hip": (d_hip / tol).tolist(), "err_over_tol_oracle_fp32": (d_o32 / tol).tolist(),
                                "row_mag_over_tensor_max": (r.max(axis=1, initial=0) / tmax).tolist()}
    print("RESULT " + json.dumps(out), flush=True)


def run_child(lib, scale, case):
    env = dict(os.environ, GS_LIB_PATH=lib, GS_ALLOW_VARIANT="1")
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(scale), str(case)], env=env, capture_output=True, text=True, timeout=1200)
    for line in p.stdout.splitlines():
        if line.startswith("RESULT "):
            return json.loads(line[7:])
    raise RuntimeError(p.stderr[-2000:])


def main():
    out_path = sys.argv[1]
    cases = [tuple(int(x) for x in a.split(":")) for a in sys.argv[2:]] or [(2, 176), (2, 1444), (2, 104), (1, 730), (1, 113), (1, 444), (1, 254)]
    pkg = os.path.join(ROOT, "easy_gaussian_splatting_amd")
    variants = os.path.join(ROOT, "build", "variants")   # (_native.VARIANT_DIR; this parent process never loads the library)
    rows = []
    for scale, case in cases:
        a = run_child(os.path.join(pkg, "libgsraster.so"), scale, case)
        b = run_child(os.path.join(variants, "libgsraster_acc64.so"), scale, case)
        c = run_child(os.path.join(variants, "libgsraster_exact.so"), scale, case)
        rec = {"scale": scale, "case": case, "n_gaussians": a["n"], "razor_fraction": round(a["razor"], 4), "tensors": {}}
        for name in NAMES:
            ta, tb, tc = a["tensors"][name], b["tensors"][name], c["tensors"][name]
            bad = ta["bad_hip"]
            rec["tensors"][name] = {
                "rows": ta["rows"], "bad_rows_fp32_sums": len(bad), "bad_rows_fp64_sums": len(tb["bad_hip"]),
                "bad_rows_fp64_sums_exact_exp2_rcp": len(tc["bad_hip"]),
                "bad_rows_oracle_fp32": len(ta["bad_oracle_fp32"]),
                # the product's bad rows, one by one: how small the row is within its tensor, and its error (in units of the
                # tolerance) with fp32 sums / fp64 sums / in the fp32 oracle
                "the_bad_rows": [{"row": i, "row_mag_over_tensor_max": float("%.3g" % ta["row_mag_over_tensor_max"][i]),
                                  "err_over_tol_fp32_sums": float("%.3g" % ta["err_over_tol_hip"][i]),
                                  "err_over_tol_fp64_sums": float("%.3g" % tb["err_over_tol_hip"][i]),
                                  "err_over_tol_fp64_sums_exact_exp2_rcp": float("%.3g" % tc["err_over_tol_hip"][i]),
                                  "err_over_tol_oracle_fp32": float("%.3g" % ta["err_over_tol_oracle_fp32"][i])} for i in bad[:40]]}
        rows.append(rec)
        print(json.dumps({k: (v if k != "tensors" else {n: {kk: vv for kk, vv in t.items() if kk != "the_bad_rows"} for n, t in v.items()}) for k, v in rec.items()}), flush=True)
    tot = lambda key: sum(t[key] for r in rows for t in r["tensors"].values())   # noqa: E731
    summary = {"bad_rows_fp32_sums": tot("bad_rows_fp32_sums"), "bad_rows_fp64_sums": tot("bad_rows_fp64_sums"),
               "bad_rows_fp64_sums_exact_exp2_rcp": tot("bad_rows_fp64_sums_exact_exp2_rcp"),
               "bad_rows_oracle_fp32": tot("bad_rows_oracle_fp32"),
               "largest_bad_row_magnitude_over_tensor_max": max([b["row_mag_over_tensor_max"] for r in rows for t in r["tensors"].values() for b in t["the_bad_rows"]] or [0.0])}
    json.dump({"what": __doc__, "summary": summary, "cases": rows}, open(out_path, "w"), indent=1)
    print("SUMMARY", json.dumps(summary))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]))
    else:
        main()
