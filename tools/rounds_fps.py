#!/usr/bin/env python3
"""Forward render fps (the reference's definition, eval.py:38-43, 70: one synchronised render per frame through the model) of a
workload of tools/config_run.py with the list stages in one round and in depth rounds (rendering.py: GS_ROUNDS):
    python tools/rounds_fps.py heavy2M [tight|gsplat|gsplat_eager] [frames]      ->  one JSON line per variant"""
import json, os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
from config_run import MAKE, model_from_scene
from easy_gaussian_splatting_amd import rendering


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "heavy2M"
    mode = sys.argv[2] if len(sys.argv) > 2 else "tight"
    frames = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    dev = torch.device("cuda:0")
    sc = MAKE[name]()
    W, H = int(sc["width"]), int(sc["height"])
    model = model_from_scene(sc, dev)
    model.tile_culling = mode
    data = {"w2c": torch.from_numpy(sc["viewmats"][0]).to(dev), "K": torch.from_numpy(sc["Ks"][0]).to(dev), "width": W, "height": H}
    ref = None
    for variant in ("auto", "off", "on", "off"):
        os.environ["GS_ROUNDS"] = variant
        rendering.reset_hints()
        with torch.no_grad():
            for _ in range(5):
                img = model(data)["render_img"]
            torch.cuda.synchronize()
            n0 = rendering.stats["round_calls"]
            t0 = time.perf_counter()
            for _ in range(frames):
                img = model(data)["render_img"]
                torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            rendering.profile_stages(True)
            for _ in range(10):
                model(data)
            st = rendering.profile_stages(False) or {}
        ref = img.clone() if ref is None else ref
        print(json.dumps({"config": name, "list_mode": mode, "rounds": variant, "fps": round(frames / dt, 1), "ms": round(1e3 * dt / frames, 4),
                          "two_round_calls": rendering.stats["round_calls"] - n0, "of": frames, "same_image": bool(torch.equal(img, ref)),
                          "binning": rendering.last_binning(dev),
                          "stage_ms_per_frame": {k[3:]: round(float(np.sum(v)) / 10, 4) for k, v in sorted(st.items())}}), flush=True)


if __name__ == "__main__":
    main()
