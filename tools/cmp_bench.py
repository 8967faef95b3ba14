import json,sys
for f in sys.argv[1:]:
    d=json.load(open(f))
    print(f, "it/s", d["value"], "fps", d.get("forward_fps"))
    print("  stage", {k:v for k,v in d["stage_ms"].items() if "bin" in k or "project_fwd" in k})
    ll=d.get("long_lists",{})
    for m in ("gsplat_eager","tight"):
        if m in ll: print("  long", m, ll[m]["n_isects"], ll[m]["fwd_bwd_ms"]["median"], {k:v for k,v in ll[m]["stage_ms"].items() if "bin" in k})
