#!/bin/bash
# Runs tools/micro/traffic_cal under two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; the program sits
# directly after `--`) and joins them with the exact counts the program prints -> gpurun_out/<tag>_traffic_calibration.json
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp
BIN=$GRAFT_REPO_ROOT/tools/micro/traffic_cal
[ -x $BIN ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $BIN $GRAFT_REPO_ROOT/tools/micro/traffic_cal.hip
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/cal_$c -o $c -- $BIN > /tmp/cal_$c.log 2>&1
done
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
python3 - $tag <<'PY'
import csv, glob, json, os, sys
tag = sys.argv[1]
exact = None
for line in open("/tmp/cal_FETCH_SIZE.log"):
    if line.startswith("{"):
        exact = json.loads(line)
raw = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for path in glob.glob(f"/tmp/cal_{c}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            k = row["Kernel_Name"].split("(")[0].strip()
            if k.startswith("cal_") and row["Counter_Name"] == c:
                raw.setdefault(k, {}).setdefault(c, []).append(float(row["Counter_Value"]))
out = {}
for k, e in (exact or {}).items():
    r = raw.get(k, {})
    f = 1024.0 * sum(r.get("FETCH_SIZE", [0])) / max(len(r.get("FETCH_SIZE", [])), 1)
    w = 1024.0 * sum(r.get("WRITE_SIZE", [0])) / max(len(r.get("WRITE_SIZE", [])), 1)
    o = dict(e, fetch_raw_bytes=f, write_raw_bytes=w)
    ref = f if ("read" in k or "gather" in k) else w
    for g in ("bytes", "sectors32", "sectors64", "lines128"):
        if g in e:
            mult = {"bytes": 1, "sectors32": 32, "sectors64": 64, "lines128": 128}[g]
            o[f"raw_over_{g}"] = round(ref / (e[g] * mult), 4)
    out[k] = o
json.dump(out, open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", f"{tag}_traffic_calibration.json"), "w"), indent=1)
print(json.dumps({k: {g: v for g, v in o.items() if g.startswith("raw_over")} for k, o in out.items()}, indent=1))
PY
