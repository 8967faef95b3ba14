#!/bin/bash
# HBM traffic per launch of the kernels of the CAPTURED step (project_bwd_kernel<DEG, true> = fused Adam etc.): two rocprofv3
# --pmc passes (FETCH_SIZE, WRITE_SIZE) over a short graph bench -> gpurun_out/<tag>_pmc_traffic.json
tag=${1:-pmcg}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/${tag}_$c -o $c -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 3 > /tmp/${tag}_$c.log 2>&1 || tail -3 /tmp/${tag}_$c.log
done
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py traffic /tmp/${tag}_FETCH_SIZE /tmp/${tag}_WRITE_SIZE > $GRAFT_REPO_ROOT/gpurun_out/${tag}_pmc_traffic.json
python3 - $GRAFT_REPO_ROOT/gpurun_out/${tag}_pmc_traffic.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    if any(x in k for x in ("project_bwd", "project_fwd", "adam")):
        print(k[:60], {c: round(x / 1e6, 1) for c, x in v.items() if "bytes" in c}, v.get("launches_fetch"))
PY
