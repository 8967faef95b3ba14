#!/bin/bash
# SQ counters of the kernels of the CAPTURED step (project_bwd with fused Adam etc.): two rocprofv3 --pmc passes over a short graph bench
tag=${1:-sqg}
cd /tmp && export TMPDIR=/tmp
dirs=""; i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  i=$((i+1)); d=/tmp/${tag}_sq$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -o x -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 3 > $d.log 2>&1 || tail -3 $d.log
  dirs="$dirs $d"
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py counters $dirs > $GRAFT_REPO_ROOT/gpurun_out/${tag}_sq_counters.json
python3 - $GRAFT_REPO_ROOT/gpurun_out/${tag}_sq_counters.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    if any(x in k for x in ("project_bwd", "project_fwd", "l1_ssim", "radix", "emit")):
        print(k[:60], {c: round(x) for c, x in v.items()})
PY
