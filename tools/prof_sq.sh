#!/bin/bash
# SQ counters of our kernels over a short bench run (separate rocprofv3 --pmc passes, program directly after --):
#   tools/prof_sq.sh <tag>   -> gpurun_out/<tag>_sq_counters.json  (per kernel: per-launch averages)
tag=${1:-sq}
cd /tmp && export TMPDIR=/tmp
dirs=""
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA"; do
  i=$((i+1)); d=/tmp/${tag}_sq$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -o x -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --no-graph --steps 3 --warmup 2 > $d.log 2>&1 || tail -3 $d.log
  dirs="$dirs $d"
done
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py counters $dirs > $GRAFT_REPO_ROOT/gpurun_out/${tag}_sq_counters.json
python3 - $GRAFT_REPO_ROOT/gpurun_out/${tag}_sq_counters.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    if "blend" in k:
        print(k, {c: round(x) for c, x in v.items()})
PY
