#!/bin/bash
# SQ counters of our kernels over a short bench run (separate rocprofv3 --pmc passes, program directly after --):
#   tools/prof_sq.sh <tag> [tool.py args...]  -> gpurun_out/<tag>_sq_counters.json  (per kernel: per-launch averages; default
#   program: the bench; GS_SQ_EXTRA="C1 C2 C3 C4": one more counter set, GS_SQ_SHOW: substring of the kernels to print)
tag=${1:-sq}; shift
prog="bench.py --no-cpu-baseline --no-extras --no-graph --steps 3 --warmup 2"
[ $# -gt 0 ] && prog="$*"
cd /tmp && export TMPDIR=/tmp
dirs=""
i=0
sets=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA")
[ -n "$GS_SQ_EXTRA" ] && sets+=("$GS_SQ_EXTRA")
for set in "${sets[@]}"; do
  i=$((i+1)); d=/tmp/${tag}_sq$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -o x -- python3 $GRAFT_REPO_ROOT/$prog > $d.log 2>&1 || tail -3 $d.log
  dirs="$dirs $d"
done
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py counters $dirs > $GRAFT_REPO_ROOT/gpurun_out/${tag}_sq_counters.json
python3 - $GRAFT_REPO_ROOT/gpurun_out/${tag}_sq_counters.json <<'PY'
import json, os, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    if os.environ.get("GS_SQ_SHOW", "blend") in k:
        print(k, {c: round(x) for c, x in v.items()})
PY
