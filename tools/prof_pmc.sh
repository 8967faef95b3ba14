#!/bin/bash
# HBM traffic per launch of our kernels: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over a
# short bench run, summarised by tools/pmc_summary.py -> gpurun_out/<tag>_pmc_traffic.json  (run via gpurun)
#   tools/prof_pmc.sh <tag> [script.py args...]     default: the bench;  e.g. tools/prof_pmc.sh ll tools/long_lists_run.py tight 3
tag=${1:-pmc}; shift
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  if [ $# -gt 0 ]; then
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/${tag}_$c -o $c -- python3 $GRAFT_REPO_ROOT/$@ > /tmp/${tag}_$c.log 2>&1
  else
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/${tag}_$c -o $c -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --no-graph --steps 4 --warmup 2 > /tmp/${tag}_$c.log 2>&1
  fi
done
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py traffic /tmp/${tag}_FETCH_SIZE /tmp/${tag}_WRITE_SIZE > $GRAFT_REPO_ROOT/gpurun_out/${tag}_pmc_traffic.json
head -c 400 $GRAFT_REPO_ROOT/gpurun_out/${tag}_pmc_traffic.json
