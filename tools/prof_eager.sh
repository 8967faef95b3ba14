#!/bin/bash
# rocprofv3 kernel stats of the eager bench step: tools/prof_eager.sh <tag>   (GS_BINNING etc. from the environment)
tag=${1:-eager}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/$tag -o $tag -- python3 $GRAFT_REPO_ROOT/bench.py --no-graph --no-cpu-baseline --no-extras --steps 30 --warmup 10 > /tmp/$tag.log 2>&1
f=$(find /tmp/$tag -name "*kernel_stats.csv" | head -1)
cp "$f" $GRAFT_REPO_ROOT/gpurun_out/${tag}_kernel_stats.csv
python3 - "$f" <<PY
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:26]:
    print(r["Name"][:72].ljust(72), r["Calls"].rjust(5), f'{float(r["AverageNs"])/1e3:9.1f}', r["Percentage"])
PY
