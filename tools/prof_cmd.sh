#!/bin/bash
# rocprofv3 kernel stats of an arbitrary python tool: tools/prof_cmd.sh <tag> <script.py> [args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/$tag -o $tag -- python3 $GRAFT_REPO_ROOT/$@ > /tmp/$tag.log 2>&1
f=$(find /tmp/$tag -name "*kernel_stats.csv" | head -1)
cp "$f" $GRAFT_REPO_ROOT/gpurun_out/${tag}_kernel_stats.csv
cut -d, -f1-4 "$f" | head -14
