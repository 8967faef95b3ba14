#!/bin/bash
# rocprofv3 kernel stats of the default bench (run on the GPU box via gpurun); summary -> gpurun_out/<tag>_kernel_stats.csv
tag=${1:-prof}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/$tag -o $tag -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --steps 50 --warmup 10 > /tmp/$tag.log 2>&1
f=$(find /tmp/$tag -name "*kernel_stats.csv" | head -1)
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
cp "$f" $GRAFT_REPO_ROOT/gpurun_out/${tag}_kernel_stats.csv
grep "^{\"metric\"" /tmp/$tag.log | tail -1 > $GRAFT_REPO_ROOT/gpurun_out/${tag}_bench_line.json
cut -d, -f1-4 "$f" | head -16
