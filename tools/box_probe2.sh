#!/bin/bash
# Is this one of the boxes on which the eager blend_fwd stage reads 0.6-2.3 ms instead of 0.28?  Eager step rate, stage times,
# and -- if it is -- the kernel trace of the same eager loop (are the kernels long, or only the event brackets?).
cd $GRAFT_REPO_ROOT
rocm-smi --showuniqueid 2>/dev/null | grep -i "GPU\[" | head -2
out=$(timeout 200 python bench.py --no-cpu-baseline --no-extras --no-graph --steps 50 --warmup 10 2>/dev/null | grep '^{"metric"')
echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('eager value', d['value'], 'fps', d['forward_fps'], d['stage_ms'], d['blend_kernel_ms'], d['host'])"
slow=$(echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(1 if d['stage_ms']['gs_blend_fwd'] > 0.5 else 0)")
if [ "$slow" = "1" ]; then
  echo "ANOMALOUS BOX"
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/anom -o a -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --no-graph --steps 30 --warmup 5 > /tmp/anom.log 2>&1
  f=$(find /tmp/anom -name "*kernel_stats.csv" | head -1)
  cut -d, -f1-4,6,7 "$f" | head -12
  grep '^{"metric"' /tmp/anom.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('under rocprof: eager value', d['value'], d['stage_ms'])"
  mkdir -p $GRAFT_REPO_ROOT/gpurun_out; cp "$f" $GRAFT_REPO_ROOT/gpurun_out/anomalous_box_kernel_stats.csv
fi
