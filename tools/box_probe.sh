#!/bin/bash
# Which kind of box is this?  Quick bench + GPU identity; on a box whose blend_fwd is slow, collect more.
cd $GRAFT_REPO_ROOT
rocm-smi --showuniqueid --showcomputepartition --showmemorypartition 2>/dev/null | grep -i "GPU\[" | head -4
cat /proc/cpuinfo | grep "model name" | head -1; nproc
out=$(timeout 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | grep '^{"metric"')
echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('value', d['value'], 'fps', d['forward_fps'], d['stage_ms'], d['host'])"
slow=$(echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(1 if d['stage_ms']['gs_blend_fwd'] > 0.5 else 0)")
if [ "$slow" = "1" ]; then
  echo "SLOW BOX: extra diagnostics"
  rocm-smi 2>/dev/null | head -20
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --kernel-trace --output-format csv -d /tmp/slowpmc -o s -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 2 > /tmp/slowpmc.log 2>&1
  f=$(find /tmp/slowpmc -name "*counter_collection.csv" | head -1)
  grep "blend_fwd" "$f" | head -8
fi
