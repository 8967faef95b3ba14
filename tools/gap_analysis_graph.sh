#!/bin/bash
# GPU idle gaps inside the CAPTURED train step: kernel trace of a short graph bench -> per-step span / busy / idle and every
# hand-over gap (steps delimited by adam_hyper_kernel, the one eager launch in front of each replay).
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/gapsg -o g -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --steps 30 --warmup 10 > /tmp/gapsg.log 2>&1
f=$(find /tmp/gapsg -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-44:]) for r in rows), key=lambda x: x[0])
import os
key = os.environ.get("GS_GAP_KEY", "adam_hyper")
hyp = [i for i, e in enumerate(ev) if key in e[2]]
# the headline's timed steps: the longest run of equally spaced hyper launches with a fused project_bwd in between
steps = []
for a, b in zip(hyp[:-1], hyp[1:]):
    seg = ev[a:b]
    if any("project_bwd_kernel<3, true>" in e[2] for e in seg) and len(seg) < 40:
        steps.append(seg + [ev[b]])
steps = steps[len(steps) // 3:]
gaps = collections.defaultdict(list); dur = collections.defaultdict(list)
span = busy = 0
for seg in steps:
    span += seg[-1][0] - seg[0][0]
    busy += sum(e[1] - e[0] for e in seg[:-1])
    for (s0, e0, n0), (s1, e1, n1) in zip(seg[:-1], seg[1:]):
        gaps[(n0, n1)].append(max(0, s1 - e0)); dur[n0].append(e0 - s0)
n = len(steps)
print(f"captured steps {n}: span {span/n/1e3:.1f} us, busy {busy/n/1e3:.1f} us, idle {(span-busy)/n/1e3:.1f} us per step")
for (a, b), v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:24]:
    print(f"  gap {sum(v)/n/1e3:6.1f} us/step   {a} ({sum(dur[a])/len(dur[a])/1e3:.1f} us)  ->  {b}")
PY
