#!/bin/bash
# GPU idle gaps inside a train step: kernel trace of a short bench run -> per-step busy/idle and the largest gaps.
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/gaps -o g -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 20 --warmup 5 > /tmp/gaps.log 2>&1
f=$(find /tmp/gaps -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:]) for r in rows), key=lambda x: x[0])
# train steps: delimited by adam_step_kernel
adam = [i for i, e in enumerate(ev) if "adam_step" in e[2]]
gaps = collections.defaultdict(list)
tot_busy, tot_span = 0, 0
for a, b in zip(adam[8:-1], adam[9:]):   # skip warm-up steps
    seg = ev[a + 1: b + 1]
    span = seg[-1][1] - ev[a][1]
    busy = sum(e[1] - e[0] for e in seg)
    tot_busy += busy; tot_span += span
    prev_end, prev_name = ev[a][1], "adam_step"
    for s, e, n in seg:
        gaps[(prev_name, n)].append(max(0, s - prev_end))
        prev_end, prev_name = e, n
nst = len(adam[8:-1])
print(f"steps {nst}: span {tot_span/nst/1e3:.1f} us, busy {tot_busy/nst/1e3:.1f} us, idle {(tot_span-tot_busy)/nst/1e3:.1f} us per step")
top = sorted(((sum(v) / nst / 1e3, k) for k, v in gaps.items()), reverse=True)[:12]
for g, (a, b) in top:
    print(f"  {g:7.1f} us/step  {a}  ->  {b}")
PY
