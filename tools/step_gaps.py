#!/usr/bin/env python3
"""Idle time between consecutive kernels of the captured step, from a rocprofv3 kernel trace of the bench:
   rocprofv3 --kernel-trace --output-format csv -d /tmp/gaps -o g -- python3 bench.py --no-cpu-baseline --no-extras --steps 60 --warmup 10
   python3 tools/step_gaps.py /tmp/gaps        -> per (kernel -> next kernel) pair: median gap in us, over the last 40 steps"""
import csv, glob, statistics, sys, collections
f = [p for p in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)][0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "").replace("gs::", "")[:34]
# the timed loop: the last 40 occurrences of step_status_kernel delimit steps
ends = [i for i, r in enumerate(rows) if "step_status_kernel" in r["Kernel_Name"]]
lo, hi = ends[-41], ends[-1]
gaps = collections.defaultdict(list); durs = collections.defaultdict(list)
for i in range(lo + 1, hi + 1):
    a, b = rows[i - 1], rows[i]
    gaps[(name(a), name(b))].append((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3)
    durs[name(b)].append((int(b["End_Timestamp"]) - int(b["Start_Timestamp"])) / 1e3)
step = (int(rows[hi]["End_Timestamp"]) - int(rows[lo]["End_Timestamp"])) / 40 / 1e3
print(f"step {step:.1f} us; kernels {sum(statistics.median(v) for v in durs.values()):.1f} us; gaps {sum(statistics.median(v) for v in gaps.values() if len(v) >= 30):.1f} us")
for k, v in sorted(gaps.items(), key=lambda kv: -statistics.median(kv[1])):
    if len(v) >= 30:
        print(f"{statistics.median(v):7.1f} us  {k[0]:>34s} -> {k[1]}")
