# The kernel sequence of ONE eager train step (model mirror + fused loss + fused Adam): durations and the gap in front of each
# kernel (kernel trace of a short --no-graph bench; steps delimited by adam_step_kernel).
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/seq -o g -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --no-graph --steps 12 --warmup 5 > /tmp/seq.log 2>&1
f=$(find /tmp/seq -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda x: x[0])
adam = [i for i, e in enumerate(ev) if "adam_step" in e[2]]
a, b = adam[9], adam[10]
prev = ev[a][1]; busy = 0
for s, e, n in ev[a + 1: b + 1]:
    short = n.split("(")[0][-60:] if "at::native" not in n else n[n.find("at::native"):][:90]
    print(f"{(e - s) / 1e3:7.1f} us  (+{max(0, s - prev) / 1e3:5.1f} idle)  {short}")
    busy += e - s; prev = e
print(f"span {(ev[b][1] - ev[a][1]) / 1e3:.1f} us, busy {busy / 1e3:.1f} us, kernels {b - a}")
PY
