cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/seq -o g -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 6 --warmup 3 > /tmp/seq.log 2>&1
f=$(find /tmp/seq -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda x: x[0])
adam = [i for i, e in enumerate(ev) if "adam_step" in e[2]]
a, b = adam[5], adam[6]
for s, e, n in ev[a + 1: b + 1]:
    short = n.split("(")[0][-60:] if "at::native" not in n else n[n.find("at::native"):][:90]
    print(f"{(e - s) / 1e3:7.1f} us  {short}")
PY
