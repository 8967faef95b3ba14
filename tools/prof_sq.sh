#!/bin/bash
# SQ counters of one kernel over a short bench run: tools/prof_sq.sh <tag> <kernel-substring>
tag=${1:-sq}; kern=${2:-blend_bwd}
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LEVEL_WAVES"; do
  d=/tmp/${tag}_$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -o x -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 2 > $d.log 2>&1
  f=$(find $d -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$kern" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in row["Kernel_Name"]:
        acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in acc.items():
    print(f"{k:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
done
