#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV passes per kernel (template arguments kept, so the training and inference
instantiations of blend_fwd_kernel and the two stages of project_fwd_kernel stay separate).

    python tools/pmc_summary.py traffic <dir FETCH_SIZE> <dir WRITE_SIZE>   > profiles/rNN_pmc_traffic.json
    python tools/pmc_summary.py counters <dir> [<dir> ...]                   > profiles/rNN_sq_counters.json

traffic: L2 <-> fabric bytes per launch.  rocprofv3 reports FETCH_SIZE / WRITE_SIZE in units of 1024 B, collected in
separate passes (TCC slots: FETCH_SIZE 3 + WRITE_SIZE 2 > 4).  Calibration on gfx950 (tools/micro/traffic_cal.hip,
profiles/r03_traffic_calibration.json): FETCH_SIZE reads exactly HALF of the bytes of a wide coalesced stream (16 B / lane;
MI355X_MICROARCH.md, HBM section) but reads a GATHER in full at its 64-byte sector granularity (one random 16-byte record =
64 B counted; one random 48-byte record = 80 B counted: 1.42 sectors); WRITE_SIZE is exact for streams and counts a
scattered 1-byte store as 32 B and a scattered 48-byte row as 67 B.  A kernel's true fetch therefore lies between the raw
figure (all gathers) and twice the raw figure (all streams): both are reported, `traffic_bytes_per_launch` keeps the x2
figure (an upper bound for the blend kernels, exact for the streaming ones), `traffic_bytes_per_launch_lower` the raw one.
counters: plain per-launch averages of every counter found (SQ_* passes of tools/prof_sq.sh).
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def kernel_key(full: str) -> str:
    """'void gs::blend_fwd_kernel<true, 4>(gs::BlendFwdArgs)' -> 'blend_fwd_kernel<true, 4>'."""
    name = re.sub(r"\(.*$", "", full).strip()
    name = re.sub(r"^void\s+", "", name)
    return name.replace("gs::", "").strip()


def load(dirname):
    out = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if "gs::" in row["Kernel_Name"]:   # our kernels only
                    out[kernel_key(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return out


def avg(v):
    return sum(v) / max(len(v), 1)


def main():
    mode = sys.argv[1]
    if mode == "traffic":
        fetch, write = load(sys.argv[2]), load(sys.argv[3])
        res = {}
        for k in sorted(set(fetch) | set(write)):
            f, w = fetch.get(k, {}).get("FETCH_SIZE"), write.get(k, {}).get("WRITE_SIZE")
            fb = 2.0 * 1024.0 * avg(f) if f else None
            wb = 1024.0 * avg(w) if w else None
            res[k] = {"launches_fetch": len(f or []), "launches_write": len(w or []),
                      "fetch_bytes_per_launch_raw": None if fb is None else fb / 2.0,
                      "fetch_bytes_per_launch_x2_corrected": fb, "write_bytes_per_launch": wb,
                      "traffic_bytes_per_launch": None if fb is None or wb is None else fb + wb,
                      "traffic_bytes_per_launch_lower": None if fb is None or wb is None else fb / 2.0 + wb}
    else:
        res = defaultdict(dict)
        for d in sys.argv[2:]:
            for k, counters in load(d).items():
                for c, v in counters.items():
                    res[k][c] = avg(v)
                    res[k]["launches"] = len(v)
        res = {k: res[k] for k in sorted(res)}
    json.dump(res, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
