#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV passes into HBM bytes per launch for our kernels.

    python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write > profiles/r01_pmc_traffic.json

FETCH_SIZE / WRITE_SIZE are reported in KiB-like units of 1024 B?  No: rocprofv3 reports them in
kilobytes (1 unit = 1024 B).  Per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): on gfx950
FETCH_SIZE reads exactly half of the bytes of a wide coalesced stream -> doubled here; WRITE_SIZE
is exact.  Collected in separate passes (TCC slots: FETCH_SIZE 3 + WRITE_SIZE 2 > 4).
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def load(dirname, counter):
    out = defaultdict(list)
    for path in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") == counter and "gs::" in row["Kernel_Name"]:   # our kernels only
                    name = row["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0].strip()
                    out[name].append(float(row["Counter_Value"]))
    return out


def main():
    fetch = load(sys.argv[1], "FETCH_SIZE")
    write = load(sys.argv[2], "WRITE_SIZE")
    res = {}
    for k in sorted(set(fetch) | set(write)):
        fb = 2.0 * 1024.0 * (sum(fetch[k]) / max(len(fetch[k]), 1)) if k in fetch else None
        wb = 1024.0 * (sum(write[k]) / max(len(write[k]), 1)) if k in write else None
        res[k] = {"launches_fetch": len(fetch.get(k, [])), "launches_write": len(write.get(k, [])),
                  "fetch_bytes_per_launch_x2_corrected": fb, "write_bytes_per_launch": wb,
                  "traffic_bytes_per_launch": None if fb is None or wb is None else fb + wb}
    json.dump(res, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
