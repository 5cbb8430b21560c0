#!/usr/bin/env python3
"""Sum a rocprofv3 --pmc counter per kernel: pmc_summary.py <dir-or-csv> [counter]  ->  JSON {kernel: {launches, sum}}"""
import csv
import glob
import json
import os
import sys

path = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else None
files = [path] if os.path.isfile(path) else glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
out = {}
for fn in files:
    with open(fn) as f:
        for row in csv.DictReader(f):
            name = row.get("Kernel_Name") or row.get("Kernel Name") or ""
            cn = row.get("Counter_Name") or row.get("Counter Name") or ""
            if want and cn != want:
                continue
            v = float(row.get("Counter_Value") or row.get("Counter Value") or 0)
            k = name.split("(")[0].replace("void ", "").replace("itsx::", "")
            d = out.setdefault(k, {"counter": cn, "launches": 0, "sum": 0.0})
            d["launches"] += 1
            d["sum"] += v
print(json.dumps(out, indent=1, sort_keys=True))
