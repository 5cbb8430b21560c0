#!/usr/bin/env python3
"""rocprofv3 --pmc SQ_* counter_collection.csv -> a markdown table of per-kernel fractions of SQ_WAVE_CYCLES"""
import collections
import csv
import glob
import os
import sys

path = sys.argv[1]
files = [path] if os.path.isfile(path) else glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for fn in files:
    with open(fn) as f:
        for r in csv.DictReader(f):
            n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("itsx::", "").replace("<12>", "")
            acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
cols = ["SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM"]
cols = [c for c in cols if any(c in d for d in acc.values())]
print("| kernel | wave-cycles (G) | " + " | ".join(c.replace("SQ_", "") for c in cols) + " |")
print("|---|---|" + "---|" * len(cols))
for n, d in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    wc = d.get("SQ_WAVE_CYCLES", 0)
    if wc < 1e7 or n.startswith("__amd"):
        continue
    print("| %s | %.1f | " % (n, wc / 1e9) + " | ".join("%.3f" % (d.get(c, 0) / wc) for c in cols) + " |")
