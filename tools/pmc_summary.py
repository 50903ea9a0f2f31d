#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output (…_counter_collection.csv) per kernel: dispatches, and per counter the sum and the
mean per dispatch.   usage: pmc_summary.py <dir-or-csv> [out.csv]"""
import csv
import glob
import os
import sys
from collections import defaultdict

src = sys.argv[1]
files = [src] if os.path.isfile(src) else glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)
acc = defaultdict(lambda: defaultdict(float))
disp = defaultdict(set)
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("wtk::", "").split("(")[0][-60:].replace(",", ";")
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k].add((f, r["Dispatch_Id"]))
counters = sorted({c for k in acc for c in acc[k]})
lines = ["kernel,dispatches," + ",".join(f"{c}_sum,{c}_per_dispatch" for c in counters)]
for k in sorted(acc, key=lambda k: -sum(acc[k].values())):
    n = max(1, len(disp[k]))
    lines.append(f"{k},{n}," + ",".join(f"{acc[k].get(c, 0):.6g},{acc[k].get(c, 0) / n:.6g}" for c in counters))
out = "\n".join(lines)
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
