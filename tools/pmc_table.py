#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 --pmc counters.  usage: pmc_table.py <dir-or-csv> [...] [--filter substr]"""
import collections, csv, glob, os, sys

paths, flt = [], "palace::"
a = sys.argv[1:]
while a:
    x = a.pop(0)
    if x == "--filter":
        flt = a.pop(0)
    elif os.path.isdir(x):
        paths += glob.glob(os.path.join(x, "**", "*counter_collection.csv"), recursive=True)
    else:
        paths.append(x)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for p in paths:
    for r in csv.DictReader(open(p)):
        if flt in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("palace::", "")[:40]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"    {c:34s} {sum(v)/len(v):16.0f}  (n={len(v)})")
