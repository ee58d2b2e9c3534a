#!/usr/bin/env python3
"""The stage-04 kernels (selection, arcs, the decomposition's rounds) of the LAST step of a rocprofv3 kernel trace, in launch order:
usage: stage04_timeline.py <kernel_trace.csv> [min_us]  -- kernels of at least min_us (default 25), per-kernel and per-round sums"""
import collections
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
floor = float(sys.argv[2]) if len(sys.argv) > 2 else 25.0
idx = [i for i, r in enumerate(rows) if "st4_begin_kernel" in r["Kernel_Name"]]
start = idx[-1]
t0 = int(rows[start]["Start_Timestamp"])
seq = []
for r in rows[start:]:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "")
    if "palace::" not in n:
        continue
    k = n.split("palace::")[1].split("(")[0].split("<")[0]
    if k.startswith(("st4_", "dec_", "scan_")):
        seq.append((k, (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
print(len(seq), "launches")
for k, at, d in seq:
    if d >= floor:
        print(f"{k:28s} at {at:8.1f} us  dur {d:7.1f} us")
print("sum of durations us", round(sum(d for _, _, d in seq)), "span us", round(seq[-1][1] + seq[-1][2]))
c = collections.Counter()
for k, at, d in seq:
    c[k] += d
print({k: round(v) for k, v in c.most_common()})
rnd, per = -1, collections.Counter()
for k, at, d in seq:
    if k == "dec_round_begin_kernel":
        rnd += 1
    per[rnd] += d
print("per round (us; -1 = selection + arcs):", {k: round(v) for k, v in per.items()})
