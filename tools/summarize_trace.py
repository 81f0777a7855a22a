#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --kernel-trace CSV that separates the working launches from
the empty size-class launches (the chain kernels are launched once per LDS size class; a class with
no stream of its size exits in microseconds).  Usage: summarize_trace.py <kernel_trace.csv> [min_us]"""
import csv
import collections
import sys

path = sys.argv[1]
min_ns = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 1e6
rows = collections.defaultdict(list)
for r in csv.DictReader(open(path)):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if name.startswith("k_"):
        rows[name].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print("kernel,launches,working_launches,avg_working_ms,total_ms,empty_launch_avg_us")
for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    big = [x for x in v if x >= min_ns]
    small = [x for x in v if x < min_ns]
    print(f"{k},{len(v)},{len(big)},{(sum(big)/len(big)/1e6 if big else 0):.3f},{sum(v)/1e6:.3f},"
          f"{(sum(small)/len(small)/1e3 if small else 0):.1f}")
