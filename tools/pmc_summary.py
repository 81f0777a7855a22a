#!/usr/bin/env python3
"""Per-kernel SQ counter summary from rocprofv3 --pmc CSV output (largest dispatch of each chain kernel).
usage: pmc_summary.py <counter_collection.csv>... [--steps N]   (N = steps per wave, for per-step figures)"""
import csv, sys, collections
steps = None
files = []
for a in sys.argv[1:]:
    if a.startswith("--steps="): steps = int(a.split("=")[1])
    else: files.append(a)
tot = collections.defaultdict(dict)
for f in files:
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "chain" not in k: continue
        agg[(k.split("(")[0], r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    byk = collections.defaultdict(list)
    for (k, d), v in agg.items(): byk[k].append(v)
    for k, l in byk.items():
        best = max(l, key=lambda v: max(v.values()))
        tot[k].update(best)
for k, v in tot.items():
    if v.get("SQ_INSTS_VALU", 0) < 1e6 and v.get("SQ_ACTIVE_INST_VALU", 0) < 1e6: continue
    print(k)
    waves = v.get("SQ_WAVES", 0) / 2          # the counter reports twice the launched waves on this part
    for n in sorted(v):
        line = f"   {n:24s} {v[n]:16.0f}"
        if steps and waves and n != "SQ_WAVES":
            per = v[n] / waves / steps
            line += f"   per wave-step {per:8.2f}" + (f"  (= {4*per:7.1f} cycles)" if "CYCLES" in n or "WAIT" in n or "ACTIVE" in n else "")
        print(line)
